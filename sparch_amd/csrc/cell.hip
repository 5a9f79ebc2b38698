// G3/G4 (non-recurrent kinds), G5, G6, G7: fused spiking-cell kernels with the time
// loop inside the kernel.
//
// Forward replaces _lif_cell (snns.py:282-303) and _adlif_cell (419-445) together
// with SpikeFunctionBoxcar.forward (26-29), the bidirectional flip/cat glue
// (252-254, 272-275), nn.Dropout (278) and the firing-rate reduction (174).
// Backward replaces the autograd replay of those loops through
// SpikeFunctionBoxcar.backward (31-36); recurrences in SURVEY.md §8a.
//
// Mapping: one thread owns VEC (4 or 1) adjacent neurons of one virtual batch row and
// walks T in registers; lanes run along H so every global access of a wave is one
// contiguous segment (1 KiB at VEC=4).  Loads of the U next time steps are issued
// before the dependent arithmetic of the current ones (the inputs do not depend on the
// recurrence), which is what hides HBM latency at one or two waves per SIMD.
// Arithmetic keeps the reference's operation order with -ffp-contract=off, so given
// identical inputs the spikes are bit-identical to the eager CPU path.
// HBM-bound: 12H (LIF) / 16H (adLIF) bytes per sample-step forward, the same backward.
#include "common.h"

namespace {

// time steps of loads in flight per thread: the small-problem (VEC = 1) variant runs at ~4 waves per CU and
// needs a deep prefetch to cover HBM latency; the vector variant has 4x the bytes per load
template <int VEC> struct Depth { static constexpr int U = VEC == 1 ? 16 : 8; };

struct CellArgs {
    int B, dirs, T, H;
    const float* Wx; const float* scale; const float* shift;
    const float* alpha; const float* beta; const float* a; const float* b;
    const float* u0; const float* w0; const float* s0;
    float theta, p_drop, inv_keep; uint64_t seed;
    float* s_out; uint16_t* s16_out; float* u_save; float* w_save; uint32_t* spike_count;
    // backward
    const float* g_out; const float* g_rate; float g_rate_scale;
    float* dWx; float* dparam_ws;
    // BatchNorm backward folded in (nullable): the raw projection and its per-column statistics; the kernel
    // then also leaves sum_t dWx and sum_t dWx*xhat per (row, column) in two more planes of dparam_ws
    const float* bn_x; const float* bn_mean; const float* bn_invstd;
    int save16;  // u_save / w_save hold bf16 (common.h save_u16) instead of fp32
};

template <int VEC>
__device__ __forceinline__ void ldv(float (&d)[VEC], const float* p) {
    if constexpr (VEC == 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    } else {
        d[0] = p[0];
    }
}
// saved states (u, w): element index i of an fp32 or bf16 array
template <int VEC>
__device__ __forceinline__ void ld_saved(float (&d)[VEC], const float* base, size_t i, bool s16) {
    if (!s16) { ldv<VEC>(d, base + i); return; }
    const unsigned short* q = reinterpret_cast<const unsigned short*>(base) + i;
    if constexpr (VEC == 4) {
        const unsigned long long raw = *reinterpret_cast<const unsigned long long*>(q);
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = bf16_to_f32((unsigned short)(raw >> (16 * e)));
    } else {
        d[0] = bf16_to_f32(q[0]);
    }
}
template <int VEC, bool IS_U>
__device__ __forceinline__ void st_saved(float* base, size_t i, const float (&d)[VEC], bool s16, float theta) {
    if (!s16) {
        if constexpr (VEC == 4) *reinterpret_cast<f32x4*>(base + i) = f32x4{d[0], d[1], d[2], d[3]};
        else base[i] = d[0];
        return;
    }
    unsigned short* q = reinterpret_cast<unsigned short*>(base) + i;
    if constexpr (VEC == 4) {
        unsigned long long raw = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            raw |= (unsigned long long)(IS_U ? save_u16(d[e], theta) : f32_to_bf16_rne(d[e])) << (16 * e);
        *reinterpret_cast<unsigned long long*>(q) = raw;
    } else {
        q[0] = IS_U ? save_u16(d[0], theta) : f32_to_bf16_rne(d[0]);
    }
}
template <int VEC>
__device__ __forceinline__ void stv(float* p, const float (&d)[VEC]) {
    if constexpr (VEC == 4) {
        f32x4 v; v.x = d[0]; v.y = d[1]; v.z = d[2]; v.w = d[3];
        *reinterpret_cast<f32x4*>(p) = v;
    } else {
        p[0] = d[0];
    }
}

template <bool ADAPT, int VEC, bool S16>
__global__ __launch_bounds__(256) void cell_fwd_kernel(CellArgs c) {
    constexpr int U = Depth<VEC>::U;
    const int HQ = c.H / VEC;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int Bp = c.B * c.dirs;
    if (idx >= (long long)Bp * HQ) return;
    const int bp = (int)(idx / HQ), h = (int)(idx % HQ) * VEC;
    const int d = bp / c.B, b = bp - d * c.B;
    const int T = c.T, H = c.H, HO = c.H * c.dirs;

    float al[VEC], oma[VEC], be[VEC], pa[VEC], pb[VEC], sc[VEC], sh[VEC];
    float u[VEC], w[VEC], s[VEC];
    uint32_t cnt[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        al[e] = clampf(c.alpha[h + e], SP_ALPHA_LO, SP_ALPHA_HI);
        oma[e] = 1.0f - al[e];
        if (ADAPT) {
            be[e] = clampf(c.beta[h + e], SP_BETA_LO, SP_BETA_HI);
            pa[e] = clampf(c.a[h + e], SP_A_LO, SP_A_HI);
            pb[e] = clampf(c.b[h + e], SP_B_LO, SP_B_HI);
        }
        sc[e] = c.scale ? c.scale[h + e] : 1.0f;
        sh[e] = c.scale ? c.shift[h + e] : 0.0f;
        cnt[e] = 0;
    }
    ldv<VEC>(u, c.u0 + (size_t)bp * H + h);
    ldv<VEC>(s, c.s0 + (size_t)bp * H + h);
    if (ADAPT) ldv<VEC>(w, c.w0 + (size_t)bp * H + h);
    const bool has_norm = c.scale != nullptr;
    const bool drop = c.p_drop > 0.0f;
    const uint64_t seed = drop ? resolve_seed(c.seed) : 0;

    for (int t0 = 0; t0 < T; t0 += U) {
        float x[U][VEC];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int t = t0 + j;
            if (t < T) {
                const int tt = d ? (T - 1 - t) : t;
                ldv<VEC>(x[j], c.Wx + ((size_t)b * T + tt) * H + h);
            }
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int t = t0 + j;
            if (t >= T) break;
            const int tt = d ? (T - 1 - t) : t;
            float so[VEC];
            const size_t o = ((size_t)b * T + tt) * HO + (size_t)d * H + h;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float xn = x[j][e];
                if (has_norm) xn = bn_affine(xn, sc[e], sh[e]);
                float drive = xn;
                if (ADAPT) {
                    w[e] = (be[e] * w[e] + pa[e] * u[e]) + pb[e] * s[e];  // snns.py:438
                    drive = xn - w[e];
                }
                u[e] = al[e] * (u[e] - s[e]) + oma[e] * drive;           // snns.py:297 / 439
                s[e] = (u[e] - c.theta) > 0.0f ? 1.0f : 0.0f;            // snns.py:29
                const float k = drop ? keep_scale(seed, o + e, c.p_drop, c.inv_keep) : 1.0f;
                so[e] = s[e] * k;
                cnt[e] += (so[e] != 0.0f) ? 1u : 0u;
            }
            stv<VEC>(c.s_out + o, so);
            if (c.s16_out) {  // the same spikes as a bf16 plane (0 / 1.0) for the GEMMs that consume them
#pragma unroll
                for (int e = 0; e < VEC; ++e) c.s16_out[o + e] = so[e] != 0.0f ? (uint16_t)0x3F80 : (uint16_t)0;
            }
            if (c.u_save) st_saved<VEC, true>(c.u_save, ((size_t)bp * T + t) * H + h, u, S16, c.theta);
            if (ADAPT && c.w_save) st_saved<VEC, false>(c.w_save, ((size_t)bp * T + t) * H + h, w, S16, c.theta);
        }
    }
    if (c.spike_count) {
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            if (cnt[e]) atomicAdd(c.spike_count + (size_t)d * H + h + e, cnt[e]);
    }
}

template <bool ADAPT, int VEC, bool S16>
__global__ __launch_bounds__(256) void cell_bwd_kernel(CellArgs c) {
    constexpr int U = Depth<VEC>::U;
    const int HQ = c.H / VEC;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int Bp = c.B * c.dirs;
    if (idx >= (long long)Bp * HQ) return;
    const int bp = (int)(idx / HQ), h = (int)(idx % HQ) * VEC;
    const int d = bp / c.B, b = bp - d * c.B;
    const int T = c.T, H = c.H, HO = c.H * c.dirs;

    float al[VEC], oma[VEC], be[VEC], pa[VEC], pb[VEC], gr[VEC];
    float du_n[VEC], dw_n[VEC], u_t[VEC];
    float acc_al[VEC], acc_be[VEC], acc_a[VEC], acc_b[VEC];
    const bool bn = c.bn_x != nullptr;
    float bn_mu[VEC], bn_is[VEC], acc_dy[VEC], acc_dyx[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        bn_mu[e] = bn ? c.bn_mean[h + e] : 0.f;
        bn_is[e] = bn ? c.bn_invstd[h + e] : 0.f;
        acc_dy[e] = acc_dyx[e] = 0.f;
        al[e] = clampf(c.alpha[h + e], SP_ALPHA_LO, SP_ALPHA_HI);
        oma[e] = 1.0f - al[e];
        if (ADAPT) {
            be[e] = clampf(c.beta[h + e], SP_BETA_LO, SP_BETA_HI);
            pa[e] = clampf(c.a[h + e], SP_A_LO, SP_A_HI);
            pb[e] = clampf(c.b[h + e], SP_B_LO, SP_B_HI);
        }
        gr[e] = c.g_rate ? c.g_rate[(size_t)d * H + h + e] * c.g_rate_scale : 0.0f;
        du_n[e] = dw_n[e] = 0.f;
        acc_al[e] = acc_be[e] = acc_a[e] = acc_b[e] = 0.f;
    }
    constexpr bool s16 = S16;  // compile-time: a run-time flag here breaks up the batched prefetch loads
    ld_saved<VEC>(u_t, c.u_save, ((size_t)bp * T + (T - 1)) * H + h, s16);
    const bool drop = c.p_drop > 0.0f;
    const uint64_t seed = drop ? resolve_seed(c.seed) : 0;

    for (int t0 = T - 1; t0 >= 0; t0 -= U) {
        float g[U][VEC], up[U][VEC], wp[U][VEC], xr[U][VEC];
        // bf16 saves: the raw words are fetched here and unpacked at their use, so that the U steps' loads stay
        // one batch (unpacking next to each load made hipcc wait for every load in turn: 2x the kernel time)
        [[maybe_unused]] unsigned long long upr[U], wpr[U];
        if constexpr (S16 && VEC == 4) {
            // branch-free batch (steps below 0 re-read step 0: in range, never used): with the per-step `if (t >= 0)`
            // the 64-bit raw words crossed basic blocks and hipcc waited for every load in turn
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int t = max(t0 - j, 0);
                const int tt = d ? (T - 1 - t) : t;
                ldv<VEC>(g[j], c.g_out + ((size_t)b * T + tt) * HO + (size_t)d * H + h);
                if (bn) ldv<VEC>(xr[j], c.bn_x + ((size_t)b * T + tt) * H + h);
                const size_t i = ((size_t)bp * T + (t > 0 ? t - 1 : 0)) * H + h;  // t = 0 reads u0 / w0 below
                upr[j] = *reinterpret_cast<const unsigned long long*>(reinterpret_cast<const unsigned short*>(c.u_save) + i);
                if (ADAPT) wpr[j] = *reinterpret_cast<const unsigned long long*>(reinterpret_cast<const unsigned short*>(c.w_save) + i);
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int t = t0 - j;
                if (t >= 0) {
                    const int tt = d ? (T - 1 - t) : t;
                    ldv<VEC>(g[j], c.g_out + ((size_t)b * T + tt) * HO + (size_t)d * H + h);
                    if (bn) ldv<VEC>(xr[j], c.bn_x + ((size_t)b * T + tt) * H + h);
                    if (t > 0) {
                        const size_t i = ((size_t)bp * T + (t - 1)) * H + h;
                        ld_saved<VEC>(up[j], c.u_save, i, s16);
                        if (ADAPT) ld_saved<VEC>(wp[j], c.w_save, i, s16);
                    } else {
                        ldv<VEC>(up[j], c.u0 + (size_t)bp * H + h);
                        if (ADAPT) ldv<VEC>(wp[j], c.w0 + (size_t)bp * H + h);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int t = t0 - j;
            if (t < 0) break;
            const int tt = d ? (T - 1 - t) : t;
            const size_t o = ((size_t)b * T + tt) * HO + (size_t)d * H + h;
            float sp[VEC], dwx[VEC];
            if constexpr (S16 && VEC == 4) {
                if (t > 0) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        up[j][e] = bf16_to_f32((unsigned short)(upr[j] >> (16 * e)));
                        if (ADAPT) wp[j][e] = bf16_to_f32((unsigned short)(wpr[j] >> (16 * e)));
                    }
                } else {  // last step of the reverse pass: the exact fp32 initial states
                    ldv<VEC>(up[j], c.u0 + (size_t)bp * H + h);
                    if (ADAPT) ldv<VEC>(wp[j], c.w0 + (size_t)bp * H + h);
                }
            }
            if (t > 0) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) sp[e] = (up[j][e] - c.theta) > 0.0f ? 1.0f : 0.0f;
            } else {
                ldv<VEC>(sp, c.s0 + (size_t)bp * H + h);
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float k = drop ? keep_scale(seed, o + e, c.p_drop, c.inv_keep) : 1.0f;
                const float gs = (g[j][e] + gr[e]) * k;
                float ds = gs - al[e] * du_n[e];
                if (ADAPT) ds = ds + pb[e] * dw_n[e];
                const float xs = u_t[e] - c.theta;
                float du = boxcar_gate(ds, xs) + al[e] * du_n[e];            // snns.py:33-35
                if (ADAPT) du = du + pa[e] * dw_n[e];
                dwx[e] = oma[e] * du;
                if (bn) {  // BatchNorm backward's column sums (dy = dWx, xhat = (x - mean) * invstd)
                    acc_dy[e] += dwx[e];
                    acc_dyx[e] += dwx[e] * ((xr[j][e] - bn_mu[e]) * bn_is[e]);
                }
                const float q = up[j][e] - sp[e];
                acc_al[e] += du * (q - u_t[e]);  // d u_t / d alpha = (q - u_t)/(1-alpha); scaled at the end
                if (ADAPT) {
                    const float dw = be[e] * dw_n[e] - dwx[e];
                    acc_be[e] += dw * wp[j][e];
                    acc_a[e] += dw * up[j][e];
                    acc_b[e] += dw * sp[e];
                    dw_n[e] = dw;
                }
                du_n[e] = du;
                u_t[e] = up[j][e];
            }
            stv<VEC>(c.dWx + ((size_t)bp * T + tt) * H + h, dwx);
        }
    }
    const size_t plane = (size_t)Bp * H;
    float* ws = c.dparam_ws + (size_t)bp * H + h;
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc_al[e] = acc_al[e] / oma[e];
    stv<VEC>(ws, acc_al);
    if (ADAPT) {
        stv<VEC>(ws + plane, acc_be);
        stv<VEC>(ws + 2 * plane, acc_a);
        stv<VEC>(ws + 3 * plane, acc_b);
    }
    if (bn) {
        stv<VEC>(ws + 4 * plane, acc_dy);
        stv<VEC>(ws + 5 * plane, acc_dyx);
    }
}

// ------------------------------------------------------------------ readout cell
// Readout layer (snns.py:815-825): u_t = alpha u_{t-1} + (1-alpha) x_t, out = sum_t softmax_c(u_t).
// One wave per batch row, time in chunks of RT steps staged through LDS, and the lane's meaning changes
// per phase so that nothing needs a cross-lane reduction:
//   lane = class : the linear recurrence over t (one FMA per step), u_t -> LDS [t][class]
//   lane = time  : softmax over the classes of "its" time step, sequentially over the C values in LDS
//                  (rows are CS = C|1 floats apart: odd stride, conflict-free for lane = time)
//   lane = class : out_c += p[t][c] in time order (the same summation order as the reference's loop)
// (The first version kept lane = class throughout and paid two butterfly reductions through the LDS
// crossbar per time step: 137 us forward / 194 us backward for 256 x 250 x 35.)
constexpr int RT = 128;  // time steps per chunk
constexpr int RU = 8;    // global loads in flight per lane in the recurrence phases
// More than 64 classes: NW waves per batch row (lane -> thread index, wave barrier -> workgroup barrier).
template <int NW>
__device__ __forceinline__ void ro_barrier() {
    if (NW == 1) __builtin_amdgcn_wave_barrier();
    else __syncthreads();
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void readout_fwd_kernel(int B, int T, int C, const float* __restrict__ Wx,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift,
                                                         const float* __restrict__ alpha,
                                                         const float* __restrict__ u0, float* __restrict__ out,
                                                         float* __restrict__ u_save) {
    __shared__ float us[RT * (64 * NW + 1)];
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    const int CS = C | 1;
    const bool act = lane < C;
    const int cc = act ? lane : 0;
    const float al = clampf(alpha[cc], SP_ALPHA_LO, SP_ALPHA_HI), oma = 1.0f - al;
    const float sc = scale ? scale[cc] : 1.0f, sh = scale ? shift[cc] : 0.0f;
    float u = u0[(size_t)b * C + cc], acc = 0.f;
    const float* xr = Wx + (size_t)b * T * C + cc;
    for (int c0 = 0; c0 < T; c0 += RT) {
        const int len = min(RT, T - c0);
        // lane = class: recurrence
        for (int t0 = 0; t0 < len; t0 += RU) {
            float x[RU];
#pragma unroll
            for (int j = 0; j < RU; ++j) x[j] = (t0 + j < len) ? xr[(size_t)(c0 + t0 + j) * C] : 0.f;
#pragma unroll
            for (int j = 0; j < RU; ++j) {
                if (t0 + j >= len) break;
                float xn = x[j];
                if (scale) xn = bn_affine(xn, sc, sh);
                u = al * u + oma * xn;                                   // snns.py:822
                if (act) {
                    us[(t0 + j) * CS + cc] = u;
                    if (u_save) u_save[((size_t)b * T + c0 + t0 + j) * C + cc] = u;
                }
            }
        }
        ro_barrier<NW>();
        // lane = time: softmax over classes, in place
        for (int tl = lane; tl < len; tl += 64 * NW) {
            float* row = us + tl * CS;
            float m = row[0];
            for (int c = 1; c < C; ++c) m = fmaxf(m, row[c]);
            float den = 0.f;
            for (int c = 0; c < C; ++c) {
                const float e = expf(row[c] - m);
                row[c] = e;
                den += e;
            }
            for (int c = 0; c < C; ++c) row[c] = row[c] / den;
        }
        ro_barrier<NW>();
        // lane = class: out += softmax(u_t) in time order                 // snns.py:823
        if (act) {
            for (int t0 = 0; t0 < len; t0 += RU) {  // reads first (independent), then the ordered adds
                float pv[RU];
#pragma unroll
                for (int j = 0; j < RU; ++j) pv[j] = (t0 + j < len) ? us[(t0 + j) * CS + cc] : 0.f;
#pragma unroll
                for (int j = 0; j < RU; ++j)
                    if (t0 + j < len) acc = acc + pv[j];
            }
        }
        ro_barrier<NW>();
    }
    if (act) out[(size_t)b * C + cc] = acc;
}

// Backward of the above.  With p_t = softmax(u_t) and g = dL/dout:
//   e_t = p_t * (g - <p_t, g>)            (lane = time; independent over t)
//   du_t = alpha du_{t+1} + e_t,  dWx_t = (1-alpha) du_t,  dalpha += du_t (u_{t-1} - u_t) / (1-alpha)
//                                          (lane = class; reverse time)
template <int NW>
__global__ __launch_bounds__(64 * NW) void readout_bwd_kernel(int B, int T, int C, const float* __restrict__ g_out,
                                                         const float* __restrict__ u_save,
                                                         const float* __restrict__ alpha,
                                                         const float* __restrict__ u0, float* __restrict__ dWx,
                                                         float* __restrict__ dalpha_ws,
                                                         const float* __restrict__ bn_x,
                                                         const float* __restrict__ bn_mean,
                                                         const float* __restrict__ bn_invstd) {
    __shared__ float us[RT * (64 * NW + 1)];
    __shared__ float gs[64 * NW];
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    const int CS = C | 1;
    const bool act = lane < C;
    const int cc = act ? lane : 0;
    const float al = clampf(alpha[cc], SP_ALPHA_LO, SP_ALPHA_HI), oma = 1.0f - al;
    gs[lane] = act ? g_out[(size_t)b * C + cc] : 0.f;
    const float* ur = u_save + (size_t)b * T * C + cc;
    float du = 0.f, acc = 0.f;
    // BatchNorm backward's column sums folded in (nullable): sum_t dWx and sum_t dWx*xhat per (row, class)
    const bool bn = bn_x != nullptr;
    const float bn_mu = bn ? bn_mean[cc] : 0.f, bn_is = bn ? bn_invstd[cc] : 0.f;
    const float* xr = bn ? bn_x + (size_t)b * T * C + cc : nullptr;
    float acc_dy = 0.f, acc_dyx = 0.f;
    const int nchunk = (T + RT - 1) / RT;
    for (int ch = nchunk - 1; ch >= 0; --ch) {
        const int c0 = ch * RT, len = min(RT, T - c0);
        // lane = class: u_t of the chunk -> LDS
        for (int t0 = 0; t0 < len; t0 += RU) {
            float x[RU];
#pragma unroll
            for (int j = 0; j < RU; ++j) x[j] = (t0 + j < len) ? ur[(size_t)(c0 + t0 + j) * C] : 0.f;
#pragma unroll
            for (int j = 0; j < RU; ++j)
                if (t0 + j < len && act) us[(t0 + j) * CS + cc] = x[j];
        }
        ro_barrier<NW>();
        // lane = time: e_t in place
        for (int tl = lane; tl < len; tl += 64 * NW) {
            float* row = us + tl * CS;
            float m = row[0];
            for (int c = 1; c < C; ++c) m = fmaxf(m, row[c]);
            float den = 0.f;
            for (int c = 0; c < C; ++c) {
                const float e = expf(row[c] - m);
                row[c] = e;
                den += e;
            }
            float dot = 0.f;
            for (int c = 0; c < C; ++c) {
                const float pc = row[c] / den;
                row[c] = pc;
                dot += pc * gs[c];
            }
            for (int c = 0; c < C; ++c) row[c] = row[c] * (gs[c] - dot);
        }
        ro_barrier<NW>();
        // lane = class: reverse recurrence; u_{t-1} - u_t re-read from u_save (coalesced, prefetched)
        for (int t0 = len - 1; t0 >= 0; t0 -= RU) {
            float uc[RU + 1];  // uc[j] = u_{t0-j}, uc[RU] = u_{t0-RU}
#pragma unroll
            for (int j = 0; j <= RU; ++j) {
                const int t = c0 + t0 - j;
                uc[j] = t >= 0 ? ur[(size_t)t * C] : (t == -1 ? u0[(size_t)b * C + cc] : 0.f);
            }
            float ev[RU], xv[RU];
#pragma unroll
            for (int j = 0; j < RU; ++j) ev[j] = (t0 - j >= 0) ? us[(t0 - j) * CS + cc] : 0.f;
#pragma unroll
            for (int j = 0; j < RU; ++j) xv[j] = (bn && t0 - j >= 0) ? xr[(size_t)(c0 + t0 - j) * C] : 0.f;
#pragma unroll
            for (int j = 0; j < RU; ++j) {
                const int tl = t0 - j;
                if (tl < 0) break;
                du = al * du + ev[j];
                const float dwx = oma * du;
                if (act) dWx[((size_t)b * T + c0 + tl) * C + cc] = dwx;
                acc += du * (uc[j + 1] - uc[j]);
                acc_dy += dwx;
                acc_dyx += dwx * ((xv[j] - bn_mu) * bn_is);
            }
        }
        ro_barrier<NW>();
    }
    if (act) {
        dalpha_ws[(size_t)b * C + cc] = acc / oma;
        if (bn) {
            dalpha_ws[((size_t)B + b) * C + cc] = acc_dy;
            dalpha_ws[((size_t)2 * B + b) * C + cc] = acc_dyx;
        }
    }
}

template <bool ADAPT>
int launch_cell(bool bwd, CellArgs& c, hipStream_t st) {
    const long long work = (long long)c.B * c.dirs * c.H;
    const bool vec_ok = (c.H % 4 == 0);
    // VEC=4 only when it still leaves >= 8 waves per CU; small problems favour more threads
    const bool vec4 = vec_ok && work >= (long long)256 * 8 * 64 * 4;
    const long long threads = vec4 ? work / 4 : work;
    const unsigned blocks = (unsigned)((threads + 255) / 256);
#define SP_CELL_LAUNCH(KERNEL, S16)                                                                     \
    do {                                                                                                \
        if (vec4) hipLaunchKernelGGL((KERNEL<ADAPT, 4, S16>), dim3(blocks), dim3(256), 0, st, c);       \
        else      hipLaunchKernelGGL((KERNEL<ADAPT, 1, S16>), dim3(blocks), dim3(256), 0, st, c);       \
    } while (0)
    if (!bwd) {
        if (c.save16) SP_CELL_LAUNCH(cell_fwd_kernel, true); else SP_CELL_LAUNCH(cell_fwd_kernel, false);
    } else {
        if (c.save16) SP_CELL_LAUNCH(cell_bwd_kernel, true); else SP_CELL_LAUNCH(cell_bwd_kernel, false);
    }
#undef SP_CELL_LAUNCH
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

bool ptrs_aligned(std::initializer_list<const void*> ps) {
    for (const void* p : ps)
        if (p && !aligned16(p)) return false;
    return true;
}

}  // namespace

extern "C" int sparch_cell_fwd(int kind, int B, int dirs, int T, int H, const float* Wx,
                               const float* scale, const float* shift, const float* alpha,
                               const float* beta, const float* a, const float* b, const float* u0,
                               const float* w0, const float* s0, float theta, float p_drop,
                               uint64_t seed, float* s_out, uint16_t* s16_out, void* u_save, void* w_save,
                               int save_bf16, uint32_t* spike_count, void* stream) {
    SPARCH_ENTER();
    if (kind != SPARCH_KIND_LIF && kind != SPARCH_KIND_ADLIF) return SPARCH_EINVAL;
    const bool adapt = kind == SPARCH_KIND_ADLIF;
    if (B <= 0 || T <= 0 || H <= 0 || (dirs != 1 && dirs != 2) || !Wx || !alpha || !u0 || !s0 || !s_out)
        return SPARCH_EINVAL;
    if (adapt && (!beta || !a || !b || !w0)) return SPARCH_EINVAL;
    if ((scale == nullptr) != (shift == nullptr)) return SPARCH_EINVAL;
    if (!(p_drop >= 0.0f && p_drop < 1.0f)) return SPARCH_EINVAL;
    if (!ptrs_aligned({Wx, u0, w0, s0, s_out, u_save, w_save})) return SPARCH_EALIGN;
    CellArgs c{};
    c.B = B; c.dirs = dirs; c.T = T; c.H = H;
    c.Wx = Wx; c.scale = scale; c.shift = shift;
    c.alpha = alpha; c.beta = beta; c.a = a; c.b = b;
    c.u0 = u0; c.w0 = w0; c.s0 = s0;
    c.theta = theta; c.p_drop = p_drop; c.inv_keep = 1.0f / (1.0f - p_drop); c.seed = seed;
    c.s_out = s_out; c.s16_out = s16_out; c.u_save = (float*)u_save; c.w_save = (float*)w_save;
    c.save16 = save_bf16 != 0; c.spike_count = spike_count;
    return adapt ? launch_cell<true>(false, c, (hipStream_t)stream)
                 : launch_cell<false>(false, c, (hipStream_t)stream);
}

extern "C" int sparch_cell_bwd(int kind, int B, int dirs, int T, int H, const float* g_out,
                               const float* g_rate, const void* u_save, const void* w_save, int save_bf16,
                               const float* alpha, const float* beta, const float* a, const float* b,
                               const float* u0, const float* w0, const float* s0, float theta,
                               float p_drop, uint64_t seed, float* dWx, float* dparam_ws, const float* bn_x,
                               const float* bn_mean, const float* bn_invstd, void* stream) {
    SPARCH_ENTER();
    if (kind != SPARCH_KIND_LIF && kind != SPARCH_KIND_ADLIF) return SPARCH_EINVAL;
    if (bn_x && (!bn_mean || !bn_invstd || !aligned16(bn_x))) return SPARCH_EINVAL;
    const bool adapt = kind == SPARCH_KIND_ADLIF;
    if (B <= 0 || T <= 0 || H <= 0 || (dirs != 1 && dirs != 2) || !g_out || !u_save || !alpha || !u0 ||
        !s0 || !dWx || !dparam_ws)
        return SPARCH_EINVAL;
    if (adapt && (!beta || !a || !b || !w0 || !w_save)) return SPARCH_EINVAL;
    if (!(p_drop >= 0.0f && p_drop < 1.0f)) return SPARCH_EINVAL;
    if (!ptrs_aligned({g_out, u_save, w_save, u0, w0, s0, dWx, dparam_ws})) return SPARCH_EALIGN;
    CellArgs c{};
    c.B = B; c.dirs = dirs; c.T = T; c.H = H;
    c.alpha = alpha; c.beta = beta; c.a = a; c.b = b;
    c.u0 = u0; c.w0 = w0; c.s0 = s0;
    c.theta = theta; c.p_drop = p_drop; c.inv_keep = 1.0f / (1.0f - p_drop); c.seed = seed;
    c.u_save = (float*)const_cast<void*>(u_save); c.w_save = (float*)const_cast<void*>(w_save);
    c.save16 = save_bf16 != 0;
    c.g_out = g_out; c.g_rate = g_rate; c.g_rate_scale = 1.0f / ((float)B * (float)T);
    c.dWx = dWx; c.dparam_ws = dparam_ws;
    c.bn_x = bn_x; c.bn_mean = bn_mean; c.bn_invstd = bn_invstd;
    return adapt ? launch_cell<true>(true, c, (hipStream_t)stream)
                 : launch_cell<false>(true, c, (hipStream_t)stream);
}

extern "C" int sparch_readout_fwd(int B, int T, int C, const float* Wx, const float* scale,
                                  const float* shift, const float* alpha, const float* u0, float* out,
                                  float* u_save, void* stream) {
    SPARCH_ENTER();
    if (B <= 0 || T <= 0 || C <= 0 || C > 256 || !Wx || !alpha || !u0 || !out) return SPARCH_EINVAL;
    if ((scale == nullptr) != (shift == nullptr)) return SPARCH_EINVAL;
    if (C > 128)     hipLaunchKernelGGL(readout_fwd_kernel<4>, dim3(B), dim3(256), 0, (hipStream_t)stream, B, T, C, Wx, scale, shift,
                                        alpha, u0, out, u_save);
    else if (C > 64) hipLaunchKernelGGL(readout_fwd_kernel<2>, dim3(B), dim3(128), 0, (hipStream_t)stream, B, T, C, Wx, scale, shift,
                                        alpha, u0, out, u_save);
    else hipLaunchKernelGGL(readout_fwd_kernel<1>, dim3(B), dim3(64), 0, (hipStream_t)stream, B, T, C, Wx, scale, shift,
                       alpha, u0, out, u_save);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

extern "C" int sparch_readout_bwd(int B, int T, int C, const float* g_out, const float* bn_x,
                                  const float* bn_mean, const float* bn_invstd, const float* u_save,
                                  const float* alpha, const float* u0, float* dWx, float* dalpha_ws,
                                  void* stream) {
    SPARCH_ENTER();
    // (dalpha uses u_{t-1}-x_t = (u_{t-1}-u_t)/(1-alpha): the projection is re-read only for BatchNorm's sums)
    if (B <= 0 || T <= 0 || C <= 0 || C > 256 || !g_out || !u_save || !alpha || !u0 || !dWx || !dalpha_ws)
        return SPARCH_EINVAL;
    if (bn_x && (!bn_mean || !bn_invstd)) return SPARCH_EINVAL;
    if (C > 128)     hipLaunchKernelGGL(readout_bwd_kernel<4>, dim3(B), dim3(256), 0, (hipStream_t)stream, B, T, C, g_out, u_save,
                                        alpha, u0, dWx, dalpha_ws, bn_x, bn_mean, bn_invstd);
    else if (C > 64) hipLaunchKernelGGL(readout_bwd_kernel<2>, dim3(B), dim3(128), 0, (hipStream_t)stream, B, T, C, g_out, u_save,
                                        alpha, u0, dWx, dalpha_ws, bn_x, bn_mean, bn_invstd);
    else hipLaunchKernelGGL(readout_bwd_kernel<1>, dim3(B), dim3(64), 0, (hipStream_t)stream, B, T, C, g_out, u_save,
                       alpha, u0, dWx, dalpha_ws, bn_x, bn_mean, bn_invstd);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

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
#include <type_traits>

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
#ifndef CELL_NT
#define CELL_NT 1  // saved states (written by the forward, read once by the backward much later) as non-temporal accesses:
                   // they do not take the infinity cache's room from what the NEXT kernel reads (cfg2 step 0.956 -> 0.937 ms, one call)
#endif
template <int VEC, bool IS_U>
__device__ __forceinline__ void st_saved(float* base, size_t i, const float (&d)[VEC], bool s16, float theta) {
    if (!s16) {
#if CELL_NT
        if constexpr (VEC == 4) __builtin_nontemporal_store(f32x4{d[0], d[1], d[2], d[3]}, reinterpret_cast<f32x4*>(base + i));
        else __builtin_nontemporal_store(d[0], base + i);
#else
        if constexpr (VEC == 4) *reinterpret_cast<f32x4*>(base + i) = f32x4{d[0], d[1], d[2], d[3]};
        else base[i] = d[0];
#endif
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
            if (c.s_out) stv<VEC>(c.s_out + o, so);  // the fp32 copy: only for callers that read the layer's output tensor
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

// ------------------------------------------------------------------ pipelined variants (round 3)
// The kernels above fetch U steps, wait for ALL of them (`s_waitcnt vmcnt(0)`), compute U steps, and start over:
// with one wave per SIMD (65 k neurons at BASELINE configs[1] = 1024 waves on 1024 SIMDs) nothing covers the HBM
// round trip of each batch, and the nullable outputs put a scalar branch around every load and store.  These
// variants keep a register ring of the next D steps' inputs: every step issues ONE step's loads (D steps ahead,
// unconditional, clamped into range at the sequence end) and consumes the oldest slot, so the round trip sits behind
// D steps of arithmetic and stores; which outputs exist is a template parameter (no branch in the loop).  D is
// bounded by the wave's 6-bit vmcnt (in order, counts stores): with OPS vector-memory operations per step a load
// older than 63 / OPS steps is forced complete by any counted wait.  Same arithmetic, same order: bit-identical.
#ifndef CELL_PIPE
#define CELL_PIPE 1
#endif
typedef unsigned cell_u32x2 __attribute__((ext_vector_type(2)));
constexpr int pipe_depth(int vec, int ops) { return vec == 4 ? 8 : (63 / ops < 16 ? 63 / ops : 16); }

// one saved state (u or w) of VEC neurons as it comes off the wire: fp32, or bf16 words unpacked at the use
template <int VEC, bool S16> struct SavedVec;
template <int VEC> struct SavedVec<VEC, false> {
    float v[VEC];
    __device__ __forceinline__ void load(const float* base, size_t i) {
#if CELL_NT
        if constexpr (VEC == 4) {
            const f32x4 q = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + i));
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
            v[0] = __builtin_nontemporal_load(base + i);
        }
#else
        ldv<VEC>(v, base + i);
#endif
    }
    __device__ __forceinline__ void expand(float (&d)[VEC]) const {
#pragma unroll
        for (int e = 0; e < VEC; ++e) d[e] = v[e];
    }
};
template <> struct SavedVec<4, true> {
    unsigned long long raw;
    __device__ __forceinline__ void load(const float* base, size_t i) {
        raw = *reinterpret_cast<const unsigned long long*>(reinterpret_cast<const unsigned short*>(base) + i);
    }
    __device__ __forceinline__ void expand(float (&d)[4]) const {
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = bf16_to_f32((unsigned short)(raw >> (16 * e)));
    }
};
template <> struct SavedVec<1, true> {
    unsigned short raw;
    __device__ __forceinline__ void load(const float* base, size_t i) { raw = reinterpret_cast<const unsigned short*>(base)[i]; }
    __device__ __forceinline__ void expand(float (&d)[1]) const { d[0] = bf16_to_f32(raw); }
};

// SAVE: 0 = no saved states (eval), 1 = fp32, 2 = bf16.  SOUT / S16OUT: the fp32 spike tensor / the bf16 plane.
// DROP: dropout is on (a template parameter: a scalar branch around the mask's hash inside the loop makes hipcc's
// wait-count pass fall back to `s_waitcnt vmcnt(0)` at the join, once per trip — the ring would drain every D steps)
template <bool ADAPT, int VEC, int SAVE, bool SOUT, bool S16OUT, bool DROP>
__global__ __launch_bounds__(256) void cell_fwd_pipe_kernel(CellArgs c) {
    constexpr int D = pipe_depth(VEC, 1 + (SOUT ? 1 : 0) + (S16OUT ? 1 : 0) + (SAVE ? 1 + (ADAPT ? 1 : 0) : 0));
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
    constexpr bool drop = DROP;
    const uint64_t seed = drop ? resolve_seed(c.seed) : 0;

    const float* xrow = c.Wx + (size_t)b * T * H + h;
    auto fetch = [&](float (&x)[VEC], int t) __attribute__((always_inline)) {
        const int tc = min(t, T - 1);  // past the end: a step that is in range and never used
        ldv<VEC>(x, xrow + (size_t)(d ? (T - 1 - tc) : tc) * H);
    };
    auto step = [&](int t, const float (&x)[VEC]) __attribute__((always_inline)) {
        const int tt = d ? (T - 1 - t) : t;
        float so[VEC];
        const size_t o = ((size_t)b * T + tt) * HO + (size_t)d * H + h;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float xn = x[e];
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
        if (SOUT) stv<VEC>(c.s_out + o, so);
        if (S16OUT) {
            if constexpr (VEC == 4) {
                cell_u32x2 pk;
                pk.x = (so[0] != 0.0f ? 0x3F80u : 0u) | (so[1] != 0.0f ? 0x3F800000u : 0u);
                pk.y = (so[2] != 0.0f ? 0x3F80u : 0u) | (so[3] != 0.0f ? 0x3F800000u : 0u);
                *reinterpret_cast<cell_u32x2*>(c.s16_out + o) = pk;
            } else {
                c.s16_out[o] = so[0] != 0.0f ? (uint16_t)0x3F80 : (uint16_t)0;
            }
        }
        if (SAVE) {
            st_saved<VEC, true>(c.u_save, ((size_t)bp * T + t) * H + h, u, SAVE == 2, c.theta);
            if (ADAPT) st_saved<VEC, false>(c.w_save, ((size_t)bp * T + t) * H + h, w, SAVE == 2, c.theta);
        }
    };

    float ring[D][VEC];
#pragma unroll
    for (int j = 0; j < D; ++j) fetch(ring[j], j);
    // every prologue load is in before the loop is entered: hipcc's counted waits inside the loop are the merge of
    // this state and the back edge's — with the D fills still in flight here, slot 0 looks "14 operations old" on
    // every trip (`vmcnt(14)`, then 17, 19 ...) instead of a full ring old (`vmcnt(56)`)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    int t0 = 0;
    for (; t0 + D <= T; t0 += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            // consume, THEN refill the slot: issued ahead of the step's arithmetic the new value is live beside the
            // old one, the two cannot share a register, and hipcc's copies at the loop's back edge wait for every
            // load of the ring (`s_waitcnt vmcnt(4)` per trip: the pipeline drained every D steps)
            step(t0 + j, ring[j]);
            fetch(ring[j], t0 + j + D);
            // (the trip is one basic block: without this hipcc gathers the D refills at one end of it, and the first
            // slots are waited for a few operations after their issue)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int j = 0; j < D; ++j) {
        if (t0 + j >= T) break;
        step(t0 + j, ring[j]);
    }
    if (c.spike_count) {
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            if (cnt[e]) atomicAdd(c.spike_count + (size_t)d * H + h + e, cnt[e]);
    }
}

template <bool ADAPT, int VEC, bool S16, bool BN, bool DROP>
__global__ __launch_bounds__(256) void cell_bwd_pipe_kernel(CellArgs c) {
    constexpr int D = pipe_depth(VEC, 3 + (ADAPT ? 1 : 0) + (BN ? 1 : 0));
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
    float bn_mu[VEC], bn_is[VEC], acc_dy[VEC], acc_dyx[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        bn_mu[e] = BN ? c.bn_mean[h + e] : 0.f;
        bn_is[e] = BN ? c.bn_invstd[h + e] : 0.f;
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
    ld_saved<VEC>(u_t, c.u_save, ((size_t)bp * T + (T - 1)) * H + h, S16);
    // the initial states: read by cell step 0 (the last of the reverse pass) only; fetched here, off the loop
    float u0v[VEC], w0v[VEC], s0v[VEC];
    ldv<VEC>(u0v, c.u0 + (size_t)bp * H + h);
    ldv<VEC>(s0v, c.s0 + (size_t)bp * H + h);
    if (ADAPT) ldv<VEC>(w0v, c.w0 + (size_t)bp * H + h);
    constexpr bool drop = DROP;
    const uint64_t seed = drop ? resolve_seed(c.seed) : 0;

    struct Slot { float g[VEC]; float xr[BN ? VEC : 1]; SavedVec<VEC, S16> up; SavedVec<VEC, S16> wp; };
    const float* grow = c.g_out + (size_t)b * T * HO + (size_t)d * H + h;
    const float* xrow = BN ? c.bn_x + (size_t)b * T * H + h : nullptr;
    const size_t srow = (size_t)bp * T * H + h;
    auto fetch = [&](Slot& q, int t) __attribute__((always_inline)) {
        const int tc = max(t, 0);  // before the start: a step that is in range and never used
        const int tt = d ? (T - 1 - tc) : tc;
        ldv<VEC>(q.g, grow + (size_t)tt * HO);
        if constexpr (BN) ldv<VEC>(q.xr, xrow + (size_t)tt * H);
        const size_t i = srow + (size_t)max(tc - 1, 0) * H;  // cell step 0 reads row 0 (unused: u0 / w0 below)
        q.up.load(c.u_save, i);
        if (ADAPT) q.wp.load(c.w_save, i);
    };
    // FIRST: the step may be cell step 0 (tail of the reverse pass only — the main loop stops before it)
    auto step = [&](int t, const Slot& q, auto first_tag) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const int tt = d ? (T - 1 - t) : t;
        const size_t o = ((size_t)b * T + tt) * HO + (size_t)d * H + h;
        float up[VEC], wp[VEC], sp[VEC], dwx[VEC];
        q.up.expand(up);
        if (ADAPT) q.wp.expand(wp);
        const bool t0 = FIRST && t == 0;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            if (t0) { up[e] = u0v[e]; if (ADAPT) wp[e] = w0v[e]; }
            sp[e] = t0 ? s0v[e] : ((up[e] - c.theta) > 0.0f ? 1.0f : 0.0f);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float k = drop ? keep_scale(seed, o + e, c.p_drop, c.inv_keep) : 1.0f;
            const float gs = (q.g[e] + gr[e]) * k;
            float ds = gs - al[e] * du_n[e];
            if (ADAPT) ds = ds + pb[e] * dw_n[e];
            const float xs = u_t[e] - c.theta;
            float du = boxcar_gate(ds, xs) + al[e] * du_n[e];            // snns.py:33-35
            if (ADAPT) du = du + pa[e] * dw_n[e];
            dwx[e] = oma[e] * du;
            if (BN) {  // BatchNorm backward's column sums (dy = dWx, xhat = (x - mean) * invstd)
                acc_dy[e] += dwx[e];
                acc_dyx[e] += dwx[e] * ((q.xr[BN ? e : 0] - bn_mu[e]) * bn_is[e]);
            }
            const float qq = up[e] - sp[e];
            acc_al[e] += du * (qq - u_t[e]);  // d u_t / d alpha = (q - u_t)/(1-alpha); scaled at the end
            if (ADAPT) {
                const float dw = be[e] * dw_n[e] - dwx[e];
                acc_be[e] += dw * wp[e];
                acc_a[e] += dw * up[e];
                acc_b[e] += dw * sp[e];
                dw_n[e] = dw;
            }
            du_n[e] = du;
            // carried into the next step as a COPY made here: kept in the slot's own register the value outlives the
            // slot's refill, the refill needs another register, and hipcc shuffles the whole ring through `v_mov`s in
            // mid-loop — each of which waits for a load issued a few operations earlier
            asm volatile("v_mov_b32 %0, %1" : "=&v"(u_t[e]) : "v"(up[e]));
        }
        stv<VEC>(c.dWx + ((size_t)bp * T + tt) * H + h, dwx);
    };

    Slot ring[D];
#pragma unroll
    for (int j = 0; j < D; ++j) fetch(ring[j], T - 1 - j);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // see the forward kernel
    int i0 = 0;  // reverse index: step t = T - 1 - i
    for (; i0 + D <= T - 1; i0 += D) {  // steps t >= 1 only
#pragma unroll
        for (int j = 0; j < D; ++j) {  // consume, then refill (see the forward kernel)
            step(T - 1 - (i0 + j), ring[j], std::false_type{});
            fetch(ring[j], T - 1 - (i0 + j + D));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int j = 0; j < D; ++j) {
        if (i0 + j >= T) break;
        step(T - 1 - (i0 + j), ring[j], std::true_type{});
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
    if (BN) {
        stv<VEC>(ws + 4 * plane, acc_dy);
        stv<VEC>(ws + 5 * plane, acc_dyx);
    }
}

// ------------------------------------------------------------------ readout cell
// Readout layer (snns.py:815-825): u_t = alpha u_{t-1} + (1-alpha) x_t, out = sum_t softmax_c(u_t).
// One workgroup of 256 threads per batch row; a chunk of up to 256 time steps (the whole sequence for the
// reference's shapes) is staged in LDS, and the thread's meaning changes per phase so that nothing needs a
// cross-lane reduction:
//   thread = element : the chunk's projection, one coalesced sweep with every load in flight at once -> LDS
//   thread = class   : the linear recurrence over t (one FMA per step, in place: x_t -> u_t)
//   thread = time    : softmax over the classes of "its" time step, sequentially over the C values in LDS
//                      (rows are CS floats apart: odd stride, conflict-free for thread = time)
//   thread = class   : out_c += p[t][c] in time order (the same summation order as the reference's loop)
// History (256 x 250 x 35): class-per-lane throughout with two butterfly reductions per step 137 / 194 us
// (forward / backward); one wave per row with these phases on 128-step chunks 65 / 103 us, of which most was
// global-load round trips (8 loads in flight per lane) and then the 64-rows-per-wave softmax phase running on one
// SIMD of the CU.
constexpr int RO_NT = 256;           // threads per batch row (and the most time steps per chunk)
constexpr int RO_LPT = 36;           // elements per thread and load batch of the staging sweep
constexpr int RO_FWD_FLOATS = 15872; // LDS floats of the forward kernel (62 KiB)
constexpr int RO_BWD_FLOATS = 39168; // of the backward kernel: three chunk images (153 KiB, dynamic)
constexpr unsigned RO_OOB = 0xFFFFFFF0u;
__host__ __device__ inline int ro_row_stride(int C) { return (C & 1) ? C + 2 : C + 1; }  // odd, > C: column C is a dump slot
inline int ro_chunk_steps(int T, int C, int lds_floats) {
    const int fit = lds_floats / ro_row_stride(C);
    return T < RO_NT ? (T < fit ? T : fit) : (RO_NT < fit ? RO_NT : fit);
}
// workgroup barrier for the LDS hand-offs between the phases: LDS-only fences (no vmcnt(0): the u_save / dWx
// stores of a phase stay in flight across it, which __syncthreads() would wait for)
__device__ __forceinline__ void ro_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ro_row(const float* base, size_t row_elems, size_t row) {
    // a null base gives an empty resource: loads return 0, stores are dropped
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base ? base + row * row_elems : nullptr), 0,
                                             base ? (int)(row_elems * sizeof(float)) : 0, 0x00020000);
}
// bounds-checked fp32 accesses (the builtins move raw 32-bit words: bit casts, not conversions)
__device__ __forceinline__ float ro_load(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ void ro_store(float v, __amdgpu_buffer_rsrc_t r, unsigned off) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, off, 0, 0);
}
// (time, class) of the linear element index idx = first + 256 k of a chunk, advanced without divisions
struct RoCursor {
    int t, c, dq, dr, C;
    __device__ RoCursor(int first, int C_) : t(first / C_), c(first % C_), dq(RO_NT / C_), dr(RO_NT % C_), C(C_) {}
    __device__ __forceinline__ void next() {
        t += dq; c += dr;
        if (c >= C) { c -= C; ++t; }
    }
};
__device__ __forceinline__ float ro_softmax_row(float* row, int C) {  // in place; returns nothing useful
    float m = row[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, row[c]);
    float den = 0.f;
    for (int c = 0; c < C; ++c) {
        const float e = expf(row[c] - m);
        row[c] = e;
        den += e;
    }
    return den;
}

__global__ __launch_bounds__(RO_NT) void readout_fwd_kernel(int B, int T, int C, int rows, const float* __restrict__ Wx,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ alpha,
                                                            const float* __restrict__ u0, float* __restrict__ out,
                                                            float* __restrict__ u_save) {
    __shared__ float us[RO_FWD_FLOATS];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int CS = ro_row_stride(C);
    const bool act = tid < C;
    const bool wave_has_class = (tid & ~63) < C;  // wave-uniform: waves past the classes skip the class phases
    const int cc = act ? tid : C;
    const int cp = act ? tid : 0;
    const float al = clampf(alpha[cp], SP_ALPHA_LO, SP_ALPHA_HI), oma = 1.0f - al;
    const bool has_bn = scale != nullptr;
    const float sc = has_bn ? scale[cp] : 1.0f, sh = has_bn ? shift[cp] : 0.0f;
    float u = u0[(size_t)b * C + cp], acc = 0.f;
    const __amdgpu_buffer_rsrc_t rx = ro_row(Wx, (size_t)T * C, b), ru = ro_row(u_save, (size_t)T * C, b);
    for (int c0 = 0; c0 < T; c0 += rows) {
        const int len = min(rows, T - c0), n = len * C;
        // thread = element: the chunk's projection -> LDS
        {
            RoCursor cur(tid, C);
            for (int base = 0; base < n; base += RO_NT * RO_LPT) {
                float v[RO_LPT];
#pragma unroll
                for (int k = 0; k < RO_LPT; ++k) {
                    const int idx = base + tid + RO_NT * k;
                    v[k] = ro_load(rx, idx < n ? (unsigned)(((size_t)c0 * C + idx) * sizeof(float)) : RO_OOB);
                }
#pragma unroll
                for (int k = 0; k < RO_LPT; ++k) {
                    const int idx = base + tid + RO_NT * k;
                    us[idx < n ? cur.t * CS + cur.c : C] = v[k];
                    cur.next();
                }
            }
        }
        ro_barrier();
        // thread = class: recurrence, in place (x_t -> u_t).  Whole groups of 8 steps run without predicates on
        // running LDS / row offsets (the step is a handful of instructions: its issue time is the phase);
        // the ragged tail goes step by step.
        if (wave_has_class) {
            float* p = us + cc;                                                      // -> us[t][cc]
            unsigned go = act ? (unsigned)(((size_t)c0 * C + tid) * sizeof(float)) : RO_OOB;  // -> u_save[c0 + t][tid]
            const unsigned gstep = act ? (unsigned)(C * sizeof(float)) : 0u;
            int t0 = 0;
            for (; t0 + 8 <= len; t0 += 8) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = p[j * CS];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xn = has_bn ? bn_affine(x[j], sc, sh) : x[j];
                    u = al * u + oma * xn;                               // snns.py:822
                    p[j * CS] = u;
                    ro_store(u, ru, go + j * gstep);
                }
                p += 8 * CS;
                go += 8 * gstep;
            }
            for (; t0 < len; ++t0) {
                const float x = *p;
                const float xn = has_bn ? bn_affine(x, sc, sh) : x;
                u = al * u + oma * xn;
                *p = u;
                ro_store(u, ru, go);
                p += CS;
                go += gstep;
            }
        }
        ro_barrier();
        // thread = time: softmax over classes, in place
        for (int tl = tid; tl < len; tl += RO_NT) {
            float* row = us + tl * CS;
            const float den = ro_softmax_row(row, C);
            for (int c = 0; c < C; ++c) row[c] = row[c] / den;
        }
        ro_barrier();
        // thread = class: out += softmax(u_t) in time order             // snns.py:823
        if (wave_has_class) {
            const float* p = us + cc;
            int t0 = 0;
            for (; t0 + 8 <= len; t0 += 8) {
                float pv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) pv[j] = p[j * CS];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc = acc + pv[j];
                p += 8 * CS;
            }
            for (; t0 < len; ++t0, p += CS) acc = acc + *p;
        }
        ro_barrier();
    }
    if (act) out[(size_t)b * C + cc] = acc;
}

// Backward of the above.  With p_t = softmax(u_t) and g = dL/dout:
//   e_t = p_t * (g - <p_t, g>)            (thread = time; independent over t)
//   du_t = alpha du_{t+1} + e_t,  dWx_t = (1-alpha) du_t,  dalpha += du_t (u_{t-1} - u_t) / (1-alpha)
//                                          (thread = class; reverse time)
// Three chunk images in LDS: u (overwritten by e), u_{t-1} - u_t, and BatchNorm's xhat of the raw projection.
extern __shared__ __attribute__((aligned(16))) float ro_dyn_lds[];
__global__ __launch_bounds__(RO_NT) void readout_bwd_kernel(int B, int T, int C, int rows, const float* __restrict__ g_out,
                                                            const float* __restrict__ u_save,
                                                            const float* __restrict__ alpha,
                                                            const float* __restrict__ u0, float* __restrict__ dWx,
                                                            float* __restrict__ dalpha_ws,
                                                            const float* __restrict__ bn_x,
                                                            const float* __restrict__ bn_mean,
                                                            const float* __restrict__ bn_invstd) {
    __shared__ float gs[RO_NT];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int CS = ro_row_stride(C);
    float* const ue = ro_dyn_lds;            // [rows][CS] u_t, then e_t
    float* const dl = ue + rows * CS;        // [rows][CS] u_{t-1} - u_t
    float* const xh = dl + rows * CS;        // [rows][CS] xhat_t
    const bool act = tid < C;
    const bool wave_has_class = (tid & ~63) < C;
    const int cc = act ? tid : C;
    const int cp = act ? tid : 0;
    const float al = clampf(alpha[cp], SP_ALPHA_LO, SP_ALPHA_HI), oma = 1.0f - al;
    gs[tid] = act ? g_out[(size_t)b * C + cp] : 0.f;
    float du = 0.f, acc = 0.f;
    // BatchNorm backward's column sums folded in (nullable): sum_t dWx and sum_t dWx*xhat per (row, class)
    const bool bn = bn_x != nullptr;
    float acc_dy = 0.f, acc_dyx = 0.f;
    const __amdgpu_buffer_rsrc_t ru = ro_row(u_save, (size_t)T * C, b), rb = ro_row(bn_x, (size_t)T * C, b),
                                 rd = ro_row(dWx, (size_t)T * C, b), r0 = ro_row(u0, (size_t)C, b);
    const int nchunk = (T + rows - 1) / rows;
    for (int ch = nchunk - 1; ch >= 0; --ch) {
        const int c0 = ch * rows, len = min(rows, T - c0), n = len * C;
        // thread = element: u_t, u_{t-1} - u_t (u_{-1} = u0) and xhat_t of the chunk -> LDS
        {
            RoCursor cur(tid, C);
            constexpr int LPT = RO_LPT / 2;
            for (int base = 0; base < n; base += RO_NT * LPT) {
                float vu[LPT], vx[LPT];
                unsigned vp[LPT], v0[LPT];
#pragma unroll
                for (int k = 0; k < LPT; ++k) {
                    const int idx = base + tid + RO_NT * k;
                    const bool in = idx < n;
                    const size_t e = (size_t)c0 * C + idx;           // element of the batch row
                    const bool first = in && e < (size_t)C;          // time step 0: the step before is u0
                    vu[k] = ro_load(ru, in ? (unsigned)(e * sizeof(float)) : RO_OOB);
                    vx[k] = ro_load(rb, in ? (unsigned)(e * sizeof(float)) : RO_OOB);
                    vp[k] = __builtin_amdgcn_raw_buffer_load_b32(ru, (in && !first) ? (unsigned)((e - C) * sizeof(float)) : RO_OOB, 0, 0);
                    v0[k] = __builtin_amdgcn_raw_buffer_load_b32(r0, first ? (unsigned)(e * sizeof(float)) : RO_OOB, 0, 0);
                }
#pragma unroll
                for (int k = 0; k < LPT; ++k) {
                    const int idx = base + tid + RO_NT * k;
                    const int at = idx < n ? cur.t * CS + cur.c : C;
                    const int cl = idx < n ? cur.c : 0;
                    ue[at] = vu[k];
                    dl[at] = __uint_as_float(vp[k] | v0[k]) - vu[k];   // exactly one of the two loads was in range
                    xh[at] = bn ? (vx[k] - bn_mean[cl]) * bn_invstd[cl] : 0.f;
                    cur.next();
                }
            }
        }
        ro_barrier();
        // thread = time: e_t in place
        for (int tl = tid; tl < len; tl += RO_NT) {
            float* row = ue + tl * CS;
            const float den = ro_softmax_row(row, C);
            float dot = 0.f;
            for (int c = 0; c < C; ++c) {
                const float pc = row[c] / den;
                row[c] = pc;
                dot += pc * gs[c];
            }
            for (int c = 0; c < C; ++c) row[c] = row[c] * (gs[c] - dot);
        }
        ro_barrier();
        // thread = class: reverse recurrence (groups of 8 steps without predicates, then the ragged rest)
        if (wave_has_class) {
            int at = (len - 1) * CS + cc;                                            // -> [t][cc] of the three images
            unsigned go = act ? (unsigned)(((size_t)(c0 + len - 1) * C + tid) * sizeof(float)) : RO_OOB;  // -> dWx[c0 + t][tid]
            const unsigned gstep = act ? (unsigned)(C * sizeof(float)) : 0u;
            int left = len;
            for (; left >= 8; left -= 8) {
                float ev[8], dv[8], xv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { ev[j] = ue[at - j * CS]; dv[j] = dl[at - j * CS]; xv[j] = xh[at - j * CS]; }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    du = al * du + ev[j];
                    const float dwx = oma * du;
                    ro_store(dwx, rd, go - j * gstep);
                    acc = acc + du * dv[j];
                    acc_dy = acc_dy + dwx;
                    acc_dyx = acc_dyx + dwx * xv[j];
                }
                at -= 8 * CS;
                go -= 8 * gstep;
            }
            for (; left > 0; --left) {
                du = al * du + ue[at];
                const float dwx = oma * du;
                ro_store(dwx, rd, go);
                acc = acc + du * dl[at];
                acc_dy = acc_dy + dwx;
                acc_dyx = acc_dyx + dwx * xh[at];
                at -= CS;
                go -= gstep;
            }
        }
        ro_barrier();
    }
    if (act) {
        dalpha_ws[(size_t)b * C + cc] = acc / oma;
        if (bn) {
            dalpha_ws[((size_t)B + b) * C + cc] = acc_dy;
            dalpha_ws[((size_t)2 * B + b) * C + cc] = acc_dyx;
        }
    }
}

template <bool ADAPT, int VEC, bool DROP>
void launch_cell_pipe(bool bwd, const CellArgs& c, unsigned blocks, hipStream_t st) {
    if (bwd) {
        auto go = [&](auto s16, auto bn) {
            hipLaunchKernelGGL((cell_bwd_pipe_kernel<ADAPT, VEC, decltype(s16)::value, decltype(bn)::value, DROP>),
                               dim3(blocks), dim3(256), 0, st, c);
        };
        if (c.save16) { if (c.bn_x) go(std::true_type{}, std::true_type{}); else go(std::true_type{}, std::false_type{}); }
        else          { if (c.bn_x) go(std::false_type{}, std::true_type{}); else go(std::false_type{}, std::false_type{}); }
        return;
    }
    auto go = [&](auto save, auto sout, auto s16out) {
        hipLaunchKernelGGL((cell_fwd_pipe_kernel<ADAPT, VEC, decltype(save)::value, decltype(sout)::value,
                                                 decltype(s16out)::value, DROP>),
                           dim3(blocks), dim3(256), 0, st, c);
    };
    auto outs = [&](auto save) {
        if (c.s_out && c.s16_out) go(save, std::true_type{}, std::true_type{});
        else if (c.s_out) go(save, std::true_type{}, std::false_type{});
        else go(save, std::false_type{}, std::true_type{});
    };
    if (!c.u_save) outs(std::integral_constant<int, 0>{});
    else if (c.save16) outs(std::integral_constant<int, 2>{});
    else outs(std::integral_constant<int, 1>{});
}

template <bool ADAPT>
int launch_cell(bool bwd, CellArgs& c, hipStream_t st) {
    const long long work = (long long)c.B * c.dirs * c.H;
    const bool vec_ok = (c.H % 4 == 0);
    // VEC=4 only when it still leaves >= 8 waves per CU; small problems favour more threads
    const bool vec4 = vec_ok && work >= (long long)256 * 8 * 64 * 4;
    const long long threads = vec4 ? work / 4 : work;
    const unsigned blocks = (unsigned)((threads + 255) / 256);
#if CELL_PIPE
    const bool drop = c.p_drop > 0.0f;
    if (vec4) { if (drop) launch_cell_pipe<ADAPT, 4, true>(bwd, c, blocks, st); else launch_cell_pipe<ADAPT, 4, false>(bwd, c, blocks, st); }
    else      { if (drop) launch_cell_pipe<ADAPT, 1, true>(bwd, c, blocks, st); else launch_cell_pipe<ADAPT, 1, false>(bwd, c, blocks, st); }
#else
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
#endif
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
    if (B <= 0 || T <= 0 || H <= 0 || (dirs != 1 && dirs != 2) || !Wx || !alpha || !u0 || !s0 || (!s_out && !s16_out))
        return SPARCH_EINVAL;
    if (adapt && (!beta || !a || !b || !w0)) return SPARCH_EINVAL;
    if (adapt && ((u_save == nullptr) != (w_save == nullptr))) return SPARCH_EINVAL;  // saved together or not at all
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
    hipLaunchKernelGGL(readout_fwd_kernel, dim3(B), dim3(RO_NT), 0, (hipStream_t)stream, B, T, C,
                       ro_chunk_steps(T, C, RO_FWD_FLOATS), Wx, scale, shift, alpha, u0, out, u_save);
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
    const int rows = ro_chunk_steps(T, C, RO_BWD_FLOATS / 3);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(readout_bwd_kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       (int)(RO_BWD_FLOATS * sizeof(float)));
    if (attr != hipSuccess) { sparch_note_hip_error((int)attr); return SPARCH_ELAUNCH; }
    hipLaunchKernelGGL(readout_bwd_kernel, dim3(B), dim3(RO_NT), (size_t)3 * rows * ro_row_stride(C) * sizeof(float),
                       (hipStream_t)stream, B, T, C, rows, g_out, u_save, alpha, u0, dWx, dalpha_ws, bn_x, bn_mean,
                       bn_invstd);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

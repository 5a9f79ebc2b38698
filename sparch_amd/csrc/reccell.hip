// G3/G4 (recurrent kinds): RLIF / RadLIF cells, forward and reverse-time backward, with
// the time loop inside a persistent kernel and the recurrent product on MFMA.
//
// Replaces _rlif_cell (snns.py:554-578) and _radlif_cell (696-727) — per step
//     w = beta*w + a*u + b*s ;  u = alpha*(u-s) + (1-alpha)*(Wx_t + s@V - w) ;  s = H(u-theta)
// with V = V.weight, diagonal zeroed (566/712), `s @ V` un-transposed — and their autograd
// replay (reverse recurrences in SURVEY.md §8a), including du_{t+1} @ V^T per step.
//
// Decomposition (gfx950, 256 CUs in 8 XCDs):
//   * a workgroup (256 threads, one wave per SIMD) owns a 32-row x 32-column tile of the
//     (B', H) state for the whole sequence; membrane/adaptation state lives in registers;
//   * its 1024x32 slice of V (forward) / V^T (backward) is resident in VGPRs for the whole
//     launch as MFMA B-operands (128 VGPRs per wave at H=1024); the four waves split K and
//     the partial 32x32 tiles meet in LDS (fixed order => reproducible);
//   * the 32 workgroups of one batch tile exchange the step's output every step:
//       forward  — spikes, bit-packed, as 8-byte {tag = t+1, 32 spike bits} granules written
//                  with one agent-scope (sc1, write-through) store each and polled with sc1
//                  loads: the data is the flag, no fence (MI355X guide, G16 form R2);
//       backward — the 32x32 fp32 tile of dWx_t, stored straight into the dWx output tensor
//                  with 16-byte sc1 stores, drained (vmcnt(0)) + workgroup barrier, then one
//                  sc1 flag store; consumers poll the flag and read the tile with sc1 loads
//                  (G16 form R1).  No slot is ever reused inside a call (one slot per time
//                  step), so there is no back-pressure protocol;
//   * block -> tile mapping keeps a batch tile's workgroups at equal blockIdx % n_row_tiles,
//     i.e. on one XCD under round-robin dispatch.  That is a speed choice only: every
//     hand-off is agent-scope and placement-independent.  Every spin is bounded by a
//     wall-clock timeout that raises *status and unwinds the launch.
//   * steps_per_launch = T gives one persistent launch; = 1 degenerates to one launch per
//     time step, where every wait is already satisfied at launch (safe fallback, and the
//     path for shapes whose grid cannot be co-resident).
// MFMA: v_mfma_f32_32x32x2_f32, exact fp32 products; A = spikes (0/1) or dWx, B = V slice.
#include "common.h"

namespace {

typedef unsigned long long u64;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

constexpr int RT = 32;       // rows per batch tile
constexpr int CT = 32;       // columns per workgroup (= one k-group of its consumers)
constexpr int RED_LD = 33;   // padded row of the cross-wave reduction tiles
constexpr u64 TIMEOUT_TICKS = 200000000ull;  // 2 s of s_memrealtime (100 MHz)

struct RecArgs {
    int B, dirs, T, H, Bp;
    int n_ct, nkg, n_rt_total;
    int rt_base, n_rt_launch;
    int t_begin, t_end;
    const float* Wx; const float* scale; const float* shift;
    const float* alpha; const float* beta; const float* a; const float* b;
    const float* vpack; const float* rec0;
    const float* u0; const float* w0; const float* s0;
    float theta, p_drop, inv_keep; uint64_t seed;
    float* s_out; float* u_save; float* w_save; uint32_t* spike_count;
    // backward
    const float* g_out; const float* g_rate; float g_rate_scale;
    float* dWx; float* s_prev; float* dparam_ws;
    // hand-off
    u64* chan; unsigned* flags; unsigned* status;
};

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

__device__ __forceinline__ void raise_timeout(unsigned* status, int* abort_slot) {
    __hip_atomic_store((gu32*)status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *(volatile int*)abort_slot = 1;
}

// ------------------------------------------------------------------------------ forward
template <bool ADAPT, int KGW>
__global__ __launch_bounds__(256, 1) void rec_fwd_kernel(RecArgs a) {
    __shared__ __attribute__((aligned(16))) float red[2][4][RT * RED_LD];
    __shared__ int abort_flag[2];
    __shared__ unsigned cnt_lds[2][CT];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hh = lane >> 5;
    const int rt = a.rt_base + (int)(blockIdx.x % a.n_rt_launch);
    const int ct = (int)(blockIdx.x / a.n_rt_launch);
    const int T = a.T, H = a.H, HO = a.H * a.dirs;

    // element ownership for the pointwise update: row r, 4 columns
    const int r = tid >> 3, cq = tid & 7;
    const int bp = rt * RT + r, col = ct * CT + cq * 4;
    const bool valid = bp < a.Bp && col < H;
    const int bpc = min(bp, a.Bp - 1), colc = min(col, H - 4);
    const int d = bpc / a.B, b = bpc - d * a.B;

    // V slice -> registers (B operand of the 32x32x2 MFMA: lane = (column li, k-half hh))
    float vreg[KGW][16];
#pragma unroll
    for (int kk = 0; kk < KGW; ++kk) {
        const int kg = wave + 4 * kk;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = ld4(a.vpack + ((((size_t)ct * a.nkg + kg) * 4 + q) * 64 + lane) * 4);
            vreg[kk][4 * q + 0] = v.x; vreg[kk][4 * q + 1] = v.y;
            vreg[kk][4 * q + 2] = v.z; vreg[kk][4 * q + 3] = v.w;
        }
    }

    float al[4], oma[4], be[4], pa[4], pb[4], sc[4], sh[4], u[4], w[4], s[4];
    uint32_t cnt[4] = {0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        al[e] = clampf(a.alpha[colc + e], SP_ALPHA_LO, SP_ALPHA_HI);
        oma[e] = 1.0f - al[e];
        be[e] = ADAPT ? clampf(a.beta[colc + e], SP_BETA_LO, SP_BETA_HI) : 0.f;
        pa[e] = ADAPT ? clampf(a.a[colc + e], SP_A_LO, SP_A_HI) : 0.f;
        pb[e] = ADAPT ? clampf(a.b[colc + e], SP_B_LO, SP_B_HI) : 0.f;
        sc[e] = a.scale ? a.scale[colc + e] : 1.0f;
        sh[e] = a.scale ? a.shift[colc + e] : 0.0f;
        w[e] = 0.f;
    }
    {
        f32x4 v;
        if (a.t_begin == 0) {
            v = ld4(a.u0 + (size_t)bpc * H + colc); u[0] = v.x; u[1] = v.y; u[2] = v.z; u[3] = v.w;
            v = ld4(a.s0 + (size_t)bpc * H + colc); s[0] = v.x; s[1] = v.y; s[2] = v.z; s[3] = v.w;
            if (ADAPT) { v = ld4(a.w0 + (size_t)bpc * H + colc); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
        } else {
            const size_t o = ((size_t)bpc * T + (a.t_begin - 1)) * H + colc;
            v = ld4(a.u_save + o); u[0] = v.x; u[1] = v.y; u[2] = v.z; u[3] = v.w;
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] = (u[e] - a.theta) > 0.0f ? 1.0f : 0.0f;
            if (ADAPT) { v = ld4(a.w_save + o); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
        }
    }
    if (tid < 2) abort_flag[tid] = 0;
    if (tid < 2 * CT) cnt_lds[tid / CT][tid % CT] = 0;
    __syncthreads();

    const bool has_norm = a.scale != nullptr;
    const bool drop = a.p_drop > 0.0f;
    auto wx_ptr = [&](int t) {
        const int tt = d ? (T - 1 - t) : t;
        return a.Wx + ((size_t)b * T + tt) * H + colc;
    };
    f32x4 x_next = ld4(wx_ptr(a.t_begin));

    for (int t = a.t_begin; t < a.t_end; ++t) {
        const f32x4 xv = x_next;
        if (t + 1 < a.t_end) x_next = ld4(wx_ptr(t + 1));
        float rec[4] = {0.f, 0.f, 0.f, 0.f};

        if (t == 0) {
            const f32x4 v = ld4(a.rec0 + (size_t)bpc * H + colc);
            rec[0] = v.x; rec[1] = v.y; rec[2] = v.z; rec[3] = v.w;
        } else {
            // ---- gather the 32 x H spike bits of step t-1 (tag == t) for this wave's k-groups
            const gu64* base = (const gu64*)a.chan + ((size_t)(t - 1) * a.n_rt_total + rt) * a.n_ct * 32;
            u64 gran[KGW];
            const u64 t_start = __builtin_amdgcn_s_memrealtime();
            for (unsigned spins = 0;; ++spins) {
                bool ok = true;
#pragma unroll
                for (int kk = 0; kk < KGW; ++kk) {
                    const int kg = wave + 4 * kk;
                    if (kg < a.n_ct) {
                        gran[kk] = __hip_atomic_load(base + (size_t)kg * 32 + li, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT);
                        ok = ok && ((unsigned)(gran[kk] >> 32) == (unsigned)t);
                    } else {
                        gran[kk] = 0;
                    }
                }
                if (__all(ok)) break;
                if ((spins & 63u) == 63u &&
                    __builtin_amdgcn_s_memrealtime() - t_start > TIMEOUT_TICKS) {
                    raise_timeout(a.status, &abort_flag[t & 1]);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int kk = 0; kk < KGW; ++kk) {
                const int kg = wave + 4 * kk;
                if (kg < a.n_ct) {
                    const unsigned bits = ((unsigned)gran[kk]) >> (16 * hh);
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float av = ((bits >> i) & 1u) ? 1.0f : 0.0f;
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, vreg[kk][i], acc, 0, 0, 0);
                    }
                }
            }
            float* rd = red[t & 1][wave];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
                rd[row * RED_LD + li] = acc[i];
            }
        }
        __syncthreads();
        if (*(volatile int*)&abort_flag[t & 1]) break;
        if (t > 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int o = r * RED_LD + cq * 4 + e;
                rec[e] = ((red[t & 1][0][o] + red[t & 1][1][o]) + red[t & 1][2][o]) + red[t & 1][3][o];
            }
        }

        // ---- pointwise membrane update for this thread's 4 neurons
        const int tt = d ? (T - 1 - t) : t;
        const size_t o_out = ((size_t)b * T + tt) * HO + (size_t)d * H + colc;
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
        f32x4 so, uo, wo;
        unsigned nib = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float xn = xs[e];
            if (has_norm) xn = xn * sc[e] + sh[e];
            float drive = xn + rec[e];                                      // snns.py:572 / 720
            if (ADAPT) {
                w[e] = (be[e] * w[e] + pa[e] * u[e]) + pb[e] * s[e];        // snns.py:718
                drive = drive - w[e];
            }
            u[e] = al[e] * (u[e] - s[e]) + oma[e] * drive;                  // snns.py:572 / 719
            s[e] = (u[e] - a.theta) > 0.0f ? 1.0f : 0.0f;                   // snns.py:29
            const float k = drop ? keep_scale(a.seed, o_out + e, a.p_drop, a.inv_keep) : 1.0f;
            so[e] = s[e] * k;
            uo[e] = u[e];
            wo[e] = w[e];
            if (valid) {
                cnt[e] += (so[e] != 0.0f) ? 1u : 0u;
                nib |= (s[e] != 0.0f ? 1u : 0u) << e;
            }
        }
        if (valid) {
            st4(a.s_out + o_out, so);
            st4(a.u_save + ((size_t)bp * T + t) * H + col, uo);
            if (ADAPT) st4(a.w_save + ((size_t)bp * T + t) * H + col, wo);
        }
        // ---- publish this tile's spikes: one tagged granule per row
        unsigned word = nib << (cq * 4);
        word |= __shfl_xor(word, 1);
        word |= __shfl_xor(word, 2);
        word |= __shfl_xor(word, 4);
        if (cq == 0 && t + 1 < T) {
            gu64* slot = (gu64*)a.chan + (((size_t)t * a.n_rt_total + rt) * a.n_ct + ct) * 32 + r;
            __hip_atomic_store(slot, ((u64)(unsigned)(t + 1) << 32) | (u64)word, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    // ---- spike counts (post-dropout) -> one integer atomic per (direction, column) per workgroup
    if (a.spike_count) {
        if (valid) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (cnt[e]) atomicAdd(&cnt_lds[d][cq * 4 + e], cnt[e]);
        }
        __syncthreads();
        if (tid < 2 * CT) {
            const int dd = tid / CT, c = tid % CT;
            const unsigned v = cnt_lds[dd][c];
            if (v && dd < a.dirs && ct * CT + c < H) atomicAdd(a.spike_count + (size_t)dd * H + ct * CT + c, v);
        }
    }
}

// ------------------------------------------------------------------------------ backward
template <bool ADAPT, int KGW>
__global__ __launch_bounds__(256, 1) void rec_bwd_kernel(RecArgs a) {
    __shared__ __attribute__((aligned(16))) float red[2][4][RT * RED_LD];
    __shared__ int abort_flag[2];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hh = lane >> 5;
    const int rt = a.rt_base + (int)(blockIdx.x % a.n_rt_launch);
    const int ct = (int)(blockIdx.x / a.n_rt_launch);
    const int T = a.T, H = a.H, HO = a.H * a.dirs;

    const int r = tid >> 3, cq = tid & 7;
    const int bp = rt * RT + r, col = ct * CT + cq * 4;
    const bool valid = bp < a.Bp && col < H;
    const int bpc = min(bp, a.Bp - 1), colc = min(col, H - 4);
    const int d = bpc / a.B, b = bpc - d * a.B;

    // A-operand row of this lane (row li of the batch tile), clamped into the tensor
    const int arow = min(rt * RT + li, a.Bp - 1);
    const int ad = arow / a.B;

    // V^T slice -> registers
    float vreg[KGW][16];
#pragma unroll
    for (int kk = 0; kk < KGW; ++kk) {
        const int kg = wave + 4 * kk;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = ld4(a.vpack + ((((size_t)ct * a.nkg + kg) * 4 + q) * 64 + lane) * 4);
            vreg[kk][4 * q + 0] = v.x; vreg[kk][4 * q + 1] = v.y;
            vreg[kk][4 * q + 2] = v.z; vreg[kk][4 * q + 3] = v.w;
        }
    }

    float al[4], oma[4], ioma[4], be[4], pa[4], pb[4], gr[4];
    float du_n[4], dw_n[4], u_t[4], acc_al[4], acc_be[4], acc_a[4], acc_b[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        al[e] = clampf(a.alpha[colc + e], SP_ALPHA_LO, SP_ALPHA_HI);
        oma[e] = 1.0f - al[e];
        ioma[e] = 1.0f / oma[e];
        be[e] = ADAPT ? clampf(a.beta[colc + e], SP_BETA_LO, SP_BETA_HI) : 0.f;
        pa[e] = ADAPT ? clampf(a.a[colc + e], SP_A_LO, SP_A_HI) : 0.f;
        pb[e] = ADAPT ? clampf(a.b[colc + e], SP_B_LO, SP_B_HI) : 0.f;
        gr[e] = a.g_rate ? a.g_rate[(size_t)d * H + colc + e] * a.g_rate_scale : 0.0f;
        du_n[e] = dw_n[e] = 0.f;
        acc_al[e] = acc_be[e] = acc_a[e] = acc_b[e] = 0.f;
    }
    const size_t plane = (size_t)a.Bp * H;
    float* ws = a.dparam_ws + (size_t)bpc * H + colc;
    if (a.t_end < T) {  // resume a chunked pass: carried state + partial sums
        f32x4 v;
        v = ld4(ws); acc_al[0] = v.x; acc_al[1] = v.y; acc_al[2] = v.z; acc_al[3] = v.w;
        v = ld4(ws + 4 * plane); du_n[0] = v.x; du_n[1] = v.y; du_n[2] = v.z; du_n[3] = v.w;
        if (ADAPT) {
            v = ld4(ws + plane); acc_be[0] = v.x; acc_be[1] = v.y; acc_be[2] = v.z; acc_be[3] = v.w;
            v = ld4(ws + 2 * plane); acc_a[0] = v.x; acc_a[1] = v.y; acc_a[2] = v.z; acc_a[3] = v.w;
            v = ld4(ws + 3 * plane); acc_b[0] = v.x; acc_b[1] = v.y; acc_b[2] = v.z; acc_b[3] = v.w;
            v = ld4(ws + 5 * plane); dw_n[0] = v.x; dw_n[1] = v.y; dw_n[2] = v.z; dw_n[3] = v.w;
        }
    }
    {
        const f32x4 v = ld4(a.u_save + ((size_t)bpc * T + (a.t_end - 1)) * H + colc);
        u_t[0] = v.x; u_t[1] = v.y; u_t[2] = v.z; u_t[3] = v.w;
    }
    if (tid < 2) abort_flag[tid] = 0;
    __syncthreads();

    // the dWx rows of this batch tile as a buffer resource (hand-off loads/stores carry sc1)
    const int rows_here = min(RT, a.Bp - rt * RT);
    float* tile_base = a.dWx + (size_t)rt * RT * T * H;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        tile_base, 0, (int)((size_t)rows_here * T * H * sizeof(float)), 0x00020000);

    const bool drop = a.p_drop > 0.0f;
    auto load_step = [&](int t, f32x4& g, f32x4& up, f32x4& wp) {
        const int tt = d ? (T - 1 - t) : t;
        g = ld4(a.g_out + ((size_t)b * T + tt) * HO + (size_t)d * H + colc);
        if (t > 0) {
            up = ld4(a.u_save + ((size_t)bpc * T + (t - 1)) * H + colc);
            if (ADAPT) wp = ld4(a.w_save + ((size_t)bpc * T + (t - 1)) * H + colc);
        } else {
            up = ld4(a.u0 + (size_t)bpc * H + colc);
            if (ADAPT) wp = ld4(a.w0 + (size_t)bpc * H + colc);
        }
    };
    f32x4 g_nx, up_nx, wp_nx = {0.f, 0.f, 0.f, 0.f};
    load_step(a.t_end - 1, g_nx, up_nx, wp_nx);

    for (int t = a.t_end - 1; t >= a.t_begin; --t) {
        const f32x4 gv = g_nx, upv = up_nx, wpv = wp_nx;
        if (t - 1 >= a.t_begin) load_step(t - 1, g_nx, up_nx, wp_nx);
        float rec[4] = {0.f, 0.f, 0.f, 0.f};
        const int par = t & 1;

        if (t + 1 < T) {
            // ---- wait for dWx_{t+1} tiles of this wave's producers, then read them (sc1)
            const gu32* fl = (const gu32*)a.flags + ((size_t)(t + 1) * a.n_rt_total + rt) * a.n_ct;
            const u64 t_start = __builtin_amdgcn_s_memrealtime();
            for (unsigned spins = 0;; ++spins) {
                bool ok = true;
#pragma unroll
                for (int kk = 0; kk < KGW; ++kk) {
                    const int kg = wave + 4 * kk;
                    if (kg < a.n_ct)
                        ok = ok && (__hip_atomic_load(fl + kg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u);
                }
                if (__all(ok)) break;
                if ((spins & 63u) == 63u &&
                    __builtin_amdgcn_s_memrealtime() - t_start > TIMEOUT_TICKS) {
                    raise_timeout(a.status, &abort_flag[par]);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            const int att = ad ? (T - 1 - (t + 1)) : (t + 1);
            const unsigned row_off = (unsigned)(((size_t)(arow - rt * RT) * T + att) * H);
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int kk = 0; kk < KGW; ++kk) {
                const int kg = wave + 4 * kk;
                if (kg < a.n_ct) {
                    u32x4 av[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int kcol = kg * 32 + 16 * hh + 4 * q;
                        const unsigned off = (row_off + (unsigned)min(kcol, H - 4)) * 4u;
                        av[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 16 /* sc1 */);
                        if (kcol >= H) av[q] = u32x4{0u, 0u, 0u, 0u};
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(av[q][e]),
                                                                       vreg[kk][4 * q + e], acc, 0, 0, 0);
                }
            }
            float* rd = red[par][wave];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
                rd[row * RED_LD + li] = acc[i];
            }
        }
        __syncthreads();
        if (*(volatile int*)&abort_flag[par]) break;
        if (t + 1 < T) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int o = r * RED_LD + cq * 4 + e;
                rec[e] = ((red[par][0][o] + red[par][1][o]) + red[par][2][o]) + red[par][3][o];
            }
        }

        // ---- pointwise reverse step
        const int tt = d ? (T - 1 - t) : t;
        const size_t o_out = ((size_t)b * T + tt) * HO + (size_t)d * H + colc;
        float sp[4];
        if (t > 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) sp[e] = (upv[e] - a.theta) > 0.0f ? 1.0f : 0.0f;
        } else {
            const f32x4 v = ld4(a.s0 + (size_t)bpc * H + colc);
            sp[0] = v.x; sp[1] = v.y; sp[2] = v.z; sp[3] = v.w;
        }
        f32x4 dwx, spv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float k = drop ? keep_scale(a.seed, o_out + e, a.p_drop, a.inv_keep) : 1.0f;
            const float gs = (gv[e] + gr[e]) * k;
            float ds = gs - al[e] * du_n[e];
            if (ADAPT) ds = ds + pb[e] * dw_n[e];
            ds = ds + rec[e];
            const float xs = u_t[e] - a.theta;
            const float box = (xs > -0.5f && xs <= 0.5f) ? 1.0f : 0.0f;       // snns.py:34-35
            float du = ds * box + al[e] * du_n[e];
            if (ADAPT) du = du + pa[e] * dw_n[e];
            dwx[e] = oma[e] * du;
            const float q = upv[e] - sp[e];
            acc_al[e] += (du * (q - u_t[e])) * ioma[e];
            if (ADAPT) {
                const float dw = be[e] * dw_n[e] - dwx[e];
                acc_be[e] += dw * wpv[e];
                acc_a[e] += dw * upv[e];
                acc_b[e] += dw * sp[e];
                dw_n[e] = dw;
            }
            du_n[e] = du;
            u_t[e] = upv[e];
            spv[e] = sp[e];
        }
        // ---- publish dWx_t (write-through), drain, barrier, flag
        if (valid) {
            const unsigned off = (unsigned)((((size_t)r * T + tt) * H + col) * sizeof(float));
            u32x4 raw;
#pragma unroll
            for (int e = 0; e < 4; ++e) raw[e] = __float_as_uint(dwx[e]);
            __builtin_amdgcn_raw_buffer_store_b128(raw, rsrc, off, 0, 16 /* sc1 */);
            st4(a.s_prev + ((size_t)bp * T + tt) * H + col, spv);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0 && t > 0) {
            gu32* f = (gu32*)a.flags + ((size_t)t * a.n_rt_total + rt) * a.n_ct + ct;
            __hip_atomic_store(f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    if (valid) {
        f32x4 v;
        v.x = acc_al[0]; v.y = acc_al[1]; v.z = acc_al[2]; v.w = acc_al[3]; st4(ws, v);
        v.x = du_n[0]; v.y = du_n[1]; v.z = du_n[2]; v.w = du_n[3]; st4(ws + 4 * plane, v);
        if (ADAPT) {
            v.x = acc_be[0]; v.y = acc_be[1]; v.z = acc_be[2]; v.w = acc_be[3]; st4(ws + plane, v);
            v.x = acc_a[0]; v.y = acc_a[1]; v.z = acc_a[2]; v.w = acc_a[3]; st4(ws + 2 * plane, v);
            v.x = acc_b[0]; v.y = acc_b[1]; v.z = acc_b[2]; v.w = acc_b[3]; st4(ws + 3 * plane, v);
            v.x = dw_n[0]; v.y = dw_n[1]; v.z = dw_n[2]; v.w = dw_n[3]; st4(ws + 5 * plane, v);
        }
    }
}

// ------------------------------------------------------------------------------ V prepack
// vpack[ct][kg][q][lane][e] = Vm[k][col] (forward) or Vm[col][k] (backward), Vm = V with a
// zero diagonal, k = kg*32 + 16*(lane>>5) + 4q + e, col = ct*32 + (lane&31); zero padded.
__global__ void vpack_kernel(int H, int n_ct, int nkg, int transpose, const float* __restrict__ V,
                             float* __restrict__ vpack) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)n_ct * nkg * 1024;
    if (idx >= total) return;
    const int e = (int)(idx & 3), lane = (int)((idx >> 2) & 63), q = (int)((idx >> 8) & 3);
    const int kg = (int)((idx >> 10) % nkg), ct = (int)((idx >> 10) / nkg);
    const int k = kg * 32 + 16 * (lane >> 5) + 4 * q + e, col = ct * 32 + (lane & 31);
    float v = 0.f;
    if (k < H && col < H && k != col) v = transpose ? V[(size_t)col * H + k] : V[(size_t)k * H + col];
    vpack[idx] = v;
}
__global__ void vmask_kernel(int H, const float* __restrict__ V, float* __restrict__ Vm) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)H * H) return;
    const int i = (int)(idx / H), j = (int)(idx % H);
    Vm[idx] = (i == j) ? 0.f : V[idx];
}

int pick_kgw(int H) {
    const int need = cdiv(cdiv(H, 32), 4);
    for (int k : {1, 2, 4, 8, 16})
        if (need <= k) return k;
    return 0;
}

template <bool BWD, bool ADAPT>
int launch_rec(int kgw, const RecArgs& a, unsigned grid, hipStream_t st) {
#define SP_LAUNCH(K)                                                                            \
    if (BWD) hipLaunchKernelGGL((rec_bwd_kernel<ADAPT, K>), dim3(grid), dim3(256), 0, st, a);   \
    else     hipLaunchKernelGGL((rec_fwd_kernel<ADAPT, K>), dim3(grid), dim3(256), 0, st, a);
    switch (kgw) {
        case 1: SP_LAUNCH(1) break;
        case 2: SP_LAUNCH(2) break;
        case 4: SP_LAUNCH(4) break;
        case 8: SP_LAUNCH(8) break;
        case 16: SP_LAUNCH(16) break;
        default: return SPARCH_EINVAL;
    }
#undef SP_LAUNCH
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

template <bool BWD>
int run_rec(int kind, RecArgs& a, size_t chan_bytes, int steps_per_launch, hipStream_t st) {
    const bool adapt = kind == SPARCH_KIND_RADLIF;
    const int kgw = pick_kgw(a.H);
    if (kgw == 0) return SPARCH_EINVAL;
    a.n_ct = cdiv(a.H, CT);
    a.nkg = 4 * kgw;
    a.n_rt_total = cdiv(a.Bp, RT);
    const size_t need = sparch_rec_chan_bytes(a.Bp, a.T, a.H);
    if (!a.chan || chan_bytes < need) return SPARCH_EWORKSPACE;
    if (hipMemsetAsync(a.chan, 0, need, st) != hipSuccess) return SPARCH_ELAUNCH;
    a.flags = reinterpret_cast<unsigned*>(a.chan);

    int L = steps_per_launch;
    if (L < 1) L = 1;
    if (L > a.T) L = a.T;
    int cus = sparch_device_cus();
    if (cus <= 0) cus = 256;
    int rt_per_launch;
    if (L == 1) {
        rt_per_launch = a.n_rt_total;  // nothing waits inside a launch: any grid size is fine
    } else {
        rt_per_launch = cus / a.n_ct;  // one workgroup per CU must be co-resident
        if (rt_per_launch < 1) { L = 1; rt_per_launch = a.n_rt_total; }
    }
    for (int rt0 = 0; rt0 < a.n_rt_total; rt0 += rt_per_launch) {
        a.rt_base = rt0;
        a.n_rt_launch = min(rt_per_launch, a.n_rt_total - rt0);
        const unsigned grid = (unsigned)(a.n_ct * a.n_rt_launch);
        if (!BWD) {
            for (int t0 = 0; t0 < a.T; t0 += L) {
                a.t_begin = t0; a.t_end = min(a.T, t0 + L);
                int rc = adapt ? launch_rec<false, true>(kgw, a, grid, st) : launch_rec<false, false>(kgw, a, grid, st);
                if (rc != SPARCH_OK) return rc;
            }
        } else {
            for (int t1 = a.T; t1 > 0; t1 -= L) {
                a.t_end = t1; a.t_begin = max(0, t1 - L);
                int rc = adapt ? launch_rec<true, true>(kgw, a, grid, st) : launch_rec<true, false>(kgw, a, grid, st);
                if (rc != SPARCH_OK) return rc;
            }
        }
    }
    return SPARCH_OK;
}

bool al16(std::initializer_list<const void*> ps) {
    for (const void* p : ps)
        if (p && !aligned16(p)) return false;
    return true;
}

}  // namespace

extern "C" size_t sparch_vpack_bytes(int H) {
    const int kgw = pick_kgw(H);
    if (H <= 0 || kgw == 0) return 0;
    return (size_t)cdiv(H, CT) * (4 * kgw) * 1024 * sizeof(float);
}

extern "C" int sparch_vpack(int H, const float* V, int transpose, float* vpack, float* vmasked, void* stream) {
    SPARCH_ENTER();
    const int kgw = pick_kgw(H);
    if (H <= 0 || kgw == 0 || !V || !vpack) return SPARCH_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int n_ct = cdiv(H, CT), nkg = 4 * kgw;
    const size_t total = (size_t)n_ct * nkg * 1024;
    hipLaunchKernelGGL(vpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, H, n_ct, nkg,
                       transpose, V, vpack);
    SPARCH_CHECK_LAUNCH();
    if (vmasked) {
        const size_t n = (size_t)H * H;
        hipLaunchKernelGGL(vmask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, H, V, vmasked);
        SPARCH_CHECK_LAUNCH();
    }
    return SPARCH_OK;
}

extern "C" size_t sparch_rec_chan_bytes(int Bp, int T, int H) {
    if (Bp <= 0 || T <= 0 || H <= 0) return 0;
    // forward: T x row tiles x column tiles x 32 granules of 8 B (the backward's flags fit inside)
    return (size_t)T * cdiv(Bp, RT) * cdiv(H, CT) * 32 * sizeof(u64);
}

extern "C" int sparch_rec_cell_fwd(int kind, int B, int dirs, int T, int H, const float* Wx,
                                   const float* scale, const float* shift, const float* alpha,
                                   const float* beta, const float* a, const float* b,
                                   const float* vpack, const float* rec0, const float* u0,
                                   const float* w0, const float* s0, float theta, float p_drop,
                                   uint64_t seed, float* s_out, float* u_save, float* w_save,
                                   uint32_t* spike_count, void* chan, size_t chan_bytes,
                                   uint32_t* status, int steps_per_launch, void* stream) {
    SPARCH_ENTER();
    if (kind != SPARCH_KIND_RLIF && kind != SPARCH_KIND_RADLIF) return SPARCH_EINVAL;
    const bool adapt = kind == SPARCH_KIND_RADLIF;
    if (B <= 0 || T <= 0 || H < 4 || (H % 4) != 0 || (dirs != 1 && dirs != 2)) return SPARCH_EINVAL;
    if (!Wx || !alpha || !vpack || !rec0 || !u0 || !s0 || !s_out || !u_save || !status) return SPARCH_EINVAL;
    if (adapt && (!beta || !a || !b || !w0 || !w_save)) return SPARCH_EINVAL;
    if ((scale == nullptr) != (shift == nullptr)) return SPARCH_EINVAL;
    if (!(p_drop >= 0.0f && p_drop < 1.0f)) return SPARCH_EINVAL;
    if (!al16({Wx, vpack, rec0, u0, w0, s0, s_out, u_save, w_save, chan})) return SPARCH_EALIGN;
    RecArgs r{};
    r.B = B; r.dirs = dirs; r.T = T; r.H = H; r.Bp = B * dirs;
    r.Wx = Wx; r.scale = scale; r.shift = shift;
    r.alpha = alpha; r.beta = beta; r.a = a; r.b = b;
    r.vpack = vpack; r.rec0 = rec0; r.u0 = u0; r.w0 = w0; r.s0 = s0;
    r.theta = theta; r.p_drop = p_drop; r.inv_keep = 1.0f / (1.0f - p_drop); r.seed = seed;
    r.s_out = s_out; r.u_save = u_save; r.w_save = w_save; r.spike_count = spike_count;
    r.chan = (u64*)chan; r.status = status;
    return run_rec<false>(kind, r, chan_bytes, steps_per_launch, (hipStream_t)stream);
}

extern "C" int sparch_rec_cell_bwd(int kind, int B, int dirs, int T, int H, const float* g_out,
                                   const float* g_rate, const float* u_save, const float* w_save,
                                   const float* alpha, const float* beta, const float* a,
                                   const float* b, const float* vpack_t, const float* u0,
                                   const float* w0, const float* s0, float theta, float p_drop,
                                   uint64_t seed, float* dWx, float* s_prev, float* dparam_ws,
                                   void* chan, size_t chan_bytes, uint32_t* status,
                                   int steps_per_launch, void* stream) {
    SPARCH_ENTER();
    if (kind != SPARCH_KIND_RLIF && kind != SPARCH_KIND_RADLIF) return SPARCH_EINVAL;
    const bool adapt = kind == SPARCH_KIND_RADLIF;
    if (B <= 0 || T <= 0 || H < 4 || (H % 4) != 0 || (dirs != 1 && dirs != 2)) return SPARCH_EINVAL;
    if (!g_out || !u_save || !alpha || !vpack_t || !u0 || !s0 || !dWx || !s_prev || !dparam_ws || !status)
        return SPARCH_EINVAL;
    if (adapt && (!beta || !a || !b || !w0 || !w_save)) return SPARCH_EINVAL;
    if (!(p_drop >= 0.0f && p_drop < 1.0f)) return SPARCH_EINVAL;
    if (!al16({g_out, u_save, w_save, vpack_t, u0, w0, s0, dWx, s_prev, dparam_ws, chan})) return SPARCH_EALIGN;
    // the per-tile dWx window (32 rows x T x H floats) must be addressable by a 32-bit byte offset
    if ((size_t)RT * T * H * sizeof(float) >= ((size_t)1 << 31)) return SPARCH_EINVAL;
    RecArgs r{};
    r.B = B; r.dirs = dirs; r.T = T; r.H = H; r.Bp = B * dirs;
    r.alpha = alpha; r.beta = beta; r.a = a; r.b = b;
    r.vpack = vpack_t; r.u0 = u0; r.w0 = w0; r.s0 = s0;
    r.theta = theta; r.p_drop = p_drop; r.inv_keep = 1.0f / (1.0f - p_drop); r.seed = seed;
    r.u_save = const_cast<float*>(u_save); r.w_save = const_cast<float*>(w_save);
    r.g_out = g_out; r.g_rate = g_rate; r.g_rate_scale = 1.0f / ((float)B * (float)T);
    r.dWx = dWx; r.s_prev = s_prev; r.dparam_ws = dparam_ws;
    r.chan = (u64*)chan; r.status = status;
    return run_rec<true>(kind, r, chan_bytes, steps_per_launch, (hipStream_t)stream);
}

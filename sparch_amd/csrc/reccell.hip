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
//   * its 1024x32 slice of V (forward) / V^T (backward) stays in VGPRs for the whole launch
//     as MFMA B-operands; the four waves split K and the partial 32x32 tiles meet in LDS in
//     a fixed order (reproducible);
//   * EXACT low-precision operands: every fp32 V element is split into three bf16 values
//     V = hi + mid + lo (8+8+8 significand bits, exact).  Spikes are 0/1, exact in bf16, so
//     s@V = s@hi + s@mid + s@lo runs on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate) with
//     exact products and fp32 accumulation: same accuracy class as an fp32 fmaf chain at 3/16
//     of its cost.  In the backward dWx is split as well (x = t1 + t2 + t3 exactly) and the six
//     largest cross terms (t1*hi, t1*mid, t2*hi, t1*lo, t3*hi, t2*mid; the dropped ones are
//     t3*mid <= 2^-22 and t2*lo <= 2^-23 of |x||V| per product in the worst case; measured aggregate
//     3e-9 of sum|x||V| against 1e-7 for an fp32 sgemm's own rounding) are accumulated: fp32-faithful
//     at 6/16 of the fp32 MFMA cost;
//   * the 32 workgroups of one batch tile exchange the step's output every step:
//       forward  — spikes, bit-packed, as 8-byte {tag = t+1, 32 spike bits} granules written
//                  with one agent-scope (sc1, write-through) store each and polled with sc1
//                  loads: the data is the flag, no fence (MI355X guide, G16 form R2).  One slot
//                  per (step, row): never reused inside a call;
//       backward — the 32x32 tile of dWx_t as three PRE-SPLIT bf16 planes (x = t1 + t2 + t3 exactly: truncation
//                  split by v_perm / AND / SUB in the producer), stored in MFMA-FRAGMENT ORDER (16-byte piece
//                  ((ks*3 + p)*64 + h*32 + row) = plane p, k = 16 ks + 8 h + 0..7) into a depth-4 ring, and
//                  NOTHING ELSE: no drain, no tag.  The data is the flag here too: every ring slot holds a
//                  SENTINEL bit pattern (a signalling NaN, which no fp32 arithmetic result can be — and no
//                  plane word either: its upper half would be a signalling-NaN bf16) until its piece lands,
//                  and consumers simply load the tile (16-byte sc1 loads, every wave-load one contiguous
//                  1 KiB = the MFMA A fragments of one k16-step and plane) and re-load the pieces that still
//                  read as the sentinel (first and last word checked: a producer thread owns 4 consecutive k,
//                  i.e. half a piece, written with one 8-byte store per plane; 8- and 16-byte stores are
//                  observed untorn on gfx950).  A producer puts the sentinel back into its slot of step t+2
//                  while it works on step t: by then it has seen every peer's step t+1 tile, which they
//                  produced after consuming every step t+2 tile, and its next vmcnt(0) (the tile loads of
//                  step t-1) retires that store before it publishes step t-1 — the store a consumer must see
//                  before it can ask for step t-2, the slot's next content.  Against the first version (tile
//                  stores, drain, barrier, tag store; consumers poll the tag, barrier, then load): one
//                  store->load visibility hop instead of two per step and two workgroup barriers fewer.
//                  Until the XCD-local stores the wire carried fp32 (4 KB per tile) and each of the 32
//                  consumers split the same values (2.8 k VALU cycles per SIMD and step); with the tiles
//                  served by the XCD's own L2 the 6 KB plane tiles are the faster trade (1.34 -> 1.27 ms per
//                  launch; timing ablations: no split at 4 KB 1.16, no split at 6 KB 1.27);
//   * block -> tile mapping keeps a batch tile's workgroups at equal blockIdx % n_row_tiles,
//     i.e. on one XCD under round-robin dispatch.  That is a speed choice only: every
//     hand-off is agent-scope and placement-independent.  Every spin is bounded by a
//     wall-clock timeout that raises *status and unwinds the launch.
//   * steps_per_launch = T gives one persistent launch; = 1 degenerates to one launch per
//     time step, where every wait is already satisfied at launch (safe fallback, and the
//     path for shapes whose grid cannot be co-resident).
#include "common.h"

#include "rec_common.h"

#ifndef REC_ACC_REGS
#define REC_ACC_REGS 0  /* backward: parameter / BatchNorm partial sums in LDS (1: in registers — measured no faster: 1.37 vs 1.36 ms per launch) */
#endif

namespace {

struct RecArgs {
    int B, dirs, T, H, Bp;
    int n_ct, nkg, n_rt_total;
    int rt_base, n_rt_launch;
    int t_begin, t_end;
    const float* Wx; const float* scale; const float* shift;
    const float* alpha; const float* beta; const float* a; const float* b;
    const u32x4* vpack; const float* rec0;
    // step variants (EXT): the recurrent product of the ONE step [t_begin, t_begin+1) is supplied by the caller
    // in rec0 (Bp,H); the step's raw spikes (forward, bf16 0/1) / dWx (backward, fp32) also go to a contiguous
    // (Bp,H) buffer, the operand of the caller's next product
    uint16_t* s_step16; float* dwx_step;
    const float* u0; const float* w0; const float* s0;
    float theta, p_drop, inv_keep; uint64_t seed;
    float* s_out; uint16_t* s16_out; float* u_save; float* w_save; uint32_t* spike_count;
    // backward
    const float* g_out; const float* g_rate; float g_rate_scale;
    float* dWx; uint16_t* s_prev16; float* dparam_ws;
    // BatchNorm backward folded in (nullable): raw projection (B,T,H) + per-column statistics; the kernel then
    // also leaves sum_t dWx and sum_t dWx*xhat per (row, column) in planes 6 and 7 of dparam_ws
    const float* bn_x; const float* bn_mean; const float* bn_invstd;
    const float* bn_src;  // backward: bn_x, or (without BatchNorm) any readable (B,T,H) range — the loop loads unconditionally
    int save16;  // u_save / w_save hold bf16 (common.h save_u16); whole-sequence launches only
    // hand-off
    u64* chan; char* ring; unsigned* status;
    unsigned* xcd_tab;  // [n_rt_total][n_ct] agreement table of the XCD-local stores (whole-sequence launches), or null
};

// Diagnostic build only (-DSPARCH_REC_PROF, never shipped): per-workgroup sums of s_memtime
// deltas for the segments of a time step, written to a private buffer no kernel reads.
#ifdef SPARCH_REC_PROF
__device__ u64 g_rec_prof[2][512][12];
#define PROF_DECL u64 pf_t = 0, pf_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define PROF_STAMP(i)                                                                           \
    do {                                                                                        \
        u64 now_;                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");            \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        if ((i) >= 0) pf_acc[(i) < 0 ? 0 : (i)] += now_ - pf_t;                                 \
        pf_t = now_;                                                                            \
    } while (0)
#ifdef SPARCH_REC_PROF_ALLWAVES  /* every wave of the first 64 workgroups: slot = workgroup * 8 + wave */
#define PROF_FLUSH(which)                                                                       \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 64)                                             \
        for (int i_ = 0; i_ < 10; ++i_) g_rec_prof[which][blockIdx.x * 8 + (threadIdx.x >> 6)][i_] += pf_acc[i_];
#else
#define PROF_FLUSH(which)                                                                       \
    if ((threadIdx.x == 0 || threadIdx.x == 256) && blockIdx.x < 256)  /* wave 0, and wave 4 in slots 256.. */ \
        for (int i_ = 0; i_ < 10; ++i_) g_rec_prof[which][blockIdx.x + (threadIdx.x ? 256 : 0)][i_] += pf_acc[i_];
#endif
#else
#define PROF_DECL
#define PROF_STAMP(i)
#define PROF_FLUSH(which)
#endif

// ------------------------------------------------------------------------------ forward
// NW waves per workgroup, each taking the k-groups kg = wave + NW*kk of the contraction (partial tiles
// summed through LDS); NW = 8 puts two waves on each SIMD so that one's LUT reads / poll latency sit
// under the other's MFMAs.  Pointwise update, publish and stores stay on the first 256 threads.
// NP: planes of V — 3 = exact split (default), 1 = the bf16 operand mode (V rounded once by the pack kernel).
// CW (round 3): 32-column groups per workgroup.  1: a workgroup owns 32 columns and its first 256 threads the
// pointwise state (above).  2 (bf16 operand mode, 8 waves): it owns 64 columns — with ONE plane of V the slice is
// 128 KiB, a quarter of the register file — and all 512 threads own pointwise state (thread = (column group j,
// row, column quad)); a batch of 512 virtual rows (a bidirectional B = 256 layer, BASELINE configs[4]) is then
// 16 row tiles x 16 workgroups = ONE persistent launch on 256 CUs instead of two half-filled ones run back to
// back.  The hand-off format does not change: the workgroup publishes the granules of k-groups 2 ctw and
// 2 ctw + 1, consumers read k-groups as before.
template <bool ADAPT, int KGW, int NW, bool EXT = false, int NP = 3, int CW = 1>
__global__ __launch_bounds__(64 * NW, 1) void rec_fwd_kernel(RecArgs a) {
    static_assert(CW == 1 || (CW == 2 && NW == 8 && !EXT), "64-column workgroups: the 8-wave persistent kernels");
    __shared__ __attribute__((aligned(16))) float red[2][NW][CW][RT * RED_LD4];
    __shared__ __attribute__((aligned(16))) u32x4 lut[256];  // byte of 8 spikes -> 8 bf16 (0 / 1.0)
    __shared__ int abort_flag[2];
    __shared__ int xcd_local_flag;
    __shared__ unsigned cnt_lds[2][CW * CT];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hh = lane >> 5;
    const int rt = a.rt_base + (int)(blockIdx.x % a.n_rt_launch);
    const int ctw = (int)(blockIdx.x / a.n_rt_launch);  // the workgroup's index among its row tile's workgroups
    const int T = a.T, H = a.H, HO = a.H * a.dirs;

    // element ownership for the pointwise update: row r, 4 columns (of column group ct)
    const bool pw = tid < 256 * CW;
    const bool pw_wave = NW == 4 || CW == 2 || wave < 4;  // the same, as a scalar (wave-uniform branches)
    const int ct = ctw * CW + (CW == 2 ? tid >> 8 : 0);   // this THREAD's 32-column group
    const int r = (tid & 255) >> 3, cq = tid & 7;
    const int bp = rt * RT + r, col = ct * CT + cq * 4;
    const bool valid = pw && bp < a.Bp && col < H;
    const int bpc = min(bp, a.Bp - 1), colc = min(col, H - 4);
    const int d = bpc / a.B, b = bpc - d * a.B;

    u32x4 vb[CW][KGW][2][NP];
    if (!EXT) {
#pragma unroll
        for (int c = 0; c < CW; ++c) load_vslice<KGW, NW, NP>(vb[c], a.vpack, ctw * CW + c, a.nkg, wave, lane);
    }
    if (tid < 256) {
        u32x4 e;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            e[j] = (((unsigned)tid >> (2 * j)) & 1u ? 0x3F80u : 0u) | (((unsigned)tid >> (2 * j + 1)) & 1u ? 0x3F800000u : 0u);
        lut[tid] = e;
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
    if (tid < 2 * CW * CT) cnt_lds[tid / (CW * CT)][tid % (CW * CT)] = 0;
    const unsigned my_xcc = xcc_id();
    xcd_agree((EXT || !a.xcd_tab) ? nullptr : (gu32*)a.xcd_tab + (size_t)rt * a.n_ct, 0u, a.n_ct / CW, ctw, my_xcc, tid, &xcd_local_flag);
    __syncthreads();
    const bool xcd_local = xcd_local_flag != 0;  // this row tile's workgroups share an XCD: plain hand-off stores

    const bool has_norm = a.scale != nullptr;
    const bool drop = a.p_drop > 0.0f;
    const uint64_t seed = drop ? resolve_seed(a.seed) : 0;
    auto wx_ptr = [&](int t) {
        const int tt = d ? (T - 1 - t) : t;
        return a.Wx + ((size_t)b * T + tt) * H + colc;
    };
    f32x4 x_next = {0.f, 0.f, 0.f, 0.f};
    if (pw) x_next = ld4s(wx_ptr(a.t_begin));
    // The recurrent drive of the launch's FIRST step (t = 0: s0 @ V from the host; step variants: the caller's
    // product) and the projection row of its second step are loaded HERE, not inside the loop: a global load on
    // one path of the loop body only (round 2 had `if (t == 0) rec = ld4(rec0)`) makes hipcc's wait-count pass
    // merge the two paths conservatively — an `s_waitcnt vmcnt(0)` behind the reduction barrier of EVERY step,
    // where the pointwise waves then sat out the acknowledgements of the previous step's bulk HBM stores.
    const bool first_ext = a.t_begin == 0 || EXT;
    f32x4 rec_first = {0.f, 0.f, 0.f, 0.f}, x_second = rec_first;
    if (first_ext) {
        rec_first = ld4(a.rec0 + (size_t)bpc * H + colc);
        if (pw && a.t_begin + 1 < a.t_end) x_second = ld4s(wx_ptr(a.t_begin + 1));
    }
    // Bulk HBM stores of a step are held back (12 VGPRs) and issued only after the NEXT step's poll loads:
    // vmcnt retires in order and counts stores, so stores (and the Wx prefetch) issued ahead of the poll
    // would put their HBM latency in front of the sweep.  Issued behind it they retire under the MFMA phase.
    f32x4 pend_s = {0.f, 0.f, 0.f, 0.f}, pend_u = pend_s, pend_w = pend_s;
    int pend_t = -1;
    // XSTORE (build option REC_XSTORE=1, 8-wave kernels; OFF: measured 0.76 -> 0.81 ms per launch, bit-identical
    // results): the stores issued by the four waves that hold NO pointwise state.  A step's (s, u, w) go through
    // an LDS staging buffer; after the next step's reduction barrier the upper waves read them and issue the
    // global stores while the pointwise waves run their update — the stores then sit in THOSE waves' memory
    // queues, not in front of the pointwise waves' poll (`vmcnt` is in order and counts stores).  But the upper
    // waves poll too, with LESS lead over their stores (a pointwise phase instead of a whole step), and the
    // matrix phase waits for its slowest wave: the timing ablation's 0.75 k cycles (no stores at all) cannot be
    // had by moving the stores between waves of the same workgroup.
    constexpr bool XSTORE = REC_XSTORE && NW == 8 && !EXT && CW == 1;
    __shared__ __attribute__((aligned(16))) f32x4 stage[XSTORE ? 2 : 1][3][XSTORE ? 256 : 1];
    const bool valid_hi = !pw && bp < a.Bp && col < H;  // an upper-wave thread, same (row, columns) as tid - 256
    auto store_step = [&](int st_t, const f32x4& vs, const f32x4& vu, const f32x4& vw) {
        const int ptt = d ? (T - 1 - st_t) : st_t;
        const size_t o_s = ((size_t)b * T + ptt) * HO + (size_t)d * H + colc;
        if (a.s_out) st4(a.s_out + o_s, vs);  // the fp32 copy: only for callers that read the layer's output tensor
        if (a.s16_out) {  // the same spikes as a bf16 plane (0 / 1.0) for the GEMMs that consume them
            u32x2 h;
            h.x = (vs[0] != 0.f ? 0x3F80u : 0u) | (vs[1] != 0.f ? 0x3F800000u : 0u);
            h.y = (vs[2] != 0.f ? 0x3F80u : 0u) | (vs[3] != 0.f ? 0x3F800000u : 0u);
            st2(a.s16_out + o_s, h);
        }
        st4_saved<true>(a.u_save, ((size_t)bp * T + st_t) * H + col, vu, a.save16, a.theta);
        if (ADAPT) st4_saved<false>(a.w_save, ((size_t)bp * T + st_t) * H + col, vw, a.save16, a.theta);
    };
    auto flush_pending = [&]() {
#if defined(SPARCH_REC_PROF) && defined(FA_NO_BULK)  // timing ablation (no outputs): the step's HBM stores dropped
        pend_t = -1;
        return;
#endif
        if (!XSTORE && pend_t >= 0 && valid) store_step(pend_t, pend_s, pend_u, pend_w);
        pend_t = -1;
    };
    auto flush_staged = [&](int st_t) {  // upper waves: the staged step st_t -> HBM
#if defined(SPARCH_REC_PROF) && defined(FA_NO_BULK)
        return;
#endif
        if (valid_hi) {
            const int q = tid & 255;
            const f32x4 vs = stage[st_t & 1][0][q], vu = stage[st_t & 1][1][q];
            f32x4 vw = vs;
            if (ADAPT) vw = stage[st_t & 1][2][q];
            store_step(st_t, vs, vu, vw);
        }
    };
    PROF_DECL

    for (int t = a.t_begin; t < a.t_end; ++t) {
        PROF_STAMP(-1);
        // this step's projection row.  (The take-over copy of the prefetch register ends up at the loop latch with
        // an `s_waitcnt vmcnt(0)` in front of the next poll's issue.  Measured against it in one call, 40 launches
        // each: the prefetch by LDS-DMA — no register destination, no wait anywhere but the poll's own — 0.679 vs
        // 0.669 ms per launch, its LDS read hoisted in front of the matrix phase 0.69; 256-1024 cycles of s_sleep
        // ahead of the first poll sweep +0.02 ms per 256.  The round-2 code, with a FLAT flag read and a
        // conditional load in the loop body, waited vmcnt(0) behind the reduction barrier instead: 0.699.)
        f32x4 xv;
        float rec[4] = {0.f, 0.f, 0.f, 0.f};

        if (t == 0 || EXT) {
            rec[0] = rec_first.x; rec[1] = rec_first.y; rec[2] = rec_first.z; rec[3] = rec_first.w;
            xv = x_next;
            x_next = x_second;
        } else {
            // ---- gather the 32 x H spike bits of step t-1 (tag == t) for this wave's k-groups
            const gu64* base = (const gu64*)a.chan + ((size_t)(t - 1) * a.n_rt_total + rt) * a.n_ct * 32;
            u64 gran[KGW];
            const u64 t_start = __builtin_amdgcn_s_memrealtime();
            for (unsigned spins = 0;; ++spins) {
                // all loads of the sweep are issued back to back (clamped index, no control flow between
                // them) and only then compared: ONE round trip per sweep.  (A short-circuit `ok && tag==t`
                // inside a per-k-group `if` made hipcc wait vmcnt(0) after every load: 8 serialized round
                // trips, ~5.5k cycles per step — the largest single cost of the first versions.)
#pragma unroll
                for (int kk = 0; kk < KGW; ++kk) {
                    const int kgc = min(wave + NW * kk, a.n_ct - 1);
                    gran[kk] = __hip_atomic_load(base + (size_t)kgc * 32 + li, __ATOMIC_RELAXED, REC_LD_SCOPE);
                }
                unsigned bad = 0;
#pragma unroll
                for (int kk = 0; kk < KGW; ++kk) {
                    const unsigned m = (wave + NW * kk < a.n_ct) ? 0xFFFFFFFFu : 0u;
                    bad |= ((unsigned)(gran[kk] >> 32) ^ (unsigned)t) & m;
                }
                if (__all(bad == 0)) break;
                if ((spins & 63u) == 63u &&
                    __builtin_amdgcn_s_memrealtime() - t_start > TIMEOUT_TICKS) {
                    raise_timeout(a.status, &abort_flag[t & 1], SPARCH_STATUS_REC_FWD, t);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            PROF_STAMP(0);  // poll wait
            // (the empty asm makes the take-over a use of the prefetch register AT THIS POINT: as a plain copy the
            // register allocator placed it at the loop latch, with a `vmcnt(0)` in front of the next poll)
#pragma unroll
            for (int e = 0; e < 4; ++e) xv[e] = x_next[e];
            if (pw && t + 1 < a.t_end) x_next = ld4s(wx_ptr(t + 1));
            flush_pending();
            // ---- s_{t-1} @ V on the bf16 MFMA: spikes expanded through the LDS table (lane
            //      (row li, k-half hh) takes byte 2*ks + hh of its row's 32-bit word)
            u32x4 af[KGW][2];
#pragma unroll
            for (int kk = 0; kk < KGW; ++kk) {
                // k-groups beyond H (padding of the register layout) contribute zero spikes; keeping the whole
                // section free of branches lets the scheduler interleave LUT reads with the MFMA chain
                const unsigned wbits = (wave + NW * kk < a.n_ct) ? (unsigned)gran[kk] : 0u;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#if defined(SPARCH_REC_PROF) && defined(FA_NO_LUT)  // timing ablation (wrong results): no LDS table read
                    const unsigned by = (wbits >> (16 * ks + 8 * hh)) & 0xFFu;
                    af[kk][ks] = u32x4{by, by ^ 0x3F80u, by, by};
#else
                    af[kk][ks] = lut[(wbits >> (16 * ks + 8 * hh)) & 0xFFu];
#endif
                }
            }
            f32x16 acc[CW];
#pragma unroll
            for (int c = 0; c < CW; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
#pragma unroll
            for (int kk = 0; kk < KGW; ++kk)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
#ifdef REC_FWD_ABL2  // timing ablation (wrong results): two of the three planes
                    for (int p = NP - 1; p >= (NP == 3 ? 1 : 0); --p)
#else
                    for (int p = NP - 1; p >= 0; --p)
#endif
#pragma unroll
                        for (int c = 0; c < CW; ++c) {
#if defined(SPARCH_REC_PROF) && defined(FA_NO_MFMA)  // timing ablation (wrong results): operands kept alive, no MFMA
                            asm volatile("" ::"v"(af[kk][ks]), "v"(vb[c][kk][ks][p]));
#else
                            acc[c] = mfma_bf16(af[kk][ks], vb[c][kk][ks][p], acc[c]);
#endif
                        }
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                float* rd = red[t & 1][wave][c];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
                    rd[row * RED_LD4 + li] = acc[c][i];
                }
            }
            PROF_STAMP(1);  // expand + MFMA + LDS write
        }
        if (xcd_local && xcc_id() != my_xcc) raise_timeout(a.status, &abort_flag[t & 1], SPARCH_STATUS_REC_FWD, t);  // moved to another XCD
        lds_barrier();
        PROF_STAMP(2);  // barrier
        if (lds_flag_read(&abort_flag[t & 1])) break;
        if (XSTORE && t > a.t_begin) flush_staged(t - 1);  // (the barrier above ordered the staging writes)
        // Everything from here to the end of the step belongs to the waves that own pointwise state.  A WAVE-UNIFORM
        // branch (round 3): until then the upper four waves of an 8-wave workgroup ran the whole update on dead
        // values — per-thread `valid` only masked its stores — and took every other vector issue slot of the
        // pointwise waves they share a SIMD with (stamped: 1.44 k cycles of "pointwise" on wave 4 beside 1.54 k on
        // wave 0).  They now go straight on to the next step's poll.
        if (pw_wave) {
#if defined(SPARCH_REC_PROF) && defined(FA_NO_RED)  // timing ablation (wrong results): one partial tile instead of NW
        if (t > 0 && !EXT) {
#pragma unroll
            for (int e = 0; e < 4; ++e) rec[e] = red[t & 1][0][CW == 2 ? tid >> 8 : 0][r * RED_LD4 + cq * 4 + e];
        }
#else
        if (t > 0 && !EXT) {
            // unpadded 128-byte rows: a thread's four columns are one aligned ds_read_b128 per partial tile, and the
            // b128 lane groups (rows r, r+1 of column quads 0-3 / 4-7) fall on all 64 banks (conflict-free, like the
            // MFMA lanes' ds_write_b32 of 32 consecutive floats) — 8 LDS instructions instead of 32 on this chain
            const int cj = CW == 2 ? tid >> 8 : 0;
            f32x4 sum = *reinterpret_cast<const f32x4*>(&red[t & 1][0][cj][r * RED_LD4 + cq * 4]);
#pragma unroll
            for (int w_ = 1; w_ < NW; ++w_) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(&red[t & 1][w_][cj][r * RED_LD4 + cq * 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) sum[e] = sum[e] + v[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) rec[e] = sum[e];
        }
#endif

        // ---- pointwise membrane update for this thread's 4 neurons
        // (Hoisting the rec-independent part of this update in front of the poll was measured: 0.75 -> 0.81 ms
        // per launch — its use of x_t there waits, in order, for the previous step's bulk stores.)
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
        f32x4 uo, wo;
        unsigned nib = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float xn = xs[e];
            if (has_norm) xn = bn_affine(xn, sc[e], sh[e]);
            float drive = xn + rec[e];                                      // snns.py:572 / 720
            if (ADAPT) {
                w[e] = (be[e] * w[e] + pa[e] * u[e]) + pb[e] * s[e];        // snns.py:718
                drive = drive - w[e];
            }
            u[e] = al[e] * (u[e] - s[e]) + oma[e] * drive;                  // snns.py:572 / 719
            s[e] = (u[e] - a.theta) > 0.0f ? 1.0f : 0.0f;                   // snns.py:29
            uo[e] = u[e];
            wo[e] = w[e];
            if (valid) nib |= (s[e] != 0.0f ? 1u : 0u) << e;
        }
        // ---- publish this tile's spikes first (it is on every consumer's critical path): one
        //      tagged granule per row; dropout and the bulk stores then overlap the peers' step
        // OR over the 8 lanes of a row with DPP moves (quad xor 1, quad xor 2, then the mirrored half row
        // brings in the other quad) — __shfl_xor would go through the LDS crossbar three times in a row
        unsigned word = nib << (cq * 4);
        word |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)word, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
        word |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)word, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
        word |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)word, 0x141, 0xF, 0xF, true);  // row_half_mirror
        if (EXT) {
            if (valid) {  // the step's raw (pre-dropout) spikes: operand of the caller's s_t @ V product
                u32x2 h;
                h.x = (s[0] != 0.f ? 0x3F80u : 0u) | (s[1] != 0.f ? 0x3F800000u : 0u);
                h.y = (s[2] != 0.f ? 0x3F80u : 0u) | (s[3] != 0.f ? 0x3F800000u : 0u);
                *reinterpret_cast<u32x2*>(a.s_step16 + (size_t)bp * H + col) = h;
            }
        } else if (pw && cq == 0 && t + 1 < T) {
            gu64* slot = (gu64*)a.chan + (((size_t)t * a.n_rt_total + rt) * a.n_ct + ct) * 32 + r;
            const u64 granule = ((u64)(unsigned)(t + 1) << 32) | (u64)word;
            if (xcd_local) __hip_atomic_store(slot, granule, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_store(slot, granule, __ATOMIC_RELAXED, REC_ST_SCOPE);
        }
#if !REC_FWD_UPPER_SLEEP
        if (NW == 8 && CW == 1) lds_barrier();  // releases the upper waves into the next step's poll (see the else branch)
#endif
        PROF_STAMP(3);  // pointwise + publish
        if (valid) {
            const int tt = d ? (T - 1 - t) : t;
            const size_t o_out = ((size_t)b * T + tt) * HO + (size_t)d * H + colc;
            f32x4 so;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float k = drop ? keep_scale(seed, o_out + e, a.p_drop, a.inv_keep) : 1.0f;
                so[e] = s[e] * k;
                cnt[e] += (so[e] != 0.0f) ? 1u : 0u;
            }
            if (XSTORE) {
                stage[t & 1][0][tid] = so; stage[t & 1][1][tid] = uo;
                if (ADAPT) stage[t & 1][2][tid] = wo;
            } else {
                pend_s = so; pend_u = uo; pend_w = wo;
            }
        }
        } else {
            // ... but not at once: the peers' granules of this step cannot exist before their pointwise phase is
            // over, and sweeps that must fail load the one L2 channel that holds the row tile's granule lines
            // (measured, 40 launches each in one call: polling at once 0.70 ms per launch; 1024 / 1536 / 2048 cycles
            // of s_sleep first 0.665 / 0.646 / 0.72 — past the pointwise phase the upper waves are late themselves;
            // spinning on an LDS word the first wave sets when it has published 0.68: the spin takes issue slots
            // and LDS cycles from the pointwise waves).  So they park in a second barrier that the pointwise waves
            // join right behind their publish store: no instruction issues while a wave waits there, and the
            // wait adapts to the cell kind and the clock.
#if REC_FWD_UPPER_SLEEP
            __builtin_amdgcn_s_sleep(REC_FWD_UPPER_SLEEP);
#else
            lds_barrier();
#endif
        }
        pend_t = t;
        PROF_STAMP(4);  // dropout
    }
    if (XSTORE) {  // the last step of the launch (skipped after an abort: the step is discarded anyway)
        const bool aborted = (lds_flag_read(&abort_flag[0]) | lds_flag_read(&abort_flag[1])) != 0;
        __syncthreads();
        if (!aborted && a.t_end > a.t_begin) flush_staged(a.t_end - 1);
    } else {
        flush_pending();
    }
    PROF_FLUSH(0)

    // ---- spike counts (post-dropout) -> one integer atomic per (direction, column) per workgroup
    if (a.spike_count) {
        if (valid) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (cnt[e]) atomicAdd(&cnt_lds[d][(CW == 2 ? (tid >> 8) * CT : 0) + cq * 4 + e], cnt[e]);
        }
        __syncthreads();
        if (tid < 2 * CW * CT) {
            const int dd = tid / (CW * CT), c = tid % (CW * CT);
            const unsigned v = cnt_lds[dd][c];
            if (v && dd < a.dirs && ctw * CW * CT + c < H) atomicAdd(a.spike_count + (size_t)dd * H + ctw * CW * CT + c, v);
        }
    }
}

// ------------------------------------------------------------------------------ backward
// NW waves per workgroup, each contracting over its own k-groups (kg = wave + NW*kk) into a partial
// 32 x 32 tile that is summed through LDS.  NW = 8 (two waves per SIMD) whenever there are enough k-groups:
// per step a SIMD has 112 MFMAs, ~770 VALU instructions of splitting and 32 tile loads to issue, and with
// a single wave those queue behind each other (and behind the first load's latency); two waves fill each
// other's stalls.  The pointwise reverse step and the stores stay on the first 256 threads.
// NP: planes of the hand-off tiles and of V^T — 3 = exact split, six cross terms (default); 1 = the bf16 operand
// mode: the producer rounds its dWx once, 2 KB tiles, ONE MFMA per k16-step.
// S16: the saved states u / w are bf16 (common.h save_u16) — a template parameter so that every global load of the
// time loop is ONE unconditional instruction (see `load_step`).
// CW: 32-column groups per workgroup (see rec_fwd_kernel): 2 = the bf16 operand mode's 64-column workgroups, all 512
// threads own pointwise state, the workgroup publishes the tiles of k-groups 2 ctw and 2 ctw + 1.
template <bool ADAPT, int KGW, int NW, bool EXT = false, int NP = 3, bool S16 = false, int CW = 1>
__global__ __launch_bounds__(64 * NW, 1) void rec_bwd_kernel(RecArgs a) {
    static_assert(CW == 1 || (CW == 2 && NW == 8 && !EXT && NP == 1), "64-column workgroups: bf16 operand mode, 8 waves");
    // cross-wave reduction tiles: written before the step's first barrier, read after it by the pointwise
    // waves, which reach the second (publish) barrier only when done with them -> one buffer
    __shared__ __attribute__((aligned(16))) float red[NW][CW][RT * RED_LD4];
    // lo plane of the V^T slice lives in LDS (64 KiB at H=1024) so the register file holds the
    // hi/mid planes (128 VGPRs) plus all 32 in-flight fp32 dWx tile loads (128 VGPRs) without spilling
    __shared__ __attribute__((aligned(16))) u32x4 vlo[NP == 3 ? NW : 1][NP == 3 ? KGW : 1][2][64];
    // neuron constants (alpha, beta, a, b, rate gradient, BatchNorm mean / invstd) of the tile's 8 column quads,
    // per direction (the rate gradient depends on it): kept in LDS and re-read each step — the 8-wave kernel's
    // 256-register budget has no room to hold them.  (Rounds 1-2 kept a copy per THREAD: 28 KB for 1.8 KB of data.)
    __shared__ __attribute__((aligned(16))) f32x4 pcol[7][2][8 * CW];
    // initial states of the tile (u0, w0, s0): read by cell step 0 only — from LDS, so that the loop body holds no
    // global load on one path only (hipcc's wait-count pass merges such paths with an `s_waitcnt vmcnt(0)`)
    __shared__ __attribute__((aligned(16))) f32x4 first_tile[3][256 * CW];
    // running parameter-gradient partial sums (alpha, beta, a, b) of the thread's 4 columns: touched once per
    // step, off the critical path -> LDS, so that the hot loop's registers do not spill
#if REC_ACC_REGS
    f32x4 pacc_r[6] = {};         // the six accumulators in registers (the k-group pipeline freed the room)
#define PACC(j, i) pacc_r[j]
#else
    __shared__ __attribute__((aligned(16))) f32x4 pacc[6][256 * CW];  // + BatchNorm's sum dWx, sum dWx*xhat
#define PACC(j, i) pacc[j][i]
#endif
    __shared__ int abort_flag[2];
    __shared__ int xcd_local_flag;
    // PARK: the saved states of a step (u_{t-1}, w_{t-1}, the raw projection) wait in LDS from the loop top to the
    // points that use them — the spike of s_{t-1} and the parameter sums behind the publish barrier, u_t in the next
    // step's box-car gate — instead of in 16 VGPRs across the whole tile phase (the registers a second k-group of
    // tile loads in flight needs)
    constexpr bool PARK = REC_BWD_PARK && NW == 8 && CW == 1;
    __shared__ __attribute__((aligned(16))) f32x4 park_u[PARK ? 2 : 1][PARK ? 256 : 1];
    __shared__ __attribute__((aligned(16))) f32x4 park_wx[PARK ? 2 : 1][PARK ? 256 : 1];
    // BXS (8-wave kernels): the step's bulk HBM stores (dWx for the GEMMs, the bf16 plane of s_{t-1}) are issued
    // by the four waves that hold NO pointwise state, one step later: the pointwise waves stage the 24 bytes per
    // thread in LDS after the publish barrier, the upper waves pick them up behind the NEXT step's reduction
    // barrier — where they would otherwise idle through the pointwise phase — and issue the stores there, ~2.4 k
    // cycles ahead of their own next tile loads.  On the pointwise waves the stores sat ~0.8 k cycles in front of
    // the next step's tile loads, and `vmcnt` (in order, counts stores) made the first k-group wait for their
    // acknowledgement: timing ablation without the stores 1.29 -> 1.16 ms per launch.
    constexpr bool BXS = REC_BWD_XSTORE && NW == 8 && !EXT && CW == 1;
    __shared__ __attribute__((aligned(16))) f32x4 stage_dwx[BXS ? 256 : 1];
    __shared__ __attribute__((aligned(8))) u32x2 stage_sp[BXS ? 256 : 1];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hh = lane >> 5;
    const int rt = a.rt_base + (int)(blockIdx.x % a.n_rt_launch);
    const int ctw = (int)(blockIdx.x / a.n_rt_launch);  // the workgroup's index among its row tile's workgroups
    const int T = a.T, H = a.H, HO = a.H * a.dirs;

    const bool pw = tid < 256 * CW;  // threads that own a (row, 4 columns) piece of the tile's pointwise state
    const bool pw_wave = NW == 4 || CW == 2 || wave < 4;  // the same, as a scalar
    const int cj = CW == 2 ? tid >> 8 : 0;                // this thread's column group within the workgroup
    const int ct = ctw * CW + cj;                         // ... as a 32-column group of the layer
    const int cqx = cj * 8 + (tid & 7);                   // index of its column quad in the workgroup's tables
    const int r = (tid & 255) >> 3, cq = tid & 7;
    const int bp = rt * RT + r, col = ct * CT + cq * 4;
    const bool valid = pw && bp < a.Bp && col < H;
    const bool valid_hi = !pw && bp < a.Bp && col < H;  // upper-wave thread: same (row, columns) as tid - 256
    const int bpc = min(bp, a.Bp - 1), colc = min(col, H - 4);
    const int d = bpc / a.B, b = bpc - d * a.B;

    u32x4 vb[CW][KGW][2][NP == 3 ? 2 : 1];
#pragma unroll
    for (int c = 0; c < CW; ++c)
#pragma unroll
    for (int kk = 0; kk < (EXT ? 0 : KGW); ++kk) {
        const int kg = wave + NW * kk;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const u32x4* src = a.vpack + ((((size_t)(ctw * CW + c) * a.nkg + kg) * 2 + ks) * 3) * 64 + lane;
            vb[c][kk][ks][0] = src[0];
            if constexpr (NP == 3) {
                vb[c][kk][ks][1] = src[64];
                vlo[wave][kk][ks][lane] = src[128];
            }
        }
    }

    float du_n[4], dw_n[4], u_t[4];
    if (tid < 16 * CW) {  // thread = (direction, column quad of the workgroup)
        const int dq = tid / (8 * CW), qx = tid % (8 * CW);
        const int cc = min(ctw * CW * CT + qx * 4, H - 4), dd = min(dq, a.dirs - 1);
        f32x4 c_al, c_be, c_a, c_b, c_gr;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            c_al[e] = clampf(a.alpha[cc + e], SP_ALPHA_LO, SP_ALPHA_HI);
            c_be[e] = ADAPT ? clampf(a.beta[cc + e], SP_BETA_LO, SP_BETA_HI) : 0.f;
            c_a[e] = ADAPT ? clampf(a.a[cc + e], SP_A_LO, SP_A_HI) : 0.f;
            c_b[e] = ADAPT ? clampf(a.b[cc + e], SP_B_LO, SP_B_HI) : 0.f;
            c_gr[e] = a.g_rate ? a.g_rate[(size_t)dd * H + cc + e] * a.g_rate_scale : 0.0f;
        }
        pcol[0][dq][qx] = c_al; pcol[1][dq][qx] = c_be; pcol[2][dq][qx] = c_a;
        pcol[3][dq][qx] = c_b; pcol[4][dq][qx] = c_gr;
        if (a.bn_x) { pcol[5][dq][qx] = ld4(a.bn_mean + cc); pcol[6][dq][qx] = ld4(a.bn_invstd + cc); }
    }
    if (pw && a.t_begin == 0) {
        first_tile[0][tid] = ld4(a.u0 + (size_t)bpc * H + colc);
        if (ADAPT) first_tile[1][tid] = ld4(a.w0 + (size_t)bpc * H + colc);
        first_tile[2][tid] = ld4(a.s0 + (size_t)bpc * H + colc);
    }
    const bool bn = a.bn_x != nullptr;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        du_n[e] = dw_n[e] = 0.f;
    }
    const size_t plane = (size_t)a.Bp * H;
    float* ws = a.dparam_ws + (size_t)bpc * H + colc;
    if (pw) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        f32x4 v_al = z4, v_be = z4, v_a = z4, v_b = z4;
        if (a.t_end < T) {  // resume a chunked pass: carried state + partial sums
            f32x4 v;
            v_al = ld4(ws);
            v = ld4(ws + 4 * plane); du_n[0] = v.x; du_n[1] = v.y; du_n[2] = v.z; du_n[3] = v.w;
            if (ADAPT) {
                v_be = ld4(ws + plane); v_a = ld4(ws + 2 * plane); v_b = ld4(ws + 3 * plane);
                v = ld4(ws + 5 * plane); dw_n[0] = v.x; dw_n[1] = v.y; dw_n[2] = v.z; dw_n[3] = v.w;
            }
        }
        PACC(0, tid) = v_al; PACC(1, tid) = v_be; PACC(2, tid) = v_a; PACC(3, tid) = v_b;
        if (bn) {
            PACC(4, tid) = a.t_end < T ? ld4(ws + 6 * plane) : z4;
            PACC(5, tid) = a.t_end < T ? ld4(ws + 7 * plane) : z4;
        }
    }
    {
        const f32x4 v = expand_saved(ld_saved_raw<S16>(a.u_save, ((size_t)bpc * T + (a.t_end - 1)) * H + colc));
        u_t[0] = v.x; u_t[1] = v.y; u_t[2] = v.z; u_t[3] = v.w;
        if (PARK && pw) park_u[a.t_end & 1][tid] = v;
    }
    if (tid < 2) abort_flag[tid] = 0;
    const unsigned my_xcc = xcc_id();
    xcd_agree((EXT || !a.xcd_tab) ? nullptr : (gu32*)a.xcd_tab + (size_t)rt * a.n_ct, SENTINEL, a.n_ct / CW, ctw, my_xcc, tid, &xcd_local_flag);
    __syncthreads();
    const bool xcd_local = xcd_local_flag != 0;  // this row tile's workgroups share an XCD: plain hand-off stores

    // hand-off ring: ring[slot][rt][ct] = one 4 KiB fp32 tile in fragment order; one buffer resource
    constexpr unsigned PT = (unsigned)ptile_bytes<NP>();  // bytes of one hand-off tile
    const unsigned slot_bytes = (unsigned)((size_t)a.n_rt_total * a.n_ct * PT);
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(a.ring, 0, (int)(RING * slot_bytes), 0x00020000);
    const unsigned rt_off = (unsigned)((size_t)rt * a.n_ct * PT);

    const bool drop = a.p_drop > 0.0f;
    const uint64_t seed = drop ? resolve_seed(a.seed) : 0;
    // The inputs of a reverse step (incoming gradient, raw projection for BatchNorm's sums, the saved u / w of the
    // step before) as FOUR UNCONDITIONAL loads (three without adaptation): every path through the time loop then
    // issues the same vector-memory instructions and hipcc's counted waits are exact.  (Rounds 1-2: `if (bn)`,
    // `if (t > 0) .. else u0`, and a width switch for bf16 saves around them — on the merged paths the wait-count
    // pass fell back to `vmcnt(0)`, and register reuse between the arms serialized the loads.)  Without BatchNorm
    // the host points `bn_src` at the incoming gradient (a readable range of the same extent); cell step 0 reads row 0
    // of the saves and is given u0 / w0 from LDS at its use; bf16 saves stay packed until then (`expand_saved`).
    typedef typename SavedRaw<S16>::type saved_raw;
    auto load_step = [&](int t, f32x4& g, saved_raw& up, saved_raw& wp, f32x4& xr) {
        const int tt = d ? (T - 1 - t) : t;
        const float* gp = a.g_out + ((size_t)b * T + tt) * HO + (size_t)d * H + colc;
        g = ld4s(gp);
        xr = ld4s(a.bn_src + ((size_t)b * T + tt) * H + colc);
        const size_t o = ((size_t)bpc * T + (size_t)max(t - 1, 0)) * H + colc;
        up = ld_saved_raw<S16>(a.u_save, o);
        if (ADAPT) wp = ld_saved_raw<S16>(a.w_save, o);
    };
    f32x4 g_nx = {0.f, 0.f, 0.f, 0.f}, xr_nx = g_nx;
    saved_raw up_nx = {}, wp_nx = {};
    if (pw) load_step(a.t_end - 1, g_nx, up_nx, wp_nx, xr_nx);
    vm_settled();  // every prologue load is in: the loop is entered with nothing outstanding
    // (Unlike the forward, holding this kernel's fp32 stores / prefetch back until after the tag poll does
    // not pay: measured 16.8k -> 18.2k cycles per step, the deferred traffic then competes with the tile loads.)
    PROF_DECL

    int t_stop = -1;  // the step an abort was seen at (diagnostics of the status word)
    for (int t = a.t_end - 1; t >= a.t_begin; --t) {
        PROF_STAMP(-1);
        const int pt = CW == 2 ? tid : (tid & 255);
        const f32x4 gv = g_nx, xrv = xr_nx;
        f32x4 upv = expand_saved(up_nx), wpv = expand_saved(wp_nx);
        if (t == 0) {  // cell step 0: the initial states (LDS; filled in the prologue when the launch ends at t = 0)
            upv = first_tile[0][pt];
            if (ADAPT) wpv = first_tile[1][pt];
        }
        if (PARK && pw) {
            park_u[t & 1][pt] = upv;
            if (ADAPT) park_wx[0][pt] = wpv;
            if (bn) park_wx[1][pt] = xrv;
        }
        float rec[4] = {0.f, 0.f, 0.f, 0.f};
        const int par = t & 1;
        // The next step's inputs (g, u, w, raw projection: HBM loads) are prefetched IN FRONT of this step's tile
        // loads.  (Build option REC_BWD_LATE_PREFETCH=1 issues them behind the last tile load instead, so that no
        // HBM round trip sits between the pointwise waves and their first k-group, `vmcnt` being in order —
        // measured 1.20 -> 1.26 ms per launch: the tiles have not landed at that point anyway, and the late loads
        // then arrive into the pointwise phase.)
#if REC_BWD_LATE_PREFETCH == 2
#elif REC_BWD_LATE_PREFETCH
        if (!(t + 1 < T && !EXT) && pw && t - 1 >= a.t_begin) load_step(t - 1, g_nx, up_nx, wp_nx, xr_nx);
#else
        if (pw && t - 1 >= a.t_begin) load_step(t - 1, g_nx, up_nx, wp_nx, xr_nx);
#endif
        // Everything of the reverse step that does not depend on the recurrent product — the dropout factor's
        // hash, the incoming gradient, alpha * du_{t+1}, the adaptation terms — is computed BEFORE the reduction
        // barrier, behind the first tile loads' issue (the pointwise waves would only wait there): the chain
        // behind the barrier is then `+ rec`, the box-car gate, two adds and a multiply.  Same operations in the
        // same order as the reference's autograd replay: bit-identical results.
        const int tt = d ? (T - 1 - t) : t;
        const size_t o_out = ((size_t)b * T + tt) * HO + (size_t)d * H + colc;
        float pre_ds[4], pre_aldu[4], pre_padw[4];
        auto pre_pointwise = [&]() __attribute__((always_inline)) {
            const f32x4 al = pcol[0][d][cqx], pa = pcol[2][d][cqx], pb = pcol[3][d][cqx], gr = pcol[4][d][cqx];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float k = drop ? keep_scale(seed, o_out + e, a.p_drop, a.inv_keep) : 1.0f;
                const float gs = (gv[e] + gr[e]) * k;
                pre_aldu[e] = al[e] * du_n[e];
                float ds = gs - pre_aldu[e];
                pre_padw[e] = ADAPT ? pa[e] * dw_n[e] : 0.f;
                if (ADAPT) ds = ds + pb[e] * dw_n[e];
                pre_ds[e] = ds;
                asm volatile("" : "+v"(pre_ds[e]), "+v"(pre_aldu[e]), "+v"(pre_padw[e]));  // not sunk behind the barrier
            }
        };
        if (!(t + 1 < T && !EXT) && pw) pre_pointwise();

        if (t + 1 < T && !EXT) {
            // ---- the dWx_{t+1} tiles of this wave's producers: load, re-load what has not landed yet
            const unsigned slot = (unsigned)((t + 1) % RING);
            const unsigned base = slot * slot_bytes + rt_off + (unsigned)lane * 16u;
            // The CU's fill rate from L2 (~70 GB/s: 128 KiB of tiles per step take ~3.7 k cycles) holds a wave
            // in the ISSUE of its loads once the memory pipeline is full, so the issue is spread out: two
            // k-groups ahead of the one being multiplied, the rest of the loads interleaved with the MFMAs
            // (the partner wave on the SIMD computes while this one is stuck issuing).
            // Tried on top (round 2, not kept): splitting group kk+1 INSIDE the MFMAs of group kk (software
            // pipeline with sched_group_barrier 1 MFMA : 8 VALU, and hand-placed half-pairs between the MFMAs
            // with scheduling barriers) — hipcc hoists the ~90 split instructions above the MFMAs in every block
            // that ends in the next group's branch, and the launch time did not move (1.37 vs 1.34-1.39 ms); pinned
            // with one asm statement per MFMA + 5-6 split instructions (bit-exact, 255 VGPRs) it was 1.39-1.43 ms:
            // the SIMD's issue port is NOT what bounds the phase.
            // With the producers' data always ready (consuming step t+2's tiles, a timing experiment) the step is
            // 9.7 k cycles against 10.2 k: the phase is throughput-, not hand-off-latency-bound — L2 port
            // 3.7 k, and MFMA (3.1 k) + split VALU (2.8 k) add up on the SIMD instead of overlapping.
            // (bf16 operand mode, NP = 1: a third of the bytes — all k-groups are issued at once; one group ahead
            // made that mode's tile phase four sequential L2 round trips, 3.2 k cycles with 16 MFMAs per SIMD)
            constexpr int AHEAD = NP == 1 ? KGW : (KGW < REC_AHEAD ? KGW : REC_AHEAD);
            u32x4 raw[KGW][2][NP];  // [k-group][k16-step][plane]: MFMA A fragments as they come off the wire
#if REC_BWD_PROBE
            // PROBE, then burst (round 3).  A tile load issued before its producer's stores have landed returns
            // sentinels and costs a second, serialized round trip in `settle_ptile` — which is why more than one
            // k-group in flight at the step's start only lost time (round 2: 10.3 k -> 11.4 k cycles per step with
            // all loads up front; 1.04 -> 1.34 ms per launch with two groups ahead): the early groups were
            // speculative.  So a wave first watches ONE dword per sample: lane (kk, j) reads the first word that
            // producer thread (row 2j+1, column quad j & 7) of tile kk stores into the LAST plane — 16 samples per
            // tile from all four producer waves, 4 cache lines per probe instruction instead of 6 KiB per k-group —
            // and issues the tile loads only when every sample has landed.  The full sentinel check of
            // `settle_ptile` stays: a piece that is still missing then is re-loaded as before.
            if (pw) pre_pointwise();
            {
                const int pk = lane >> 4, pj = lane & 15, prow = 2 * pj + 1, pcq = pj & 7;
                const int pkg = wave + NW * pk;
                const unsigned poff = (pk < KGW && pkg < a.n_ct)
                    ? slot * slot_bytes + rt_off + (unsigned)pkg * PT +
                      (unsigned)((((pcq >> 2) * NP + (NP - 1)) * 64 + ((pcq >> 1) & 1) * 32 + prow) * 16 + (pcq & 1) * 8)
                    : 0xFFFFFF00u;  // beyond the buffer resource: reads 0, never "missing"
                const u64 t_start = __builtin_amdgcn_s_memrealtime();
                for (unsigned spins = 0;; ++spins) {
                    const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(rsrc, poff, 0, REC_LD_AUX);
                    if (__all(v != SENTINEL)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if ((spins & 63u) == 63u && __builtin_amdgcn_s_memrealtime() - t_start > TIMEOUT_TICKS) {
                        lds_flag_set(&abort_flag[par]);
                        break;
                    }
                }
            }
#endif
#if REC_BWD_UPPER_DELAY
            // (experiment: the waves without pointwise state reach this point a few hundred cycles before the others;
            // if their first k-group is requested before the peers' tiles have landed it costs them a second round
            // trip, and the pointwise waves then wait for them at the reduction barrier)
            if (NW == 8 && CW == 1 && !pw_wave) __builtin_amdgcn_s_sleep(REC_BWD_UPPER_DELAY);
#endif
#pragma unroll
            for (int kk = 0; kk < AHEAD; ++kk) issue_ptile<NW, NP>(raw[kk], rsrc, base, wave + NW * kk, a.n_ct);
            PROF_STAMP(0);  // first tile load issue
#if !REC_BWD_PROBE && REC_BWD_LATE_PREFETCH != 3
            if (pw) pre_pointwise();
#endif
            f32x16 acc[CW];
#pragma unroll
            for (int c = 0; c < CW; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
#pragma unroll
            for (int kk = 0; kk < KGW; ++kk) {
                // k-group by k-group: the MFMAs of one run while the loads of the next are still in flight
                // (scheduling barriers: hipcc otherwise hoists the next group's check, and its wait, into this
                // group's MFMAs)
                __builtin_amdgcn_sched_barrier(0);
                settle_ptile<NP>(raw[kk], rsrc, base + (unsigned)(wave + NW * kk) * PT, &abort_flag[par]);
                if (kk + AHEAD < KGW) issue_ptile<NW, NP>(raw[kk + AHEAD], rsrc, base, wave + NW * (kk + AHEAD), a.n_ct);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    if constexpr (NP == 1) {  // bf16 operand mode: bf16(dWx) x bf16(V^T), one product
#pragma unroll
                        for (int c = 0; c < CW; ++c) acc[c] = mfma_bf16(raw[kk][ks][0], vb[c][kk][ks][0], acc[c]);
                    } else {
                    const u32x4 p1 = raw[kk][ks][0], p2 = raw[kk][ks][NP - 2], p3 = raw[kk][ks][NP - 1];
                    const u32x4 vl = vlo[wave][kk][ks][lane];
                    // six largest cross terms of (t1+t2+t3)(V_hi+V_mid+V_lo), small first.  Dropped:
                    // t3*mid (|t3| < 2^-14 |x| after two 8-bit truncations, |V_mid| <= 2^-8 |V|: <= 2^-22 of
                    // |x||V|), t2*lo (<= 2^-23) and t3*lo.  Worst-case per product; summed over a row the
                    // dropped part measures 3e-9 of sum|x||V| (7 terms: 1.4e-9; an fp32 sgemm's own
                    // rounding: 1e-7) — the same kind of cut as the dense 6-term GEMM.
#ifndef REC_BWD_ABL3  // (timing ablation, wrong results: the three largest terms only)
                    acc[0] = mfma_bf16(p2, vb[0][kk][ks][NP == 3], acc[0]);  // t2*mid
                    acc[0] = mfma_bf16(p3, vb[0][kk][ks][0], acc[0]);  // t3*hi
                    acc[0] = mfma_bf16(p1, vl, acc[0]);       // t1*lo
#else
                    asm volatile("" ::"v"(p3), "v"(vl));
#endif
                    acc[0] = mfma_bf16(p2, vb[0][kk][ks][0], acc[0]);  // t2*hi
                    acc[0] = mfma_bf16(p1, vb[0][kk][ks][NP == 3], acc[0]);  // t1*mid
                    acc[0] = mfma_bf16(p1, vb[0][kk][ks][0], acc[0]);  // t1*hi
                    }
                }
            }
#if REC_BWD_LATE_PREFETCH == 1
            if (pw && t - 1 >= a.t_begin) load_step(t - 1, g_nx, up_nx, wp_nx, xr_nx);
#endif
#if REC_BWD_LATE_PREFETCH == 3
            if (pw) pre_pointwise();
#endif
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                float* rd = red[wave][c];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
                    rd[row * RED_LD4 + li] = acc[c][i];
                }
            }
            PROF_STAMP(1);  // per k-group: wait, split, MFMA; LDS write
        }
        if (xcd_local && xcc_id() != my_xcc) lds_flag_set(&abort_flag[par]);  // moved to another XCD
        lds_barrier();
        vm_settled();  // tile loads are in, last step's stores and this step's prefetch long complete
        PROF_STAMP(2);  // barrier
        if (lds_flag_read(&abort_flag[par])) { t_stop = t; break; }
#if REC_BWD_LATE_PREFETCH == 2
        if (pw && t - 1 >= a.t_begin) load_step(t - 1, g_nx, up_nx, wp_nx, xr_nx);
#endif
#ifndef REC_NO_RESET
        if (BXS && REC_BWD_UPPER_RESET && !pw) {
            // The sentinels go back into this workgroup's tile of step t+2 (every peer has consumed it: see the
            // header) from the waves WITHOUT pointwise state, which idle from here to the publish barrier — as
            // 16-byte stores, one or two per thread, instead of three 8-byte stores per pointwise thread between
            // the publish stores and the publish barrier, i.e. on the step's critical chain (round 3).  These
            // waves' `vm_settled()` behind the next reduction barrier retires them two barriers before the slot
            // is written again (the publish of step t-2).
            const u32x4 sent4 = {SENTINEL, SENTINEL, SENTINEL, SENTINEL};
            const unsigned so = (unsigned)((t + 2) % RING) * slot_bytes + rt_off + (unsigned)ct * PT;
            constexpr int PIECES = NP * 128;  // 16-byte pieces of one tile
#pragma unroll
            for (int q = 0; q < (PIECES + 255) / 256; ++q) {
                const int piece = (tid - 256) + 256 * q;
                // (nothing to reset: an offset beyond the buffer resource — chosen per piece, a base of 0xFFFFF000
                // plus 16 * piece would wrap around into the ring's first tile)
                const unsigned o = (t + 2 < T && piece < PIECES) ? so + (unsigned)piece * 16u : 0xFFFFF000u;
                if (xcd_local) __builtin_amdgcn_raw_buffer_store_b128(sent4, rsrc, o, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b128(sent4, rsrc, o, 0, REC_ST_AUX);
            }
        }
#endif
        if (BXS && valid_hi && t + 1 < a.t_end) {  // the previous step's staged outputs -> HBM (upper waves)
            const int t1 = t + 1, tt1 = d ? (T - 1 - t1) : t1;
            st4(a.dWx + ((size_t)bp * T + tt1) * H + col, stage_dwx[tid & 255]);
            st2(a.s_prev16 + ((size_t)bp * T + tt1) * H + col, stage_sp[tid & 255]);
        }
        // From here to the publish barrier: the waves that own pointwise state only (a WAVE-UNIFORM branch, round 3
        // — until then the upper four waves ran the reduction reads and the whole reverse step on dead values and
        // took vector issue slots and LDS cycles from the pointwise waves they share a SIMD with).
        float sp[4] = {0.f, 0.f, 0.f, 0.f}, du_new[4] = {0.f, 0.f, 0.f, 0.f}, dw_new[4] = {0.f, 0.f, 0.f, 0.f};
        f32x4 dwx = {0.f, 0.f, 0.f, 0.f}, spv = dwx;
        PROF_STAMP(6);  // behind the barrier: settle, flag read, prefetch issue, staged stores
        f32x4 up_use = upv, ut_use = {u_t[0], u_t[1], u_t[2], u_t[3]};
        if (PARK && pw) { up_use = park_u[t & 1][pt]; ut_use = park_u[(t + 1) & 1][pt]; }
        if (pw_wave) {
        if (t + 1 < T && EXT) {
            const f32x4 v = ld4(a.rec0 + (size_t)bpc * H + colc);
            rec[0] = v.x; rec[1] = v.y; rec[2] = v.z; rec[3] = v.w;
        } else if (t + 1 < T) {
            f32x4 sum = *reinterpret_cast<const f32x4*>(&red[0][cj][r * RED_LD4 + cq * 4]);  // (see the forward)
#pragma unroll
            for (int w = 1; w < NW; ++w) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(&red[w][cj][r * RED_LD4 + cq * 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) sum[e] = sum[e] + v[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) rec[e] = sum[e];
        }

        PROF_STAMP(7);  // partial-tile reduction
        // ---- pointwise reverse step (its rec-independent part: pre_pointwise above)
        if (t > 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) sp[e] = (up_use[e] - a.theta) > 0.0f ? 1.0f : 0.0f;
        } else {
            const f32x4 v = first_tile[2][pt];
            sp[0] = v.x; sp[1] = v.y; sp[2] = v.z; sp[3] = v.w;
        }
        const f32x4 al = pcol[0][d][cqx], be = pcol[1][d][cqx];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ds = pre_ds[e] + rec[e];
            const float xs = ut_use[e] - a.theta;
            float du = boxcar_gate(ds, xs) + pre_aldu[e];                     // snns.py:33-35
            if (ADAPT) du = du + pre_padw[e];
            dwx[e] = valid ? (1.0f - al[e]) * du : 0.0f;
            du_new[e] = du;
            dw_new[e] = ADAPT ? be[e] * dw_n[e] - dwx[e] : 0.f;
            spv[e] = (t > 0) ? sp[e] : 0.0f;  // binary rows only: the s0 term of dV is added by the host
        }
        PROF_STAMP(8);  // reverse-step arithmetic
        // ---- publish dWx_t first: this thread's 4 values are one 16-byte piece of the tile in fragment
        //      order (columns cq*4.. -> k16-step ks = cq>>2, k-half h = (cq>>1)&1, quad q = cq&1), one
        //      write-through store; then the sentinel goes back into the slot of step t+2 (see header)
        if (EXT) {
            if (valid) st4(a.dwx_step + (size_t)bp * H + col, dwx);  // operand of the caller's dWx_t @ V^T product
        } else if (pw) {
            // this thread's 4 values = k 4 cq .. +3 of its row: k16-step cq >> 2, k-half (cq >> 1) & 1, first or
            // second 8 bytes of that piece; the three planes of a piece are 1 KiB apart
            const unsigned piece = (unsigned)(((cq >> 2) * NP * 64 + ((cq >> 1) & 1) * 32 + r) * 16 + (cq & 1) * 8);
            const unsigned tile_off = rt_off + (unsigned)ct * PT + piece;
            {
                u32x2 w[3];
                if constexpr (NP == 1) {  // one nearest-even rounding (v_cvt_pk_bf16_f32)
                    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                    w[0].x = __builtin_bit_cast(unsigned, bf16x2{(__bf16)dwx[0], (__bf16)dwx[1]});
                    w[0].y = __builtin_bit_cast(unsigned, bf16x2{(__bf16)dwx[2], (__bf16)dwx[3]});
                } else {
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const unsigned x0 = __float_as_uint(dwx[2 * pr]), x1 = __float_as_uint(dwx[2 * pr + 1]);
                    const float r0 = dwx[2 * pr] - __uint_as_float(x0 & 0xFFFF0000u);
                    const float r1 = dwx[2 * pr + 1] - __uint_as_float(x1 & 0xFFFF0000u);
                    const unsigned y0 = __float_as_uint(r0), y1 = __float_as_uint(r1);
                    const float q0 = r0 - __uint_as_float(y0 & 0xFFFF0000u);
                    const float q1 = r1 - __uint_as_float(y1 & 0xFFFF0000u);
                    w[0][pr] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
                    w[1][pr] = __builtin_amdgcn_perm(y1, y0, 0x07060302u);
                    w[2][pr] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
                }
                }
                // (no branch around the stores: where there is nothing to publish — cell step 0 — or to reset, the
                // offset lies beyond the buffer resource and the hardware drops the store; the instruction count
                // of a step is then the same on every path)
                const unsigned so = t > 0 ? (unsigned)(t % RING) * slot_bytes + tile_off : 0xFFFFF000u;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    if (xcd_local) __builtin_amdgcn_raw_buffer_store_b64(w[p], rsrc, so + (unsigned)(p * 1024), 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b64(w[p], rsrc, so + (unsigned)(p * 1024), 0, REC_ST_AUX);
                }
            }
#ifndef REC_NO_RESET
            if (!(BXS && REC_BWD_UPPER_RESET)) {  // (8-wave kernels: the upper waves put the sentinels back, see below)
                const u32x2 sent = {SENTINEL, SENTINEL};
                const unsigned so = t + 2 < T ? (unsigned)((t + 2) % RING) * slot_bytes + tile_off : 0xFFFFF000u;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    if (xcd_local) __builtin_amdgcn_raw_buffer_store_b64(sent, rsrc, so + (unsigned)(p * 1024), 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b64(sent, rsrc, so + (unsigned)(p * 1024), 0, REC_ST_AUX);
                }
            }
#endif
        }
        }
#if REC_BWD_LATE_PREFETCH == 4
        if (pw && t - 1 >= a.t_begin) load_step(t - 1, g_nx, up_nx, wp_nx, xr_nx);
#endif
        PROF_STAMP(3);  // pointwise + tile store issue
#ifndef REC_NO_PUBLISH_BARRIER
        lds_barrier();  // the non-pointwise waves start polling only once this workgroup's own tile is on its way
#endif
        PROF_STAMP(4);  // publish barrier
#if REC_BWD_LATE_PREFETCH == 3
        if (pw && t - 1 >= a.t_begin) load_step(t - 1, g_nx, up_nx, wp_nx, xr_nx);
#endif
        // ---- off the critical path: fp32 outputs for the following GEMMs, parameter partial sums
#if defined(SPARCH_REC_PROF) && defined(BA_NO_BULK)  // timing ablation (no outputs): the step's HBM stores dropped
        if (false) {
#else
        if (valid) {
#endif
            u32x2 h;  // s_{t-1} (binary for t >= 1, zero row at t = 0) as a bf16 plane for the dV product
            h.x = (spv[0] != 0.f ? 0x3F80u : 0u) | (spv[1] != 0.f ? 0x3F800000u : 0u);
            h.y = (spv[2] != 0.f ? 0x3F80u : 0u) | (spv[3] != 0.f ? 0x3F800000u : 0u);
            if (BXS) {
                stage_dwx[tid] = dwx; stage_sp[tid] = h;
            } else {
                st4(a.dWx + ((size_t)bp * T + tt) * H + col, dwx);
                st2(a.s_prev16 + ((size_t)bp * T + tt) * H + col, h);
            }
        }
        if (pw) {
            f32x4 wp_use = wpv, xr_use = xrv;
            if (PARK) { if (ADAPT) wp_use = park_wx[0][pt]; if (bn) xr_use = park_wx[1][pt]; }
            f32x4 v_al = PACC(0, pt), v_be, v_a, v_b;
            if (ADAPT) { v_be = PACC(1, pt); v_a = PACC(2, pt); v_b = PACC(3, pt); }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float q = up_use[e] - sp[e];
                v_al[e] += du_new[e] * (q - ut_use[e]);  // x 1/(1-alpha) once, at the end
                if (ADAPT) {
                    v_be[e] += dw_new[e] * wp_use[e];
                    v_a[e] += dw_new[e] * up_use[e];
                    v_b[e] += dw_new[e] * sp[e];
                }
            }
            PACC(0, pt) = v_al;
            if (ADAPT) { PACC(1, pt) = v_be; PACC(2, pt) = v_a; PACC(3, pt) = v_b; }
            if (bn) {  // BatchNorm backward's column sums (dy = dWx, xhat = (x - mean) * invstd)
                f32x4 v_dy = PACC(4, pt), v_dyx = PACC(5, pt);
                const f32x4 mu = pcol[5][d][cqx], is = pcol[6][d][cqx];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v_dy[e] += dwx[e];
                    v_dyx[e] += dwx[e] * ((xr_use[e] - mu[e]) * is[e]);
                }
                PACC(4, pt) = v_dy; PACC(5, pt) = v_dyx;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (ADAPT) dw_n[e] = dw_new[e];
            du_n[e] = du_new[e];
            u_t[e] = PARK ? 0.f : upv[e];
        }
        PROF_STAMP(5);  // fp32 stores + partial sums
    }
    PROF_FLUSH(1)
    if (BXS) {  // the launch's last step is still staged (after an abort the step is discarded anyway)
        const bool aborted = (lds_flag_read(&abort_flag[0]) | lds_flag_read(&abort_flag[1])) != 0;
        __syncthreads();
#if !(defined(SPARCH_REC_PROF) && defined(BA_NO_BULK))
        if (!aborted && valid_hi && a.t_end > a.t_begin) {
            const int t1 = a.t_begin, tt1 = d ? (T - 1 - t1) : t1;
            st4(a.dWx + ((size_t)bp * T + tt1) * H + col, stage_dwx[tid & 255]);
            st2(a.s_prev16 + ((size_t)bp * T + tt1) * H + col, stage_sp[tid & 255]);
        }
#endif
    }
    if (tid == 0 && (lds_flag_read(&abort_flag[0]) | lds_flag_read(&abort_flag[1])))
        status_raise(a.status, SPARCH_STATUS_REC_BWD, t_stop);

    if (valid) {
        f32x4 v = PACC(0, tid);
        if (a.t_begin == 0) {  // last chunk of the pass: d u_t / d alpha = (q - u_t) / (1 - alpha)
            const f32x4 al = pcol[0][d][cqx];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] / (1.0f - al[e]);
        }
        st4(ws, v);
        v.x = du_n[0]; v.y = du_n[1]; v.z = du_n[2]; v.w = du_n[3]; st4(ws + 4 * plane, v);
        if (ADAPT) {
            st4(ws + plane, PACC(1, tid));
            st4(ws + 2 * plane, PACC(2, tid));
            st4(ws + 3 * plane, PACC(3, tid));
            v.x = dw_n[0]; v.y = dw_n[1]; v.z = dw_n[2]; v.w = dw_n[3]; st4(ws + 5 * plane, v);
        }
        if (bn) {
            st4(ws + 6 * plane, PACC(4, tid));
            st4(ws + 7 * plane, PACC(5, tid));
        }
    }
}


// ------------------------------------------------------------------ dense recurrent cell (ANN baselines, f-4)
// RNNLayer._rnn_cell (anns.py:328-339): y_t = act(Wx_t + y_{t-1} V^T), y_{-1} = 0, and its reverse pass
//     dpre_t = (g_t + dpre_{t+1} V) * act'(y_t).
// Both directions are the same machine as the spiking backward above: per step a workgroup contracts the
// PREVIOUS step's dense fp32 row tile (handed over through the tagged ring, split exactly into three bf16
// planes by the consumer) with its resident slice of V (six cross terms), then applies a pointwise rule and
// publishes its own 32 x 32 tile.  `s` counts steps in processing order (forward: t = s, backward:
// t = T-1-s); the tile of step s goes to ring slot s % RING (sentinel protocol of the spiking backward).
struct AnnArgs {
    int B, dirs, T, H, Bp;
    int n_ct, nkg, n_rt_total;
    int rt_base, n_rt_launch;
    int s_begin, s_end;
    const float* Wx; const float* scale; const float* shift;
    const u32x4* vpack;
    float p_drop, inv_keep; uint64_t seed;
    float* y_state; float* y_out;                  // forward outputs
    const float* g_out; const float* y_in;         // backward inputs (y_in = the forward's y_state)
    float* dpre; float* y_prev;                    // backward outputs
    char* ring; unsigned* status;
    unsigned* xcd_tab;  // agreement table of the XCD-local stores (whole-sequence launches), or null
    // step variant (EXT): this ONE step's recurrent product comes from the caller (Bp,H); the step's y / dpre
    // also goes to a contiguous (Bp,H) buffer, the operand of the caller's next product
    const float* rec_ext; float* step_out;
};

__device__ __forceinline__ float ann_act(int kind, float v) {
    if (kind == SPARCH_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    if (kind == SPARCH_ACT_RELU) return fmaxf(v, 0.0f);
    return tanhf(v);
}
__device__ __forceinline__ float ann_dact(int kind, float a) {  // through the activation's output
    if (kind == SPARCH_ACT_SIGMOID) return a * (1.0f - a);
    if (kind == SPARCH_ACT_RELU) return a > 0.0f ? 1.0f : 0.0f;
    return 1.0f - a * a;
}

template <int ACT, bool BWD, int KGW, int NW, bool EXT = false>
__global__ __launch_bounds__(64 * NW, 1) void ann_rec_kernel(AnnArgs a) {
    __shared__ __attribute__((aligned(16))) float red[2][NW][RT * RED_LD];
    __shared__ __attribute__((aligned(16))) u32x4 vlo[NW][KGW][2][64];
    __shared__ int abort_flag[2];
    __shared__ int xcd_local_flag;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hh = lane >> 5;
    const int rt = a.rt_base + (int)(blockIdx.x % a.n_rt_launch);
    const int ct = (int)(blockIdx.x / a.n_rt_launch);
    const int T = a.T, H = a.H, HO = a.H * a.dirs;

    const bool pw = tid < 256;
    const int r = (tid & 255) >> 3, cq = tid & 7;
    const int bp = rt * RT + r, col = ct * CT + cq * 4;
    const bool valid = pw && bp < a.Bp && col < H;
    const int bpc = min(bp, a.Bp - 1), colc = min(col, H - 4);
    const int d = bpc / a.B, b = bpc - d * a.B;

    u32x4 vb[KGW][2][2];
#pragma unroll
    for (int kk = 0; kk < (EXT ? 0 : KGW); ++kk) {
        const int kg = wave + NW * kk;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const u32x4* src = a.vpack + ((((size_t)ct * a.nkg + kg) * 2 + ks) * 3) * 64 + lane;
            vb[kk][ks][0] = src[0];
            vb[kk][ks][1] = src[64];
            vlo[wave][kk][ks][lane] = src[128];
        }
    }
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (!BWD && a.scale) { sc = ld4(a.scale + colc); sh = ld4(a.shift + colc); }
    if (tid < 2) abort_flag[tid] = 0;
    const unsigned my_xcc = xcc_id();
    xcd_agree((EXT || !a.xcd_tab) ? nullptr : (gu32*)a.xcd_tab + (size_t)rt * a.n_ct, SENTINEL, a.n_ct, ct, my_xcc, tid, &xcd_local_flag);
    __syncthreads();
    const bool xcd_local = xcd_local_flag != 0;  // this row tile's workgroups share an XCD: plain hand-off stores

    // hand-off tiles as pre-split bf16 planes (6 KB), as in the spiking backward
    const unsigned slot_bytes = (unsigned)((size_t)a.n_rt_total * a.n_ct * PTILE_BYTES);
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(a.ring, 0, (int)(RING * slot_bytes), 0x00020000);
    const unsigned rt_off = (unsigned)((size_t)rt * a.n_ct * PTILE_BYTES);
    const bool drop = a.p_drop > 0.0f;
    const uint64_t seed = drop ? resolve_seed(a.seed) : 0;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // per-step operands of the pointwise rule: forward x = Wx[b, tt]; backward g = g_out[b, tt, d*H..],
    // y_t and y_{t-1} of this row in cell time
    auto load_step = [&](int s, f32x4& v0, f32x4& v1, f32x4& v2) {
        const int t = BWD ? (T - 1 - s) : s;
        const int tt = d ? (T - 1 - t) : t;
        if (!BWD) {
            v0 = ld4(a.Wx + ((size_t)b * T + tt) * H + colc);
        } else {
            v0 = ld4(a.g_out + ((size_t)b * T + tt) * HO + (size_t)d * H + colc);
            v1 = ld4(a.y_in + ((size_t)bpc * T + t) * H + colc);
            v2 = t > 0 ? ld4(a.y_in + ((size_t)bpc * T + (t - 1)) * H + colc) : zero4;
        }
    };
    f32x4 n0 = zero4, n1 = zero4, n2 = zero4;
    if (pw) load_step(a.s_begin, n0, n1, n2);

    for (int s = a.s_begin; s < a.s_end; ++s) {
        const f32x4 c0 = n0, c1 = n1, c2 = n2;
        float rec[4] = {0.f, 0.f, 0.f, 0.f};
        const int par = s & 1;
        if (pw && s + 1 < a.s_end) load_step(s + 1, n0, n1, n2);

        if (s > 0 && !EXT) {
            const unsigned slot = (unsigned)((s - 1) % RING);
            const unsigned base = slot * slot_bytes + rt_off + (unsigned)lane * 16u;
            constexpr int AHEAD = KGW < REC_AHEAD ? KGW : REC_AHEAD;  // see rec_bwd_kernel
            u32x4 raw[KGW][2][3];  // [k-group][k16-step][plane]
#pragma unroll
            for (int kk = 0; kk < AHEAD; ++kk) issue_ptile<NW>(raw[kk], rsrc, base, wave + NW * kk, a.n_ct);
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int kk = 0; kk < KGW; ++kk) {
                __builtin_amdgcn_sched_barrier(0);
                settle_ptile(raw[kk], rsrc, base + (unsigned)(wave + NW * kk) * PTILE_BYTES, &abort_flag[par]);
                if (kk + AHEAD < KGW) issue_ptile<NW>(raw[kk + AHEAD], rsrc, base, wave + NW * (kk + AHEAD), a.n_ct);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const u32x4 p1 = raw[kk][ks][0], p2 = raw[kk][ks][1], p3 = raw[kk][ks][2];
                    const u32x4 vl = vlo[wave][kk][ks][lane];
                    acc = mfma_bf16(p2, vb[kk][ks][1], acc);  // t2*mid
                    acc = mfma_bf16(p3, vb[kk][ks][0], acc);  // t3*hi
                    acc = mfma_bf16(p1, vl, acc);             // t1*lo
                    acc = mfma_bf16(p2, vb[kk][ks][0], acc);  // t2*hi
                    acc = mfma_bf16(p1, vb[kk][ks][1], acc);  // t1*mid
                    acc = mfma_bf16(p1, vb[kk][ks][0], acc);  // t1*hi
                }
            }
            float* rd = red[par][wave];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
                rd[row * RED_LD + li] = acc[i];
            }
        }
        lds_barrier();
        vm_settled();
        if (lds_flag_read(&abort_flag[par])) break;
        if (s > 0 && EXT) {
            const f32x4 v = ld4(a.rec_ext + (size_t)bpc * H + colc);
            rec[0] = v.x; rec[1] = v.y; rec[2] = v.z; rec[3] = v.w;
        } else if (s > 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int o = r * RED_LD + cq * 4 + e;
                float sum = red[par][0][o];
#pragma unroll
                for (int w = 1; w < NW; ++w) sum = sum + red[par][w][o];
                rec[e] = sum;
            }
        }

        // ---- pointwise rule
        const int t = BWD ? (T - 1 - s) : s;
        const int tt = d ? (T - 1 - t) : t;
        const size_t o_out = ((size_t)b * T + tt) * HO + (size_t)d * H + colc;
        f32x4 val, aux;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float k = drop ? keep_scale(seed, o_out + e, a.p_drop, a.inv_keep) : 1.0f;
            if (!BWD) {
                float xn = c0[e];
                if (a.scale) xn = bn_affine(xn, sc[e], sh[e]);
                const float y = ann_act(ACT, xn + rec[e]);                      // anns.py:336
                val[e] = valid ? y : 0.0f;
                aux[e] = y * k;                                                 // dropout after the cell (323-324)
            } else {
                const float dp = (c0[e] * k + rec[e]) * ann_dact(ACT, c1[e]);
                val[e] = valid ? dp : 0.0f;
                aux[e] = c2[e];
            }
        }
        // ---- publish this step's tile (fragment order, write-through); put the sentinel back into the slot
        //      of step s-2 (every peer has consumed it: they have all published step s-1 since)
        if (EXT) {
            if (valid) st4(a.step_out + (size_t)bp * H + col, val);
        } else if (pw) {
            const unsigned piece = (unsigned)(((cq >> 2) * 3 * 64 + ((cq >> 1) & 1) * 32 + r) * 16 + (cq & 1) * 8);
            const unsigned tile_off = rt_off + (unsigned)ct * PTILE_BYTES + piece;
            if (s + 1 < T) {
                u32x2 w[3];
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const unsigned x0 = __float_as_uint(val[2 * pr]), x1 = __float_as_uint(val[2 * pr + 1]);
                    const float r0 = val[2 * pr] - __uint_as_float(x0 & 0xFFFF0000u);
                    const float r1 = val[2 * pr + 1] - __uint_as_float(x1 & 0xFFFF0000u);
                    const unsigned y0 = __float_as_uint(r0), y1 = __float_as_uint(r1);
                    const float q0 = r0 - __uint_as_float(y0 & 0xFFFF0000u);
                    const float q1 = r1 - __uint_as_float(y1 & 0xFFFF0000u);
                    w[0][pr] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
                    w[1][pr] = __builtin_amdgcn_perm(y1, y0, 0x07060302u);
                    w[2][pr] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
                }
                const unsigned so = (unsigned)(s % RING) * slot_bytes + tile_off;
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    if (xcd_local) __builtin_amdgcn_raw_buffer_store_b64(w[p], rsrc, so + (unsigned)(p * 1024), 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b64(w[p], rsrc, so + (unsigned)(p * 1024), 0, REC_ST_AUX);
                }
            }
            if (s >= 2) {
                const u32x2 sent = {SENTINEL, SENTINEL};
                const unsigned so = (unsigned)((s - 2) % RING) * slot_bytes + tile_off;
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    if (xcd_local) __builtin_amdgcn_raw_buffer_store_b64(sent, rsrc, so + (unsigned)(p * 1024), 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b64(sent, rsrc, so + (unsigned)(p * 1024), 0, REC_ST_AUX);
                }
            }
        }
        lds_barrier();
        // ---- off the critical path: outputs
        if (valid) {
            if (!BWD) {
                st4(a.y_state + ((size_t)bp * T + t) * H + col, val);
                st4(a.y_out + ((size_t)b * T + tt) * HO + (size_t)d * H + col, aux);
            } else {
                st4(a.dpre + ((size_t)bp * T + tt) * H + col, val);
                st4(a.y_prev + ((size_t)bp * T + tt) * H + col, aux);
            }
        }
    }
    if (tid == 0 && (lds_flag_read(&abort_flag[0]) | lds_flag_read(&abort_flag[1])))
        status_raise(a.status, BWD ? SPARCH_STATUS_ANN_REC_BWD : SPARCH_STATUS_ANN_REC_FWD, -1);
}

// ------------------------------------------------------------------------------ V prepack
// vpack[ct][kg][ks][p][lane] = 8 bf16 (16 B): plane p (0 hi, 1 mid, 2 lo) of Vm[k][col] (forward) or
// Vm[col][k] (backward) for k = kg*32 + 16*ks + 8*(lane>>5) + j, j = 0..7, col = ct*32 + (lane&31);
// Vm = V with a zero diagonal, zero padded.  This is the B-operand fragment of v_mfma_f32_32x32x16_bf16.
__global__ void vpack_kernel(int H, int n_ct, int nkg, int transpose, int rne, const float* __restrict__ V,
                             u32x4* __restrict__ vpack) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)n_ct * nkg * 2 * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63), ks = (int)((idx >> 6) & 1);
    const int kg = (int)((idx >> 7) % nkg), ct = (int)((idx >> 7) / nkg);
    const int col = ct * 32 + (lane & 31);
    const bool tr = transpose & 1, keep_diag = transpose & 2;  // bit 1: dense matrix of an ANN cell, no mask
    unsigned short pl[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = kg * 32 + 16 * ks + 8 * (lane >> 5) + j;
        float v = 0.f;
        if (k < H && col < H && (keep_diag || k != col)) v = tr ? V[(size_t)col * H + k] : V[(size_t)k * H + col];
        vsplit(v, rne, pl[0][j], pl[1][j], pl[2][j]);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = (unsigned)pl[p][2 * q] | ((unsigned)pl[p][2 * q + 1] << 16);
        vpack[((((size_t)ct * nkg + kg) * 2 + ks) * 3 + p) * 64 + lane] = o;
    }
}
// forward fragments, backward (transposed) fragments and the masked copy of a spiking layer's V in ONE launch (a
// training step needs all three; three launches were ~25 us of launch + queue gaps per layer)
__global__ void vpack_both_kernel(int H, int n_ct, int nkg, int rne, const float* __restrict__ V, u32x4* __restrict__ vpack_f,
                                  u32x4* __restrict__ vpack_b, float* __restrict__ Vm) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)n_ct * nkg * 2 * 64;
    if (idx < total) {
        const int lane = (int)(idx & 63), ks = (int)((idx >> 6) & 1);
        const int kg = (int)((idx >> 7) % nkg), ct = (int)((idx >> 7) / nkg);
        const int col = ct * 32 + (lane & 31);
        unsigned short pf[3][8], pb[3][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = kg * 32 + 16 * ks + 8 * (lane >> 5) + j;
            const bool in = k < H && col < H && k != col;
            vsplit(in ? V[(size_t)k * H + col] : 0.f, rne, pf[0][j], pf[1][j], pf[2][j]);
            vsplit(in ? V[(size_t)col * H + k] : 0.f, rne, pb[0][j], pb[1][j], pb[2][j]);
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            u32x4 of, ob;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                of[q] = (unsigned)pf[p][2 * q] | ((unsigned)pf[p][2 * q + 1] << 16);
                ob[q] = (unsigned)pb[p][2 * q] | ((unsigned)pb[p][2 * q + 1] << 16);
            }
            const size_t o = ((((size_t)ct * nkg + kg) * 2 + ks) * 3 + p) * 64 + lane;
            vpack_f[o] = of;
            vpack_b[o] = ob;
        }
    }
    if (Vm) {
        const size_t n = (size_t)H * H, stride = (size_t)gridDim.x * blockDim.x;
        for (size_t e = idx; e < n; e += stride) {
            const int i = (int)(e / H), j = (int)(e % H);
            Vm[e] = (i == j) ? 0.f : V[e];
        }
    }
}
__global__ void vmask_kernel(int H, const float* __restrict__ V, float* __restrict__ Vm) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)H * H) return;
    const int i = (int)(idx / H), j = (int)(idx % H);
    Vm[idx] = (i == j) ? 0.f : V[idx];
}

int pick_kgw(int H) {
    const int need = cdiv(cdiv(H, 32), 4);
    for (int k : {1, 2, 4, 8})
        if (need <= k) return k;
    return 0;  // H > 1024: V slice no longer fits the register file in this layout
}

size_t fwd_chan_bytes(int Bp, int T, int H) {
    return (size_t)T * cdiv(Bp, RT) * cdiv(H, CT) * 32 * sizeof(u64);
}
size_t bwd_ring_bytes(int Bp, int H) {  // dense fp32 tiles (the RNN cell)
    return (size_t)RING * cdiv(Bp, RT) * cdiv(H, CT) * TILE_BYTES;
}
size_t bwd_pring_bytes(int Bp, int H) {  // pre-split plane tiles (the spiking backward)
    return (size_t)RING * cdiv(Bp, RT) * cdiv(H, CT) * PTILE_BYTES;
}
// agreement table of the XCD-local stores: one word per workgroup, behind the granules / the ring
size_t xcd_tab_bytes(int Bp, int H) { return (size_t)cdiv(Bp, RT) * cdiv(H, CT) * sizeof(unsigned); }
int g_xcd_local = -1;  // -1: SPARCH_XCD_LOCAL (default on); 0 / 1: set by sparch_set_xcd_local
bool xcd_local_enabled() {
    static const bool env_on = [] { const char* e = getenv("SPARCH_XCD_LOCAL"); return !e || atoi(e) != 0; }();
    return g_xcd_local < 0 ? env_on : g_xcd_local != 0;
}

// NP = 1: the bf16 operand mode (sparch_set_operand_precision) — the V pack then holds one rounded plane
template <bool BWD, bool ADAPT, int NP = 3>
int launch_rec(int kgw, const RecArgs& a, unsigned grid, hipStream_t st, int cw = 1) {
    if constexpr (NP == 1) {
        if (cw == 2) {  // 64-column workgroups (bf16 operand mode, 8-wave kernels: kgw >= 2)
#define SP_LAUNCH2(KB)                                                                                \
    if (BWD && a.save16) hipLaunchKernelGGL((rec_bwd_kernel<ADAPT, KB, 8, false, 1, true, 2>), dim3(grid), dim3(512), 0, st, a); \
    else if (BWD) hipLaunchKernelGGL((rec_bwd_kernel<ADAPT, KB, 8, false, 1, false, 2>), dim3(grid), dim3(512), 0, st, a); \
    else     hipLaunchKernelGGL((rec_fwd_kernel<ADAPT, KB, 8, false, 1, 2>), dim3(grid), dim3(512), 0, st, a);
            switch (kgw) {
                case 2: SP_LAUNCH2(1) break;
                case 4: SP_LAUNCH2(2) break;
                case 8: SP_LAUNCH2(4) break;
                default: return SPARCH_EINVAL;
            }
#undef SP_LAUNCH2
            SPARCH_CHECK_LAUNCH();
            return SPARCH_OK;
        }
    }
    // 8 waves of K/2 k-groups each once there are at least 8 k-groups, else 4 waves
#define SP_LAUNCH(K, KB, NWB)                                                                        \
    if (BWD && a.save16) hipLaunchKernelGGL((rec_bwd_kernel<ADAPT, KB, NWB, false, NP, true>), dim3(grid), dim3(64 * NWB), 0, st, a); \
    else if (BWD) hipLaunchKernelGGL((rec_bwd_kernel<ADAPT, KB, NWB, false, NP, false>), dim3(grid), dim3(64 * NWB), 0, st, a); \
    else     hipLaunchKernelGGL((rec_fwd_kernel<ADAPT, KB, NWB, false, NP>), dim3(grid), dim3(64 * NWB), 0, st, a);
    switch (kgw) {
        case 1: SP_LAUNCH(1, 1, 4) break;
        case 2: SP_LAUNCH(2, 1, 8) break;
        case 4: SP_LAUNCH(4, 2, 8) break;
        case 8: SP_LAUNCH(8, 4, 8) break;
        default: return SPARCH_EINVAL;
    }
#undef SP_LAUNCH
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

template <bool BWD, bool ADAPT, int NP = 3>
bool rec_co_resident(int kgw, unsigned grid, int cus, int cw = 1) {
    if constexpr (NP == 1) {
        if (cw == 2) {
#define SP_RES2(KB) \
    return BWD ? grid_is_co_resident<rec_bwd_kernel<ADAPT, KB, 8, false, 1, false, 2>>(grid, 512, cus) \
               : grid_is_co_resident<rec_fwd_kernel<ADAPT, KB, 8, false, 1, 2>>(grid, 512, cus);
            switch (kgw) {
                case 2: SP_RES2(1)
                case 4: SP_RES2(2)
                case 8: SP_RES2(4)
                default: return false;
            }
#undef SP_RES2
        }
    }
#define SP_RES(KB, NWB) \
    return BWD ? grid_is_co_resident<rec_bwd_kernel<ADAPT, KB, NWB, false, NP>>(grid, 64 * NWB, cus) \
               : grid_is_co_resident<rec_fwd_kernel<ADAPT, KB, NWB, false, NP>>(grid, 64 * NWB, cus);
    switch (kgw) {
        case 1: SP_RES(1, 4)
        case 2: SP_RES(1, 8)
        case 4: SP_RES(2, 8)
        case 8: SP_RES(4, 8)
        default: return false;
    }
#undef SP_RES
}

template <bool BWD>
int run_rec(int kind, RecArgs& a, size_t chan_bytes, int steps_per_launch, hipStream_t st) {
    const bool adapt = kind == SPARCH_KIND_RADLIF;
    const bool low = sparch_operand_bf16() != 0;  // the V pack must have been made in the same mode
    const int kgw = pick_kgw(a.H);
    if (kgw == 0) return SPARCH_EINVAL;
    a.n_ct = cdiv(a.H, CT);
    a.nkg = 4 * kgw;
    a.n_rt_total = cdiv(a.Bp, RT);
    if (!a.chan || chan_bytes < sparch_rec_chan_bytes(a.Bp, a.T, a.H)) return SPARCH_EWORKSPACE;
    // (the agreement table of the XCD-local stores sits behind the granules / the ring and is cleared with them:
    // "empty" is 0 in the forward's zeroed buffer, the sentinel word in the backward's)
    const size_t tabb = xcd_tab_bytes(a.Bp, a.H);
    if (!BWD) {
        const size_t fb = fwd_chan_bytes(a.Bp, a.T, a.H);
        if (hipMemsetAsync(a.chan, 0, fb + tabb, st) != hipSuccess) return SPARCH_ELAUNCH;
        a.xcd_tab = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(a.chan) + fb);
    } else {  // every ring piece reads "not written yet" until its producer's store lands
        const size_t rb = bwd_pring_bytes(a.Bp, a.H);
        if (hipMemsetD32Async((hipDeviceptr_t)a.chan, (int)SENTINEL, (rb + tabb) / 4, st) != hipSuccess)
            return SPARCH_ELAUNCH;
        a.ring = reinterpret_cast<char*>(a.chan);
        a.xcd_tab = reinterpret_cast<unsigned*>(a.ring + rb);
    }

    int L = steps_per_launch;
    if (L < 1) L = 1;
    if (L > a.T) L = a.T;
    int cus = sparch_device_cus();
    if (cus <= 0) cus = 256;
    int rt_per_launch;
    // 64-column workgroups (bf16 operand mode): when the row tiles do not fit one persistent launch at 32 columns
    // per workgroup (512 virtual rows at H = 1024: 16 x 32 workgroups) but do at 64 (16 x 16), the whole batch runs
    // as ONE launch instead of two half-machine launches back to back — the steps are latency chains, so a launch
    // over all rows takes about as long as one over half of them.  SPARCH_REC_CW=1 forces the 32-column kernels.
    static const int cw_env = [] { const char* e = getenv("SPARCH_REC_CW"); return e ? atoi(e) : 0; }();
    int cw = 1;
    if (low && L > 1 && kgw >= 2 && a.n_ct % 2 == 0 && cw_env != 1 &&
        (cw_env == 2 || ((long long)a.n_rt_total * a.n_ct > cus && (long long)a.n_rt_total * (a.n_ct / 2) <= cus)))
        cw = 2;
    const int wg_per_rt = a.n_ct / cw;  // workgroups of one row tile
    if (L == 1) {
        rt_per_launch = a.n_rt_total;  // nothing waits inside a launch: any grid size is fine
    } else {
        rt_per_launch = cus / wg_per_rt;  // one workgroup per CU must be co-resident
        if (rt_per_launch < 1) { L = 1; rt_per_launch = a.n_rt_total; }
    }
    if (L > 1) {  // ask the runtime's occupancy calculator instead of assuming one workgroup per CU fits
        const unsigned g = (unsigned)(wg_per_rt * min(rt_per_launch, a.n_rt_total));
        const bool ok = low ? (adapt ? rec_co_resident<BWD, true, 1>(kgw, g, cus, cw) : rec_co_resident<BWD, false, 1>(kgw, g, cus, cw))
                            : (adapt ? rec_co_resident<BWD, true>(kgw, g, cus) : rec_co_resident<BWD, false>(kgw, g, cus));
        if (!ok) {
            if (a.save16 && !BWD) return SPARCH_EINVAL;  // bf16 saves need the whole-sequence forward launch
            L = 1; rt_per_launch = a.n_rt_total;
        }
    }
    if (L == 1) cw = 1;
    if (L < a.T || !xcd_local_enabled()) a.xcd_tab = nullptr;  // whole-sequence launches only (one agreement per launch)
    for (int rt0 = 0; rt0 < a.n_rt_total; rt0 += rt_per_launch) {
        a.rt_base = rt0;
        a.n_rt_launch = min(rt_per_launch, a.n_rt_total - rt0);
        const unsigned grid = (unsigned)((a.n_ct / cw) * a.n_rt_launch);
        if (!BWD) {
            for (int t0 = 0; t0 < a.T; t0 += L) {
                a.t_begin = t0; a.t_end = min(a.T, t0 + L);
                int rc = low ? (adapt ? launch_rec<false, true, 1>(kgw, a, grid, st, cw) : launch_rec<false, false, 1>(kgw, a, grid, st, cw))
                             : (adapt ? launch_rec<false, true>(kgw, a, grid, st) : launch_rec<false, false>(kgw, a, grid, st));
                if (rc != SPARCH_OK) return rc;
            }
        } else {
            for (int t1 = a.T; t1 > 0; t1 -= L) {
                a.t_end = t1; a.t_begin = max(0, t1 - L);
                int rc = low ? (adapt ? launch_rec<true, true, 1>(kgw, a, grid, st, cw) : launch_rec<true, false, 1>(kgw, a, grid, st, cw))
                             : (adapt ? launch_rec<true, true>(kgw, a, grid, st) : launch_rec<true, false>(kgw, a, grid, st));
                if (rc != SPARCH_OK) return rc;
            }
        }
    }
    return SPARCH_OK;
}

template <int ACT, bool BWD>
int launch_ann(int kgw, const AnnArgs& a, unsigned grid, hipStream_t st) {
#define SP_LAUNCH_ANN(KB, NWB) \
    hipLaunchKernelGGL((ann_rec_kernel<ACT, BWD, KB, NWB>), dim3(grid), dim3(64 * NWB), 0, st, a);
    switch (kgw) {
        case 1: SP_LAUNCH_ANN(1, 4) break;
        case 2: SP_LAUNCH_ANN(1, 8) break;
        case 4: SP_LAUNCH_ANN(2, 8) break;
        case 8: SP_LAUNCH_ANN(4, 8) break;
        default: return SPARCH_EINVAL;
    }
#undef SP_LAUNCH_ANN
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

template <bool BWD>
int run_ann(int act, AnnArgs& a, void* chan, size_t chan_bytes, int steps_per_launch, hipStream_t st) {
    const int kgw = pick_kgw(a.H);
    if (kgw == 0) return SPARCH_EINVAL;
    a.n_ct = cdiv(a.H, CT);
    a.nkg = 4 * kgw;
    a.n_rt_total = cdiv(a.Bp, RT);
    if (!chan || chan_bytes < sparch_rec_chan_bytes(a.Bp, a.T, a.H)) return SPARCH_EWORKSPACE;
    const size_t rb = bwd_pring_bytes(a.Bp, a.H), tabb = xcd_tab_bytes(a.Bp, a.H);  // plane tiles + agreement table
    if (hipMemsetD32Async((hipDeviceptr_t)chan, (int)SENTINEL, (rb + tabb) / 4, st) != hipSuccess)
        return SPARCH_ELAUNCH;
    a.ring = reinterpret_cast<char*>(chan);
    a.xcd_tab = reinterpret_cast<unsigned*>(a.ring + rb);
    int L = steps_per_launch;
    if (L < 1) L = 1;
    if (L > a.T) L = a.T;
    if (L < a.T || !xcd_local_enabled()) a.xcd_tab = nullptr;  // whole-sequence launches only
    int cus = sparch_device_cus();
    if (cus <= 0) cus = 256;
    int rt_per_launch;
    if (L == 1) {
        rt_per_launch = a.n_rt_total;
    } else {
        rt_per_launch = cus / a.n_ct;  // one workgroup per CU must be co-resident
        if (rt_per_launch < 1) { L = 1; rt_per_launch = a.n_rt_total; }
    }
    for (int rt0 = 0; rt0 < a.n_rt_total; rt0 += rt_per_launch) {
        a.rt_base = rt0;
        a.n_rt_launch = min(rt_per_launch, a.n_rt_total - rt0);
        const unsigned grid = (unsigned)(a.n_ct * a.n_rt_launch);
        for (int s0 = 0; s0 < a.T; s0 += L) {
            a.s_begin = s0; a.s_end = min(a.T, s0 + L);
            int rc;
            switch (act) {
                case SPARCH_ACT_SIGMOID: rc = launch_ann<SPARCH_ACT_SIGMOID, BWD>(kgw, a, grid, st); break;
                case SPARCH_ACT_RELU: rc = launch_ann<SPARCH_ACT_RELU, BWD>(kgw, a, grid, st); break;
                case SPARCH_ACT_TANH: rc = launch_ann<SPARCH_ACT_TANH, BWD>(kgw, a, grid, st); break;
                default: return SPARCH_EINVAL;
            }
            if (rc != SPARCH_OK) return rc;
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

#ifdef SPARCH_REC_PROF
extern "C" int sparch_rec_prof_read(unsigned long long* host_out, int reset) {
    static unsigned long long zero[2 * 512 * 12];
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_rec_prof), sizeof(zero)) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_rec_prof), zero, sizeof(zero)) != hipSuccess) return -1;
    return 0;
}
#endif

extern "C" size_t sparch_vpack_bytes(int H) {
    const int kgw = pick_kgw(H);
    if (H <= 0 || kgw == 0) return 0;
    return (size_t)cdiv(H, CT) * (4 * kgw) * 2 * 3 * 64 * sizeof(u32x4);
}

extern "C" int sparch_vpack(int H, const float* V, int transpose, float* vpack, float* vmasked, void* stream, int precision) {
    SPARCH_ENTER();
    PrecisionScope prec_scope_(precision);
    if (!prec_scope_.ok) return SPARCH_EINVAL;
    const int kgw = pick_kgw(H);
    if (H <= 0 || kgw == 0 || !V || !vpack) return SPARCH_EINVAL;
    if (!aligned16(vpack)) return SPARCH_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int n_ct = cdiv(H, CT), nkg = 4 * kgw;
    const size_t total = (size_t)n_ct * nkg * 2 * 64;
    hipLaunchKernelGGL(vpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, H, n_ct, nkg,
                       transpose, sparch_operand_bf16(), V, reinterpret_cast<u32x4*>(vpack));
    SPARCH_CHECK_LAUNCH();
    if (vmasked) {
        const size_t n = (size_t)H * H;
        hipLaunchKernelGGL(vmask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, H, V, vmasked);
        SPARCH_CHECK_LAUNCH();
    }
    return SPARCH_OK;
}

extern "C" int sparch_vpack_both(int H, const float* V, float* vpack_fwd, float* vpack_bwd, float* vmasked, void* stream, int precision) {
    SPARCH_ENTER();
    PrecisionScope prec_scope_(precision);
    if (!prec_scope_.ok) return SPARCH_EINVAL;
    const int kgw = pick_kgw(H);
    if (H <= 0 || kgw == 0 || !V || !vpack_fwd || !vpack_bwd) return SPARCH_EINVAL;
    if (!aligned16(vpack_fwd) || !aligned16(vpack_bwd)) return SPARCH_EALIGN;
    const int n_ct = cdiv(H, CT), nkg = 4 * kgw;
    const size_t total = (size_t)n_ct * nkg * 2 * 64;
    hipLaunchKernelGGL(vpack_both_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, H, n_ct,
                       nkg, sparch_operand_bf16(), V, reinterpret_cast<u32x4*>(vpack_fwd), reinterpret_cast<u32x4*>(vpack_bwd), vmasked);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

extern "C" int sparch_set_xcd_local(int on) {
    const int was = xcd_local_enabled() ? 1 : 0;
    g_xcd_local = on < 0 ? -1 : (on != 0);
    return was;
}

extern "C" int sparch_vmask(int H, const float* V, float* vmasked, void* stream) {
    SPARCH_ENTER();
    if (H <= 0 || !V || !vmasked) return SPARCH_EINVAL;
    const size_t n = (size_t)H * H;
    hipLaunchKernelGGL(vmask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, H, V, vmasked);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

extern "C" size_t sparch_rec_chan_bytes(int Bp, int T, int H) {
    if (Bp <= 0 || T <= 0 || H <= 0) return 0;
    // forward: T x row tiles x column tiles x 32 granules of 8 B; backward: fp32 tile ring
    const size_t f = fwd_chan_bytes(Bp, T, H), b = bwd_pring_bytes(Bp, H);
    // rounded up to 16 bytes: the agreement table is 4 bytes per workgroup, and a caller that sizes its buffer in
    // 8-byte words (round 2: `nbytes // 8` in the Python host) must not come out 4 bytes short of the table's
    // last word when the number of workgroups is odd
    return ((f > b ? f : b) + xcd_tab_bytes(Bp, H) + 15) & ~(size_t)15;
}

extern "C" int sparch_rec_cell_fwd(int kind, int B, int dirs, int T, int H, const float* Wx,
                                   const float* scale, const float* shift, const float* alpha,
                                   const float* beta, const float* a, const float* b,
                                   const float* vpack, const float* rec0, const float* u0,
                                   const float* w0, const float* s0, float theta, float p_drop,
                                   uint64_t seed, float* s_out, uint16_t* s16_out, void* u_save, void* w_save,
                                   int save_bf16, uint32_t* spike_count, void* chan, size_t chan_bytes,
                                   uint32_t* status, int steps_per_launch, void* stream, int precision) {
    SPARCH_ENTER();
    PrecisionScope prec_scope_(precision);
    if (!prec_scope_.ok) return SPARCH_EINVAL;
    // bf16 saved states cannot carry the exact state from one launch of a chunked forward to the next
    if (save_bf16 && steps_per_launch < T) return SPARCH_EINVAL;
    if (kind != SPARCH_KIND_RLIF && kind != SPARCH_KIND_RADLIF) return SPARCH_EINVAL;
    const bool adapt = kind == SPARCH_KIND_RADLIF;
    if (B <= 0 || T <= 0 || H < 4 || (H % 4) != 0 || (dirs != 1 && dirs != 2)) return SPARCH_EINVAL;
    if (!Wx || !alpha || !vpack || !rec0 || !u0 || !s0 || (!s_out && !s16_out) || !u_save || !status) return SPARCH_EINVAL;
    if (adapt && (!beta || !a || !b || !w0 || !w_save)) return SPARCH_EINVAL;
    if ((scale == nullptr) != (shift == nullptr)) return SPARCH_EINVAL;
    if (!(p_drop >= 0.0f && p_drop < 1.0f)) return SPARCH_EINVAL;
    if (!al16({Wx, vpack, rec0, u0, w0, s0, s_out, s16_out, u_save, w_save, chan})) return SPARCH_EALIGN;
    RecArgs r{};
    r.B = B; r.dirs = dirs; r.T = T; r.H = H; r.Bp = B * dirs;
    r.Wx = Wx; r.scale = scale; r.shift = shift;
    r.alpha = alpha; r.beta = beta; r.a = a; r.b = b;
    r.vpack = reinterpret_cast<const u32x4*>(vpack); r.rec0 = rec0; r.u0 = u0; r.w0 = w0; r.s0 = s0;
    r.theta = theta; r.p_drop = p_drop; r.inv_keep = 1.0f / (1.0f - p_drop); r.seed = seed;
    r.s_out = s_out; r.s16_out = s16_out; r.u_save = (float*)u_save; r.w_save = (float*)w_save;
    r.save16 = save_bf16 != 0; r.spike_count = spike_count;
    r.chan = (u64*)chan; r.status = status;
    return run_rec<false>(kind, r, chan_bytes, steps_per_launch, (hipStream_t)stream);
}

extern "C" int sparch_rec_cell_bwd(int kind, int B, int dirs, int T, int H, const float* g_out,
                                   const float* g_rate, const void* u_save, const void* w_save, int save_bf16,
                                   const float* alpha, const float* beta, const float* a,
                                   const float* b, const float* vpack_t, const float* u0,
                                   const float* w0, const float* s0, float theta, float p_drop,
                                   uint64_t seed, float* dWx, uint16_t* s_prev16, float* dparam_ws,
                                   const float* bn_x, const float* bn_mean, const float* bn_invstd,
                                   void* chan, size_t chan_bytes, uint32_t* status,
                                   int steps_per_launch, void* stream, int precision) {
    SPARCH_ENTER();
    PrecisionScope prec_scope_(precision);
    if (!prec_scope_.ok) return SPARCH_EINVAL;
    if (bn_x && (!bn_mean || !bn_invstd || !aligned16(bn_x) || !aligned16(bn_mean) || !aligned16(bn_invstd)))
        return SPARCH_EINVAL;
    if (kind != SPARCH_KIND_RLIF && kind != SPARCH_KIND_RADLIF) return SPARCH_EINVAL;
    const bool adapt = kind == SPARCH_KIND_RADLIF;
    if (B <= 0 || T <= 0 || H < 4 || (H % 4) != 0 || (dirs != 1 && dirs != 2)) return SPARCH_EINVAL;
    if (!g_out || !u_save || !alpha || !vpack_t || !u0 || !s0 || !dWx || !s_prev16 || !dparam_ws || !status)
        return SPARCH_EINVAL;
    if (adapt && (!beta || !a || !b || !w0 || !w_save)) return SPARCH_EINVAL;
    if (!(p_drop >= 0.0f && p_drop < 1.0f)) return SPARCH_EINVAL;
    if (!al16({g_out, u_save, w_save, vpack_t, u0, w0, s0, dWx, s_prev16, dparam_ws, chan})) return SPARCH_EALIGN;
    if (bwd_pring_bytes(B * dirs, H) >= ((size_t)1 << 31)) return SPARCH_EINVAL;  // 32-bit buffer offsets
    RecArgs r{};
    r.B = B; r.dirs = dirs; r.T = T; r.H = H; r.Bp = B * dirs;
    r.alpha = alpha; r.beta = beta; r.a = a; r.b = b;
    r.vpack = reinterpret_cast<const u32x4*>(vpack_t); r.u0 = u0; r.w0 = w0; r.s0 = s0;
    r.theta = theta; r.p_drop = p_drop; r.inv_keep = 1.0f / (1.0f - p_drop); r.seed = seed;
    r.u_save = (float*)const_cast<void*>(u_save); r.w_save = (float*)const_cast<void*>(w_save);
    r.save16 = save_bf16 != 0;
    r.g_out = g_out; r.g_rate = g_rate; r.g_rate_scale = 1.0f / ((float)B * (float)T);
    r.dWx = dWx; r.s_prev16 = s_prev16; r.dparam_ws = dparam_ws;
    r.bn_x = bn_x; r.bn_mean = bn_mean; r.bn_invstd = bn_invstd; r.bn_src = bn_x ? bn_x : g_out;
    r.chan = (u64*)chan; r.status = status;
    return run_rec<true>(kind, r, chan_bytes, steps_per_launch, (hipStream_t)stream);
}

// ---- f-4: dense recurrent cell of the RNN baseline (anns.py:328-339)
extern "C" int sparch_ann_rec_fwd(int act, int B, int dirs, int T, int H, const float* Wx, const float* scale,
                                  const float* shift, const float* vpack, float p_drop, uint64_t seed,
                                  float* y_out, float* y_state, void* chan, size_t chan_bytes, uint32_t* status,
                                  int steps_per_launch, void* stream) {
    SPARCH_ENTER();
    if (B <= 0 || T <= 0 || H <= 0 || H % 4 != 0 || (dirs != 1 && dirs != 2)) return SPARCH_EINVAL;
    if (!Wx || !vpack || !y_out || !y_state || !status) return SPARCH_EINVAL;
    if ((scale == nullptr) != (shift == nullptr)) return SPARCH_EINVAL;
    if (!(p_drop >= 0.0f && p_drop < 1.0f)) return SPARCH_EINVAL;
    if (!al16({Wx, scale, shift, vpack, y_out, y_state, chan})) return SPARCH_EALIGN;
    AnnArgs a{};
    a.B = B; a.dirs = dirs; a.T = T; a.H = H; a.Bp = B * dirs;
    a.Wx = Wx; a.scale = scale; a.shift = shift; a.vpack = reinterpret_cast<const u32x4*>(vpack);
    a.p_drop = p_drop; a.inv_keep = 1.0f / (1.0f - p_drop); a.seed = seed;
    a.y_state = y_state; a.y_out = y_out; a.status = status;
    return run_ann<false>(act, a, chan, chan_bytes, steps_per_launch, (hipStream_t)stream);
}

extern "C" int sparch_ann_rec_bwd(int act, int B, int dirs, int T, int H, const float* g_out, const float* y_state,
                                  const float* vpack, float p_drop, uint64_t seed, float* dpre, float* y_prev,
                                  void* chan, size_t chan_bytes, uint32_t* status, int steps_per_launch,
                                  void* stream) {
    SPARCH_ENTER();
    if (B <= 0 || T <= 0 || H <= 0 || H % 4 != 0 || (dirs != 1 && dirs != 2)) return SPARCH_EINVAL;
    if (!g_out || !y_state || !vpack || !dpre || !y_prev || !status) return SPARCH_EINVAL;
    if (!(p_drop >= 0.0f && p_drop < 1.0f)) return SPARCH_EINVAL;
    if (!al16({g_out, y_state, vpack, dpre, y_prev, chan})) return SPARCH_EALIGN;
    AnnArgs a{};
    a.B = B; a.dirs = dirs; a.T = T; a.H = H; a.Bp = B * dirs;
    a.g_out = g_out; a.y_in = y_state; a.vpack = reinterpret_cast<const u32x4*>(vpack);
    a.p_drop = p_drop; a.inv_keep = 1.0f / (1.0f - p_drop); a.seed = seed;
    a.dpre = dpre; a.y_prev = y_prev; a.status = status;
    return run_ann<true>(act, a, chan, chan_bytes, steps_per_launch, (hipStream_t)stream);
}

// ---- one time step with the recurrent product supplied by the caller: the path for hidden sizes whose V slice
//      does not fit the register-resident layout (H > 1024), any grid size, no inter-workgroup hand-off
extern "C" int sparch_rec_cell_step_fwd(int kind, int B, int dirs, int T, int H, int t, const float* Wx,
                                        const float* scale, const float* shift, const float* alpha,
                                        const float* beta, const float* a, const float* b, const float* rec,
                                        const float* u0, const float* w0, const float* s0, float theta,
                                        float p_drop, uint64_t seed, float* s_out, uint16_t* s16_out,
                                        float* u_save, float* w_save, uint32_t* spike_count,
                                        uint16_t* s_step16, void* stream) {
    SPARCH_ENTER();
    if (kind != SPARCH_KIND_RLIF && kind != SPARCH_KIND_RADLIF) return SPARCH_EINVAL;
    const bool adapt = kind == SPARCH_KIND_RADLIF;
    if (B <= 0 || T <= 0 || H < 4 || (H % 4) != 0 || (dirs != 1 && dirs != 2) || t < 0 || t >= T) return SPARCH_EINVAL;
    if (!Wx || !alpha || !rec || !u0 || !s0 || (!s_out && !s16_out) || !u_save || !s_step16) return SPARCH_EINVAL;
    if (adapt && (!beta || !a || !b || !w0 || !w_save)) return SPARCH_EINVAL;
    if ((scale == nullptr) != (shift == nullptr)) return SPARCH_EINVAL;
    if (!(p_drop >= 0.0f && p_drop < 1.0f)) return SPARCH_EINVAL;
    if (!al16({Wx, rec, u0, w0, s0, s_out, s16_out, u_save, w_save, s_step16})) return SPARCH_EALIGN;
    RecArgs r{};
    r.B = B; r.dirs = dirs; r.T = T; r.H = H; r.Bp = B * dirs;
    r.n_ct = cdiv(H, CT); r.nkg = 4; r.n_rt_total = cdiv(r.Bp, RT); r.rt_base = 0; r.n_rt_launch = r.n_rt_total;
    r.t_begin = t; r.t_end = t + 1;
    r.Wx = Wx; r.scale = scale; r.shift = shift;
    r.alpha = alpha; r.beta = beta; r.a = a; r.b = b; r.rec0 = rec; r.u0 = u0; r.w0 = w0; r.s0 = s0;
    r.theta = theta; r.p_drop = p_drop; r.inv_keep = 1.0f / (1.0f - p_drop); r.seed = seed;
    r.s_out = s_out; r.s16_out = s16_out; r.u_save = u_save; r.w_save = w_save; r.spike_count = spike_count;
    r.s_step16 = s_step16;
    const unsigned grid = (unsigned)(r.n_ct * r.n_rt_total);
    hipStream_t st = (hipStream_t)stream;
    if (adapt) hipLaunchKernelGGL((rec_fwd_kernel<true, 1, 4, true>), dim3(grid), dim3(256), 0, st, r);
    else       hipLaunchKernelGGL((rec_fwd_kernel<false, 1, 4, true>), dim3(grid), dim3(256), 0, st, r);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

extern "C" int sparch_rec_cell_step_bwd(int kind, int B, int dirs, int T, int H, int t, const float* g_out,
                                        const float* g_rate, const float* u_save, const float* w_save,
                                        const float* alpha, const float* beta, const float* a, const float* b,
                                        const float* rec, const float* u0, const float* w0, const float* s0,
                                        float theta, float p_drop, uint64_t seed, float* dWx,
                                        uint16_t* s_prev16, float* dparam_ws, const float* bn_x,
                                        const float* bn_mean, const float* bn_invstd, float* dwx_step,
                                        void* stream) {
    SPARCH_ENTER();
    if (bn_x && (!bn_mean || !bn_invstd || !aligned16(bn_x) || !aligned16(bn_mean) || !aligned16(bn_invstd)))
        return SPARCH_EINVAL;
    if (kind != SPARCH_KIND_RLIF && kind != SPARCH_KIND_RADLIF) return SPARCH_EINVAL;
    const bool adapt = kind == SPARCH_KIND_RADLIF;
    if (B <= 0 || T <= 0 || H < 4 || (H % 4) != 0 || (dirs != 1 && dirs != 2) || t < 0 || t >= T) return SPARCH_EINVAL;
    if (!g_out || !u_save || !alpha || !u0 || !s0 || !dWx || !s_prev16 || !dparam_ws || !dwx_step) return SPARCH_EINVAL;
    if (t + 1 < T && !rec) return SPARCH_EINVAL;
    if (adapt && (!beta || !a || !b || !w0 || !w_save)) return SPARCH_EINVAL;
    if (!(p_drop >= 0.0f && p_drop < 1.0f)) return SPARCH_EINVAL;
    if (!al16({g_out, u_save, w_save, rec, u0, w0, s0, dWx, s_prev16, dparam_ws, dwx_step})) return SPARCH_EALIGN;
    RecArgs r{};
    r.B = B; r.dirs = dirs; r.T = T; r.H = H; r.Bp = B * dirs;
    r.n_ct = cdiv(H, CT); r.nkg = 4; r.n_rt_total = cdiv(r.Bp, RT); r.rt_base = 0; r.n_rt_launch = r.n_rt_total;
    r.t_begin = t; r.t_end = t + 1;
    r.alpha = alpha; r.beta = beta; r.a = a; r.b = b; r.rec0 = rec; r.u0 = u0; r.w0 = w0; r.s0 = s0;
    r.theta = theta; r.p_drop = p_drop; r.inv_keep = 1.0f / (1.0f - p_drop); r.seed = seed;
    r.u_save = const_cast<float*>(u_save); r.w_save = const_cast<float*>(w_save);
    r.g_out = g_out; r.g_rate = g_rate; r.g_rate_scale = 1.0f / ((float)B * (float)T);
    r.dWx = dWx; r.s_prev16 = s_prev16; r.dparam_ws = dparam_ws; r.dwx_step = dwx_step;
    r.bn_x = bn_x; r.bn_mean = bn_mean; r.bn_invstd = bn_invstd; r.bn_src = bn_x ? bn_x : g_out;
    const unsigned grid = (unsigned)(r.n_ct * r.n_rt_total);
    hipStream_t st = (hipStream_t)stream;
    if (adapt) hipLaunchKernelGGL((rec_bwd_kernel<true, 1, 4, true>), dim3(grid), dim3(256), 0, st, r);
    else       hipLaunchKernelGGL((rec_bwd_kernel<false, 1, 4, true>), dim3(grid), dim3(256), 0, st, r);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

namespace {
template <bool BWD>
int launch_ann_step(int act, const AnnArgs& a, hipStream_t st) {
    const unsigned grid = (unsigned)(a.n_ct * a.n_rt_total);
    switch (act) {
        case SPARCH_ACT_SIGMOID:
            hipLaunchKernelGGL((ann_rec_kernel<SPARCH_ACT_SIGMOID, BWD, 1, 4, true>), dim3(grid), dim3(256), 0, st, a); break;
        case SPARCH_ACT_RELU:
            hipLaunchKernelGGL((ann_rec_kernel<SPARCH_ACT_RELU, BWD, 1, 4, true>), dim3(grid), dim3(256), 0, st, a); break;
        case SPARCH_ACT_TANH:
            hipLaunchKernelGGL((ann_rec_kernel<SPARCH_ACT_TANH, BWD, 1, 4, true>), dim3(grid), dim3(256), 0, st, a); break;
        default: return SPARCH_EINVAL;
    }
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}
}  // namespace

/* `s` counts steps in processing order (forward: t = s; backward: t = T-1-s); rec = y_{t-1} V^T (forward) /
 * dpre_{t+1} V (backward), ignored at s = 0. */
extern "C" int sparch_ann_rec_step_fwd(int act, int B, int dirs, int T, int H, int s, const float* Wx,
                                       const float* scale, const float* shift, const float* rec, float p_drop,
                                       uint64_t seed, float* y_out, float* y_state, float* y_step, void* stream) {
    SPARCH_ENTER();
    if (B <= 0 || T <= 0 || H <= 0 || H % 4 != 0 || (dirs != 1 && dirs != 2) || s < 0 || s >= T) return SPARCH_EINVAL;
    if (!Wx || !y_out || !y_state || !y_step || (s > 0 && !rec)) return SPARCH_EINVAL;
    if ((scale == nullptr) != (shift == nullptr)) return SPARCH_EINVAL;
    if (!(p_drop >= 0.0f && p_drop < 1.0f)) return SPARCH_EINVAL;
    if (!al16({Wx, scale, shift, rec, y_out, y_state, y_step})) return SPARCH_EALIGN;
    AnnArgs a{};
    a.B = B; a.dirs = dirs; a.T = T; a.H = H; a.Bp = B * dirs;
    a.n_ct = cdiv(H, CT); a.nkg = 4; a.n_rt_total = cdiv(a.Bp, RT); a.rt_base = 0; a.n_rt_launch = a.n_rt_total;
    a.s_begin = s; a.s_end = s + 1;
    a.Wx = Wx; a.scale = scale; a.shift = shift; a.p_drop = p_drop; a.inv_keep = 1.0f / (1.0f - p_drop); a.seed = seed;
    a.y_state = y_state; a.y_out = y_out; a.rec_ext = rec; a.step_out = y_step;
    return launch_ann_step<false>(act, a, (hipStream_t)stream);
}

extern "C" int sparch_ann_rec_step_bwd(int act, int B, int dirs, int T, int H, int s, const float* g_out,
                                       const float* y_state, const float* rec, float p_drop, uint64_t seed,
                                       float* dpre, float* y_prev, float* dpre_step, void* stream) {
    SPARCH_ENTER();
    if (B <= 0 || T <= 0 || H <= 0 || H % 4 != 0 || (dirs != 1 && dirs != 2) || s < 0 || s >= T) return SPARCH_EINVAL;
    if (!g_out || !y_state || !dpre || !y_prev || !dpre_step || (s > 0 && !rec)) return SPARCH_EINVAL;
    if (!(p_drop >= 0.0f && p_drop < 1.0f)) return SPARCH_EINVAL;
    if (!al16({g_out, y_state, rec, dpre, y_prev, dpre_step})) return SPARCH_EALIGN;
    AnnArgs a{};
    a.B = B; a.dirs = dirs; a.T = T; a.H = H; a.Bp = B * dirs;
    a.n_ct = cdiv(H, CT); a.nkg = 4; a.n_rt_total = cdiv(a.Bp, RT); a.rt_base = 0; a.n_rt_launch = a.n_rt_total;
    a.s_begin = s; a.s_end = s + 1;
    a.g_out = g_out; a.y_in = y_state; a.p_drop = p_drop; a.inv_keep = 1.0f / (1.0f - p_drop); a.seed = seed;
    a.dpre = dpre; a.y_prev = y_prev; a.rec_ext = rec; a.step_out = dpre_step;
    return launch_ann_step<true>(act, a, (hipStream_t)stream);
}

// G1 for spike operands: GEMMs in which one operand is a spike tensor, on the bf16 MFMA with
// EXACT products.
//
// In every layer after the first, the projection's input is the previous layer's spike train
// (snns.py:169 feeds layer i with layer i-1's output), so `self.W(x)` (snns.py:261...), its weight
// gradient dW = dWx^T x and the recurrent weight gradient dV = s_prev^T dWx all have one operand
// whose entries are 0 or one constant c (c = 1/(1-p) after dropout, snns.py:278).  Such an operand
// is exact in bf16 once c is factored out; the other (fp32) operand is split exactly into three
// bf16 planes x = hi + mid + lo.  Then
//     C = c * ( E*S_hi + E*S_mid + E*S_lo )
// has exact products and fp32 accumulation (v_mfma_f32_32x32x16_bf16): the same accuracy class as
// an fp32 fmaf chain at 3/16 of the fp32-MFMA cost.
//
// Structure mirrors gemm.hip (128x128 tile, 4 waves as 2x2, BK = 32, register-prefetched staging)
// with the conversion done ONCE per element while staging into LDS:
//   KC operand (element (row,k) at p[row*ld+k]): LDS image [row][32 k] bf16, 80-byte rows
//       -> MFMA fragment = one ds_read_b128 (conflict-free);
//   KM operand (element (k,col) at p[k*ld+col]): LDS image [k][128 col] bf16, 320-byte rows
//       -> MFMA fragment = two ds_read_b64_tr_b16 (hardware transpose read, conflict-free).
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

#ifndef GEMM_TAIL2
#define GEMM_TAIL2 2
#endif
constexpr int BK = 32;
constexpr int KC_ROW = 40;    // bf16 per LDS row of a KC image (32 + 8 pad = 80 B)
// bf16 per LDS row of a KM image of a ROWS-wide operand tile: ROWS + 32 pad (128 -> 320 B, 256 -> 576 B;
// both are 16 dwords mod the 64 banks, which is what makes the transposed reads conflict-free)
template <int ROWS> constexpr int km_row() { return ROWS + 32; }
template <bool KM, int ROWS> constexpr int plane_elems() { return KM ? BK * km_row<ROWS>() : ROWS * KC_ROW; }

enum Epi { EPI_NONE = 0, EPI_BIAS = 1, EPI_STATS = 2 };

// Diagnostic build only (-DSPARCH_REC_PROF): s_memtime stamps of the phases of one K-tile iteration
#if defined(SPARCH_REC_PROF) && !defined(GA_NO_STAMPS)
__device__ unsigned long long g_gemm_prof[8 + 8 * 2];  // [8..]: per wave of the sampled workgroup: phase, barrier
#define GP_DECL                                                                                  \
    unsigned long long gp_t = 0, gp_acc[5] = {0, 0, 0, 0, 0}, gp_c0, gp_r0;                      \
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(gp_c0), "=s"(gp_r0)::"memory");
#define GP_STAMP(i)                                                                              \
    do {                                                                                         \
        unsigned long long now_;                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");           \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        if ((i) >= 0) gp_acc[(i) < 0 ? 0 : (i)] += now_ - gp_t;                                  \
        gp_t = now_;                                                                             \
    } while (0)
#define GP_FLUSH()                                                                               \
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 7) {                                            \
        atomicAdd(&g_gemm_prof[8 + 2 * (threadIdx.x >> 6)], gp_acc[3]);                          \
        atomicAdd(&g_gemm_prof[9 + 2 * (threadIdx.x >> 6)], gp_acc[4]);                          \
    }                                                                                            \
    if (threadIdx.x == 0 && blockIdx.x == 7) {                                                   \
        unsigned long long c1_, r1_;                                                             \
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1_), "=s"(r1_)::"memory"); \
        for (int i_ = 0; i_ < 5; ++i_) atomicAdd(&g_gemm_prof[i_], gp_acc[i_]);                  \
        atomicAdd(&g_gemm_prof[5], c1_ - gp_c0);                                                 \
        atomicAdd(&g_gemm_prof[6], r1_ - gp_r0);                                                 \
    }
#else
#define GP_DECL
#define GP_STAMP(i)
#define GP_FLUSH()
#endif

struct SArgs {
    const float* A; const float* B; float* C;
    const float* bias; float* colstat;
    int M, N, K, lda, ldb, ldc;
    int k_per_split; size_t c_split_stride;
    int a_vec, b_vec;
    float scale;
    // device-side gate: the whole launch returns at once unless *gate == gate_want (lets the host enqueue
    // both the exact single-plane kernel and the 6-term kernel for an operand whose bf16-exactness is
    // only known on the device, with no host round trip)
    const unsigned* gate; unsigned gate_want;
    int e_exact;  // spike operand conversion: 0 = (x != 0), 1 = x itself (caller guarantees bf16-exact values)
    // BPRE kernels: the dense B operand pre-split into its three exact bf16 planes (sparch_split3): plane p at
    // Bp + p * bp_stride (elements), same row layout and ldb as B.  Weight matrices are split ONCE per step
    // instead of once per workgroup that stages them (250 M-tiles re-converted the same W tile).
    const unsigned short* Bp; size_t bp_stride;
    // APRE kernels (round 3): the dense A operand pre-split likewise — the gradient of a BatchNorm'd projection,
    // which its producer (sparch_bn_bwd_apply_planes) writes as planes ONCE instead of every workgroup of the dX and
    // dW products re-converting the tiles it stages (8 column tiles each).  Plane p at Ap + p * ap_stride, A's layout.
    const unsigned short* Ap; size_t ap_stride;
    int n_splits;  // K ranges (the grid walks tiles x n_splits work items)
};

__device__ __forceinline__ unsigned short bf16_bits(float x) {
    return __builtin_bit_cast(unsigned short, (__bf16)x);
}
__device__ __forceinline__ void split3(float x, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    hi = __builtin_bit_cast(unsigned short, h);
    mid = __builtin_bit_cast(unsigned short, m);
    lo = bf16_bits(r2);
}

// ---- global -> registers (same thread/element map as gemm.hip), one 16-byte piece `p` of the thread's
// ROWS/32 pieces per call so that the K loop can spread the loads between its MFMA groups.
// `fast` (uniform): the whole ROWS x 32 tile is in range and 16-byte loads are legal.
template <bool KM, int ROWS>
__device__ __forceinline__ bool tile_is_full(int row0, int rows, int k0, int kend, int vec) {
    return vec && row0 + ROWS <= rows && k0 + BK <= kend;
}
// GEMM_NT (bit mask, experiment, OFF): non-temporal hints for what a product touches once — 1: the row-major
// activation operand of the pipelined kernels, 2: the C tiles of products with a row-major A (the big activations /
// gradients), 4: both K-major streams of the TN products.  Measured on the cfg3 step in one call (round 3): 0 -> 6.49
// ms, 1 -> 6.62-6.77, 3 -> 6.56-6.60, 7 -> 6.54-6.65: the products themselves do not change and the kernel BEHIND a
// hinted product slows down (rec_bwd after dX: 2.09 -> 2.14-2.24 ms) — its input no longer waits in the infinity cache.
#ifndef GEMM_NT
#define GEMM_NT 0
#endif
// (a weight matrix is never streamed: it is the operand every tile of the product re-reads)
template <bool A_KM, bool B_KM> constexpr bool stream_a() { return (!A_KM && (GEMM_NT & 1)) || (A_KM && B_KM && (GEMM_NT & 4)); }
// 8: the spike plane of the TN products whose spikes are the layer's INPUT (dW = dx^T * s_in, MODE 1): written by the
//    forward pass long ago and read here for the last time, beside a dx that the dX product behind reads again
template <bool A_KM, bool B_KM, int MODE = 0> constexpr bool stream_b() {
    return A_KM && B_KM && ((GEMM_NT & 4) || (MODE == 1 && (GEMM_NT & 8)));
}
template <bool STREAM>
__device__ __forceinline__ f32x4 ld16(const void* q) {
    if constexpr (STREAM) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q));
    else return *reinterpret_cast<const f32x4*>(q);
}
// Which 16-byte piece of a ROW-MAJOR (ROWS x 32) operand tile a thread moves (piece index f = tid + NT * p; PPR = 8
// pieces per row of fp32 source, 4 of a bf16 plane).  Row-major order (row = f / PPR, piece = f % PPR) makes a wave's
// LDS stores collide: the images' rows are 80 bytes apart (20 banks, the stride that keeps the fragment reads
// conflict-free), so four CONSECUTIVE rows wrap the 64 banks and the fourth lands on the first — SQ_LDS_BANK_CONFLICT
// 29 % of the LDS cycles of the NT kernels, 17 % of dX (profiles/r03_pmc_lds.md).  Rows FOUR apart start 16 banks
// apart, so the lanes that are served together take rows r, r+4, r+8, r+12: each 16 lanes of a 16-byte store and each
// 32 lanes of an 8-byte store then cover the 64 banks exactly once — and a row's pieces stay on CONSECUTIVE lanes, so
// the global loads coalesce as before (the first attempt, sixteen rows per piece column, was conflict-free too and
// 17 % slower: 16-byte requests to 16 different rows per 16 lanes).
// MEASURED, NOT KEPT (GEMM_RM_MAP=1 selects it): the conflict counter goes to 0 in every kernel and LDS busy from 43 to
// 30 % (NT) and 39 to 31 % (dX) — and the products do not get faster (tools/gemm_sweep.py, one call: NT with statistics
// 389 -> 395 us, dX 793 -> 791, dense NN 759 -> 762).  The LDS port is not what these kernels wait for.
#ifndef GEMM_RM_MAP
#define GEMM_RM_MAP 0
#endif
template <int PPR>
__device__ __forceinline__ void rm_piece(int f, int& row, int& kp) {
#if GEMM_RM_MAP
    const int l = f & 63;
    if constexpr (PPR == 4) { row = ((f >> 6) << 4) + (l >> 4) + (((l >> 2) & 3) << 2); kp = l & 3; }
    else { row = ((f >> 7) << 4) + (((f >> 6) & 1) << 1) + (l >> 5) + (((l >> 3) & 3) << 2); kp = l & 7; }
#else
    row = f / PPR; kp = f % PPR;
#endif
}
template <bool KM, int ROWS, int NT, bool STREAM = false>
__device__ __forceinline__ void load_piece(f32x4& out, int p, bool fast, const float* __restrict__ P, int ld,
                                           int row0, int rows, int k0, int kend, int vec, int tid) {
    constexpr int RQ = ROWS / 4;  // pieces per k row of a KM tile
    const int f = tid + NT * p;
    int row, k;
    if constexpr (!KM) { int r_, kp_; rm_piece<8>(f, r_, kp_); row = row0 + r_; k = k0 + (kp_ << 2); }
    else               { k = k0 + f / RQ; row = row0 + ((f % RQ) << 2); }
    if (fast) {
        if constexpr (!KM) out = ld16<STREAM>(P + (size_t)row * ld + k);
        else               out = ld16<STREAM>(P + (size_t)k * ld + row);
        return;
    }
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if constexpr (!KM) {
        if (row < rows) {
            const float* q = P + (size_t)row * ld + k;
            if (vec && k + 3 < kend) v = *reinterpret_cast<const f32x4*>(q);
            else {
                if (k + 0 < kend) v.x = q[0];
                if (k + 1 < kend) v.y = q[1];
                if (k + 2 < kend) v.z = q[2];
                if (k + 3 < kend) v.w = q[3];
            }
        }
    } else {
        if (k < kend) {
            const float* q = P + (size_t)k * ld + row;
            if (vec && row + 3 < rows) v = *reinterpret_cast<const f32x4*>(q);
            else {
                if (row + 0 < rows) v.x = q[0];
                if (row + 1 < rows) v.y = q[1];
                if (row + 2 < rows) v.z = q[2];
                if (row + 3 < rows) v.w = q[3];
            }
        }
    }
    out = v;
}
// ---- the same for a spike operand held as a bf16 plane (0 / 1.0; SURVEY f: the producers write it next to
// their fp32 output): 8 elements per 16-byte piece, half the bytes through the CU's L1 fill path — which,
// beside the matrix pipe, is what bounds these kernels.
template <bool KM, int ROWS, int NT, bool STREAM = false>
__device__ __forceinline__ void load_piece16(f32x4& out, int p, bool fast, const unsigned short* __restrict__ P,
                                             int ld, int row0, int rows, int k0, int kend, int vec, int tid) {
    constexpr int RQ = ROWS / 8;  // pieces per k row of a KM tile
    const int f = tid + NT * p;
    int row, k;
    if constexpr (!KM) { int r_, kp_; rm_piece<4>(f, r_, kp_); row = row0 + r_; k = k0 + (kp_ << 3); }
    else               { k = k0 + f / RQ; row = row0 + ((f % RQ) << 3); }
    const unsigned short* q = KM ? P + (size_t)k * ld + row : P + (size_t)row * ld + k;
    if (fast) { out = ld16<STREAM>(q); return; }
    u32x4 v = {0u, 0u, 0u, 0u};
    const bool outer_ok = KM ? (k < kend) : (row < rows);
    const int inner = KM ? row : k, inner_end = KM ? rows : kend;
    if (outer_ok) {
        if (vec && inner + 7 < inner_end) v = *reinterpret_cast<const u32x4*>(q);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (inner + e < inner_end) v[e >> 1] |= (unsigned)q[e] << (16 * (e & 1));
        }
    }
    out = __builtin_bit_cast(f32x4, v);
}
template <bool KM, int ROWS, int NT>
__device__ __forceinline__ void store_piece16(const f32x4& r, int p, unsigned short* __restrict__ S, int tid) {
    constexpr int RQ = ROWS / 8;
    const int f = tid + NT * p;
    int off;  // bf16 elements, 16-byte aligned (80-byte KC rows, 16-byte multiples for KM rows)
    if constexpr (!KM) { int r_, kp_; rm_piece<4>(f, r_, kp_); off = r_ * KC_ROW + (kp_ << 3); }
    else               off = (f / RQ) * km_row<ROWS>() + ((f % RQ) << 3);
    *reinterpret_cast<f32x4*>(S + off) = r;
}
template <bool KM, int ROWS, int NT, bool H16 = false>
__device__ __forceinline__ void stage_load(f32x4 (&r)[ROWS * (H16 ? 4 : 8) / NT], const float* __restrict__ P, int ld,
                                           int row0, int rows, int k0, int kend, int vec, int tid) {
    const bool fast = tile_is_full<KM, ROWS>(row0, rows, k0, kend, vec);
#pragma unroll
    for (int p = 0; p < ROWS * (H16 ? 4 : 8) / NT; ++p) {
        if constexpr (H16)
            load_piece16<KM, ROWS, NT>(r[p], p, fast, reinterpret_cast<const unsigned short*>(P), ld, row0, rows, k0,
                                       kend, vec, tid);
        else
            load_piece<KM, ROWS, NT>(r[p], p, fast, P, ld, row0, rows, k0, kend, vec, tid);
    }
}

// ---- registers -> LDS with conversion.  SPIKE: one plane of 0/1; else three planes hi/mid/lo.
// TRUNC: exact truncation split (x = t1 + t2 + t3; AND / SUB / v_perm, ~5.5 VALU per element against ~9 for
// round-to-nearest).  Against an exact spike plane any exact split gives the same result.  The dense 6-term
// kernel uses it too: its dropped terms (t2*u3, t3*u2: <= 2^-21 of |a||b| per product in the worst case)
// measure 1.1e-8 of sum|a||b| over a 1024-long row on random data — ten times below the rounding of an fp32
// sgemm's own accumulation (1e-7) — and the conversion VALU is what the matrix pipe waits for in that kernel.
template <bool KM, int ROWS, int NT, bool SPIKE, bool TRUNC = false>
__device__ __forceinline__ void store_piece(const f32x4& r, int p, unsigned short* __restrict__ S, int tid,
                                            int e_exact = 0) {
    constexpr int RQ = ROWS / 4;
    constexpr int PLANE = plane_elems<KM, ROWS>();
    const int f = tid + NT * p;
    int off;  // in bf16 elements, 8-byte aligned
    if constexpr (!KM) { int r_, kp_; rm_piece<8>(f, r_, kp_); off = r_ * KC_ROW + (kp_ << 2); }
    else               off = (f / RQ) * km_row<ROWS>() + ((f % RQ) << 2);
    if constexpr (SPIKE) {
        // e_exact (uniform): the values are bf16-exact and their bf16 form is the upper half of the fp32
        // word; otherwise the plane holds (x != 0).  Both forms computed and selected by mask: a branch
        // here would be a basic-block boundary between the MFMA groups of the pipelined loop.
        const unsigned keep = e_exact ? 0xFFFFFFFFu : 0u;
        u32x2 w;
        w.x = (__builtin_amdgcn_perm(__float_as_uint(r.y), __float_as_uint(r.x), 0x07060302u) & keep) |
              (((r.x != 0.f ? 0x3F80u : 0u) | (r.y != 0.f ? 0x3F800000u : 0u)) & ~keep);
        w.y = (__builtin_amdgcn_perm(__float_as_uint(r.w), __float_as_uint(r.z), 0x07060302u) & keep) |
              (((r.z != 0.f ? 0x3F80u : 0u) | (r.w != 0.f ? 0x3F800000u : 0u)) & ~keep);
        *reinterpret_cast<u32x2*>(S + off) = w;
    } else if constexpr (TRUNC) {
        u32x2 w1, w2, w3;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const unsigned x0 = __float_as_uint(r[2 * pr]), x1 = __float_as_uint(r[2 * pr + 1]);
            const float r0 = r[2 * pr] - __uint_as_float(x0 & 0xFFFF0000u);
            const float r1 = r[2 * pr + 1] - __uint_as_float(x1 & 0xFFFF0000u);
            const unsigned y0 = __float_as_uint(r0), y1 = __float_as_uint(r1);
            const float q0 = r0 - __uint_as_float(y0 & 0xFFFF0000u);
            const float q1 = r1 - __uint_as_float(y1 & 0xFFFF0000u);
            w1[pr] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
            w2[pr] = __builtin_amdgcn_perm(y1, y0, 0x07060302u);
            w3[pr] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
        }
        *reinterpret_cast<u32x2*>(S + off) = w1;
        *reinterpret_cast<u32x2*>(S + PLANE + off) = w2;
        *reinterpret_cast<u32x2*>(S + 2 * PLANE + off) = w3;
    } else {
        unsigned short h[4], m[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split3(r[e], h[e], m[e], l[e]);
        *reinterpret_cast<u32x2*>(S + off) = u32x2{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)};
        *reinterpret_cast<u32x2*>(S + PLANE + off) = u32x2{(unsigned)m[0] | ((unsigned)m[1] << 16), (unsigned)m[2] | ((unsigned)m[3] << 16)};
        *reinterpret_cast<u32x2*>(S + 2 * PLANE + off) = u32x2{(unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16)};
    }
}
// ---- the bf16 operand mode (sparch_set_operand_precision): ONE plane, each fp32 value rounded to nearest-even
// (v_cvt_pk_bf16_f32 on gfx950)
template <bool KM, int ROWS, int NT>
__device__ __forceinline__ void store_piece_rne1(const f32x4& r, int p, unsigned short* __restrict__ S, int tid) {
    constexpr int RQ = ROWS / 4;
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const int f = tid + NT * p;
    int off;
    if constexpr (!KM) { int r_, kp_; rm_piece<8>(f, r_, kp_); off = r_ * KC_ROW + (kp_ << 2); }
    else               off = (f / RQ) * km_row<ROWS>() + ((f % RQ) << 2);
    u32x2 w;
    w.x = __builtin_bit_cast(unsigned, bf16x2{(__bf16)r.x, (__bf16)r.y});
    w.y = __builtin_bit_cast(unsigned, bf16x2{(__bf16)r.z, (__bf16)r.w});
    *reinterpret_cast<u32x2*>(S + off) = w;
}
// ---- MFMA fragment (8 bf16 of row/col `idx`, k = 16*ks + 8*h + j) from an LDS plane
template <bool KM, int ROWS>
__device__ __forceinline__ u32x4 frag_read(const unsigned short* __restrict__ S, int idx_base, int lane, int ks) {
    constexpr int KM_ROW = km_row<ROWS>();
#if defined(SPARCH_REC_PROF) && defined(GA_NO_FRAG)  // ablation: no LDS fragment reads
    return u32x4{(unsigned)lane, (unsigned)ks, (unsigned)idx_base, 0x3F803F80u};
#endif
    if constexpr (!KM) {
        const int r = lane & 31, h = lane >> 5;
        return *reinterpret_cast<const u32x4*>(S + (idx_base + r) * KC_ROW + 16 * ks + 8 * h);
    } else {
        // hardware transpose read: the 16-lane group g reads rows k0..k0+3 x 16 columns and each lane
        // receives its column's four k values; two reads give k0..k0+7 (guide T10)
        const int g = lane >> 4, i = lane & 15, q = i >> 2, p4 = i & 3;
        const int col = idx_base + 16 * (g & 1) + 4 * p4;
        const int k0 = 16 * ks + 8 * (g >> 1);
        const unsigned short* a0 = S + (k0 + q) * KM_ROW + col;
        const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0)));
        const u32x2 hi = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * KM_ROW)));
        return u32x4{lo.x, lo.y, hi.x, hi.y};  // already packed: element j in bits 16*(j&1) of dword j>>1
    }
}

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
#if defined(SPARCH_REC_PROF) && defined(GA_NO_MFMA)  // ablation: keep the operands alive, drop the MFMA
    asm volatile("" ::"v"(a), "v"(b));
    return c;
#endif
#if defined(SPARCH_REC_PROF) && defined(GA_FRAG_UNUSED)  // ablation: reads issued and waited for, MFMA on constants
    asm volatile("" ::"v"(a), "v"(b));
    const u32x4 k = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, k), __builtin_bit_cast(bf16x8, k), c, 0, 0, 0);
#endif
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// MODE 0: A is the spike operand; MODE 1: B is; MODE 2: both operands are dense fp32 and both are split
// (six cross terms hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid; see store_piece for the size of the rest).
//
// Tile shape: 4 waves as 2 x 2, each wave owns WI x WJ MFMA tiles of 32 x 32, so the workgroup tile is
// (64 WI) x (64 WJ).  Per 16-deep k step a wave reads pa*WI + pb*WJ fragments (1 KiB each) from LDS for
// pa*pb-ish*WI*WJ MFMAs; at the old 2 x 2 shape that was 8 reads per 12 MFMAs = 85 B/clk/CU of LDS read
// traffic at full MFMA rate against a 128 B/clk LDS — the measured limiter (ablation: dropping the
// fragment reads took the TN kernel from 0.50 to 0.22 ms).  The spike-operand kernels therefore put 4 tiles
// on the single-plane (spike) side and 2 on the three-plane side: 10 reads per 24 MFMAs.
// BIG (TN spike kernels only): 256 x 256 workgroup tile.  What bounds these kernels beside the matrix pipe is
// the CU's fill rate from L2 / Infinity Cache (~70 / ~33 GB/s per CU measured, MI355X_MICROARCH.md), and a
// 256 x 128 tile pulls 48 KB per 48 MFMAs per SIMD; 256 x 256 pulls 64 KB per 96.  (The NT images do not fit
// two LDS stages at that size, and there the dense operand is the small, L2-resident weight matrix anyway.)
template <int MODE, bool FAST, bool BIG = false> struct Shape {
    // FAST: 8 waves (2 per SIMD, so that one wave's vector-memory issue stalls and barrier waits are
    // covered by the other's MFMAs) in one workgroup per CU; general kernel: 4 waves, 2 workgroups per CU.
    static constexpr int WM = FAST ? (MODE == 1 ? 4 : 2) : 2;   // waves along M
    static constexpr int WN = FAST ? (MODE == 1 ? 2 : 4) : 2;   // waves along N
    static constexpr int WI = FAST ? (MODE == 0 ? 4 : (MODE == 1 ? (BIG ? 2 : 1) : (BIG ? 4 : 2))) : 2;  // 32-row MFMA tiles per wave
    static constexpr int WJ = FAST ? (MODE == 1 ? 4 : (BIG ? 2 : 1)) : 2;
    static constexpr int BM = 32 * WI * WM, BN = 32 * WJ * WN;
    static constexpr int NT = 64 * WM * WN;
    // workgroups per CU the register / LDS budget is set for
    static constexpr int OCC = FAST ? 1 : (MODE == 2 ? 2 : 3);
};

// 256 x 256 workgroup tiles: the TN spike kernels of the exact mode, and EVERY pipelined kernel of the bf16 operand
// mode (NP = 1: one plane per operand, so the images of a 256 x 256 tile fit two LDS stages, and a 32-deep K tile
// of the smaller shapes holds only 4-8 MFMAs per wave between two barriers)
template <bool A_KM, bool B_KM, int MODE, bool FAST, int NP>
constexpr bool big_tile() { return FAST && ((A_KM && B_KM && MODE != 2) || NP == 1); }

extern __shared__ __attribute__((aligned(16))) unsigned short dyn_lds[];

template <bool A_KM, bool B_KM, int MODE, bool FAST, int NP = 3>
constexpr int stage_elems() {  // bf16 elements of one LDS stage (all planes of both operand tiles)
    using S = Shape<MODE, FAST, big_tile<A_KM, B_KM, MODE, FAST, NP>()>;
    return (MODE == 0 ? 1 : NP) * plane_elems<A_KM, S::BM>() + (MODE == 1 ? 1 : NP) * plane_elems<B_KM, S::BN>();
}

// FAST: one workgroup per CU (one wave per SIMD, up to 512 registers), TWO LDS stages, and the whole
// pipeline of a K tile folded into its MFMA phase: while the 48 MFMAs of tile t run from stage t&1, the
// same wave converts tile t+1 (already in registers) into the other stage and re-issues those registers'
// loads for tile t+2 — VALU, LDS stores and global loads all go into the shadow of the MFMAs, and there
// is one barrier per tile.  (With two workgroups per CU taking turns — convert phase, barrier, MFMA
// phase, barrier — the matrix pipe was busy 47 % of the time: neither workgroup's latency chain was
// short enough for two to cover each other.)
// !FAST: general shapes (small or unaligned operands): bounds-checked loads, single stage, two barriers.
// NP: planes of a dense operand — 3 = the exact split (default), 1 = the bf16 operand mode (one rounding, one
// MFMA per product; MODE 2 then has ONE term).
template <bool A_KM, bool B_KM, int MODE, int EPI, bool FAST, bool S16, bool BPRE = false, int NP = 3, bool APRE = false>
__global__ __launch_bounds__((Shape<MODE, FAST>::NT), (Shape<MODE, FAST>::OCC)) void gemm_spike_kernel(SArgs g) {
    static_assert(!S16 || MODE != 2, "a bf16 plane is a spike operand");
    static_assert(!BPRE || (FAST && MODE != 1), "pre-split B: pipelined kernel, B is the dense operand");
    static_assert(!APRE || (FAST && MODE != 0 && NP == 3), "pre-split A: pipelined exact kernel, A is the dense operand");
    static_assert(NP == 3 || (NP == 1 && !BPRE), "dense operands: three exact planes or one rounded plane");
    constexpr bool BIGT = big_tile<A_KM, B_KM, MODE, FAST, NP>();
    using S = Shape<MODE, FAST, BIGT>;
    constexpr bool SPIKE_A = MODE == 0;
    constexpr bool SPIKE_B = MODE == 1;
    constexpr int WI = S::WI, WJ = S::WJ, WN = S::WN, BM = S::BM, BN = S::BN, NT = S::NT;
    constexpr int A_PLANES = SPIKE_A ? 1 : NP;
    [[maybe_unused]] constexpr int B_PLANES = SPIKE_B ? 1 : NP;
    constexpr int PLANE_A = plane_elems<A_KM, BM>(), PLANE_B = plane_elems<B_KM, BN>();
    constexpr int STAGE = stage_elems<A_KM, B_KM, MODE, FAST, NP>();
    unsigned short* const lds = dyn_lds;  // [FAST ? 2 : 1][A planes | B planes]

    if (g.gate != nullptr && *g.gate != g.gate_want) return;  // uniform: every workgroup reads the same word
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_n = (g.N + BN - 1) / BN;
    // Workgroups go to the 8 XCDs round-robin by linear id, and each XCD has its own L2.  An XCD owns a
    // CONTIGUOUS range of (split, tile_m, tile_n): tiles that share an operand panel then share one L2 instead
    // of pulling the panel through all eight.  (Any grid size: the first grid % 8 XCDs have one workgroup
    // more, the first total % 8 ranges one tile more.)
    // PERSISTENT launches (grid < tiles: one workgroup per CU): a workgroup walks its XCD's range with the
    // stride of that XCD's workgroups — no dispatch gap and no wave start-up between its tiles (measured per
    // workgroup of the NT product: 38 us of life, ~5 us to the next one's first instruction).
    const int tiles_all = tiles_n * ((g.M + BM - 1) / BM);
    const int total = tiles_all * g.n_splits;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int wgs_here = (gridDim.x >> 3) + (xcd < (int)(gridDim.x & 7) ? 1 : 0);
    const int lin_end = (xcd + 1) * (total >> 3) + min(xcd + 1, total & 7);
    // (the general kernels and the 256 x 256 TN kernels — long K ranges, one item per workgroup — have no
    // registers to spare for a loop: theirs is a single trip at compile time)
    constexpr bool WALKS = FAST && !BIGT;
    int lin = xcd * (total >> 3) + min(xcd, total & 7) + idx;
    if (lin >= lin_end) return;
    do {
    const int split = lin / tiles_all, tile = lin - split * tiles_all;
    const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
    // FAST (host guarantees M >= BM, N >= BN, 16-byte loadable rows): an edge tile is SHIFTED back inside
    // the matrix instead of being bounds-checked; it recomputes a few rows/columns of its neighbour and
    // stores the same values again (same operands, same order: bit-identical), so the steady-state loop
    // has no per-element predicates.
    const int m0 = FAST ? min(tile_m * BM, g.M - BM) : tile_m * BM;
    const int n0 = FAST ? min(tile_n * BN, g.N - BN) : tile_n * BN;
    const int k_begin = split * g.k_per_split;
    const int k_end = min(g.K, k_begin + g.k_per_split);

    f32x16 acc[WI][WJ];
#pragma unroll
    for (int i = 0; i < WI; ++i)
#pragma unroll
        for (int j = 0; j < WJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr bool A16 = S16 && SPIKE_A, B16 = S16 && SPIKE_B;     // operand arrives as a bf16 plane
    constexpr int NPB1 = BN * 4 / NT;  // 16-byte pieces per thread of ONE bf16 plane of the B tile
    constexpr int NPA1 = BM * 4 / NT;  // ... of ONE bf16 plane of the A tile
    constexpr int NPA = APRE ? 3 * NPA1 : BM * (A16 ? 4 : 8) / NT, NPB = BPRE ? 3 * NPB1 : BN * (B16 ? 4 : 8) / NT;  // pieces per thread and tile
    // FAST: two register sets, tiles in flight two K tiles ahead of the MFMAs (a load issued in phase t is
    // converted in phase t + 2; one phase of a 256 x 128 tile is ~0.8 us, less than a loaded HBM round trip)
    // (not the 256 x 256 TN kernels: their phase is twice as long and they have no registers to spare)
    constexpr bool AHEAD2 = FAST && !BIGT;
    f32x4 ra[NPA], rb[NPB];
    [[maybe_unused]] f32x4 ra2[AHEAD2 ? NPA : 1], rb2[AHEAD2 ? NPB : 1];

    // registers -> LDS stage at `st` (piece q of the NPA + NPB pieces a thread owns)
    auto convert_piece = [&](auto& ra, auto& rb, int q, unsigned short* st) __attribute__((always_inline)) {
        if (q < NPA) {
            if constexpr (APRE)  // plane q / NPA1 straight into its LDS image: no conversion
                store_piece16<A_KM, BM, NT>(ra[q], q % NPA1, st + (q / NPA1) * PLANE_A, tid);
            else if constexpr (A16) store_piece16<A_KM, BM, NT>(ra[q], q, st, tid);
            else if constexpr (NP == 1 && !SPIKE_A) store_piece_rne1<A_KM, BM, NT>(ra[q], q, st, tid);
            else store_piece<A_KM, BM, NT, SPIKE_A, true>(ra[q], q, st, tid, g.e_exact);
        } else if (q < NPA + NPB) {
#ifdef GEMM_ABL_B
            if (MODE == 2 || MODE == 0) { asm volatile("" ::"v"(rb[q - NPA])); return; }
#endif
            if constexpr (BPRE)  // plane (q - NPA) / NPB1 straight into its LDS image: no conversion
                store_piece16<B_KM, BN, NT>(rb[q - NPA], (q - NPA) % NPB1, st + A_PLANES * PLANE_A + ((q - NPA) / NPB1) * PLANE_B, tid);
            else if constexpr (B16) store_piece16<B_KM, BN, NT>(rb[q - NPA], q - NPA, st + A_PLANES * PLANE_A, tid);
            else if constexpr (NP == 1 && !SPIKE_B) store_piece_rne1<B_KM, BN, NT>(rb[q - NPA], q - NPA, st + A_PLANES * PLANE_A, tid);
            else store_piece<B_KM, BN, NT, SPIKE_B, true>(rb[q - NPA], q - NPA, st + A_PLANES * PLANE_A, tid, g.e_exact);
        }
    };
    // global -> registers, full in-range tile at K offset k (FAST only)
    constexpr bool STR_A = stream_a<A_KM, B_KM>(), STR_B = stream_b<A_KM, B_KM, MODE>();
    auto fetch_piece = [&](auto& ra, auto& rb, int q, int k) __attribute__((always_inline)) {
        if (q < NPA) {
            if constexpr (APRE)
                load_piece16<A_KM, BM, NT, STR_A>(ra[q], q % NPA1, true, g.Ap + (q / NPA1) * g.ap_stride, g.lda, m0, g.M, k, k_end, 1, tid);
            else if constexpr (A16)
                load_piece16<A_KM, BM, NT, STR_A>(ra[q], q, true, reinterpret_cast<const unsigned short*>(g.A), g.lda, m0, g.M, k, k_end, 1, tid);
            else load_piece<A_KM, BM, NT, STR_A>(ra[q], q, true, g.A, g.lda, m0, g.M, k, k_end, 1, tid);
        } else if (q < NPA + NPB) {
            if constexpr (BPRE)
                load_piece16<B_KM, BN, NT, STR_B>(rb[q - NPA], (q - NPA) % NPB1, true, g.Bp + ((q - NPA) / NPB1) * g.bp_stride, g.ldb, n0,
                                           g.N, k, k_end, 1, tid);
            else if constexpr (B16)
                load_piece16<B_KM, BN, NT, STR_B>(rb[q - NPA], q - NPA, true, reinterpret_cast<const unsigned short*>(g.B), g.ldb, n0, g.N, k, k_end, 1, tid);
            else load_piece<B_KM, BN, NT, STR_B>(rb[q - NPA], q - NPA, true, g.B, g.ldb, n0, g.N, k, k_end, 1, tid);
        }
    };

    // The MFMA phase of one K tile held in the LDS stage `cur`.  Fragment reads run one group AHEAD of the
    // MFMAs that use them, and the fragments of a phase's FIRST group are already in registers when it
    // starts (pre_read).  `side(q)` is called once per piece q, spread evenly over the groups, in the same
    // scheduling region as that group's MFMAs (FAST: convert piece q of the next tile and re-issue its
    // load).  No branch inside: a branch is a basic-block boundary, and the compiler's waitcnt insertion
    // drains every outstanding load at one.
    // HAS_NEXT (pipelined loop): the workgroup barrier that hands the other stage `nxt` over sits INSIDE the
    // phase, before its last TAIL groups — by then every side piece is stored and every fragment of `cur`
    // is read — and the next phase's first fragments are read from `nxt` right behind it, so the matrix
    // pipe has those groups' MFMAs (both waves of the SIMD) to run while the reads are in flight.  With the
    // barrier at the phase boundary all eight waves started each K tile on LDS latency: per-wave stamps
    // showed the SIMDs done 1950 cycles after the barrier against 1536 cycles of MFMAs.
    GP_DECL
    constexpr int WS = SPIKE_A ? WI : WJ, WD = SPIKE_A ? WJ : WI;   // spike-side / dense-side tiles per wave
    constexpr int SB = FAST ? 2 : 1;
    [[maybe_unused]] u32x4 fa[MODE == 2 ? 2 : 1][MODE == 2 ? WI : 1][NP], fb[MODE == 2 ? 2 : 1][MODE == 2 ? WJ : 1][NP];
    [[maybe_unused]] u32x4 fs[MODE != 2 ? SB : 1][MODE != 2 ? WS : 1], fd[2][MODE != 2 ? WD : 1];
    // MODE 2: all fragments of a 16-deep k step, double buffered across the two steps
    auto read_step = [&](const unsigned short* st, int ks) __attribute__((always_inline)) {
        const unsigned short* As = st;
        const unsigned short* Bs = st + A_PLANES * PLANE_A;
#pragma unroll
        for (int i = 0; i < WI; ++i)
#pragma unroll
            for (int p = 0; p < NP; ++p)
                fa[ks][i][p] = frag_read<A_KM, BM>(As + p * PLANE_A, (wm * WI + i) * 32, lane, ks);
#ifdef GEMM_ABL_B  // timing ablation (wrong results): the weight operand's LDS traffic (staging stores + fragment reads)
                   // dropped.  Standalone on random operands (tools/gemm_sweep.py, round 3): NT with statistics 383 ->
                   // 330 us, dX 746 -> 635 us — the bound on what W fragments loaded straight from a fragment-ordered pack
                   // in L2 could give (-14 %; inside the training step the ablated run reads 0.71 -> 0.54 ms, but there
                   // its garbage output turns the next steps' operands into NaNs, which draw less power: clocks, not LDS)
        if (MODE == 2) { asm volatile("" : "+v"(fb[ks][0][0]), "+v"(fb[ks][0][1]), "+v"(fb[ks][0][NP - 1])); return; }
#endif
#pragma unroll
        for (int j = 0; j < WJ; ++j)
#pragma unroll
            for (int p = 0; p < NP; ++p)
                fb[ks][j][p] = frag_read<B_KM, BN>(Bs + p * PLANE_B, (wn * WJ + j) * 32, lane, ks);
    };
    // MODE 0 / 1: dense fragments double buffered by group, spike fragments by k step (the general kernel's
    // 2-workgroup register budget has no room for the second spike buffer: there the spike fragments are
    // re-read in place at the k-step boundary)
    auto read_spike = [&](const unsigned short* st, int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < WS; ++i) {
            if constexpr (SPIKE_A) fs[ks % SB][i] = frag_read<A_KM, BM>(st, (wm * WI + i) * 32, lane, ks);
            else fs[ks % SB][i] = frag_read<B_KM, BN>(st + A_PLANES * PLANE_A, (wn * WJ + i) * 32, lane, ks);
        }
    };
    auto read_dense = [&](const unsigned short* st, int buf, int ks, int p) __attribute__((always_inline)) {
#ifdef GEMM_ABL_B  // (the same ablation for the spike x weight products)
        if (SPIKE_A) { asm volatile("" : "+v"(fd[buf][0])); return; }
#endif
#pragma unroll
        for (int j = 0; j < WD; ++j) {
            if constexpr (SPIKE_A)
                fd[buf][j] = frag_read<B_KM, BN>(st + A_PLANES * PLANE_A + p * PLANE_B, (wn * WJ + j) * 32, lane, ks);
            else fd[buf][j] = frag_read<A_KM, BM>(st + p * PLANE_A, (wm * WI + j) * 32, lane, ks);
        }
    };
    // the fragments of a phase's first group, from the stage it will run on
    auto pre_read = [&](const unsigned short* st) __attribute__((always_inline)) {
        if constexpr (MODE == 2) read_step(st, 0);
        else { read_spike(st, 0); read_dense(st, 0, 0, NP - 1); }
    };
    auto mfma_phase = [&](const unsigned short* cur, const unsigned short* nxt, auto has_next, auto side)
                          __attribute__((always_inline)) {
        constexpr bool HAS_NEXT = decltype(has_next)::value;
        constexpr int NTERM = NP == 3 ? 6 : 1;              // cross terms of a dense x dense product
        constexpr int NG = MODE == 2 ? 2 * NTERM : 2 * NP;  // MFMA groups of WI x WJ per K tile
        constexpr int TAIL = !HAS_NEXT ? 0 : (MODE == 2 && NP == 3 ? GEMM_TAIL2 : 1);  // groups behind the barrier
        constexpr int PPG = (NPA + NPB + NG - TAIL - 1) / (NG - TAIL);
        auto handover = [&]() __attribute__((always_inline)) {
            GP_STAMP(3);  // MFMA groups with the next tile's conversion and the loads after it
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            GP_STAMP(4);  // barrier
            pre_read(nxt);
        };
        if constexpr (MODE == 2) {
            // both operands dense: six cross terms per 16-deep k step; a group = one term
            // (pa, pb) pairs, small terms first: mid*mid, lo*hi, hi*lo, mid*hi, hi*mid, hi*hi
            constexpr int PA[6] = {NP == 3 ? 1 : 0, 2, 0, 1, 0, 0};
            constexpr int PB[6] = {NP == 3 ? 1 : 0, 0, 2, 0, 1, 0};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int c = 0; c < NTERM; ++c) {
                    const int gi = ks * NTERM + c;
                    if (gi == 0) read_step(cur, 1);
                    if (HAS_NEXT && gi == NG - TAIL) handover();
                    if (gi < NG - TAIL) {
#pragma unroll
                        for (int q = 0; q < PPG; ++q) side(gi * PPG + q);
                    }
#pragma unroll
                    for (int i = 0; i < WI; ++i)
#pragma unroll
                        for (int j = 0; j < WJ; ++j)
                            acc[i][j] = mfma_bf16(fa[ks][i][PA[c]], fb[ks][j][PB[c]], acc[i][j]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            // one spike-side plane, three dense-side planes: a group = one plane of one k step
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                const int ks = gi / NP, p = NP - 1 - gi % NP;  // smallest plane first
                if (p > 0) read_dense(cur, (gi + 1) & 1, ks, p - 1);
                else if (ks == 0) { if constexpr (SB == 2) read_spike(cur, 1); read_dense(cur, (gi + 1) & 1, 1, NP - 1); }
                if (HAS_NEXT && gi == NG - TAIL) handover();
                if (gi < NG - TAIL) {
#pragma unroll
                    for (int q = 0; q < PPG; ++q) side(gi * PPG + q);
                }
#pragma unroll
                for (int i = 0; i < WI; ++i)
#pragma unroll
                    for (int j = 0; j < WJ; ++j) {
                        if constexpr (SPIKE_A) acc[i][j] = mfma_bf16(fs[ks % SB][i], fd[gi & 1][j], acc[i][j]);
                        else                   acc[i][j] = mfma_bf16(fd[gi & 1][i], fs[ks % SB][j], acc[i][j]);
                    }
                if constexpr (SB == 1) { if (p == 0 && ks == 0) read_spike(cur, 1); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    constexpr std::true_type with_next{};
    constexpr std::false_type last_tile{};
    auto no_side = [](int) __attribute__((always_inline)) {};

    if constexpr (FAST) {
        const int nt = (k_end - k_begin) / BK;  // full tiles
        if (nt > 0) {
            // prologue: tile 0 -> stage 0 (through set 1), tile 1 -> set 2, tile 2 -> set 1
            // (past the last tile: fetch the last tile again — in range, never used)
            auto k_of = [&](int t) { return k_begin + min(t, nt - 1) * BK; };
#pragma unroll
            for (int q = 0; q < NPA + NPB; ++q) fetch_piece(ra, rb, q, k_begin);
            if constexpr (AHEAD2) {
#pragma unroll
                for (int q = 0; q < NPA + NPB; ++q) fetch_piece(ra2, rb2, q, k_of(1));
            }
#pragma unroll
            for (int q = 0; q < NPA + NPB; ++q) { convert_piece(ra, rb, q, lds); fetch_piece(ra, rb, q, k_of(AHEAD2 ? 2 : 1)); }
            __syncthreads();
            pre_read(lds);
            // phase t: MFMAs of tile t from stage t & 1; tile t + 1 (set 2 for even t, set 1 for odd t)
            // -> the other stage, and that set's loads re-issued for tile t + 3
            auto phase = [&](auto& xa, auto& xb, int t) __attribute__((always_inline)) {
                GP_STAMP(-1);
                const unsigned short* cur = lds + (t & 1) * STAGE;
                unsigned short* nxt = lds + ((t + 1) & 1) * STAGE;
                const int k3 = k_of(t + (AHEAD2 ? 3 : 2));
                mfma_phase(cur, nxt, with_next, [&](int q) __attribute__((always_inline)) {
#if !(defined(SPARCH_REC_PROF) && defined(GA_NO_STORE))
                    convert_piece(xa, xb, q, nxt);
#endif
#if !(defined(SPARCH_REC_PROF) && defined(GA_NO_GLOAD))
                    fetch_piece(xa, xb, q, k3);
#endif
                });
            };
            if constexpr (AHEAD2) {
                int t = 0;
                for (; t + 2 < nt; t += 2) {
                    phase(ra2, rb2, t);
                    phase(ra, rb, t + 1);
                }
                if (t + 1 < nt) phase(ra2, rb2, t);
            } else {  // one set: tile t + 1 converted while its loads for tile t + 2 are re-issued
                for (int t = 0; t + 1 < nt; ++t) phase(ra, rb, t);
            }
            mfma_phase(lds + ((nt - 1) & 1) * STAGE, nullptr, last_tile, no_side);
            __syncthreads();
        }
        const int k_tail = k_begin + nt * BK;
        if constexpr (!BPRE && !APRE)   // (pre-split operands: the host launches these kernels for K % 32 == 0 only)
        if (k_tail < k_end) {  // K tail (< 32 deep): element-wise bounds-checked loads, zero filled
            stage_load<A_KM, BM, NT, A16>(ra, g.A, g.lda, m0, g.M, k_tail, k_end, 0, tid);
            stage_load<B_KM, BN, NT, B16>(rb, g.B, g.ldb, n0, g.N, k_tail, k_end, 0, tid);
#pragma unroll
            for (int q = 0; q < NPA + NPB; ++q) convert_piece(ra, rb, q, lds);
            __syncthreads();
            pre_read(lds);
            mfma_phase(lds, nullptr, last_tile, no_side);
            __syncthreads();
        }
    } else {
        stage_load<A_KM, BM, NT, A16>(ra, g.A, g.lda, m0, g.M, k_begin, k_end, g.a_vec, tid);
        stage_load<B_KM, BN, NT, B16>(rb, g.B, g.ldb, n0, g.N, k_begin, k_end, g.b_vec, tid);
        for (int k0 = k_begin; k0 < k_end; k0 += BK) {
#pragma unroll
            for (int q = 0; q < NPA + NPB; ++q) convert_piece(ra, rb, q, lds);
            __syncthreads();
            if (k0 + BK < k_end) {
                stage_load<A_KM, BM, NT, A16>(ra, g.A, g.lda, m0, g.M, k0 + BK, k_end, g.a_vec, tid);
                stage_load<B_KM, BN, NT, B16>(rb, g.B, g.ldb, n0, g.N, k0 + BK, k_end, g.b_vec, tid);
            }
            pre_read(lds);
            mfma_phase(lds, nullptr, last_tile, no_side);
            __syncthreads();
        }
    }
    GP_FLUSH()

    // ---- epilogue (C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*h)
    float* Cz = g.C + (size_t)split * g.c_split_stride;
    float csum[WJ], csq[WJ];
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
        csum[j] = 0.f; csq[j] = 0.f;
        const int col = n0 + (wn * WJ + j) * 32 + li;
        float bj = 0.f;
        if constexpr (EPI & EPI_BIAS) bj = (g.bias != nullptr && (FAST || col < g.N)) ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < WI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wm * WI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                float v = acc[i][j][r] * g.scale;
                if constexpr (EPI & EPI_BIAS) v = v + bj;
                // FAST: the (shifted) tile lies inside C — no predicates, the 64 stores of a lane go out
                // back to back (a predicated store is its own basic block, and hipcc then waits for the
                // previous store's acknowledgement, vmcnt(0), in each one)
                if (FAST || (row < g.M && col < g.N)) {
                    if constexpr (FAST && !A_KM && (GEMM_NT & 2)) __builtin_nontemporal_store(v, Cz + (size_t)row * g.ldc + col);
                    else Cz[(size_t)row * g.ldc + col] = v;
                    if constexpr (EPI & EPI_STATS) { csum[j] += v; csq[j] += v * v; }
                }
            }
        }
    }
    if constexpr (EPI & EPI_STATS) {
        // column sum / sum of squares per 128-row block of C (the layout sparch_bn_finalize reads:
        // [2 (sum|sq)][ceil(M/128)][N]).  A wave covers 32*WI rows: at WI = 4 that is one whole block,
        // at WI = 2 the two wm halves of the workgroup add up to one.
        float* red = reinterpret_cast<float*>(lds);  // [2 (sum|sq)][2 (wm)][BN] (<= 4 KiB)
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            csum[j] += __shfl_xor(csum[j], 32);
            csq[j] += __shfl_xor(csq[j], 32);
        }
        if (h == 0) {
#pragma unroll
            for (int j = 0; j < WJ; ++j) {
                const int c = (wn * WJ + j) * 32 + li;
                red[(0 * 2 + wm) * BN + c] = csum[j];
                red[(1 * 2 + wm) * BN + c] = csq[j];
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < g.N && g.colstat != nullptr) {
            const int blocks = (g.M + 127) / 128;
            if constexpr (WI == 2) {
                g.colstat[(size_t)tile_m * g.N + n0 + tid] = red[0 * BN + tid] + red[1 * BN + tid];
                g.colstat[(size_t)(blocks + tile_m) * g.N + n0 + tid] = red[2 * BN + tid] + red[3 * BN + tid];
            } else {
                static_assert(WI == 2 || WI == 4, "colstat blocks are 128 rows");
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const int blk = 2 * tile_m + w;
                    if (blk < blocks) {
                        g.colstat[(size_t)blk * g.N + n0 + tid] = red[(0 * 2 + w) * BN + tid];
                        g.colstat[(size_t)(blocks + blk) * g.N + n0 + tid] = red[(1 * 2 + w) * BN + tid];
                    }
                }
            }
        }
    }
    __syncthreads();  // the next tile's staging overwrites this tile's statistics rows / last stage
    } while (WALKS && (lin += wgs_here) < lin_end);  // tile loop
}

// (ld_ws: row stride of the slabs, 0 = N; the ragged-plane TN product keeps its slabs at the padded width)
__global__ void splitk_reduce_kernel2(const float* __restrict__ ws, float* __restrict__ C, int M, int N, int ldc,
                                      int splits, int zero_diag, int accumulate, int ld_ws = 0) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)M * N;
    if (i >= total) return;
    const int m = (int)(i / N), n = (int)(i % N);
    const size_t lw = ld_ws ? (size_t)ld_ws : (size_t)N, slab = (size_t)M * lw, at = (size_t)m * lw + n;
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += ws[(size_t)z * slab + at];
    if (zero_diag && m == n) s = 0.f;
    float* c = C + (size_t)m * ldc + n;
    *c = accumulate ? (*c + s) : s;
}

// Split count of a TN product computed by the MODE kernel: one full round of co-resident workgroups (a
// partial second round costs more than the shorter K range per workgroup gains).
int target_wgs(int occ) {
    static const int env = [] { const char* e = getenv("SPARCH_GEMM_TARGET_WGS"); return e ? atoi(e) : 0; }();
    if (env > 0) return env;
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n;
    }();
    return occ * cus;
}
template <int MODE>
int choose_splits(int M, int N, int K) {  // TN products only
    if (sparch_operand_bf16()) {  // every pipelined kernel of that mode works on 256 x 256 tiles
        using SB = Shape<MODE, true, true>;
        const int tiles_b = cdiv(M, SB::BM) * cdiv(N, SB::BN);
        int sb = target_wgs(1) / tiles_b;
        if (sb > cdiv(K, BK) / 8) sb = cdiv(K, BK) / 8;
        return sb < 1 ? 1 : sb;
    }
    using S = Shape<MODE, true, MODE != 2>;
    const int tiles = cdiv(M, S::BM) * cdiv(N, S::BN);
    const int kt = cdiv(K, BK);
    int s = target_wgs(1) / tiles;  // the pipelined kernel runs one workgroup per CU
    if (s > kt / 8) s = kt / 8;
    return s < 1 ? 1 : s;
}

template <bool A_KM, bool B_KM, int MODE, int EPI, bool S16, int NP = 3>
bool fast_ok(const SArgs& g) {
    using S = Shape<MODE, true, big_tile<A_KM, B_KM, MODE, true, NP>()>;
    constexpr int BM = S::BM, BN = S::BN;
    constexpr int QA = (S16 && MODE == 0) ? 8 : 4, QB = (S16 && MODE == 1) ? 8 : 4;  // elements per 16 bytes
    // the shifted-edge-tile kernel needs whole tiles to exist, 16-byte rows, and (for the BatchNorm
    // statistics, which are kept per 128-row block) no partial row tile
    return g.a_vec && g.b_vec && g.M >= BM && g.N >= BN && (!A_KM || g.M % QA == 0) && (!B_KM || g.N % QB == 0) &&
           (!(EPI & EPI_STATS) || g.M % BM == 0) && g.k_per_split % BK == 0 &&
           g.k_per_split >= 8 * BK;  // a short K range never fills the pipeline: general kernel, 2 workgroups per CU
}

template <bool A_KM, bool B_KM, int MODE, int EPI, bool FAST, bool S16, bool BPRE = false, int NP = 3, bool APRE = false>
int launch_variant(SArgs& g, int splits, hipStream_t st) {
    constexpr bool BIGT = big_tile<A_KM, B_KM, MODE, FAST, NP>();
    using S = Shape<MODE, FAST, BIGT>;
    const int work = cdiv(g.M, S::BM) * cdiv(g.N, S::BN) * splits;
    g.n_splits = splits;
    // pipelined kernels (one workgroup per CU): more work items than CUs -> a persistent grid of one workgroup
    // per CU walking them (SPARCH_GEMM_PERSISTENT=0: one workgroup per item)
    static const bool persistent = [] { const char* e = getenv("SPARCH_GEMM_PERSISTENT"); return !e || atoi(e) != 0; }();
    const int cus = target_wgs(1);
    const int wgs = (FAST && !BIGT && persistent && work > cus) ? cus : work;
    constexpr size_t lds_bytes = (size_t)(FAST ? 2 : 1) * stage_elems<A_KM, B_KM, MODE, FAST, NP>() * sizeof(unsigned short);
    auto kernel = gemm_spike_kernel<A_KM, B_KM, MODE, EPI, FAST, S16, BPRE, NP, APRE>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (attr != hipSuccess) { sparch_note_hip_error((int)attr); return SPARCH_ELAUNCH; }
    hipLaunchKernelGGL(kernel, dim3(wgs, 1, 1), dim3(Shape<MODE, FAST>::NT), lds_bytes, st, g);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

template <bool A_KM, bool B_KM, int MODE, int EPI, bool S16 = false>
int launch(SArgs& g, int splits, hipStream_t st) {
    if (sparch_operand_bf16()) {  // bf16 operand mode: the same kernels with ONE rounded plane per dense operand
        if (fast_ok<A_KM, B_KM, MODE, EPI, S16, 1>(g))
            return launch_variant<A_KM, B_KM, MODE, EPI, true, S16, false, 1>(g, splits, st);
        return launch_variant<A_KM, B_KM, MODE, EPI, false, S16, false, 1>(g, splits, st);
    }
    if (fast_ok<A_KM, B_KM, MODE, EPI, S16>(g)) return launch_variant<A_KM, B_KM, MODE, EPI, true, S16>(g, splits, st);
    return launch_variant<A_KM, B_KM, MODE, EPI, false, S16>(g, splits, st);
}
// the same with B's pre-split planes when the pipelined kernel applies (whole K tiles, 16-byte plane rows);
// otherwise the ordinary kernels convert the fp32 B on the fly — same truncation split, identical results
template <bool A_KM, bool B_KM, int MODE, int EPI, bool S16 = false>
int launch_wp(SArgs& g, int splits, hipStream_t st) {
    const bool planes_ok = g.Bp != nullptr && aligned16(g.Bp) && g.ldb % 8 == 0 && (g.bp_stride % 8) == 0 &&
                           g.k_per_split % BK == 0 && g.K % BK == 0 && (!B_KM || g.N % 8 == 0);
    if (planes_ok && !sparch_operand_bf16() && fast_ok<A_KM, B_KM, MODE, EPI, S16>(g))
        return launch_variant<A_KM, B_KM, MODE, EPI, true, S16, true>(g, splits, st);
    return launch<A_KM, B_KM, MODE, EPI, S16>(g, splits, st);
}

// ... and with A's pre-split planes (B's too where BPRE says so), under the same conditions; otherwise the ordinary
// kernels convert A on the fly from its fp32 form, which the caller must then have supplied (g.A)
template <bool A_KM, bool B_KM, int MODE, int EPI, bool S16, bool BPRE>
int launch_ap(SArgs& g, int splits, hipStream_t st) {
    const bool a_ok = g.Ap != nullptr && aligned16(g.Ap) && g.lda % 8 == 0 && (g.ap_stride % 8) == 0 &&
                      g.k_per_split % BK == 0 && g.K % BK == 0 && (!A_KM || g.M % 8 == 0);
    const bool b_ok = !BPRE || (g.Bp != nullptr && aligned16(g.Bp) && g.ldb % 8 == 0 && (g.bp_stride % 8) == 0 &&
                                (!B_KM || g.N % 8 == 0));
    if (a_ok && b_ok && !sparch_operand_bf16() && fast_ok<A_KM, B_KM, MODE, EPI, S16>(g))
        return launch_variant<A_KM, B_KM, MODE, EPI, true, S16, BPRE, 3, true>(g, splits, st);
    if (g.A == nullptr) return SPARCH_EINVAL;  // planes alone cannot run the general kernels
    if constexpr (BPRE) return launch_wp<A_KM, B_KM, MODE, EPI, S16>(g, splits, st);
    else return launch<A_KM, B_KM, MODE, EPI, S16>(g, splits, st);
}

// x -> three exact bf16 planes (truncation split, as store_piece<..., TRUNC> does on the fly)
__global__ void split3_kernel(size_t n4, const u32x4* __restrict__ x, u32x2* __restrict__ p1, u32x2* __restrict__ p2,
                              u32x2* __restrict__ p3) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const u32x4 v = x[i];
    u32x2 w1, w2, w3;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        const unsigned x0 = v[2 * pr], x1 = v[2 * pr + 1];
        const float r0 = __uint_as_float(x0) - __uint_as_float(x0 & 0xFFFF0000u);
        const float r1 = __uint_as_float(x1) - __uint_as_float(x1 & 0xFFFF0000u);
        const unsigned y0 = __float_as_uint(r0), y1 = __float_as_uint(r1);
        const float q0 = r0 - __uint_as_float(y0 & 0xFFFF0000u);
        const float q1 = r1 - __uint_as_float(y1 & 0xFFFF0000u);
        w1[pr] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
        w2[pr] = __builtin_amdgcn_perm(y1, y0, 0x07060302u);
        w3[pr] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
    }
    p1[i] = w1; p2[i] = w2; p3[i] = w3;
}

}  // namespace

extern "C" int sparch_split3(size_t n, const float* x, uint16_t* planes, void* stream) {
    SPARCH_ENTER();
    if (n == 0 || n % 8 != 0 || !x || !planes) return SPARCH_EINVAL;
    if (!aligned16(x) || !aligned16(planes)) return SPARCH_EALIGN;
    const size_t n4 = n / 4;
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n4,
                       reinterpret_cast<const u32x4*>(x), reinterpret_cast<u32x2*>(planes),
                       reinterpret_cast<u32x2*>(planes + n), reinterpret_cast<u32x2*>(planes + 2 * n));
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

extern "C" int sparch_gemm_spike_nt(int M, int N, int K, const float* A_spk, int lda, float scale, const float* B,
                                    int ldb, float* C, int ldc, const float* bias, float* colstat_ws,
                                    void* stream, int precision) {
    SPARCH_ENTER();
    PrecisionScope prec_scope_(precision);
    if (!prec_scope_.ok) return SPARCH_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || !A_spk || !B || !C || lda < K || ldb < K || ldc < N) return SPARCH_EINVAL;
    SArgs g{};
    g.A = A_spk; g.B = B; g.C = C; g.bias = bias; g.colstat = colstat_ws;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.k_per_split = cdiv(K, BK) * BK; g.c_split_stride = 0; g.scale = scale;
    g.a_vec = aligned16(A_spk) && (lda % 4 == 0);
    g.b_vec = aligned16(B) && (ldb % 4 == 0);
    hipStream_t st = (hipStream_t)stream;
    if (colstat_ws) return launch<false, false, 0, EPI_BIAS | EPI_STATS>(g, 1, st);
    if (bias) return launch<false, false, 0, EPI_BIAS>(g, 1, st);
    return launch<false, false, 0, EPI_NONE>(g, 1, st);
}

extern "C" int sparch_gemm_spike_tn(int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                                    int spike_side, float scale, float* C, int ldc, int zero_diag,
                                    int accumulate, void* ws, size_t ws_bytes, void* stream, int precision) {
    SPARCH_ENTER();
    PrecisionScope prec_scope_(precision);
    if (!prec_scope_.ok) return SPARCH_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C || lda < M || ldb < N || ldc < N) return SPARCH_EINVAL;
    if (spike_side != 0 && spike_side != 1) return SPARCH_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int splits = spike_side == 0 ? choose_splits<0>(M, N, K) : choose_splits<1>(M, N, K);
    const size_t need = (size_t)splits * M * N * sizeof(float);
    if (!ws || ws_bytes < need) return SPARCH_EWORKSPACE;
    SArgs g{};
    g.A = A; g.B = B; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.scale = scale;
    g.a_vec = aligned16(A) && (lda % 4 == 0);
    g.b_vec = aligned16(B) && (ldb % 4 == 0);
    g.C = (float*)ws; g.ldc = N; g.c_split_stride = (size_t)M * N;
    g.k_per_split = cdiv(cdiv(K, splits), BK) * BK;
    int rc = spike_side == 0 ? launch<true, true, 0, EPI_NONE>(g, splits, st)
                             : launch<true, true, 1, EPI_NONE>(g, splits, st);
    if (rc != SPARCH_OK) return rc;
    const size_t total = (size_t)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel2, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const float*)ws, C, M, N, ldc, splits, zero_diag, accumulate);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

// ---- the spike operand as a bf16 plane (entries 0 / 1.0, written by the cell kernels next to their fp32
// output): same products, half the operand bytes
extern "C" int sparch_gemm_spike16_nt(int M, int N, int K, const uint16_t* A_spk16, int lda, float scale,
                                      const float* B, int ldb, float* C, int ldc, const float* bias,
                                      float* colstat_ws, void* stream, int precision) {
    SPARCH_ENTER();
    PrecisionScope prec_scope_(precision);
    if (!prec_scope_.ok) return SPARCH_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || !A_spk16 || !B || !C || lda < K || ldb < K || ldc < N) return SPARCH_EINVAL;
    SArgs g{};
    g.A = reinterpret_cast<const float*>(A_spk16); g.B = B; g.C = C; g.bias = bias; g.colstat = colstat_ws;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.k_per_split = cdiv(K, BK) * BK; g.c_split_stride = 0; g.scale = scale;
    g.a_vec = aligned16(A_spk16) && (lda % 8 == 0);
    g.b_vec = aligned16(B) && (ldb % 4 == 0);
    hipStream_t st = (hipStream_t)stream;
    if (colstat_ws) return launch<false, false, 0, EPI_BIAS | EPI_STATS, true>(g, 1, st);
    if (bias) return launch<false, false, 0, EPI_BIAS, true>(g, 1, st);
    return launch<false, false, 0, EPI_NONE, true>(g, 1, st);
}

extern "C" int sparch_gemm_spike16_nt_wp(int M, int N, int K, const uint16_t* A_spk16, int lda, float scale,
                                         const float* B, const uint16_t* B_planes, int ldb, float* C, int ldc,
                                         const float* bias, float* colstat_ws, void* stream, int precision) {
    SPARCH_ENTER();
    PrecisionScope prec_scope_(precision);
    if (!prec_scope_.ok) return SPARCH_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || !A_spk16 || !B || !C || lda < K || ldb < K || ldc < N) return SPARCH_EINVAL;
    SArgs g{};
    g.A = reinterpret_cast<const float*>(A_spk16); g.B = B; g.C = C; g.bias = bias; g.colstat = colstat_ws;
    g.Bp = B_planes; g.bp_stride = (size_t)N * ldb;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.k_per_split = cdiv(K, BK) * BK; g.c_split_stride = 0; g.scale = scale;
    g.a_vec = aligned16(A_spk16) && (lda % 8 == 0);
    g.b_vec = aligned16(B) && (ldb % 4 == 0);
    hipStream_t st = (hipStream_t)stream;
    if (colstat_ws) return launch_wp<false, false, 0, EPI_BIAS | EPI_STATS, true>(g, 1, st);
    if (bias) return launch_wp<false, false, 0, EPI_BIAS, true>(g, 1, st);
    return launch_wp<false, false, 0, EPI_NONE, true>(g, 1, st);
}

extern "C" int sparch_gemm6_nn_wp(int M, int N, int K, const float* A, int lda, const float* B,
                                  const uint16_t* B_planes, int ldb, float* C, int ldc, void* stream, int precision) {
    SPARCH_ENTER();
    PrecisionScope prec_scope_(precision);
    if (!prec_scope_.ok) return SPARCH_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C || lda < K || ldb < N || ldc < N) return SPARCH_EINVAL;
    SArgs g{};
    g.A = A; g.B = B; g.C = C; g.Bp = B_planes; g.bp_stride = (size_t)K * ldb;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.k_per_split = cdiv(K, BK) * BK; g.c_split_stride = 0; g.scale = 1.0f;
    g.a_vec = aligned16(A) && (lda % 4 == 0);
    g.b_vec = aligned16(B) && (ldb % 4 == 0);
    return launch_wp<false, true, 2, EPI_NONE>(g, 1, (hipStream_t)stream);
}

extern "C" int sparch_gemm_spike16_tn(int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                                      int spike_side, float scale, float* C, int ldc, int zero_diag,
                                      int accumulate, void* ws, size_t ws_bytes, void* stream, int precision) {
    SPARCH_ENTER();
    PrecisionScope prec_scope_(precision);
    if (!prec_scope_.ok) return SPARCH_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C || lda < M || ldb < N || ldc < N) return SPARCH_EINVAL;
    if (spike_side != 0 && spike_side != 1) return SPARCH_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int splits = spike_side == 0 ? choose_splits<0>(M, N, K) : choose_splits<1>(M, N, K);
    const size_t need = (size_t)splits * M * N * sizeof(float);
    if (!ws || ws_bytes < need) return SPARCH_EWORKSPACE;
    SArgs g{};
    g.A = static_cast<const float*>(A); g.B = static_cast<const float*>(B);
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.scale = scale;
    g.a_vec = aligned16(A) && (lda % (spike_side == 0 ? 8 : 4) == 0);
    g.b_vec = aligned16(B) && (ldb % (spike_side == 1 ? 8 : 4) == 0);
    g.C = (float*)ws; g.ldc = N; g.c_split_stride = (size_t)M * N;
    g.k_per_split = cdiv(cdiv(K, splits), BK) * BK;
    int rc = spike_side == 0 ? launch<true, true, 0, EPI_NONE, true>(g, splits, st)
                             : launch<true, true, 1, EPI_NONE, true>(g, splits, st);
    if (rc != SPARCH_OK) return rc;
    const size_t total = (size_t)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel2, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const float*)ws, C, M, N, ldc, splits, zero_diag, accumulate);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

// C[M,N] = A[M,K] * B[K,N], BOTH operands given as their three exact bf16 planes (A_planes: 3 x M x lda,
// B_planes: 3 x K x ldb); A / B themselves (fp32, same layouts) are read instead where the pipelined plane kernel
// does not apply — A may be NULL when the caller knows it does (sparch_gemm6_nn_pp_applies).
extern "C" int sparch_gemm6_nn_pp(int M, int N, int K, const float* A, const uint16_t* A_planes, int lda, const float* B,
                                  const uint16_t* B_planes, int ldb, float* C, int ldc, void* stream, int precision) {
    SPARCH_ENTER();
    PrecisionScope prec_scope_(precision);
    if (!prec_scope_.ok) return SPARCH_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || (!A && !A_planes) || !B || !C || lda < K || ldb < N || ldc < N) return SPARCH_EINVAL;
    SArgs g{};
    g.A = A; g.B = B; g.C = C; g.Bp = B_planes; g.bp_stride = (size_t)K * ldb;
    g.Ap = A_planes; g.ap_stride = (size_t)M * lda;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.k_per_split = cdiv(K, BK) * BK; g.c_split_stride = 0; g.scale = 1.0f;
    g.a_vec = (A ? aligned16(A) : true) && (lda % 4 == 0);
    g.b_vec = aligned16(B) && (ldb % 4 == 0);
    return launch_ap<false, true, 2, EPI_NONE, false, true>(g, 1, (hipStream_t)stream);
}

// C[M,N] (+)= A[K,M]^T * B[K,N] with B a bf16 spike plane (as sparch_gemm_spike16_tn, spike_side = 1) and the dense A
// given as its three exact bf16 planes (3 x K x lda); A itself (fp32) as for sparch_gemm6_nn_pp.
extern "C" int sparch_gemm_spike16_tn_ap(int M, int N, int K, const float* A, const uint16_t* A_planes, int lda,
                                         const uint16_t* B16, int ldb, float scale, float* C, int ldc, int zero_diag,
                                         int accumulate, void* ws, size_t ws_bytes, void* stream, int precision) {
    SPARCH_ENTER();
    PrecisionScope prec_scope_(precision);
    if (!prec_scope_.ok) return SPARCH_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || (!A && !A_planes) || !B16 || !C || lda < M || ldb < N || ldc < N) return SPARCH_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int splits = choose_splits<1>(M, N, K);
    const size_t need = (size_t)splits * M * N * sizeof(float);
    if (!ws || ws_bytes < need) return SPARCH_EWORKSPACE;
    SArgs g{};
    g.A = A; g.B = reinterpret_cast<const float*>(B16);
    g.Ap = A_planes; g.ap_stride = (size_t)K * lda;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.scale = scale;
    g.a_vec = (A ? aligned16(A) : true) && (lda % 4 == 0);
    g.b_vec = aligned16(B16) && (ldb % 8 == 0);
    g.C = (float*)ws; g.ldc = N; g.c_split_stride = (size_t)M * N;
    g.k_per_split = cdiv(cdiv(K, splits), BK) * BK;
    int rc = launch_ap<true, true, 1, EPI_NONE, true, false>(g, splits, st);
    if (rc != SPARCH_OK) return rc;
    const size_t total = (size_t)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel2, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const float*)ws, C, M, N, ldc, splits, zero_diag, accumulate);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

// ---- dense x dense on the exact 6-term split (same signatures as the fp32-MFMA entry points in gemm.hip)
extern "C" int sparch_gemm6_nt(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                               int ldc, const float* bias, float* colstat_ws, void* stream, int precision) {
    SPARCH_ENTER();
    PrecisionScope prec_scope_(precision);
    if (!prec_scope_.ok) return SPARCH_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C || lda < K || ldb < K || ldc < N) return SPARCH_EINVAL;
    SArgs g{};
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.colstat = colstat_ws;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.k_per_split = cdiv(K, BK) * BK; g.c_split_stride = 0; g.scale = 1.0f;
    g.a_vec = aligned16(A) && (lda % 4 == 0);
    g.b_vec = aligned16(B) && (ldb % 4 == 0);
    hipStream_t st = (hipStream_t)stream;
    if (colstat_ws) return launch<false, false, 2, EPI_BIAS | EPI_STATS>(g, 1, st);
    if (bias) return launch<false, false, 2, EPI_BIAS>(g, 1, st);
    return launch<false, false, 2, EPI_NONE>(g, 1, st);
}

extern "C" int sparch_gemm6_nn(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                               int ldc, void* stream, int precision) {
    SPARCH_ENTER();
    PrecisionScope prec_scope_(precision);
    if (!prec_scope_.ok) return SPARCH_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C || lda < K || ldb < N || ldc < N) return SPARCH_EINVAL;
    SArgs g{};
    g.A = A; g.B = B; g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.k_per_split = cdiv(K, BK) * BK; g.c_split_stride = 0; g.scale = 1.0f;
    g.a_vec = aligned16(A) && (lda % 4 == 0);
    g.b_vec = aligned16(B) && (ldb % 4 == 0);
    return launch<false, true, 2, EPI_NONE>(g, 1, (hipStream_t)stream);
}

// Split-K forms of the dense NT / NN products for SMALL M*N with a long K (the per-step recurrent products of
// the gated baselines: 256 x 2048 x 1024 is 16 tiles on 256 CUs).  splits = sparch_gemm6_splitk_count(M,N,K)
// slabs of M*N floats in `ws`, reduced in fixed order; no bias / statistics epilogue.
namespace {
int small_splits(int M, int N, int K) {
    using S = Shape<2, true>;
    const int tiles = cdiv(M, S::BM) * cdiv(N, S::BN);
    const int kt = cdiv(K, BK);
    int s = target_wgs(1) / tiles;
    if (s > kt / 8) s = kt / 8;      // at least 8 K tiles per workgroup: the pipelined kernel's minimum
    return s < 1 ? 1 : s;
}
template <bool B_KM>
int gemm6_splitk(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc, void* ws,
                 size_t ws_bytes, hipStream_t st) {
    const int splits = small_splits(M, N, K);
    SArgs g{};
    g.A = A; g.B = B; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.scale = 1.0f;
    g.a_vec = aligned16(A) && (lda % 4 == 0);
    g.b_vec = aligned16(B) && (ldb % 4 == 0);
    if (splits == 1) {
        g.C = C; g.ldc = ldc; g.k_per_split = cdiv(K, BK) * BK; g.c_split_stride = 0;
        return launch<false, B_KM, 2, EPI_NONE>(g, 1, st);
    }
    if (!ws || ws_bytes < (size_t)splits * M * N * sizeof(float)) return SPARCH_EWORKSPACE;
    g.C = (float*)ws; g.ldc = N; g.c_split_stride = (size_t)M * N;
    g.k_per_split = cdiv(cdiv(K, splits), BK) * BK;
    int rc = launch<false, B_KM, 2, EPI_NONE>(g, splits, st);
    if (rc != SPARCH_OK) return rc;
    const size_t total = (size_t)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel2, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const float*)ws, C, M, N, ldc, splits, 0, 0);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}
}  // namespace

extern "C" size_t sparch_gemm6_splitk_workspace_bytes(int M, int N, int K, int precision) {
    PrecisionScope prec_scope_(precision);
    if (!prec_scope_.ok) return 0;
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int s = small_splits(M, N, K);
    return s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
}
extern "C" int sparch_gemm6_nt_splitk(int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                                      float* C, int ldc, void* ws, size_t ws_bytes, void* stream, int precision) {
    SPARCH_ENTER();
    PrecisionScope prec_scope_(precision);
    if (!prec_scope_.ok) return SPARCH_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C || lda < K || ldb < K || ldc < N) return SPARCH_EINVAL;
    return gemm6_splitk<false>(M, N, K, A, lda, B, ldb, C, ldc, ws, ws_bytes, (hipStream_t)stream);
}
extern "C" int sparch_gemm6_nn_splitk(int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                                      float* C, int ldc, void* ws, size_t ws_bytes, void* stream, int precision) {
    SPARCH_ENTER();
    PrecisionScope prec_scope_(precision);
    if (!prec_scope_.ok) return SPARCH_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C || lda < K || ldb < N || ldc < N) return SPARCH_EINVAL;
    return gemm6_splitk<true>(M, N, K, A, lda, B, ldb, C, ldc, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int sparch_gemm6_tn(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                               int ldc, int zero_diag, int accumulate, void* ws, size_t ws_bytes, void* stream, int precision) {
    SPARCH_ENTER();
    PrecisionScope prec_scope_(precision);
    if (!prec_scope_.ok) return SPARCH_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C || lda < M || ldb < N || ldc < N) return SPARCH_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int splits = choose_splits<2>(M, N, K);
    const size_t need = (size_t)splits * M * N * sizeof(float);
    if (!ws || ws_bytes < need) return SPARCH_EWORKSPACE;
    SArgs g{};
    g.A = A; g.B = B; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.scale = 1.0f;
    g.a_vec = aligned16(A) && (lda % 4 == 0);
    g.b_vec = aligned16(B) && (ldb % 4 == 0);
    g.C = (float*)ws; g.ldc = N; g.c_split_stride = (size_t)M * N;
    g.k_per_split = cdiv(cdiv(K, splits), BK) * BK;
    int rc = launch<true, true, 2, EPI_NONE>(g, splits, st);
    if (rc != SPARCH_OK) return rc;
    const size_t total = (size_t)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel2, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const float*)ws, C, M, N, ldc, splits, zero_diag, accumulate);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

// ---- operand whose exactness in bf16 is known only on the device
__global__ void flag_bf16_exact_kernel(size_t n4, const u32x4* __restrict__ x, unsigned* __restrict__ flag) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned bad = 0;
    for (size_t j = i; j < n4; j += (size_t)gridDim.x * blockDim.x) {
        const u32x4 v = x[j];
        bad |= (v.x | v.y | v.z | v.w) & 0xFFFFu;
    }
    if (__any(bad != 0) && (threadIdx.x & 63) == 0) atomicAnd(flag, 0u);
}
__global__ void flag_set_kernel(unsigned* flag, unsigned v) { *flag = v; }

extern "C" int sparch_flag_bf16_exact(size_t n, const float* x, uint32_t* flag, void* stream) {
    SPARCH_ENTER();
    if (n == 0 || !x || !flag) return SPARCH_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = aligned16(x) && (n % 4 == 0);
    hipLaunchKernelGGL(flag_set_kernel, dim3(1), dim3(1), 0, st, flag, vec ? 1u : 0u);
    SPARCH_CHECK_LAUNCH();
    if (vec) {
        hipLaunchKernelGGL(flag_bf16_exact_kernel, dim3(2048), dim3(256), 0, st, n / 4,
                           reinterpret_cast<const u32x4*>(x), flag);
        SPARCH_CHECK_LAUNCH();
    }
    return SPARCH_OK;
}

extern "C" int sparch_gemm_auto_nt(int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                                   float* C, int ldc, const float* bias, float* colstat_ws,
                                   const uint32_t* a_exact_flag, void* stream, int precision) {
    SPARCH_ENTER();
    PrecisionScope prec_scope_(precision);
    if (!prec_scope_.ok) return SPARCH_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C || lda < K || ldb < K || ldc < N || !a_exact_flag)
        return SPARCH_EINVAL;
    SArgs g{};
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.colstat = colstat_ws;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.k_per_split = cdiv(K, BK) * BK; g.c_split_stride = 0; g.scale = 1.0f;
    g.a_vec = aligned16(A) && (lda % 4 == 0);
    g.b_vec = aligned16(B) && (ldb % 4 == 0);
    g.gate = a_exact_flag; g.e_exact = 1;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    g.gate_want = 1;  // A is bf16-exact: single plane for A, three for B
    if (colstat_ws) rc = launch<false, false, 0, EPI_BIAS | EPI_STATS>(g, 1, st);
    else rc = launch<false, false, 0, EPI_BIAS>(g, 1, st);
    if (rc != SPARCH_OK) return rc;
    g.gate_want = 0;  // otherwise: both operands split, six cross terms
    if (colstat_ws) return launch<false, false, 2, EPI_BIAS | EPI_STATS>(g, 1, st);
    return launch<false, false, 2, EPI_BIAS>(g, 1, st);
}

extern "C" int sparch_gemm_auto_tn(int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                                   float* C, int ldc, int zero_diag, int accumulate,
                                   const uint32_t* b_exact_flag, void* ws, size_t ws_bytes, void* stream, int precision) {
    SPARCH_ENTER();
    PrecisionScope prec_scope_(precision);
    if (!prec_scope_.ok) return SPARCH_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C || lda < M || ldb < N || ldc < N || !b_exact_flag)
        return SPARCH_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // one split count for both gated kernels: only one of them runs, and the reduction needs one number
    const int splits = choose_splits<1>(M, N, K);
    const size_t need = (size_t)splits * M * N * sizeof(float);
    if (!ws || ws_bytes < need) return SPARCH_EWORKSPACE;
    SArgs g{};
    g.A = A; g.B = B; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.scale = 1.0f;
    g.a_vec = aligned16(A) && (lda % 4 == 0);
    g.b_vec = aligned16(B) && (ldb % 4 == 0);
    g.C = (float*)ws; g.ldc = N; g.c_split_stride = (size_t)M * N;
    g.k_per_split = cdiv(cdiv(K, splits), BK) * BK;
    g.gate = b_exact_flag; g.e_exact = 1;
    g.gate_want = 1;
    int rc = launch<true, true, 1, EPI_NONE>(g, splits, st);
    if (rc != SPARCH_OK) return rc;
    g.gate_want = 0;
    rc = launch<true, true, 2, EPI_NONE>(g, splits, st);
    if (rc != SPARCH_OK) return rc;
    const size_t total = (size_t)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel2, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const float*)ws, C, M, N, ldc, splits, zero_diag, accumulate);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

// ---- the same check that ALSO writes the bf16 plane of x (upper halves; the exact values when the flag stays 1):
// one pass over the network input per step instead of the flag pass plus two fp32 reads by the first layer's
// GEMMs, which then read 2 bytes per element through the spike-plane kernels.  Plane rows are ldp >= K elements
// (a multiple of 8: 16-byte loadable), columns K .. ldp-1 zero.
__global__ void plane_flag_kernel(int M, int K, int ldx, int ldp, const float* __restrict__ x,
                                  unsigned short* __restrict__ plane, unsigned* __restrict__ flag) {
    const int q = ldp / 4;  // 4-element pieces per plane row
    const size_t total = (size_t)M * q;
    unsigned bad = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / q;
        const int k = (int)(i - m * q) * 4;
        unsigned w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (k + e < K) ? __float_as_uint(x[m * ldx + k + e]) : 0u;
        bad |= (w[0] | w[1] | w[2] | w[3]) & 0xFFFFu;
        u32x2 o;
        o.x = (w[0] >> 16) | (w[1] & 0xFFFF0000u);
        o.y = (w[2] >> 16) | (w[3] & 0xFFFF0000u);
        *reinterpret_cast<u32x2*>(plane + m * ldp + k) = o;
    }
    if (__any(bad != 0) && (threadIdx.x & 63) == 0) atomicAnd(flag, 0u);
}

extern "C" int sparch_plane_bf16_exact(int M, int K, const float* x, int ldx, uint16_t* plane, int ldp,
                                       uint32_t* flag, void* stream) {
    SPARCH_ENTER();
    if (M <= 0 || K <= 0 || !x || !plane || !flag || ldx < K || ldp < K || ldp % 8 != 0) return SPARCH_EINVAL;
    if (!aligned16(plane)) return SPARCH_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(flag_set_kernel, dim3(1), dim3(1), 0, st, flag, 1u);
    SPARCH_CHECK_LAUNCH();
    hipLaunchKernelGGL(plane_flag_kernel, dim3(2048), dim3(256), 0, st, M, K, ldx, ldp, x, plane, flag);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

// The device-gated products with the flagged operand's plane at hand: flag == 1 -> the spike-plane kernel reads the
// plane (2 bytes per element, no conversion); flag == 0 -> the six-term kernel on the fp32 operand, as before.
extern "C" int sparch_gemm_auto16_nt(int M, int N, int K, const float* A, int lda, const uint16_t* A16, int lda16,
                                     const float* B, int ldb, float* C, int ldc, const float* bias,
                                     float* colstat_ws, const uint32_t* a_exact_flag, void* stream, int precision) {
    SPARCH_ENTER();
    PrecisionScope prec_scope_(precision);
    if (!prec_scope_.ok) return SPARCH_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || !A || !A16 || !B || !C || lda < K || lda16 < K || ldb < K || ldc < N ||
        !a_exact_flag)
        return SPARCH_EINVAL;
    SArgs g{};
    g.B = B; g.C = C; g.bias = bias; g.colstat = colstat_ws;
    g.M = M; g.N = N; g.K = K; g.ldb = ldb; g.ldc = ldc;
    g.k_per_split = cdiv(K, BK) * BK; g.c_split_stride = 0; g.scale = 1.0f;
    g.b_vec = aligned16(B) && (ldb % 4 == 0);
    g.gate = a_exact_flag; g.e_exact = 1;
    hipStream_t st = (hipStream_t)stream;
    g.A = reinterpret_cast<const float*>(A16); g.lda = lda16;
    g.a_vec = aligned16(A16) && (lda16 % 8 == 0);
    g.gate_want = 1;
    int rc = colstat_ws ? launch<false, false, 0, EPI_BIAS | EPI_STATS, true>(g, 1, st)
                        : launch<false, false, 0, EPI_BIAS, true>(g, 1, st);
    if (rc != SPARCH_OK) return rc;
    g.A = A; g.lda = lda; g.a_vec = aligned16(A) && (lda % 4 == 0);
    g.gate_want = 0;
    if (colstat_ws) return launch<false, false, 2, EPI_BIAS | EPI_STATS>(g, 1, st);
    return launch<false, false, 2, EPI_BIAS>(g, 1, st);
}

extern "C" int sparch_gemm_auto16_tn(int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                                     const uint16_t* B16, int ldb16, float* C, int ldc, int zero_diag,
                                     int accumulate, const uint32_t* b_exact_flag, void* ws, size_t ws_bytes,
                                     void* stream, int precision) {
    SPARCH_ENTER();
    PrecisionScope prec_scope_(precision);
    if (!prec_scope_.ok) return SPARCH_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !B16 || !C || lda < M || ldb < N || ldb16 < N || ldc < N ||
        !b_exact_flag)
        return SPARCH_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // A ragged width (N % 8 != 0, e.g. 700 input channels): the plane kernel's transposed 16-byte loads want whole
    // groups of 8 columns, and the plane has them — its rows are padded with zeros to ldb16.  The product is then
    // taken at the padded width N8 (the extra columns are zeros and are never reduced), with the slabs N8 wide.
    const int N8 = (N + 7) & ~7;
    if (N8 > ldb16) return SPARCH_EINVAL;
    const int splits = choose_splits<1>(M, N8, K);
    const size_t need = (size_t)splits * M * N8 * sizeof(float);
    if (!ws || ws_bytes < need) return SPARCH_EWORKSPACE;
    SArgs g{};
    g.A = A; g.M = M; g.N = N8; g.K = K; g.lda = lda; g.scale = 1.0f;
    g.a_vec = aligned16(A) && (lda % 4 == 0);
    g.C = (float*)ws; g.ldc = N8; g.c_split_stride = (size_t)M * N8;
    g.k_per_split = cdiv(cdiv(K, splits), BK) * BK;
    g.gate = b_exact_flag; g.e_exact = 1;
    g.B = reinterpret_cast<const float*>(B16); g.ldb = ldb16;
    g.b_vec = aligned16(B16) && (ldb16 % 8 == 0);
    g.gate_want = 1;
    int rc = launch<true, true, 1, EPI_NONE, true>(g, splits, st);
    if (rc != SPARCH_OK) return rc;
    g.N = N;  // the fp32 operand has exactly N columns
    g.B = B; g.ldb = ldb; g.b_vec = aligned16(B) && (ldb % 4 == 0);
    g.gate_want = 0;
    rc = launch<true, true, 2, EPI_NONE>(g, splits, st);
    if (rc != SPARCH_OK) return rc;
    const size_t total = (size_t)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel2, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const float*)ws, C, M, N, ldc, splits, zero_diag, accumulate, N8);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

#if defined(SPARCH_REC_PROF) && !defined(GA_NO_STAMPS)
extern "C" int sparch_gemm_prof_read(unsigned long long* host_out, int reset) {
    static unsigned long long zero[8 + 8 * 2];
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_gemm_prof), sizeof(zero)) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_prof), zero, sizeof(zero)) != hipSuccess) return -1;
    return 0;
}
#endif

extern "C" size_t sparch_gemm_spike_tn_workspace_bytes(int M, int N, int K, int precision) {
    PrecisionScope prec_scope_(precision);
    if (!prec_scope_.ok) return 0;
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int s0 = choose_splits<0>(M, N, K), s1 = choose_splits<1>(M, N, K), s2 = choose_splits<2>(M, N, K);
    const int smax = s0 > s1 ? (s0 > s2 ? s0 : s2) : (s1 > s2 ? s1 : s2);
    return (size_t)smax * M * N * sizeof(float);
}

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
#include "common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int BM = 128, BN = 128, BK = 32, NT = 256;
constexpr int KC_ROW = 40;    // bf16 per LDS row of a KC image (32 + 8 pad = 80 B)
constexpr int KM_ROW = 160;   // bf16 per LDS row of a KM image (128 + 32 pad = 320 B)
constexpr int PLANE = (BM * KC_ROW > BK * KM_ROW) ? BM * KC_ROW : BK * KM_ROW;  // 5120 bf16 = 10 KiB

enum Epi { EPI_NONE = 0, EPI_BIAS = 1, EPI_STATS = 2 };

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

// ---- global -> registers (same thread/element map as gemm.hip)
template <bool KM>
__device__ __forceinline__ void stage_load(f32x4 (&r)[4], const float* __restrict__ P, int ld, int row0, int rows,
                                           int k0, int kend, int vec, int tid) {
    // fast path (uniform): the whole 128 x 32 tile is in range and 16-byte loads are legal -> straight-line
    // loads with no per-element exec-mask juggling
    if (vec && row0 + 128 <= rows && k0 + BK <= kend) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int f = tid + NT * p;
            if constexpr (!KM) r[p] = *reinterpret_cast<const f32x4*>(P + (size_t)(row0 + (f >> 3)) * ld + k0 + ((f & 7) << 2));
            else               r[p] = *reinterpret_cast<const f32x4*>(P + (size_t)(k0 + (f >> 5)) * ld + row0 + ((f & 31) << 2));
        }
        return;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int f = tid + NT * p;
        int row, k;
        if constexpr (!KM) { row = row0 + (f >> 3); k = k0 + ((f & 7) << 2); }
        else               { k = k0 + (f >> 5); row = row0 + ((f & 31) << 2); }
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
        r[p] = v;
    }
}

// ---- registers -> LDS with conversion.  SPIKE: one plane of 0/1; else three planes hi/mid/lo.
// TRUNC: exact truncation split (x = t1 + t2 + t3; AND / SUB / v_perm, ~5 VALU per element) — used when the
// other operand is an exact spike plane, where any exact split gives the same result; the dense 6-term
// kernel keeps the round-to-nearest split, whose dropped cross terms are 8x smaller.
template <bool KM, bool SPIKE, bool TRUNC = false>
__device__ __forceinline__ void stage_store(const f32x4 (&r)[4], unsigned short* __restrict__ S, int tid,
                                            int e_exact = 0) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int f = tid + NT * p;
        int off;  // in bf16 elements, 8-byte aligned
        if constexpr (!KM) off = (f >> 3) * KC_ROW + ((f & 7) << 2);
        else               off = (f >> 5) * KM_ROW + ((f & 31) << 2);
        if constexpr (SPIKE) {
            u32x2 w;
            if (e_exact) {  // values are bf16-exact: their bf16 form is the upper half of the fp32 word
                w.x = __builtin_amdgcn_perm(__float_as_uint(r[p].y), __float_as_uint(r[p].x), 0x07060302u);
                w.y = __builtin_amdgcn_perm(__float_as_uint(r[p].w), __float_as_uint(r[p].z), 0x07060302u);
            } else {
                w.x = (r[p].x != 0.f ? 0x3F80u : 0u) | (r[p].y != 0.f ? 0x3F800000u : 0u);
                w.y = (r[p].z != 0.f ? 0x3F80u : 0u) | (r[p].w != 0.f ? 0x3F800000u : 0u);
            }
            *reinterpret_cast<u32x2*>(S + off) = w;
        } else if constexpr (TRUNC) {
            u32x2 w1, w2, w3;
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const unsigned x0 = __float_as_uint(r[p][2 * pr]), x1 = __float_as_uint(r[p][2 * pr + 1]);
                const float r0 = r[p][2 * pr] - __uint_as_float(x0 & 0xFFFF0000u);
                const float r1 = r[p][2 * pr + 1] - __uint_as_float(x1 & 0xFFFF0000u);
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
            for (int e = 0; e < 4; ++e) split3(r[p][e], h[e], m[e], l[e]);
            *reinterpret_cast<u32x2*>(S + off) = u32x2{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)};
            *reinterpret_cast<u32x2*>(S + PLANE + off) = u32x2{(unsigned)m[0] | ((unsigned)m[1] << 16), (unsigned)m[2] | ((unsigned)m[3] << 16)};
            *reinterpret_cast<u32x2*>(S + 2 * PLANE + off) = u32x2{(unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16)};
        }
    }
}

// ---- MFMA fragment (8 bf16 of row/col `idx`, k = 16*ks + 8*h + j) from an LDS plane
template <bool KM>
__device__ __forceinline__ u32x4 frag_read(const unsigned short* __restrict__ S, int idx_base, int lane, int ks) {
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
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// MODE 0: A is the spike operand; MODE 1: B is; MODE 2: both operands are dense fp32 and both are split
// (six cross terms hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid; the dropped ones are <= 2^-24 relative).
template <bool A_KM, bool B_KM, int MODE, int EPI>
__global__ __launch_bounds__(NT, 3) void gemm_spike_kernel(SArgs g) {
    constexpr bool SPIKE_A = MODE == 0;
    constexpr bool SPIKE_B = MODE == 1;
    constexpr int A_PLANES = SPIKE_A ? 1 : 3, B_PLANES = SPIKE_B ? 1 : 3;
    // one array (guide: keep all LDS in one object)
    __shared__ __attribute__((aligned(16))) unsigned short lds[(A_PLANES + B_PLANES) * PLANE];
    unsigned short* As = lds;
    unsigned short* Bs = lds + A_PLANES * PLANE;

    if (g.gate != nullptr && *g.gate != g.gate_want) return;  // uniform: every workgroup reads the same word
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (g.N + BN - 1) / BN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int k_begin = blockIdx.y * g.k_per_split;
    const int k_end = min(g.K, k_begin + g.k_per_split);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[4], rb[4];
    stage_load<A_KM>(ra, g.A, g.lda, m0, g.M, k_begin, k_end, g.a_vec, tid);
    stage_load<B_KM>(rb, g.B, g.ldb, n0, g.N, k_begin, k_end, g.b_vec, tid);

    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        stage_store<A_KM, SPIKE_A, MODE != 2>(ra, As, tid, g.e_exact);
        stage_store<B_KM, SPIKE_B, MODE != 2>(rb, Bs, tid, g.e_exact);
        __syncthreads();
        if (k0 + BK < k_end) {
            stage_load<A_KM>(ra, g.A, g.lda, m0, g.M, k0 + BK, k_end, g.a_vec, tid);
            stage_load<B_KM>(rb, g.B, g.ldb, n0, g.N, k0 + BK, k_end, g.b_vec, tid);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if constexpr (MODE == 2) {
                u32x4 fa[2][3], fb[2][3];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int p = 0; p < 3; ++p) fa[i][p] = frag_read<A_KM>(As + p * PLANE, wm * 64 + i * 32, lane, ks);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int p = 0; p < 3; ++p) fb[j][p] = frag_read<B_KM>(Bs + p * PLANE, wn * 64 + j * 32, lane, ks);
                // (pa, pb) pairs, small terms first: mid*mid, lo*hi, hi*lo, mid*hi, hi*mid, hi*hi
                constexpr int PA[6] = {1, 2, 0, 1, 0, 0};
                constexpr int PB[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
                for (int c = 0; c < 6; ++c)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = mfma_bf16(fa[i][PA[c]], fb[j][PB[c]], acc[i][j]);
            } else if constexpr (SPIKE_A) {
                u32x4 fa[2], fb[2][3];
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = frag_read<A_KM>(As, wm * 64 + i * 32, lane, ks);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int p = 0; p < 3; ++p) fb[j][p] = frag_read<B_KM>(Bs + p * PLANE, wn * 64 + j * 32, lane, ks);
#pragma unroll
                for (int p = 2; p >= 0; --p)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = mfma_bf16(fa[i], fb[j][p], acc[i][j]);
            } else {
                u32x4 fa[2][3], fb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int p = 0; p < 3; ++p) fa[i][p] = frag_read<A_KM>(As + p * PLANE, wm * 64 + i * 32, lane, ks);
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = frag_read<B_KM>(Bs, wn * 64 + j * 32, lane, ks);
#pragma unroll
                for (int p = 2; p >= 0; --p)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = mfma_bf16(fa[i][p], fb[j], acc[i][j]);
            }
        }
        __syncthreads();
    }

    // ---- epilogue (C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*h)
    float* Cz = g.C + (size_t)blockIdx.y * g.c_split_stride;
    float csum[2] = {0.f, 0.f}, csq[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + li;
        float bj = 0.f;
        if constexpr (EPI & EPI_BIAS) bj = (g.bias != nullptr && col < g.N) ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                float v = acc[i][j][r] * g.scale;
                if constexpr (EPI & EPI_BIAS) v = v + bj;
                if (row < g.M && col < g.N) {
                    Cz[(size_t)row * g.ldc + col] = v;
                    if constexpr (EPI & EPI_STATS) { csum[j] += v; csq[j] += v * v; }
                }
            }
        }
    }
    if constexpr (EPI & EPI_STATS) {
        float* red = reinterpret_cast<float*>(lds);  // [2 (sum|sq)][2 (wm)][128] (2 KiB <= any LDS size here)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            csum[j] += __shfl_xor(csum[j], 32);
            csq[j] += __shfl_xor(csq[j], 32);
        }
        if (h == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = wn * 64 + j * 32 + li;
                red[(0 * 2 + wm) * 128 + c] = csum[j];
                red[(1 * 2 + wm) * 128 + c] = csq[j];
            }
        }
        __syncthreads();
        if (tid < 128 && n0 + tid < g.N && g.colstat != nullptr) {
            const int tiles_m = (g.M + BM - 1) / BM;
            g.colstat[(size_t)tile_m * g.N + n0 + tid] = red[0 * 128 + tid] + red[1 * 128 + tid];
            g.colstat[(size_t)(tiles_m + tile_m) * g.N + n0 + tid] = red[2 * 128 + tid] + red[3 * 128 + tid];
        }
    }
}

__global__ void splitk_reduce_kernel2(const float* __restrict__ ws, float* __restrict__ C, int M, int N, int ldc,
                                      int splits, int zero_diag, int accumulate) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)M * N;
    if (i >= total) return;
    const int m = (int)(i / N), n = (int)(i % N);
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += ws[(size_t)z * total + i];
    if (zero_diag && m == n) s = 0.f;
    float* c = C + (size_t)m * ldc + n;
    *c = accumulate ? (*c + s) : s;
}

int choose_splits(int M, int N, int K) {
    const int tiles = cdiv(M, BM) * cdiv(N, BN);
    const int kt = cdiv(K, BK);
    int s = 1;
    while (tiles * s < 1024 && kt / (s * 2) >= 8) s *= 2;
    return s;
}

template <bool A_KM, bool B_KM, int MODE, int EPI>
int launch(SArgs& g, int splits, hipStream_t st) {
    const int tiles = cdiv(g.M, BM) * cdiv(g.N, BN);
    hipLaunchKernelGGL((gemm_spike_kernel<A_KM, B_KM, MODE, EPI>), dim3(tiles, splits, 1), dim3(NT), 0, st, g);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

}  // namespace

extern "C" int sparch_gemm_spike_nt(int M, int N, int K, const float* A_spk, int lda, float scale, const float* B,
                                    int ldb, float* C, int ldc, const float* bias, float* colstat_ws,
                                    void* stream) {
    SPARCH_ENTER();
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
                                    int accumulate, void* ws, size_t ws_bytes, void* stream) {
    SPARCH_ENTER();
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C || lda < M || ldb < N || ldc < N) return SPARCH_EINVAL;
    if (spike_side != 0 && spike_side != 1) return SPARCH_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int splits = choose_splits(M, N, K);
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

// ---- dense x dense on the exact 6-term split (same signatures as the fp32-MFMA entry points in gemm.hip)
extern "C" int sparch_gemm6_nt(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                               int ldc, const float* bias, float* colstat_ws, void* stream) {
    SPARCH_ENTER();
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
                               int ldc, void* stream) {
    SPARCH_ENTER();
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C || lda < K || ldb < N || ldc < N) return SPARCH_EINVAL;
    SArgs g{};
    g.A = A; g.B = B; g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.k_per_split = cdiv(K, BK) * BK; g.c_split_stride = 0; g.scale = 1.0f;
    g.a_vec = aligned16(A) && (lda % 4 == 0);
    g.b_vec = aligned16(B) && (ldb % 4 == 0);
    return launch<false, true, 2, EPI_NONE>(g, 1, (hipStream_t)stream);
}

extern "C" int sparch_gemm6_tn(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                               int ldc, int zero_diag, int accumulate, void* ws, size_t ws_bytes, void* stream) {
    SPARCH_ENTER();
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C || lda < M || ldb < N || ldc < N) return SPARCH_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int splits = choose_splits(M, N, K);
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
                                   const uint32_t* a_exact_flag, void* stream) {
    SPARCH_ENTER();
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
                                   const uint32_t* b_exact_flag, void* ws, size_t ws_bytes, void* stream) {
    SPARCH_ENTER();
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C || lda < M || ldb < N || ldc < N || !b_exact_flag)
        return SPARCH_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int splits = choose_splits(M, N, K);
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

extern "C" size_t sparch_gemm_spike_tn_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return (size_t)choose_splits(M, N, K) * M * N * sizeof(float);
}

// G1: feed-forward projection GEMMs on the fp32-input MFMA (v_mfma_f32_32x32x2_f32).
//
// Replaces `self.W(x)` (nn.Linear; snns.py:261/398/533/675/796) and its autograd
// backward (dX = dWx*W, dW = dWx^T*x) plus dV = s_prev^T*dWx of the recurrent cells.
//
// Design (gfx950): 128x128 output tile per 256-thread workgroup, BK = 32, four waves
// in a 2x2 arrangement, each wave 2x2 MFMA tiles of 32x32 (64 accumulator VGPRs).
// Operands are staged global -> registers -> LDS (software-prefetched one K-tile
// ahead); two to three workgroups per CU overlap one's staging with another's MFMAs.
// An operand is either
//   KC  "K-contiguous": element (row, k) at p[row*ld + k]; LDS image [row][36]
//       (row stride 36 floats => ds_read_b128 fragment reads are bank-conflict-free);
//   KM  "K-major":      element (k, col) at p[k*ld + col]; LDS image [k][132],
//       fragment reads are 32 consecutive dwords (ds_read_b32, conflict-free).
// The MFMA contracts k in pairs; both operands use the same pairing
// k = 8g + 4h + e (g sub-step, h = lane>>5, e = element of the lane's 4-vector), so
// KC fragments are single 16-byte LDS reads with no transpose anywhere.
// fp32 MFMA is an exact k-ordered fmaf chain (one rounding per product).
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LD_KC = BK + 4;   // 36
constexpr int LD_KM = BM + 4;   // 132
constexpr int NT = 256;
constexpr int LDS_OP = (BM * LD_KC > BK * LD_KM) ? BM * LD_KC : BK * LD_KM;  // floats per operand

enum Epi { EPI_NONE = 0, EPI_BIAS = 1, EPI_STATS = 2 };

struct GemmArgs {
    const float* A; const float* B; float* C;
    const float* bias; float* colstat;
    int M, N, K;          // output M x N, contraction K (this launch's K range is [k_begin,k_end))
    int lda, ldb, ldc;
    int k_per_split;      // multiple of BK
    size_t c_split_stride;  // elements between split slabs of C (0 when not split)
    int a_vec, b_vec;     // 16-byte vector loads legal for A / B
};

// ---- global -> register staging (4 float4 per thread per operand) ----
template <bool KM>
__device__ __forceinline__ void stage_load(f32x4 (&r)[4], const float* __restrict__ P, int ld,
                                           int row0, int rows, int k0, int kend, int vec, int tid) {
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

template <bool KM>
__device__ __forceinline__ void stage_store(const f32x4 (&r)[4], float* __restrict__ S, int tid) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int f = tid + NT * p;
        float* dst;
        if constexpr (!KM) dst = S + (f >> 3) * LD_KC + ((f & 7) << 2);
        else               dst = S + (f >> 5) * LD_KM + ((f & 31) << 2);
        *reinterpret_cast<f32x4*>(dst) = r[p];
    }
}

// fragment for sub-step g: 4 k-values (k = 8g + 4h + e) of output row/col `idx`
template <bool KM>
__device__ __forceinline__ f32x4 frag_read(const float* __restrict__ S, int idx, int g, int h) {
    if constexpr (!KM) {
        return *reinterpret_cast<const f32x4*>(S + idx * LD_KC + 8 * g + 4 * h);
    } else {
        const float* q = S + (8 * g + 4 * h) * LD_KM + idx;
        f32x4 v;
        v.x = q[0]; v.y = q[LD_KM]; v.z = q[2 * LD_KM]; v.w = q[3 * LD_KM];
        return v;
    }
}

template <bool A_KM, bool B_KM, int EPI>
__global__ __launch_bounds__(NT, 2) void gemm_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float lds[2 * LDS_OP];
    float* As = lds;
    float* Bs = lds + LDS_OP;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    // Tile mapping: blockIdx.x walks N tiles fastest so that consecutive workgroups
    // (dealt round-robin to XCDs) reuse the same A row-panel out of L2 / Infinity Cache.
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
        stage_store<A_KM>(ra, As, tid);
        stage_store<B_KM>(rb, Bs, tid);
        __syncthreads();
        if (k0 + BK < k_end) {
            stage_load<A_KM>(ra, g.A, g.lda, m0, g.M, k0 + BK, k_end, g.a_vec, tid);
            stage_load<B_KM>(rb, g.B, g.ldb, n0, g.N, k0 + BK, k_end, g.b_vec, tid);
        }
#pragma unroll
        for (int gs = 0; gs < 4; ++gs) {
            f32x4 fa[2], fb[2];
            fa[0] = frag_read<A_KM>(As, wm * 64 + li, gs, h);
            fa[1] = frag_read<A_KM>(As, wm * 64 + 32 + li, gs, h);
            fb[0] = frag_read<B_KM>(Bs, wn * 64 + li, gs, h);
            fb[1] = frag_read<B_KM>(Bs, wn * 64 + 32 + li, gs, h);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*h
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
                float v = acc[i][j][r];
                if constexpr (EPI & EPI_BIAS) v = v + bj;
                if (row < g.M && col < g.N) {
                    Cz[(size_t)row * g.ldc + col] = v;
                    if constexpr (EPI & EPI_STATS) { csum[j] += v; csq[j] += v * v; }
                }
            }
        }
    }
    if constexpr (EPI & EPI_STATS) {
        // per-column partial sums of this 128-row tile, combined in a fixed order:
        // lane halves (h) by shuffle, the two M-waves through LDS.
        float* red = lds;  // [2 (sum|sq)][2 (wm)][128]
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

// out[m][n] = sum_z slab[z][m][n] in fixed z order; optional zeroed diagonal
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, int M, int N,
                                     int ldc, int splits, int zero_diag, int accumulate) {
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

__global__ void zero_diag_kernel(float* __restrict__ C, int n, int ldc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) C[(size_t)i * ldc + i] = 0.f;
}

int choose_splits(int M, int N, int K) {
    const int tiles = cdiv(M, BM) * cdiv(N, BN);
    const int kt = cdiv(K, BK);
    int s = 1;
    // aim for ~1024 workgroups (4 per CU), at least 8 K-tiles per split
    while (tiles * s < 1024 && kt / (s * 2) >= 8) s *= 2;
    return s;
}

template <bool A_KM, bool B_KM, int EPI>
int launch(GemmArgs& g, int splits, hipStream_t st) {
    const int tiles = cdiv(g.M, BM) * cdiv(g.N, BN);
    dim3 grid(tiles, splits, 1);
    hipLaunchKernelGGL((gemm_kernel<A_KM, B_KM, EPI>), grid, dim3(NT), 0, st, g);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

}  // namespace

extern "C" int sparch_gemm_nt(int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                              float* C, int ldc, const float* bias, float* colstat_ws, void* stream) {
    SPARCH_ENTER();
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C || lda < K || ldb < K || ldc < N) return SPARCH_EINVAL;
    GemmArgs g{};
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.colstat = colstat_ws;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.k_per_split = cdiv(K, BK) * BK; g.c_split_stride = 0;
    g.a_vec = aligned16(A) && (lda % 4 == 0);
    g.b_vec = aligned16(B) && (ldb % 4 == 0);
    hipStream_t st = (hipStream_t)stream;
    if (colstat_ws) return launch<false, false, EPI_BIAS | EPI_STATS>(g, 1, st);
    if (bias) return launch<false, false, EPI_BIAS>(g, 1, st);
    return launch<false, false, EPI_NONE>(g, 1, st);
}

extern "C" int sparch_gemm_nn(int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                              float* C, int ldc, void* stream) {
    SPARCH_ENTER();
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C || lda < K || ldb < N || ldc < N) return SPARCH_EINVAL;
    GemmArgs g{};
    g.A = A; g.B = B; g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.k_per_split = cdiv(K, BK) * BK; g.c_split_stride = 0;
    g.a_vec = aligned16(A) && (lda % 4 == 0);
    g.b_vec = aligned16(B) && (ldb % 4 == 0);
    return launch<false, true, EPI_NONE>(g, 1, (hipStream_t)stream);
}

extern "C" size_t sparch_gemm_tn_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return (size_t)choose_splits(M, N, K) * M * N * sizeof(float);
}

extern "C" int sparch_gemm_tn(int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                              float* C, int ldc, int zero_diag, int accumulate, void* ws, size_t ws_bytes,
                              void* stream) {
    SPARCH_ENTER();
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C || lda < M || ldb < N || ldc < N) return SPARCH_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int splits = choose_splits(M, N, K);
    GemmArgs g{};
    g.A = A; g.B = B;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb;
    g.a_vec = aligned16(A) && (lda % 4 == 0);
    g.b_vec = aligned16(B) && (ldb % 4 == 0);
    if (splits == 1 && !accumulate) {
        g.C = C; g.ldc = ldc; g.k_per_split = cdiv(K, BK) * BK; g.c_split_stride = 0;
        int rc = launch<true, true, EPI_NONE>(g, 1, st);
        if (rc != SPARCH_OK || !zero_diag) return rc;
        const int n = M < N ? M : N;
        hipLaunchKernelGGL(zero_diag_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, C, n, ldc);
        SPARCH_CHECK_LAUNCH();
        return SPARCH_OK;
    }
    const size_t need = (size_t)splits * M * N * sizeof(float);
    if (!ws || ws_bytes < need) return SPARCH_EWORKSPACE;
    g.C = (float*)ws; g.ldc = N; g.c_split_stride = (size_t)M * N;
    g.k_per_split = cdiv(cdiv(K, splits), BK) * BK;
    int rc = launch<true, true, EPI_NONE>(g, splits, st);
    if (rc != SPARCH_OK) return rc;
    const size_t total = (size_t)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const float*)ws, C, M, N, ldc, splits, zero_diag, accumulate);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

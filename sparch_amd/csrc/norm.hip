// G2: normalisation on the (M = B*T, H) view of the projection, and the small
// fixed-order column reductions used by the layer backward.
//
// Replaces nn.BatchNorm1d(H, momentum=0.05) / nn.LayerNorm(H) applied at
// snns.py:264-266 (layers built at 240 / 243).  BatchNorm never materialises the
// normalised tensor: statistics come from the projection GEMM's epilogue partials
// and are folded into a per-column (scale, shift) that the cell kernels apply on
// load:  y = x*scale + shift,  scale = gamma*invstd,  shift = beta - mean*scale.
// All column reductions are two-stage with fixed summation order (no float
// atomics) so results are bitwise reproducible run to run.
#include "common.h"

namespace {

constexpr int RB = 256;  // rows per partial block in column reductions

// Sum rows [0,n) of a row-major [n][ld] fp32 array for 16 adjacent columns per 256-thread block:
// thread (c = tid&15, q = tid>>4) adds rows q, q+16, ... in fp64 (64-B row segments), the sixteen
// row-lanes meet in LDS in a fixed order.  Returns the total to the q == 0 threads (tid < 16).
constexpr int CS_COLS = 16, CS_LANES = 16;
__device__ __forceinline__ double col_sum64(const float* __restrict__ base, int n, size_t ld, int col, bool ok,
                                            double (*red)[CS_COLS]) {
    const int c = threadIdx.x & (CS_COLS - 1), q = threadIdx.x / CS_COLS;
    double s = 0.0;
    if (ok) {
        // eight independent loads in flight per thread (a one-load-per-iteration loop is a chain of ~1 us
        // memory round trips: 25 us for 500 partial rows), summed in a fixed order
        int r = q;
        for (; r + 7 * CS_LANES < n; r += 8 * CS_LANES) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = base[(size_t)(r + j * CS_LANES) * ld + col];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (double)v[j];
        }
        for (; r < n; r += CS_LANES) s += (double)base[(size_t)r * ld + col];
    }
    red[q][c] = s;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int i = 0; i < CS_LANES; ++i) tot += red[i][c];
    __syncthreads();
    return tot;
}

// ---------------------------------------------------------------- BatchNorm forward
__global__ __launch_bounds__(256) void bn_finalize_kernel(int H, int M, int n_tiles, int dup, const float* __restrict__ ws,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ rmean, float* __restrict__ rvar, float momentum,
                                   float eps, int training, float* __restrict__ scale,
                                   float* __restrict__ shift, float* __restrict__ save_mean,
                                   float* __restrict__ save_invstd, const uint32_t* __restrict__ skip_if_nonzero,
                                   long long* __restrict__ num_batches_tracked) {
    __shared__ double red[CS_LANES][CS_COLS];
    if (training && num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0 &&
        !(skip_if_nonzero && *skip_if_nonzero != 0u))
        *num_batches_tracked += 1;  // BatchNorm1d's counter (snns.py:264), under the same guard as the statistics
    const int h = blockIdx.x * CS_COLS + (threadIdx.x & (CS_COLS - 1));
    const bool ok = h < H;
    double s = 0.0, ss = 0.0;
    if (training) {
        s = col_sum64(ws, n_tiles, H, h, ok, red);
        ss = col_sum64(ws + (size_t)n_tiles * H, n_tiles, H, h, ok, red);
    }
    if (!ok || threadIdx.x >= CS_COLS) return;
    float mean, var;
    if (training) {
        const double mu = s / (double)M;
        double v = ss / (double)M - mu * mu;
        if (v < 0.0) v = 0.0;
        mean = (float)mu;
        var = (float)v;
        const double n = (double)M * (double)dup;
        const float unbiased = (float)(v * (n / (n - 1.0)));
        if (!(skip_if_nonzero && *skip_if_nonzero != 0u)) {  // a timed-out step does not move the running statistics
            rmean[h] = momentum * mean + (1.0f - momentum) * rmean[h];
            rvar[h] = momentum * unbiased + (1.0f - momentum) * rvar[h];
        }
    } else {
        mean = rmean[h];
        var = rvar[h];
    }
    const float invstd = 1.0f / sqrtf(var + eps);
    const float sc = gamma[h] * invstd;
    scale[h] = sc;
    shift[h] = __builtin_fmaf(-mean, sc, beta[h]);
    if (save_mean) save_mean[h] = mean;
    if (save_invstd) save_invstd[h] = invstd;
}

// ---------------------------------------------------------------- column reductions
// Partial sums over RB rows for 256 columns per block (4 waves x 64 lanes x float4... here
// one column per lane-slot of a float4).  MODE 0: sum x.  MODE 1: (sum dy, sum dy*xhat) with
// per-column mean/invstd (BatchNorm).  MODE 2: same with per-row mu/rstd (LayerNorm).
template <int MODE>
__global__ __launch_bounds__(256) void colpartial_kernel(int M, int H, const float* __restrict__ dy,
                                                         const float* __restrict__ x,
                                                         const float* __restrict__ p0,
                                                         const float* __restrict__ p1,
                                                         float* __restrict__ ws, int n_rb) {
    __shared__ float red[2][4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + lane * 4;
    const int rb = blockIdx.y;
    const int r_begin = rb * RB;
    const int r_end = min(M, r_begin + RB);
    const bool vec = (H % 4 == 0);
    float a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0};
    float cm[4] = {0, 0, 0, 0}, ci[4] = {1, 1, 1, 1};
    if (MODE == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c + e < H) { cm[e] = p0[c + e]; ci[e] = p1[c + e]; }
    }
    for (int r = r_begin + wave; r < r_end; r += 4) {
        float d[4] = {0, 0, 0, 0}, xv[4] = {0, 0, 0, 0};
        const size_t off = (size_t)r * H + c;
        if (vec && c + 3 < H) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(dy + off);
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            if (MODE != 0) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(x + off);
                xv[0] = w.x; xv[1] = w.y; xv[2] = w.z; xv[3] = w.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c + e < H) { d[e] = dy[off + e]; if (MODE != 0) xv[e] = x[off + e]; }
        }
        float rm = 0.f, ri = 1.f;
        if (MODE == 2) { rm = p0[r]; ri = p1[r]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a0[e] += d[e];
            if (MODE == 1) a1[e] += d[e] * ((xv[e] - cm[e]) * ci[e]);
            if (MODE == 2) a1[e] += d[e] * ((xv[e] - rm) * ri);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][wave][lane * 4 + e] = a0[e]; red[1][wave][lane * 4 + e] = a1[e]; }
    __syncthreads();
    const int cc = blockIdx.x * 256 + threadIdx.x;
    if (cc < H) {
        const int t = threadIdx.x;
        ws[(size_t)rb * H + cc] = (red[0][0][t] + red[0][1][t]) + (red[0][2][t] + red[0][3][t]);
        if (MODE != 0) ws[(size_t)(n_rb + rb) * H + cc] = (red[1][0][t] + red[1][1][t]) + (red[1][2][t] + red[1][3][t]);
    }
}

__global__ __launch_bounds__(256) void colfinish_kernel(int H, int n_rb, int n_out, const float* __restrict__ ws,
                                                        float* __restrict__ out0, float* __restrict__ out1) {
    __shared__ double red[CS_LANES][CS_COLS];
    const int h = blockIdx.x * CS_COLS + (threadIdx.x & (CS_COLS - 1));
    const bool ok = h < H;
    const double s0 = col_sum64(ws, n_rb, H, h, ok, red);
    double s1 = 0.0;
    if (n_out > 1) s1 = col_sum64(ws + (size_t)n_rb * H, n_rb, H, h, ok, red);
    if (!ok || threadIdx.x >= CS_COLS) return;
    out0[h] = (float)s0;
    if (n_out > 1) out1[h] = (float)s1;
}

// dx = gamma*invstd * (dy - dbeta/M - xhat*dgamma/M) = k1*dy + k2*x + k3 with per-column k1,k2,k3:
// a thread keeps the coefficients of its 4 columns in registers and strides over rows.
#ifndef BN_APPLY_NT_X
#define BN_APPLY_NT_X 1
#endif
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(int M, int H, float invM, const float* __restrict__ dy,
                                                           const float* __restrict__ x,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ dgamma,
                                                           const float* __restrict__ dbeta,
                                                           float* __restrict__ dx) {
    const int HQ = H / 4;
    const int cq = blockIdx.x * blockDim.x + threadIdx.x;
    if (cq >= HQ) return;
    const int c = cq * 4;
    f32x4 k1, k2, k3;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float is = invstd[c + e], gi = gamma[c + e] * is;
        const float t = is * (dgamma[c + e] * invM);
        k1[e] = gi;
        k2[e] = -(gi * t);
        k3[e] = gi * (mean[c + e] * t - dbeta[c + e] * invM);
    }
    // 4 rows per trip, all eight 16-byte loads in flight before the first use (one pair per trip left the
    // kernel latency-bound at 2.2 TB/s)
    constexpr int UR = 4;
    for (int r = blockIdx.y * UR; r < M; r += gridDim.y * UR) {
        if (r + UR <= M) {
            f32x4 d[UR], xv[UR];
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const size_t o = (size_t)(r + u) * H + c;
                d[u] = *reinterpret_cast<const f32x4*>(dy + o);
#if BN_APPLY_NT_X  // the raw projection (saved by the forward, read here for the last time) as a streaming load: dx, which the
                   // two GEMMs behind this pass read, keeps its place in the infinity cache (cfg3 step 6.37-6.47 -> 6.28-6.33 ms)
                xv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + o));
#else
                xv[u] = *reinterpret_cast<const f32x4*>(x + o);
#endif
            }
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                f32x4 out;
#pragma unroll
                for (int e = 0; e < 4; ++e) out[e] = k1[e] * d[u][e] + k2[e] * xv[u][e] + k3[e];
                *reinterpret_cast<f32x4*>(dx + (size_t)(r + u) * H + c) = out;
            }
        } else {
            for (int q = r; q < M; ++q) {
                const size_t o = (size_t)q * H + c;
                const f32x4 d = *reinterpret_cast<const f32x4*>(dy + o);
                const f32x4 xv = *reinterpret_cast<const f32x4*>(x + o);
                f32x4 out;
#pragma unroll
                for (int e = 0; e < 4; ++e) out[e] = k1[e] * d[e] + k2[e] * xv[e] + k3[e];
                *reinterpret_cast<f32x4*>(dx + o) = out;
            }
        }
    }
}

// The same pass writing dx as its three exact bf16 planes (x = t1 + t2 + t3, the truncation split the split GEMMs
// otherwise redo in every workgroup that stages a tile of dx): 8 columns per thread, 16-byte plane stores.  dy2
// (nullable): the second direction's gradient of a bidirectional layer, added here (dy = dy + dy2) instead of by
// a separate pass (sparch_add_halves).  dx (nullable): the fp32 tensor as well.
typedef unsigned bna_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void bn_bwd_apply_planes_kernel(int M, int H, float invM, const float* __restrict__ dy,
                                                                  const float* __restrict__ dy2,
                                                                  const float* __restrict__ x,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ dgamma,
                                                                  const float* __restrict__ dbeta,
                                                                  unsigned short* __restrict__ planes,
                                                                  float* __restrict__ dx) {
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (c >= H) return;
    float k1[8], k2[8], k3[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float is = invstd[c + e], gi = gamma[c + e] * is;
        const float t = is * (dgamma[c + e] * invM);
        k1[e] = gi;
        k2[e] = -(gi * t);
        k3[e] = gi * (mean[c + e] * t - dbeta[c + e] * invM);
    }
    const size_t plane = (size_t)M * H;
    constexpr int UR = 2;
    for (int r = blockIdx.y * UR; r < M; r += gridDim.y * UR) {
        f32x4 d[UR][2], xv[UR][2], d2[UR][2];
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const size_t o = (size_t)min(r + u, M - 1) * H + c;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                d[u][h] = *reinterpret_cast<const f32x4*>(dy + o + 4 * h);
                xv[u][h] = *reinterpret_cast<const f32x4*>(x + o + 4 * h);
                if (dy2) d2[u][h] = *reinterpret_cast<const f32x4*>(dy2 + o + 4 * h);
            }
        }
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            if (r + u >= M) break;
            const size_t o = (size_t)(r + u) * H + c;
            float out[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float g = d[u][e >> 2][e & 3];
                if (dy2) g = g + d2[u][e >> 2][e & 3];
                out[e] = k1[e] * g + k2[e] * xv[u][e >> 2][e & 3] + k3[e];
            }
            bna_u32x4 w1, w2, w3;
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
                const unsigned x0 = __float_as_uint(out[2 * pr]), x1 = __float_as_uint(out[2 * pr + 1]);
                const float r0 = out[2 * pr] - __uint_as_float(x0 & 0xFFFF0000u);
                const float r1 = out[2 * pr + 1] - __uint_as_float(x1 & 0xFFFF0000u);
                const unsigned y0 = __float_as_uint(r0), y1 = __float_as_uint(r1);
                const float q0 = r0 - __uint_as_float(y0 & 0xFFFF0000u);
                const float q1 = r1 - __uint_as_float(y1 & 0xFFFF0000u);
                w1[pr] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
                w2[pr] = __builtin_amdgcn_perm(y1, y0, 0x07060302u);
                w3[pr] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
            }
            *reinterpret_cast<bna_u32x4*>(planes + o) = w1;
            *reinterpret_cast<bna_u32x4*>(planes + plane + o) = w2;
            *reinterpret_cast<bna_u32x4*>(planes + 2 * plane + o) = w3;
            if (dx) {
                *reinterpret_cast<f32x4*>(dx + o) = f32x4{out[0], out[1], out[2], out[3]};
                *reinterpret_cast<f32x4*>(dx + o + 4) = f32x4{out[4], out[5], out[6], out[7]};
            }
        }
    }
}

__global__ void bn_bwd_apply_scalar_kernel(size_t n, int H, float invM, const float* __restrict__ dy,
                                           const float* __restrict__ x, const float* __restrict__ mean,
                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                           const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                           float* __restrict__ dx) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % (size_t)H);
    const float is = invstd[c];
    const float xh = (x[i] - mean[c]) * is;
    dx[i] = gamma[c] * is * (dy[i] - dbeta[c] * invM - xh * (dgamma[c] * invM));
}

// ---------------------------------------------------------------- LayerNorm (one wave per row)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Hn <= H: the leading Hn columns are the normalised width, columns Hn..H-1 are padding (written 0 / gradient 0)
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(int M, int H, int Hn, const float* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps,
                                                            float* __restrict__ y, float* __restrict__ mu,
                                                            float* __restrict__ rstd) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * H;
    float s = 0.f;
    for (int c = lane; c < Hn; c += 64) s += xr[c];
    const float mean = wave_sum(s) / (float)Hn;
    float v = 0.f;
    for (int c = lane; c < Hn; c += 64) { const float d = xr[c] - mean; v += d * d; }
    const float var = wave_sum(v) / (float)Hn;
    const float rs = 1.0f / sqrtf(var + eps);
    float* yr = y + (size_t)row * H;
    for (int c = lane; c < Hn; c += 64) yr[c] = (xr[c] - mean) * rs * gamma[c] + beta[c];
    for (int c = Hn + lane; c < H; c += 64) yr[c] = 0.f;
    if (lane == 0) { mu[row] = mean; rstd[row] = rs; }
}

// dx = rstd * (g - mean(g) - xhat*mean(g*xhat)),  g = gamma*dy
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(int M, int H, int Hn, const float* __restrict__ dy,
                                                            const float* __restrict__ x,
                                                            const float* __restrict__ mu,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma,
                                                            float* __restrict__ dx) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float m = mu[row], rs = rstd[row];
    const float* xr = x + (size_t)row * H;
    const float* dr = dy + (size_t)row * H;
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < Hn; c += 64) {
        const float g = gamma[c] * dr[c];
        s1 += g;
        s2 += g * ((xr[c] - m) * rs);
    }
    s1 = wave_sum(s1) / (float)Hn;
    s2 = wave_sum(s2) / (float)Hn;
    float* o = dx + (size_t)row * H;
    for (int c = lane; c < Hn; c += 64) {
        const float xh = (xr[c] - m) * rs;
        o[c] = rs * (gamma[c] * dr[c] - s1 - xh * s2);
    }
    for (int c = Hn + lane; c < H; c += 64) o[c] = 0.f;
}

// ---------------------------------------------------------------- misc elementwise
__global__ void add_halves_kernel(size_t n, const float* __restrict__ x, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = x[i] + x[n + i];
}

// out[j][h] = gate_j(h) * sum_r ws[j][r][h]
struct ClampArgs {
    const float* raw[8];
    float* out[8];
    float lo[8], hi[8];
    int gated[8];
};
__global__ __launch_bounds__(256) void colsum_clamped_kernel(int n_params, int rows, int H,
                                                             const float* __restrict__ ws, ClampArgs a) {
    __shared__ double red[CS_LANES][CS_COLS];
    const int h = blockIdx.x * CS_COLS + (threadIdx.x & (CS_COLS - 1));
    const int j = blockIdx.y;
    const bool ok = h < H;
    const double s = col_sum64(ws + (size_t)j * rows * H, rows, H, h, ok, red);
    if (!ok || threadIdx.x >= CS_COLS) return;
    float v = (float)s;
    if (a.gated[j]) {
        const float x = a.raw[j][h];
        if (!(x >= a.lo[j] && x <= a.hi[j])) v = 0.f;
    }
    a.out[j][h] = v;
}

}  // namespace

extern "C" int sparch_bn_finalize(int H, int M, int n_tiles, int dup, const float* colstat_ws,
                                  const float* gamma, const float* beta, float* running_mean,
                                  float* running_var, float momentum, float eps, int training,
                                  float* scale, float* shift, float* save_mean, float* save_invstd,
                                  const uint32_t* skip_if_nonzero, int64_t* num_batches_tracked, void* stream) {
    SPARCH_ENTER();
    if (H <= 0 || !gamma || !beta || !running_mean || !running_var || !scale || !shift) return SPARCH_EINVAL;
    if (training && (M <= 0 || n_tiles <= 0 || dup < 1 || !colstat_ws)) return SPARCH_EINVAL;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(H, CS_COLS)), dim3(256), 0, (hipStream_t)stream, H, M,
                       n_tiles, dup, colstat_ws, gamma, beta, running_mean, running_var, momentum, eps,
                       training, scale, shift, save_mean, save_invstd, skip_if_nonzero,
                       reinterpret_cast<long long*>(num_batches_tracked));
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

extern "C" size_t sparch_bn_bwd_workspace_bytes(int M, int H) {
    if (M <= 0 || H <= 0) return 0;
    return (size_t)2 * cdiv(M, RB) * H * sizeof(float);
}

extern "C" int sparch_bn_bwd_reduce(int M, int H, const float* dy, const float* x, const float* mean,
                                    const float* invstd, float* dgamma, float* dbeta, void* ws,
                                    size_t ws_bytes, void* stream) {
    SPARCH_ENTER();
    if (M <= 0 || H <= 0 || !dy || !x || !mean || !invstd || !dgamma || !dbeta) return SPARCH_EINVAL;
    if (!ws || ws_bytes < sparch_bn_bwd_workspace_bytes(M, H)) return SPARCH_EWORKSPACE;
    if (!aligned16(dy) || !aligned16(x)) return SPARCH_EALIGN;
    const int n_rb = cdiv(M, RB);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colpartial_kernel<1>, dim3(cdiv(H, 256), n_rb), dim3(256), 0, st, M, H, dy, x, mean,
                       invstd, (float*)ws, n_rb);
    SPARCH_CHECK_LAUNCH();
    hipLaunchKernelGGL(colfinish_kernel, dim3(cdiv(H, CS_COLS)), dim3(256), 0, st, H, n_rb, 2, (const float*)ws,
                       dbeta, dgamma);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

extern "C" int sparch_bn_bwd_apply(int M, int H, const float* dy, const float* x, const float* mean,
                                   const float* invstd, const float* gamma, const float* dgamma,
                                   const float* dbeta, float* dx, void* stream) {
    SPARCH_ENTER();
    if (M <= 0 || H <= 0 || !dy || !x || !mean || !invstd || !gamma || !dgamma || !dbeta || !dx)
        return SPARCH_EINVAL;
    const size_t n = (size_t)M * H;
    hipStream_t st = (hipStream_t)stream;
    if (H % 4 == 0 && aligned16(dy) && aligned16(x) && aligned16(dx)) {
        const int bx = cdiv(H / 4, 256);
        const int groups = cdiv(M, 4);                       // a trip of the kernel takes 4 rows
        const int by = groups < 2048 / bx ? groups : 2048 / bx;  // ~2048 workgroups stride over the row groups
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(bx, by), dim3(256), 0, st, M, H, 1.0f / (float)M, dy, x,
                           mean, invstd, gamma, dgamma, dbeta, dx);
    } else {
        hipLaunchKernelGGL(bn_bwd_apply_scalar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n,
                           H, 1.0f / (float)M, dy, x, mean, invstd, gamma, dgamma, dbeta, dx);
    }
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

extern "C" int sparch_bn_bwd_apply_planes(int M, int H, const float* dy, const float* dy2, const float* x,
                                          const float* mean, const float* invstd, const float* gamma,
                                          const float* dgamma, const float* dbeta, uint16_t* planes, float* dx,
                                          void* stream) {
    SPARCH_ENTER();
    if (M <= 0 || H <= 0 || H % 8 != 0 || !dy || !x || !mean || !invstd || !gamma || !dgamma || !dbeta || !planes)
        return SPARCH_EINVAL;
    if (!aligned16(dy) || !aligned16(x) || !aligned16(planes) || (dy2 && !aligned16(dy2)) || (dx && !aligned16(dx)))
        return SPARCH_EALIGN;
    const int bx = cdiv(H / 8, 256);
    const int groups = cdiv(M, 2);
    const int by = groups < 4096 / bx ? groups : 4096 / bx;
    hipLaunchKernelGGL(bn_bwd_apply_planes_kernel, dim3(bx, by), dim3(256), 0, (hipStream_t)stream, M, H,
                       1.0f / (float)M, dy, dy2, x, mean, invstd, gamma, dgamma, dbeta, planes, dx);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

extern "C" int sparch_layernorm_fwd(int M, int H, int Hn, const float* x, const float* gamma, const float* beta,
                                    float eps, float* y, float* mu, float* rstd, void* stream) {
    SPARCH_ENTER();
    if (M <= 0 || H <= 0 || Hn <= 0 || Hn > H || !x || !gamma || !beta || !y || !mu || !rstd) return SPARCH_EINVAL;
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, M, H, Hn, x,
                       gamma, beta, eps, y, mu, rstd);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

extern "C" int sparch_layernorm_bwd(int M, int H, int Hn, const float* dy, const float* x, const float* mu,
                                    const float* rstd, const float* gamma, float* dx, float* dgamma,
                                    float* dbeta, void* ws, size_t ws_bytes, void* stream) {
    SPARCH_ENTER();
    if (M <= 0 || H <= 0 || Hn <= 0 || Hn > H || !dy || !x || !mu || !rstd || !gamma || !dx || !dgamma || !dbeta)
        return SPARCH_EINVAL;
    if (!ws || ws_bytes < sparch_bn_bwd_workspace_bytes(M, H)) return SPARCH_EWORKSPACE;
    if (!aligned16(dy) || !aligned16(x)) return SPARCH_EALIGN;
    const int n_rb = cdiv(M, RB);
    hipStream_t st = (hipStream_t)stream;
    // column sums first (dx may alias dy); the padding columns' sums are finite and unused (x = 0, gamma = 0 there)
    hipLaunchKernelGGL(colpartial_kernel<2>, dim3(cdiv(H, 256), n_rb), dim3(256), 0, st, M, H, dy, x, mu, rstd,
                       (float*)ws, n_rb);
    SPARCH_CHECK_LAUNCH();
    hipLaunchKernelGGL(colfinish_kernel, dim3(cdiv(H, CS_COLS)), dim3(256), 0, st, H, n_rb, 2, (const float*)ws,
                       dbeta, dgamma);
    SPARCH_CHECK_LAUNCH();
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, M, H, Hn, dy, x, mu, rstd,
                       gamma, dx);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

extern "C" int sparch_colsum(int M, int H, const float* x, float* out, void* ws, size_t ws_bytes,
                             void* stream) {
    SPARCH_ENTER();
    if (M <= 0 || H <= 0 || !x || !out) return SPARCH_EINVAL;
    if (!ws || ws_bytes < sparch_bn_bwd_workspace_bytes(M, H)) return SPARCH_EWORKSPACE;
    if (!aligned16(x)) return SPARCH_EALIGN;
    const int n_rb = cdiv(M, RB);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colpartial_kernel<0>, dim3(cdiv(H, 256), n_rb), dim3(256), 0, st, M, H, x, x,
                       (const float*)nullptr, (const float*)nullptr, (float*)ws, n_rb);
    SPARCH_CHECK_LAUNCH();
    hipLaunchKernelGGL(colfinish_kernel, dim3(cdiv(H, CS_COLS)), dim3(256), 0, st, H, n_rb, 1, (const float*)ws, out,
                       (float*)nullptr);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

extern "C" int sparch_add_halves(size_t n, const float* x, float* out, void* stream) {
    SPARCH_ENTER();
    if (n == 0 || !x || !out) return SPARCH_EINVAL;
    hipLaunchKernelGGL(add_halves_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       n, x, out);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

extern "C" int sparch_colsum_clamped(int n_params, int rows, int H, const float* ws,
                                     const float* const* raw, const float* lim_lo_hi, float* const* out,
                                     void* stream) {
    SPARCH_ENTER();
    if (n_params < 1 || n_params > 8 || rows <= 0 || H <= 0 || !ws || !out) return SPARCH_EINVAL;
    ClampArgs a{};
    for (int j = 0; j < n_params; ++j) {
        if (!out[j]) return SPARCH_EINVAL;
        a.out[j] = out[j];
        a.raw[j] = raw ? raw[j] : nullptr;
        a.gated[j] = (raw && raw[j] && lim_lo_hi) ? 1 : 0;
        if (a.gated[j]) { a.lo[j] = lim_lo_hi[2 * j]; a.hi[j] = lim_lo_hi[2 * j + 1]; }
    }
    hipLaunchKernelGGL(colsum_clamped_kernel, dim3(cdiv(H, CS_COLS), n_params), dim3(256), 0, (hipStream_t)stream,
                       n_params, rows, H, ws, a);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

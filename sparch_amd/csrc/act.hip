// f-4 (non-spiking baselines, anns.py): the element-wise tail of an MLP layer and the ANN readout.
//
//   sparch_act_fwd / _bwd      y = dropout(act(z * scale + shift))          anns.py:218-227 (MLPLayer.forward)
//   sparch_softmax_sum_fwd/bwd out[b,:] = sum_t softmax(x[b,t,:])           anns.py:658-665 (ReadoutLayerANN)
//
// Both are HBM-bound single passes (4-8 bytes per element each way).  The BatchNorm affine is folded into
// (scale, shift) exactly as for the spiking cells; dropout regenerates its mask from (seed, element index).
#include "common.h"

namespace {

__device__ __forceinline__ float act_f(int kind, float v) {
    if (kind == SPARCH_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    if (kind == SPARCH_ACT_RELU) return fmaxf(v, 0.0f);
    return tanhf(v);
}
// derivative expressed through the activation's OUTPUT a = act(v)
__device__ __forceinline__ float act_df(int kind, float a) {
    if (kind == SPARCH_ACT_SIGMOID) return a * (1.0f - a);
    if (kind == SPARCH_ACT_RELU) return a > 0.0f ? 1.0f : 0.0f;
    return 1.0f - a * a;
}

template <bool BWD>
__global__ __launch_bounds__(256) void act_kernel(int kind, size_t n4, int H, const float* __restrict__ z,
                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                  const float* __restrict__ dy, float p_drop, float inv_keep,
                                                  uint64_t seed_arg, float* __restrict__ out) {
    const bool drop = p_drop > 0.0f;
    const uint64_t seed = drop ? resolve_seed(seed_arg) : 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e0 = i * 4;
        const int h = (int)(e0 % (size_t)H);  // H % 4 == 0: the four elements share a row
        const f32x4 zv = *reinterpret_cast<const f32x4*>(z + e0);
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f}, g = {0.f, 0.f, 0.f, 0.f}, o;
        if (scale) { sc = *reinterpret_cast<const f32x4*>(scale + h); sh = *reinterpret_cast<const f32x4*>(shift + h); }
        if (BWD) g = *reinterpret_cast<const f32x4*>(dy + e0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = zv[e];
            if (scale) v = bn_affine(v, sc[e], sh[e]);
            const float a = act_f(kind, v);
            const float k = drop ? keep_scale(seed, e0 + e, p_drop, inv_keep) : 1.0f;
            o[e] = BWD ? (g[e] * k) * act_df(kind, a) : a * k;
        }
        *reinterpret_cast<f32x4*>(out + e0) = o;
    }
}

// One workgroup per batch row; per time step the 256 threads hold the row's K values (4 per thread per
// 1024-column slab), reduce max and sum through LDS, and accumulate softmax in TIME ORDER (the reference's
// `y += softmax(x_t)` loop) in registers.
constexpr int SS_MAX_SLABS = 4;  // K <= 4096

__device__ __forceinline__ float wg_reduce(float v, bool is_max, float* red, int tid) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float w = __shfl_xor(v, o);
        v = is_max ? fmaxf(v, w) : v + w;
    }
    __syncthreads();  // protects `red` from the previous use
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    const float a = red[0], b = red[1], c = red[2], d = red[3];
    return is_max ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : (a + b) + (c + d);
}

template <bool BWD>
__global__ __launch_bounds__(256) void softmax_sum_kernel(int T, int K, const float* __restrict__ x,
                                                          const float* __restrict__ g, float* __restrict__ out) {
    __shared__ float red[4];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int slabs = (K + 1023) / 1024;
    f32x4 acc[SS_MAX_SLABS], gv[SS_MAX_SLABS];
#pragma unroll
    for (int s = 0; s < SS_MAX_SLABS; ++s) {
        acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
        gv[s] = acc[s];
        const int c = s * 1024 + tid * 4;
        if (BWD && s < slabs && c < K) gv[s] = *reinterpret_cast<const f32x4*>(g + (size_t)b * K + c);
    }
    for (int t = 0; t < T; ++t) {
        const float* row = x + ((size_t)b * T + t) * K;
        f32x4 v[SS_MAX_SLABS];
        float m = -INFINITY;
#pragma unroll
        for (int s = 0; s < SS_MAX_SLABS; ++s) {
            const int c = s * 1024 + tid * 4;
            if (s < slabs && c < K) {
                v[s] = *reinterpret_cast<const f32x4*>(row + c);
                m = fmaxf(m, fmaxf(fmaxf(v[s].x, v[s].y), fmaxf(v[s].z, v[s].w)));
            }
        }
        m = wg_reduce(m, true, red, tid);
        float den = 0.f;
#pragma unroll
        for (int s = 0; s < SS_MAX_SLABS; ++s) {
            const int c = s * 1024 + tid * 4;
            if (s < slabs && c < K) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[s][e] = expf(v[s][e] - m); den += v[s][e]; }
            }
        }
        den = wg_reduce(den, false, red, tid);
        if (!BWD) {
#pragma unroll
            for (int s = 0; s < SS_MAX_SLABS; ++s) {
                const int c = s * 1024 + tid * 4;
                if (s < slabs && c < K) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[s][e] = acc[s][e] + v[s][e] / den;   // anns.py:663
                }
            }
        } else {
            float dot = 0.f;
#pragma unroll
            for (int s = 0; s < SS_MAX_SLABS; ++s) {
                const int c = s * 1024 + tid * 4;
                if (s < slabs && c < K) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[s][e] = v[s][e] / den; dot += v[s][e] * gv[s][e]; }
                }
            }
            dot = wg_reduce(dot, false, red, tid);
#pragma unroll
            for (int s = 0; s < SS_MAX_SLABS; ++s) {
                const int c = s * 1024 + tid * 4;
                if (s < slabs && c < K) {
                    f32x4 d;
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[e] = v[s][e] * (gv[s][e] - dot);
                    *reinterpret_cast<f32x4*>(out + ((size_t)b * T + t) * K + c) = d;
                }
            }
        }
    }
    if (!BWD) {
#pragma unroll
        for (int s = 0; s < SS_MAX_SLABS; ++s) {
            const int c = s * 1024 + tid * 4;
            if (s < slabs && c < K) *reinterpret_cast<f32x4*>(out + (size_t)b * K + c) = acc[s];
        }
    }
}

int act_launch(bool bwd, int kind, size_t n, int H, const float* z, const float* scale, const float* shift,
               const float* dy, float p_drop, uint64_t seed, float* out, void* stream) {
    if (kind < SPARCH_ACT_SIGMOID || kind > SPARCH_ACT_TANH || n == 0 || H <= 0 || !z || !out) return SPARCH_EINVAL;
    if (H % 4 != 0 || n % (size_t)H != 0) return SPARCH_EINVAL;
    if ((scale == nullptr) != (shift == nullptr)) return SPARCH_EINVAL;
    if (bwd && !dy) return SPARCH_EINVAL;
    if (!(p_drop >= 0.0f && p_drop < 1.0f)) return SPARCH_EINVAL;
    if (!aligned16(z) || !aligned16(out) || !aligned16(scale) || !aligned16(shift) || !aligned16(dy)) return SPARCH_EALIGN;
    const size_t n4 = n / 4;
    const unsigned grid = (unsigned)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    const float inv_keep = 1.0f / (1.0f - p_drop);
    if (bwd)
        hipLaunchKernelGGL(act_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, kind, n4, H, z, scale,
                           shift, dy, p_drop, inv_keep, seed, out);
    else
        hipLaunchKernelGGL(act_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, kind, n4, H, z, scale,
                           shift, dy, p_drop, inv_keep, seed, out);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

}  // namespace

extern "C" int sparch_act_fwd(int kind, size_t n, int H, const float* z, const float* scale, const float* shift,
                              float p_drop, uint64_t seed, float* y, void* stream) {
    SPARCH_ENTER();
    return act_launch(false, kind, n, H, z, scale, shift, nullptr, p_drop, seed, y, stream);
}

extern "C" int sparch_act_bwd(int kind, size_t n, int H, const float* z, const float* scale, const float* shift,
                              const float* dy, float p_drop, uint64_t seed, float* dz, void* stream) {
    SPARCH_ENTER();
    return act_launch(true, kind, n, H, z, scale, shift, dy, p_drop, seed, dz, stream);
}

extern "C" int sparch_softmax_sum_fwd(int B, int T, int K, const float* x, float* out, void* stream) {
    SPARCH_ENTER();
    if (B <= 0 || T <= 0 || K <= 0 || K % 4 != 0 || K > 1024 * SS_MAX_SLABS || !x || !out) return SPARCH_EINVAL;
    if (!aligned16(x) || !aligned16(out)) return SPARCH_EALIGN;
    hipLaunchKernelGGL(softmax_sum_kernel<false>, dim3(B), dim3(256), 0, (hipStream_t)stream, T, K, x, nullptr, out);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

extern "C" int sparch_softmax_sum_bwd(int B, int T, int K, const float* x, const float* g, float* dx, void* stream) {
    SPARCH_ENTER();
    if (B <= 0 || T <= 0 || K <= 0 || K % 4 != 0 || K > 1024 * SS_MAX_SLABS || !x || !g || !dx) return SPARCH_EINVAL;
    if (!aligned16(x) || !aligned16(g) || !aligned16(dx)) return SPARCH_EALIGN;
    hipLaunchKernelGGL(softmax_sum_kernel<true>, dim3(B), dim3(256), 0, (hipStream_t)stream, T, K, x, g, dx);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

// ---- a11: the loss of the train step, nn.CrossEntropyLoss()(output, y) (exp.py:100, 362; mean reduction), and its
//      gradient with respect to the logits in ONE launch: eager torch spends seven small kernels on it per step
//      (cat, log-softmax, nll forward, two fills, nll backward, log-softmax backward).  One workgroup; a thread
//      owns batch rows b, b + 256, ...: max, sum of exponentials, loss_b = log(sum) + max - x[label] and
//      dlogits = (softmax - onehot) / B; the B row losses are summed in a fixed order (deterministic).
namespace {
__global__ __launch_bounds__(256) void ce_loss_kernel(int B, int C, const float* __restrict__ x,
                                                      const long long* __restrict__ y, float* __restrict__ loss,
                                                      float* __restrict__ dx) {
    __shared__ float part[256];
    const int tid = threadIdx.x;
    const float inv_b = 1.0f / (float)B;
    float acc = 0.0f;
    for (int b = tid; b < B; b += 256) {
        const float* row = x + (size_t)b * C;
        float m = row[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, row[c]);
        float s = 0.0f;
        for (int c = 0; c < C; ++c) s += expf(row[c] - m);
        const float lse = logf(s);
        const int lab = (int)y[b];
        const bool lab_ok = lab >= 0 && lab < C;
        if (lab_ok) acc += (lse + m) - row[lab];
        for (int c = 0; c < C; ++c) {
            const float p = expf((row[c] - m) - lse);  // softmax through log-softmax, as torch's backward does
            dx[(size_t)b * C + c] = (p - ((lab_ok && c == lab) ? 1.0f : 0.0f)) * inv_b;
        }
    }
    part[tid] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) part[tid] += part[tid + o];
        __syncthreads();
    }
    if (tid == 0) loss[0] = part[0] * inv_b;
}
}  // namespace

extern "C" int sparch_ce_loss(int B, int C, const float* logits, const int64_t* labels, float* loss, float* dlogits,
                              void* stream) {
    SPARCH_ENTER();
    if (B <= 0 || C <= 0 || !logits || !labels || !loss || !dlogits) return SPARCH_EINVAL;
    hipLaunchKernelGGL(ce_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, B, C, logits,
                       reinterpret_cast<const long long*>(labels), loss, dlogits);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

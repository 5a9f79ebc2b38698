// f-2: Adam step for a whole parameter list in ONE launch (the reference calls torch.optim.Adam,
// exp.py:89, 377: ~30 small kernels per step on the device).
//
// Same arithmetic, operation by operation, as torch.optim.Adam's default (foreach, non-amsgrad) path:
//     m  = m + (g - m) * (1 - beta1)                         (Tensor.lerp_)
//     v  = v * beta2 + (g * g) * (1 - beta2)                 (mul_, addcmul_)
//     d  = sqrt(v) / sqrt(1 - beta2^t) + eps
//     p  = p + (m / d) * (-lr / (1 - beta1^t))               (addcdiv_)
// with the scalar factors formed on the host in double precision as Python does and rounded to fp32
// once.  Built with -ffp-contract=off: no fused multiply-adds the reference's separate kernels do not have.
// The tensor table travels in the kernel argument (no device-side descriptor to keep coherent with
// autograd's fresh .grad tensors each step).
#include "common.h"

namespace {

constexpr int ADAM_MAX = 24;       // tensors per launch (kernel-argument budget)
constexpr int ADAM_CHUNK = 4096;   // elements per workgroup

struct AdamBatch {
    float* p[ADAM_MAX];
    const float* g[ADAM_MAX];
    float* m[ADAM_MAX];
    float* v[ADAM_MAX];
    long long first_blk[ADAM_MAX + 1];  // prefix sum of ceil(n / ADAM_CHUNK)
    long long n[ADAM_MAX];
    int count;
    float step_size, beta1, beta2, bc2_sqrt, eps, weight_decay;
    uint32_t* skip_if_nonzero;        // device word (nullable): non-zero -> the step is a no-op, counted in word [1]
    const float* scalars;             // device {step_size, bc2_sqrt} (nullable): overrides the two arguments
    int count_skip;                   // this launch is the first of its step: it counts a skipped step
};

__global__ __launch_bounds__(256) void adam_kernel(AdamBatch a) {
    // a recurrent kernel of this training step timed out (status word raised): its gradients are invalid,
    // leave parameters and moments untouched — no host round trip needed to protect them
    if (a.skip_if_nonzero && *a.skip_if_nonzero != 0u) {
        if (a.count_skip && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(a.skip_if_nonzero + 1, 1u);
        return;
    }
    int ti = 0;
    while (ti + 1 < a.count && (long long)blockIdx.x >= a.first_blk[ti + 1]) ++ti;  // uniform, <= 24 steps
    const long long base = ((long long)blockIdx.x - a.first_blk[ti]) * ADAM_CHUNK;
    const long long n = a.n[ti];
    float* __restrict__ p = a.p[ti];
    const float* __restrict__ g = a.g[ti];
    float* __restrict__ m = a.m[ti];
    float* __restrict__ v = a.v[ti];
    const float w1 = 1.0f - a.beta1, w2 = 1.0f - a.beta2;
    // a step captured in a HIP graph cannot take its per-step factors as (frozen) kernel arguments
    const float step_size = a.scalars ? a.scalars[0] : a.step_size;
    const float bc2_sqrt = a.scalars ? a.scalars[1] : a.bc2_sqrt;
#pragma unroll 4
    for (int j = 0; j < ADAM_CHUNK / 256; ++j) {
        const long long i = base + (long long)j * 256 + threadIdx.x;
        if (i >= n) break;
        float gi = g[i];
        const float pi = p[i];
        if (a.weight_decay != 0.0f) gi = gi + pi * a.weight_decay;
        const float mi = m[i] + (gi - m[i]) * w1;
        const float vi = v[i] * a.beta2 + (gi * gi) * w2;
        const float d = sqrtf(vi) / bc2_sqrt + a.eps;
        m[i] = mi;
        v[i] = vi;
        p[i] = pi + (mi / d) * (-step_size);
    }
}

// one thread: t += 1, then the two per-step factors in double precision (the host path's formula)
__global__ void adam_scalars_kernel(double* t, const double* lr, double beta1, double beta2, float* scalars) {
    const double tt = *t + 1.0;
    *t = tt;
    const double bc1 = 1.0 - pow(beta1, tt);
    const double bc2 = 1.0 - pow(beta2, tt);
    scalars[0] = (float)(*lr / bc1);
    scalars[1] = (float)sqrt(bc2);
}

}  // namespace

extern "C" int sparch_adam_scalars(double* t_dev, const double* lr_dev, double beta1, double beta2,
                                   float* scalars_dev, void* stream) {
    SPARCH_ENTER();
    if (!t_dev || !lr_dev || !scalars_dev || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0))
        return SPARCH_EINVAL;
    hipLaunchKernelGGL(adam_scalars_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, t_dev, lr_dev, beta1, beta2,
                       scalars_dev);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

extern "C" int sparch_adam_step(int n_tensors, float* const* params, const float* const* grads,
                                float* const* exp_avg, float* const* exp_avg_sq, const int64_t* numel,
                                float step_size, float beta1, float beta2, float bc2_sqrt, float eps,
                                float weight_decay, const float* scalars_dev, uint32_t* skip_if_nonzero,
                                void* stream) {
    SPARCH_ENTER();
    if (n_tensors < 0 || (n_tensors > 0 && (!params || !grads || !exp_avg || !exp_avg_sq || !numel)))
        return SPARCH_EINVAL;
    if (!scalars_dev && !(bc2_sqrt > 0.0f)) return SPARCH_EINVAL;
    bool first = true;
    for (int t = 0; t < n_tensors;) {  // ONE running index: empty tensors are skipped without being counted
        AdamBatch a{};
        a.count = 0;
        long long blk = 0;
        for (; t < n_tensors && a.count < ADAM_MAX; ++t) {
            if (numel[t] < 0 || (numel[t] > 0 && (!params[t] || !grads[t] || !exp_avg[t] || !exp_avg_sq[t])))
                return SPARCH_EINVAL;
            if (numel[t] == 0) continue;
            const int c = a.count++;
            a.p[c] = params[t]; a.g[c] = grads[t]; a.m[c] = exp_avg[t]; a.v[c] = exp_avg_sq[t];
            a.n[c] = numel[t];
            a.first_blk[c] = blk;
            blk += (numel[t] + ADAM_CHUNK - 1) / ADAM_CHUNK;
        }
        a.first_blk[a.count] = blk;
        if (a.count == 0) continue;
        a.step_size = step_size; a.beta1 = beta1; a.beta2 = beta2; a.bc2_sqrt = bc2_sqrt; a.eps = eps;
        a.weight_decay = weight_decay; a.skip_if_nonzero = skip_if_nonzero; a.scalars = scalars_dev;
        a.count_skip = first ? 1 : 0;
        first = false;
        hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blk), dim3(256), 0, (hipStream_t)stream, a);
        SPARCH_CHECK_LAUNCH();
    }
    return SPARCH_OK;
}

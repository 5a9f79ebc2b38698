// Shared device/host helpers for libsparch_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sparch_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// hipGetLastError() is sticky per thread and other HIP users in the process (the PyTorch runtime)
// may leave a benign error behind: every entry point clears it first (SPARCH_ENTER), then each
// launch is checked and the real HIP error kept for sparch_last_hip_error().
void sparch_note_hip_error(int hip_error);
#define SPARCH_ENTER() (void)hipGetLastError()
#define SPARCH_CHECK_LAUNCH()                          \
    do {                                               \
        hipError_t e_ = hipGetLastError();             \
        if (e_ != hipSuccess) { sparch_note_hip_error((int)e_); return SPARCH_ELAUNCH; } \
    } while (0)

// Operand precision of the matrix products (the `precision` argument of the entry points): 0 = exact fp32 through
// bf16 splits, 1 = operands rounded to bf16 once (round-to-nearest-even), fp32 accumulation — BASELINE configs[4].
// An entry point opens a PrecisionScope for the call; the dispatch code below it asks sparch_operand_bf16().
int sparch_operand_bf16(void);
struct PrecisionScope {
    int prev;
    bool ok;  // false: unknown precision value (the caller returns SPARCH_EINVAL)
    explicit PrecisionScope(int precision);
    ~PrecisionScope();
    PrecisionScope(const PrecisionScope&) = delete;
    PrecisionScope& operator=(const PrecisionScope&) = delete;
};

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Clamp ranges of the neuron parameters (snns.py:229, 356-359, 631-634), as the
// float32 values torch.clamp compares against (Python doubles rounded to f32).
#define SP_ALPHA_LO 0x1.a330aep-1f /* float32(exp(-1/5))   */
#define SP_ALPHA_HI 0x1.ebec98p-1f /* float32(exp(-1/25))  */
#define SP_BETA_LO 0x1.ef36f2p-1f  /* float32(exp(-1/30))  */
#define SP_BETA_HI 0x1.fbc046p-1f  /* float32(exp(-1/120)) */
#define SP_A_LO (-1.0f)
#define SP_A_HI 1.0f
#define SP_B_LO 0.0f
#define SP_B_HI 2.0f

// The folded BatchNorm affine, in the arithmetic of torch's CPU kernel (probed bitwise on this image's build:
// alpha = invstd*weight; beta = fma(-mean, alpha, bias); y = fma(x, alpha, beta)).  The library is built with
// -ffp-contract=off, so the fused forms are spelled out.
__device__ __forceinline__ float bn_affine(float x, float sc, float sh) { return __builtin_fmaf(x, sc, sh); }

// Box-car surrogate gate of SpikeFunctionBoxcar.backward (snns.py:33-35): the reference CLEARS the entries
// where x <= -0.5 or x > 0.5 (assignment, not multiplication), so a non-finite incoming gradient is zeroed
// outside the box and a NaN x (both comparisons false) passes the gradient through.
__device__ __forceinline__ float boxcar_gate(float g, float x) { return (x <= -0.5f || x > 0.5f) ? 0.0f : g; }

// ---- optional bf16 storage of the states saved for the backward pass (u, w): half the bytes of the two
// largest tensors a spiking layer keeps.  The backward takes three DISCRETE decisions from a saved membrane
// potential — the spike u - theta > 0 and the box-car edges u - theta <= -0.5, u - theta > 0.5 — and those
// must not change, or recomputed spikes would flip.  save_u16 rounds to nearest-even and, when that moved
// the value onto or across one of the three thresholds, steps one bf16 ulp back towards u: every decision is
// then EXACTLY the fp32 one (flip rate 0 by construction); only the continuous uses of u and w (the alpha,
// beta, a gradients) see the 2^-9 relative rounding.
__device__ __forceinline__ unsigned decisions_of(float u, float theta) {
    const float x = u - theta;
    return (x > 0.0f ? 1u : 0u) | (x <= -0.5f ? 2u : 0u) | (x > 0.5f ? 4u : 0u);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16_rne(float x) {
    return __builtin_bit_cast(unsigned short, (__bf16)x);
}
__device__ __forceinline__ unsigned short save_u16(float u, float theta) {
    unsigned short b = f32_to_bf16_rne(u);
    const float v = bf16_to_f32(b);
    if (decisions_of(v, theta) != decisions_of(u, theta) && (b & 0x7FFFu) != 0u) {
        const bool up = v < u;                     // move towards u
        const bool neg = (b & 0x8000u) != 0u;
        b = (unsigned short)((up != neg) ? b + 1 : b - 1);
    }
    return b;
}

__device__ __forceinline__ float clampf(float x, float lo, float hi) {
    return fminf(fmaxf(x, lo), hi);
}

// Dropout decision for output element `idx` of a layer: a counter-based hash of (seed, idx) built
// from two rounds of a 32-bit integer finaliser (lowbias32); keep iff uniform >= p.  Forward and
// backward regenerate the same mask from (seed, idx), nothing is stored.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
// The dropout seed of a launch is a kernel ARGUMENT by default.  For a step captured in a HIP graph the argument
// is frozen at capture time, so the seed may instead live in device memory: bit 63 set = the low 63 bits are
// the device address of a uint64 holding the seed (SPARCH_SEED_IN_MEMORY; the host advances that word between
// replays).  Resolved once per thread at kernel entry.
__device__ __forceinline__ uint64_t resolve_seed(uint64_t seed_or_address) {
    // (read through the constant address space — the word only changes between launches: a generic pointer here is
    // a FLAT load, which hipcc cannot count; its wait, `vmcnt(0) lgkmcnt(0)`, lands at the seed's first use INSIDE the
    // callers' time loops and drains their prefetch pipelines once per trip)
    typedef const uint64_t __attribute__((address_space(4))) * seed_ptr;
    if (seed_or_address & SPARCH_SEED_IN_MEMORY) return *(seed_ptr)(seed_or_address & ~SPARCH_SEED_IN_MEMORY);
    return seed_or_address;
}
__device__ __forceinline__ float keep_scale(uint64_t seed, uint64_t idx, float p, float inv_keep) {
    const uint32_t a = mix32((uint32_t)idx ^ (uint32_t)seed);
    const uint32_t b = mix32(a + (uint32_t)(idx >> 32) * 0x9E3779B9U + (uint32_t)(seed >> 32));
    const float u = (float)(b >> 8) * (1.0f / 16777216.0f);  // 24-bit uniform [0,1)
    return u >= p ? inv_keep : 0.0f;
}

// Diagnostic microbenchmark (not part of the library): cost of each ingredient of the split-GEMM K loop
// beside its MFMAs.  8 waves (2 per SIMD), per iteration and wave: 24 MFMAs fed by fragment reads one group
// ahead (as gemm_spike.hip), optionally + one workgroup barrier (BAR) + 10 ds_write_b64 into the other LDS
// stage (WR) + ~100 VALU of splitting (VALU) + 6 global_load_dwordx4 of a streamed tile (GLD).
// Build: hipcc --offload-arch=gfx950 -O3 -o gemm_loop_probe gemm_loop_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
constexpr int KM_ROW = 288;
constexpr int PL = 32 * KM_ROW;           // one plane (bf16 elements)
constexpr int STAGE = 4 * PL;             // spike plane + 3 dense planes
__device__ __forceinline__ u32x4 frag_tr(const unsigned short* S, int idx_base, int lane, int ks) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p4 = i & 3;
    const int col = idx_base + 16 * (g & 1) + 4 * p4;
    const int k0 = 16 * ks + 8 * (g >> 1);
    const unsigned short* a0 = S + (k0 + q) * KM_ROW + col;
    const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0)));
    const u32x2 hi = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * KM_ROW)));
    return u32x4{lo.x, lo.y, hi.x, hi.y};
}

enum { BAR = 1, WR = 2, VALU = 4, GLD = 8, DEEP = 16 };

extern __shared__ __attribute__((aligned(16))) unsigned short lds[];

template <int F, int NLD = 6>
__global__ __launch_bounds__(512, 1) void probe(int iters, const float* __restrict__ src, size_t src_elems,
                                                unsigned long long* out, float* sink) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * STAGE; i += 512) lds[i] = (unsigned short)(0x3F80 + (i & 3));
    __syncthreads();
    const int wm = wave >> 2, wn = wave & 3;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x4 rr[6], rr2[6];
    for (int q = 0; q < 6; ++q) rr2[q] = rr[q] = f32x4{1.f + q, 2.f, 3.f, 4.f + lane};
    const float* base = src + ((size_t)blockIdx.x * 512 + tid) * 4;
    const size_t stride = (size_t)gridDim.x * 512 * 4;  // floats per "piece sweep"
    size_t off = 0;
    unsigned long long t0 = 0, t1 = 0;
    auto body = [&](int it, f32x4 (&rr)[6]) __attribute__((always_inline)) {
        const unsigned short* cur = lds + (it & 1) * STAGE;
        unsigned short* nxt = lds + ((it + 1) & 1) * STAGE;
        const unsigned short* As = cur;
        const unsigned short* Bs = cur + PL;
        u32x4 fs[2][4], fd[2][1];
        auto rs = [&](int ks) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) fs[ks][i] = frag_tr(As, (wm * 4 + i) * 32, lane, ks);
        };
        auto rd = [&](int b, int ks, int p) __attribute__((always_inline)) {
            fd[b][0] = frag_tr(Bs + p * PL, wn * 32, lane, ks);
        };
        auto side = [&](int q) __attribute__((always_inline)) {
            f32x4 r = rr[q];
            if (F & VALU) {  // truncation split of 4 values into 3 planes (as store_piece<TRUNC>)
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
                if (F & WR) {
                    const int o = (tid >> 6) * KM_ROW + ((tid & 63) << 2) + q * 4 * KM_ROW;
                    *reinterpret_cast<u32x2*>(nxt + PL + o) = w1;
                    if (q < 2) {
                        *reinterpret_cast<u32x2*>(nxt + 2 * PL + o) = w2;
                        *reinterpret_cast<u32x2*>(nxt + 3 * PL + o) = w3;
                    }
                } else {
                    asm volatile("" ::"v"(w1), "v"(w2), "v"(w3));
                }
            } else if (F & WR) {
                const int o = (tid >> 6) * KM_ROW + ((tid & 63) << 2) + q * 4 * KM_ROW;
                u32x2 w = {__float_as_uint(r.x), __float_as_uint(r.y)};
                *reinterpret_cast<u32x2*>(nxt + PL + o) = w;
                if (q < 2) {
                    *reinterpret_cast<u32x2*>(nxt + 2 * PL + o) = w;
                    *reinterpret_cast<u32x2*>(nxt + 3 * PL + o) = w;
                }
            }
            if (F & GLD) {
                size_t o = off + (size_t)q * stride;
                if (o + stride > src_elems) o = (size_t)q * stride;
                rr[q] = *reinterpret_cast<const f32x4*>(base + o);
            }
        };
        rs(0); rd(0, 0, 2);
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            const int ks = g / 3, p = 2 - g % 3;
            if (p > 0) rd((g + 1) & 1, ks, p - 1);
            else if (ks == 0) { rs(1); rd((g + 1) & 1, 1, 2); }
            if (g < NLD) side(g);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = mfma(fs[ks][i], fd[g & 1][0], acc[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
        off += 6 * stride;
        if (off + 6 * stride > src_elems) off = 0;
        if (F & BAR) __syncthreads();
    };
    for (int it = -2; it < iters; it += 2) {
        if (it == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        body(it, rr);
        if (F & DEEP) body(it + 1, rr2); else body(it + 1, rr);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int q = 0; q < 6; ++q) s += rr[q].x + rr2[q].x;
    if (s == 12345.678f) sink[tid] = s;
}

template <int F, int NLD = 6>
void run(const char* name, const float* src, size_t n) {
    const int iters = 1000, grid = 256;
    unsigned long long* out; float* sink;
    hipMalloc(&out, sizeof(unsigned long long) * grid * 8);
    hipMalloc(&sink, 4096);
    const size_t lds_bytes = 2 * STAGE * sizeof(unsigned short);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<F, NLD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL((probe<F, NLD>), dim3(grid), dim3(512), lds_bytes, 0, iters, src, n, out, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 8);
    hipMemcpy(h.data(), out, sizeof(unsigned long long) * grid * 8, hipMemcpyDeviceToHost);
    double mean = 0, mx = 0;
    for (auto v : h) { mean += (double)v; if ((double)v > mx) mx = (double)v; }
    mean /= h.size();
    printf("%-34s: %.0f cycles per K tile (mean over waves), %.0f slowest; MFMA floor 1536\n", name, mean / iters, mx / iters);
    hipFree(out); hipFree(sink);
}

int main(int argc, char** argv) {
    // floats streamed per pass: default 2 GiB (HBM); pass a small size in MiB to keep the stream L2 / MALL resident
    const size_t mib = argc > 1 ? (size_t)atoi(argv[1]) : 2048;
    const size_t n = (mib << 20) / 4;
    printf("source buffer %zu MiB\n", mib);
    float* src;
    hipMalloc(&src, n * sizeof(float));
    hipMemset(src, 0x3c, n * sizeof(float));
    run<0>("frags+MFMA", src, n);
    run<BAR>("+barrier", src, n);
    run<BAR | WR>("+barrier+LDS writes", src, n);
    run<BAR | VALU>("+barrier+split VALU", src, n);
    run<BAR | WR | VALU>("+barrier+VALU+writes", src, n);
    run<BAR | GLD>("+barrier+global loads", src, n);
    run<GLD>("+global loads (no barrier)", src, n);
    run<BAR | WR | VALU | GLD>("everything", src, n);
    run<BAR | WR | VALU | GLD | DEEP>("everything, loads 2 tiles ahead", src, n);
    run<BAR | WR | VALU | GLD, 4>("everything, 4 pieces (bf16 spikes)", src, n);
    run<BAR | WR | VALU | GLD, 3>("everything, 3 pieces (u8 spikes)", src, n);
    hipFree(src);
    return 0;
}

// Diagnostic microbenchmark (not part of the library): how many cycles per v_mfma_f32_32x32x16_bf16 does a
// wave sustain when the MFMA operands come from LDS fragment reads issued one group ahead?
//   variant 0: MFMAs on loop-invariant registers (no LDS)            -> pipe rate
//   variant 1: 10 fragments (20 ds_read_b64_tr_b16) per 24 MFMAs, operands used by the MFMAs
//   variant 2: same reads issued and waited for, MFMAs on invariant registers
//   variant 3: 10 fragments as ds_read_b128 (KC image), operands used
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_lds_probe mfma_lds_probe.hip ; run: ./mfma_lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
constexpr int KM_ROW = 288;  // bf16 per k row of a 256-wide image (576 B)
__device__ __forceinline__ u32x4 frag_tr(const unsigned short* S, int idx_base, int lane, int ks) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p4 = i & 3;
    const int col = idx_base + 16 * (g & 1) + 4 * p4;
    const int k0 = 16 * ks + 8 * (g >> 1);
    const unsigned short* a0 = S + (k0 + q) * KM_ROW + col;
    const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0)));
    const u32x2 hi = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * KM_ROW)));
    return u32x4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ u32x4 frag_kc(const unsigned short* S, int idx_base, int lane, int ks) {
    const int r = lane & 31, h = lane >> 5;
    return *reinterpret_cast<const u32x4*>(S + (idx_base + r) * 40 + 16 * ks + 8 * h);
}

template <int VAR, int NWAVES>
__global__ __launch_bounds__(64 * NWAVES, 1) void probe(int iters, unsigned long long* out, float* sink) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4 * 32 * KM_ROW];  // 4 planes, 73.7 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4 * 32 * KM_ROW; i += 64 * NWAVES) lds[i] = (unsigned short)(0x3F80 + (i & 3));
    __syncthreads();
    const int wm = wave & 1, wn = (wave >> 1) & 1;
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const unsigned short* As = lds;
    const unsigned short* Bs = lds + 32 * KM_ROW;
    constexpr int PL = 32 * KM_ROW;
    u32x4 kc = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, (unsigned)lane};
    unsigned long long t0 = 0, t1 = 0;
    for (int it = -2; it < iters; ++it) {
        if (it == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        if (VAR == 0) {
#pragma unroll
            for (int g = 0; g < 6; ++g) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma(kc, kc, acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            u32x4 fs[2][4], fd[2][2];
            auto rs = [&](int ks) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    fs[ks][i] = VAR == 3 ? frag_kc(As, (wm * 4 + i) * 32, lane, ks) : frag_tr(As, (wm * 4 + i) * 32, lane, ks);
            };
            auto rd = [&](int b, int ks, int p) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    fd[b][j] = VAR == 3 ? frag_kc(Bs + p * PL, (wn * 2 + j) * 32, lane, ks) : frag_tr(Bs + p * PL, (wn * 2 + j) * 32, lane, ks);
            };
            rs(0); rd(0, 0, 2);
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                const int ks = g / 3, p = 2 - g % 3;
                if (p > 0) rd((g + 1) & 1, ks, p - 1);
                else if (ks == 0) { rs(1); rd((g + 1) & 1, 1, 2); }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (VAR == 2) {
                            asm volatile("" ::"v"(fs[ks][i]), "v"(fd[g & 1][j]));
                            acc[i][j] = mfma(kc, kc, acc[i][j]);
                        } else {
                            acc[i][j] = mfma(fs[ks][i], fd[g & 1][j], acc[i][j]);
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) out[blockIdx.x * NWAVES + wave] = t1 - t0;
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.678f) sink[tid] = s;
}

template <int VAR, int NW>
void run(const char* name, int grid) {
    const int iters = 2000;
    unsigned long long* out; float* sink;
    hipMalloc(&out, sizeof(unsigned long long) * grid * NW);
    hipMalloc(&sink, 4096);
    hipLaunchKernelGGL((probe<VAR, NW>), dim3(grid), dim3(64 * NW), 0, 0, iters, out, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * NW);
    hipMemcpy(h.data(), out, sizeof(unsigned long long) * grid * NW, hipMemcpyDeviceToHost);
    double mean = 0, mx = 0;
    for (auto v : h) { mean += (double)v; if ((double)v > mx) mx = (double)v; }
    mean /= h.size();
    printf("%-44s waves/WG %d grid %4d : %.1f cycles per MFMA per wave (mean), %.1f (slowest wave)\n", name, NW, grid,
           mean / iters / 48.0, mx / iters / 48.0);
    hipFree(out); hipFree(sink);
}

int main() {
    for (int grid : {1, 256}) {
        run<0, 4>("0: MFMA only", grid);
        run<1, 4>("1: tr_b16 frags one group ahead, used", grid);
        run<2, 4>("2: tr_b16 frags read+waited, MFMA independent", grid);
        run<3, 4>("3: b128 frags (80 B rows), used", grid);
        run<1, 8>("1: tr_b16, used, 2 waves/SIMD", grid);
        run<3, 8>("3: b128, used, 2 waves/SIMD", grid);
        run<0, 8>("0: MFMA only, 2 waves/SIMD", grid);
    }
    return 0;
}

// Shared device helpers of the persistent recurrent kernels (reccell.hip, gatedcell.hip): types, the hand-off
// ring constants, bare LDS barriers, exact-split MFMA wrapper, tile issue / settle of the sentinel protocol.
// See the header comment of reccell.hip for the design.
#pragma once
#include "common.h"

namespace {

typedef unsigned long long u64;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

constexpr int RT = 32;       // rows per batch tile
constexpr int CT = 32;       // columns per workgroup (= one k-group of its consumers)
constexpr int RED_LD = 33;   // padded row of the cross-wave reduction tiles (read with 4-byte accesses)
constexpr int RED_LD4 = 32;  // ... of the spiking cells' tiles, read with one 16-byte access per partial tile
#ifndef REC_RING
#define REC_RING 4
#endif
constexpr int RING = REC_RING;  // depth of the backward hand-off ring (2 suffices, see header)
constexpr int TILE_BYTES = RT * CT * 4;  // one fp32 hand-off tile
// Scope of the hand-off accesses.  Experiment hooks (diagnostic builds only); the shipped values are
// agent scope / sc1, the only combination that is correct for any placement of the workgroups.
#ifndef REC_LD_SCOPE
#define REC_LD_SCOPE __HIP_MEMORY_SCOPE_AGENT
#endif
#ifndef REC_ST_SCOPE
#define REC_ST_SCOPE __HIP_MEMORY_SCOPE_AGENT
#endif
#ifndef REC_LD_AUX
#define REC_LD_AUX 16 /* sc1 */
#endif
#ifndef REC_ST_AUX
#define REC_ST_AUX 16 /* sc1 */
#endif
#ifndef REC_XSTORE
#define REC_XSTORE 0  /* forward: bulk stores issued by the waves without pointwise state (measured: slower, see reccell.hip) */
#endif
#ifndef REC_BWD_XSTORE
#define REC_BWD_XSTORE 1  /* backward: bulk stores issued one step later by the waves without pointwise state */
#endif
#ifndef REC_BWD_LATE_PREFETCH
#define REC_BWD_LATE_PREFETCH 2  /* backward: where the next step's HBM inputs are requested — 0: loop top, in front of the tile loads (1.083 ms per launch); 1: behind the last tile load (1.057); 2: behind the reduction barrier, a pointwise phase and a publish ahead of the next tile loads (1.036; round 3, A/B in one call); 3: behind the publish barrier, with the rec-independent part of the reverse step moved behind the tile loop (1.097); 4: behind the publish stores, in front of the publish barrier (1.049 against 1.031 for 2 at that time) */
#endif
#ifndef REC_FWD_UPPER_SLEEP
#define REC_FWD_UPPER_SLEEP 0  /* forward: s_sleep units (64 cycles) of the waves without pointwise state before they poll again */
#endif
#ifndef REC_BWD_UPPER_RESET
#define REC_BWD_UPPER_RESET 1  /* backward: the ring's sentinels are put back by the waves without pointwise state */
#endif
#ifndef REC_BWD_PARK
#define REC_BWD_PARK 0  /* backward: the step's saved states wait in LDS instead of VGPRs (see reccell.hip; measured 1.056 vs 1.043 ms per launch, and a second k-group of tile loads in flight still spills: off) */
#endif
#ifndef REC_BWD_PROBE
#define REC_BWD_PROBE 0  /* backward: probe one dword per producer sample before issuing a step's tile loads (measured with one k-group ahead: 1.117 vs 1.040 ms per launch — the probe is one more round trip; with two or more ahead the kernel spills) */
#endif
#ifndef REC_BWD_UPPER_DELAY
#define REC_BWD_UPPER_DELAY 0   // s_sleep units (64 cycles) before the upper waves' first tile loads of a backward step.
                                // Measured (40 launches x 3, one call): 0 -> 1.029 ms per launch, 4 -> 1.040, 8 -> 1.049,
                                // 14 -> 1.074: the upper waves are the LATE ones at the reduction barrier, not early pollers
#endif
#ifndef REC_AHEAD
#define REC_AHEAD 1  /* k-groups whose tile loads are issued ahead of the one being multiplied */
#endif
constexpr u64 TIMEOUT_TICKS = 200000000ull;  // 2 s of s_memrealtime (100 MHz)
// "not written yet" pattern of the backward / dense hand-off ring: a SIGNALLING NaN.  Arithmetic results are
// never signalling (the hardware quiets every NaN it produces or propagates, IEEE mode), so no tile value
// can equal it.
constexpr unsigned SENTINEL = 0x7FA5A5A5u;


__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
// Non-temporal hints for what a time loop touches ONCE (round 3).  The loops stream their inputs (projection rows,
// incoming gradient, saved states) and outputs (saved states, spike planes, dWx) through the same L2 that holds what
// they re-read every step — the hand-off granules / ring tiles, the V pack — and at 2-4 MB per step per XCD the
// streams push those lines out: the backward ring (6 MB, rewritten every 4 steps) was written back to HBM on every
// pass (WRITE_SIZE 734 -> 490 MiB per launch with the hints: the ring now stays in L2), and the forward's polls and
// LUT / V reads missed.  Measured in one call, per launch in isolation: forward 0.636 -> 0.615 (loads) -> 0.597 ms
// (loads + stores), backward 1.024 -> 1.004 -> 1.003; in the cfg3 step (two layers): forward 1.195 -> 1.15 (loads),
// 1.09 (stores), 1.04 ms (both), step 6.58 -> 6.38 ms.  (The same hints on the split GEMMs' activation operands and
// C tiles LOSE: 6.49 -> 6.55-6.77 ms — a GEMM's output is the next kernel's input and should stay in the 256 MB
// infinity cache; `rec_bwd` right behind the dX product slowed down by 3-7 %.)
#ifndef REC_NT_LOADS
#define REC_NT_LOADS 1
#endif
__device__ __forceinline__ f32x4 ld4s(const float* p) {  // a streaming input: read once
#if REC_NT_LOADS
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
#else
    return *reinterpret_cast<const f32x4*>(p);
#endif
}
#ifndef REC_NT_STORES
#define REC_NT_STORES 1  /* bulk output stores of the recurrent kernels as non-temporal stores (see REC_NT_LOADS) */
#endif
__device__ __forceinline__ void st4(float* p, f32x4 v) {
#if REC_NT_STORES
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
#else
    *reinterpret_cast<f32x4*>(p) = v;
#endif
}
__device__ __forceinline__ void st2(void* p, u32x2 v) {  // an 8-byte bulk output (bf16 plane quads, bf16 saves)
#if REC_NT_STORES
    __builtin_nontemporal_store(v, reinterpret_cast<u32x2*>(p));
#else
    *reinterpret_cast<u32x2*>(p) = v;
#endif
}
// saved states (u, w) as fp32 or bf16 (element index i)
__device__ __forceinline__ f32x4 ld4_saved(const float* base, size_t i, bool s16) {
    if (!s16) return ld4(base + i);
    const unsigned long long raw = *reinterpret_cast<const unsigned long long*>(reinterpret_cast<const unsigned short*>(base) + i);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = bf16_to_f32((unsigned short)(raw >> (16 * e)));
    return v;
}
// ... the same load with the storage type fixed at compile time, the packed form kept as it comes off the wire
// (expanded at the use, so that nothing waits for the load where it is issued)
template <bool S16> struct SavedRaw { typedef f32x4 type; };
template <> struct SavedRaw<true> { typedef u32x2 type; };
template <bool S16>
__device__ __forceinline__ typename SavedRaw<S16>::type ld_saved_raw(const float* base, size_t i) {
    if constexpr (S16) return *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(base) + i);
    else return ld4s(base + i);
}
__device__ __forceinline__ f32x4 expand_saved(const f32x4& r) { return r; }
__device__ __forceinline__ f32x4 expand_saved(const u32x2& r) {
    f32x4 v;
    v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xFFFF0000u);
    v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xFFFF0000u);
    return v;
}
template <bool IS_U>
__device__ __forceinline__ void st4_saved(float* base, size_t i, f32x4 v, bool s16, float theta) {
    if (!s16) { st4(base + i, v); return; }
    unsigned long long raw = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        raw |= (unsigned long long)(IS_U ? save_u16(v[e], theta) : f32_to_bf16_rne(v[e])) << (16 * e);
    st2(reinterpret_cast<unsigned short*>(base) + i, u32x2{(unsigned)raw, (unsigned)(raw >> 32)});
}

// Workgroup barrier for LDS hand-offs inside the time loops.  __syncthreads() also carries workgroup-scope
// release / acquire fences on GLOBAL memory, i.e. an `s_waitcnt vmcnt(0)`: every wave would wait for the
// acknowledgement of its write-through hand-off stores (~2 k cycles) and of its bulk output stores at every
// barrier.  The loops only exchange LDS data across these barriers (cross-workgroup data goes through sc1
// accesses that need no fence), so: LDS operations retired, then the bare barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// `s_waitcnt vmcnt(0)` as the builtin (the compiler's wait-count pass sees it and clears its scoreboard, unlike
// an asm statement).  Placed where every outstanding vector-memory operation is long complete anyway, it keeps
// hipcc from inserting its own conservative vmcnt(0) at a later join — e.g. behind the hand-off stores, where
// it would wait ~2 k cycles for their write-through acknowledgements.
__device__ __forceinline__ void vm_settled() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// The workgroup's abort flags live in LDS and are read once per step behind a barrier.  Spelled as `volatile`
// accesses through a generic pointer they compile to FLAT loads / stores, which count on vmcnt AND lgkmcnt and
// complete out of order — hipcc then waits `vmcnt(0) lgkmcnt(0)` behind every flag read, i.e. behind every
// barrier of the time loops each wave sat out the acknowledgements of the previous step's bulk HBM stores and
// of its HBM prefetch (round 3, found in the ISA).  Through an LDS-address-space pointer they are ds_read /
// ds_write and touch lgkmcnt only.
typedef __attribute__((address_space(3))) int lds_i32;
typedef __attribute__((address_space(3))) void lds_void;        // LDS destination of an LDS-DMA load
typedef __attribute__((address_space(1))) const void g_void;    // ... and its global source
// LDS-DMA: 16 bytes per lane from the lane's own global address to lds_wave_base + 16 * lane (wave-uniform base).
// Inline asm: the load is then absent from hipcc's wait-count bookkeeping (its own counted waits only get
// stricter by that), and the caller guarantees a covering `s_waitcnt vmcnt` of the ISSUING wave before the data
// is read.  M0 (the destination base) is compiler-reserved: saved and restored inside the statement
// (cdna_hip_programming.md, inline-asm rules).
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lds_void*)lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
__device__ __forceinline__ int lds_flag_read(const int* p) { return *(const volatile lds_i32*)p; }
__device__ __forceinline__ void lds_flag_set(int* p) { *(volatile lds_i32*)p = 1; }
__device__ __forceinline__ void lds_flag_store(int* p, int v) { *(volatile lds_i32*)p = v; }

// The status word of a device is FOUR uint32 (include/sparch_hip.h): [0] raised, [1] optimizer steps skipped while it
// was raised (sparch_adam_step counts them), [2] id of the kernel that raised it (SPARCH_STATUS_*), [3] the time step
// (processing order) it gave up at.  Diagnostics first, the flag last.
__device__ __forceinline__ void status_raise(unsigned* status, unsigned kernel_id, int step) {
    __hip_atomic_store((gu32*)status + 2, kernel_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store((gu32*)status + 3, (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store((gu32*)status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void raise_timeout(unsigned* status, int* abort_slot, unsigned kernel_id, int step) {
    status_raise(status, kernel_id, step);
    lds_flag_set(abort_slot);
}

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                   c, 0, 0, 0);
}

// exact 3-way bf16 split of an fp32 value: x == hi + mid + lo (round-to-nearest-even at each step)
__device__ __forceinline__ void split3(float x, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    const __bf16 l = (__bf16)r2;
    hi = __builtin_bit_cast(unsigned short, h);
    mid = __builtin_bit_cast(unsigned short, m);
    lo = __builtin_bit_cast(unsigned short, l);
}

// Packing a recurrent matrix.  rne (the bf16 operand mode, sparch_set_operand_precision): plane 0 = the value
// rounded once to bf16, planes 1 and 2 = 0 — the kernels of that mode read plane 0 only, and any three-plane
// kernel given such a pack computes products with the same rounded weights.
__device__ __forceinline__ void vsplit(float v, int rne, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
    split3(v, hi, mid, lo);
    if (rne) { mid = 0; lo = 0; }
}

// V (or V^T) slice -> registers: per k-group, 2 k16-steps x NP planes of 8 bf16 (4 VGPRs) each (the packed
// layout always has room for three planes; the bf16 operand mode keeps its one rounded plane in plane 0)
template <int KGW, int NW, int NP = 3>
__device__ __forceinline__ void load_vslice(u32x4 (&vb)[KGW][2][NP], const u32x4* __restrict__ vpack, int ct,
                                            int nkg, int wave, int lane) {
#pragma unroll
    for (int kk = 0; kk < KGW; ++kk) {
        const int kg = wave + NW * kk;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int p = 0; p < NP; ++p)
                vb[kk][ks][p] = vpack[((((size_t)ct * nkg + kg) * 2 + ks) * 3 + p) * 64 + lane];
    }
}


// Load this wave's k-groups of a row tile's fp32 hand-off tiles (fragment order) from ring slot `base`,
// re-loading every 16-byte piece that still holds the sentinel until all have landed (bounded spin).
// lane (row li, k-half hh) of k16-step ks needs k = 16*ks + 8*hh + 4q + 0..3 of producer tile kg:
// piece (ks*2+q)*64 + lane of that tile -> each wave-load is 1 KiB contiguous.
__device__ __forceinline__ bool piece_missing(const u32x4& v) { return v[0] == SENTINEL || v[3] == SENTINEL; }

// issue the four 1 KiB wave-loads of ONE k-group
template <int NW>
__device__ __forceinline__ void issue_tile(u32x4 (&g)[2][2], __amdgpu_buffer_rsrc_t rsrc, unsigned base, int kg,
                                           int n_ct) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            // padding k-groups (beyond H) read past the end of the buffer resource: the hardware returns
            // zeros for out-of-range buffer loads — no branch, and zeros are never "missing"
            const unsigned off = kg < n_ct ? base + (unsigned)kg * TILE_BYTES + (unsigned)((ks * 2 + q) * 1024)
                                           : 0xFFFFFF00u;
            g[ks][q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, REC_LD_AUX);
        }
}

// the four pieces of ONE k-group: wait for them (the others stay in flight), re-load what still reads as the
// sentinel until it has landed (bounded spin).  Fast path: four compares and one wave-uniform branch.
__device__ __forceinline__ void settle_tile(u32x4 (&g)[2][2], __amdgpu_buffer_rsrc_t rsrc, unsigned tile_base,
                                            int* abort_slot) {
    unsigned miss = 0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int q = 0; q < 2; ++q) miss |= piece_missing(g[ks][q]) ? (1u << (ks * 2 + q)) : 0u;
    if (__any(miss != 0)) {
        // slow path.  Re-loads go to temporaries and are waited for right here (builtin wait: the compiler's
        // scoreboard stays exact), then merged by select: the pending loads of the later k-groups are not
        // touched, so the fast path keeps its precise vmcnt(N) waits after the join.
        const u64 t_start = __builtin_amdgcn_s_memrealtime();
        for (unsigned spins = 0;; ++spins) {
            __builtin_amdgcn_s_sleep(1);
            u32x4 tmp[2][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    tmp[ks][q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, tile_base + (unsigned)((ks * 2 + q) * 1024),
                                                                      0, REC_LD_AUX);
            vm_settled();
            unsigned still = 0;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const unsigned bit = 1u << (ks * 2 + q);
                    const bool m = (miss & bit) != 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[ks][q][e] = m ? tmp[ks][q][e] : g[ks][q][e];
                    if (m && piece_missing(tmp[ks][q])) still |= bit;
                }
            miss = still;
            if (!__any(miss != 0)) break;
            if ((spins & 63u) == 63u && __builtin_amdgcn_s_memrealtime() - t_start > TIMEOUT_TICKS) {
                lds_flag_set(abort_slot);  // the status word is raised at the kernel's exit
                break;
            }
        }
    }
}

// ---- hand-off tiles as PRE-SPLIT bf16 planes (round 2, with the XCD-local stores).  The producer
// splits its dWx values once (x = t1 + t2 + t3, truncation split) and publishes the three planes; a consumer
// loads MFMA fragments and nothing else: 6 KB per tile instead of 4, no split VALU in the 32 consumers (each of
// which used to split the same 32 x 1024 values: 2.8 k VALU cycles per SIMD and step).  Tile layout: 16-byte
// piece ((ks*3 + p)*64 + h*32 + row) = plane p, k = 16 ks + 8 h + 0..7 of that row — the wave-load of one
// (k16-step, plane) is one contiguous 1 KiB.  A producer thread owns 4 consecutive k: half a piece, written
// with 8-byte stores (first / last word of a piece sit in different halves: the sentinel check covers both).
// No plane word can equal the sentinel: its upper half would be a signalling-NaN bf16, and every plane value
// is the upper half of an arithmetic fp32 result.
// NP planes per tile: 3 = the exact split, 1 = the bf16 operand mode (one nearest-even rounding by the producer,
// 2 KB per tile; the same layout with NP in place of 3).  A rounded plane word cannot equal the sentinel either:
// v_cvt_pk_bf16_f32 quiets every NaN it converts.
constexpr int PTILE_BYTES = RT * CT * 6;  // what the host sizes the ring for (three planes)
template <int NP> constexpr int ptile_bytes() { return RT * CT * 2 * NP; }
template <int NW, int NP = 3>
__device__ __forceinline__ void issue_ptile(u32x4 (&g)[2][NP], __amdgpu_buffer_rsrc_t rsrc, unsigned base, int kg, int n_ct) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const unsigned off = kg < n_ct ? base + (unsigned)kg * ptile_bytes<NP>() + (unsigned)((ks * NP + p) * 1024) : 0xFFFFFF00u;
            g[ks][p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, REC_LD_AUX);
        }
}
template <int NP = 3>
__device__ __forceinline__ void settle_ptile(u32x4 (&g)[2][NP], __amdgpu_buffer_rsrc_t rsrc, unsigned tile_base, int* abort_slot) {
    unsigned miss = 0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int p = 0; p < NP; ++p) miss |= piece_missing(g[ks][p]) ? (1u << (ks * NP + p)) : 0u;
    if (__any(miss != 0)) {  // slow path as settle_tile: temporaries, builtin wait, merge by select
        const u64 t_start = __builtin_amdgcn_s_memrealtime();
        for (unsigned spins = 0;; ++spins) {
            __builtin_amdgcn_s_sleep(1);
            u32x4 tmp[2][NP];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    tmp[ks][p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, tile_base + (unsigned)((ks * NP + p) * 1024), 0, REC_LD_AUX);
            vm_settled();
            unsigned still = 0;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const unsigned bit = 1u << (ks * NP + p);
                    const bool m = (miss & bit) != 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[ks][p][e] = m ? tmp[ks][p][e] : g[ks][p][e];
                    if (m && piece_missing(tmp[ks][p])) still |= bit;
                }
            miss = still;
            if (!__any(miss != 0)) break;
            if ((spins & 63u) == 63u && __builtin_amdgcn_s_memrealtime() - t_start > TIMEOUT_TICKS) {
                lds_flag_set(abort_slot);
                break;
            }
        }
    }
}

// exact truncation split of a thread's 4 fp32 values into three words-pairs of bf16 (plane p: w[p] = 4 bf16)
__device__ __forceinline__ void split4_planes(const f32x4& v, u32x2 (&w)[3]) {
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        const unsigned x0 = __float_as_uint(v[2 * pr]), x1 = __float_as_uint(v[2 * pr + 1]);
        const float r0 = v[2 * pr] - __uint_as_float(x0 & 0xFFFF0000u);
        const float r1 = v[2 * pr + 1] - __uint_as_float(x1 & 0xFFFF0000u);
        const unsigned y0 = __float_as_uint(r0), y1 = __float_as_uint(r1);
        const float q0 = r0 - __uint_as_float(y0 & 0xFFFF0000u);
        const float q1 = r1 - __uint_as_float(y1 & 0xFFFF0000u);
        w[0][pr] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
        w[1][pr] = __builtin_amdgcn_perm(y1, y0, 0x07060302u);
        w[2][pr] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
    }
}
// one plane-tile half piece per plane (8 bytes each, 1 KiB apart) at byte offset `off`; `plain`: XCD-local stores
__device__ __forceinline__ void store_planes(const u32x2 (&w)[3], __amdgpu_buffer_rsrc_t rsrc, unsigned off, bool plain) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        if (plain) __builtin_amdgcn_raw_buffer_store_b64(w[p], rsrc, off + (unsigned)(p * 1024), 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b64(w[p], rsrc, off + (unsigned)(p * 1024), 0, REC_ST_AUX);
    }
}

// exact truncation split of one k-group's tile (2 k16-steps x 8 fp32 values per lane) into the three bf16
// fragments per k16-step: P[ks][0..2] = t1, t2, t3 with x = t1 + t2 + t3 (v_perm for the packing, AND + SUB for
// the residuals: ~5.5 VALU instructions per value)
__device__ __forceinline__ void split_tile(const u32x4 (&g)[2][2], u32x4 (&P)[2][3]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const unsigned x0 = g[ks][q][2 * pr], x1 = g[ks][q][2 * pr + 1];
                const float r0 = __uint_as_float(x0) - __uint_as_float(x0 & 0xFFFF0000u);
                const float r1 = __uint_as_float(x1) - __uint_as_float(x1 & 0xFFFF0000u);
                const unsigned y0 = __float_as_uint(r0), y1 = __float_as_uint(r1);
                const float q0 = r0 - __uint_as_float(y0 & 0xFFFF0000u);
                const float q1 = r1 - __uint_as_float(y1 & 0xFFFF0000u);
                P[ks][0][2 * q + pr] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
                P[ks][1][2 * q + pr] = __builtin_amdgcn_perm(y1, y0, 0x07060302u);
                P[ks][2][2 * q + pr] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
            }
}

// ---- XCD-local hand-off stores (speed option, VERIFIED at run time).
// Agent-scope (sc1) stores are write-through and drop the line from the XCD's L2: a consumer on the same XCD
// then fetches it at the cross-XCD rate and latency.  A plain store keeps the line in that L2, where an sc1
// load of any CU of the XCD finds it (sc1 loads bypass L1 only) — but it is invisible to the other XCDs.
// Placement is not part of HIP's contract, so nothing may ASSUME that a row tile's workgroups share an XCD:
// they establish it.  Every workgroup publishes the XCC id of the XCD it runs on (HW_REG_XCC_ID) with an
// agent-scope store into a table the host cleared, reads the entries of all workgroups of its row tile with
// agent-scope loads (bounded spin), and only if all are there and equal do these workgroups — all of which
// see the same entries — use plain stores among themselves.  Anything else (different XCDs, a workgroup
// that never arrives, chunked launches, an absent table) keeps the sc1 stores.  A wave that finds itself
// on another XCD later (queue preemption with save / restore) raises the launch's abort flag: the step is
// discarded and the host goes to per-step launches, exactly as after a timeout.
constexpr int GETREG_XCC_ID = (3 << 11) | 20;  // hwreg(HW_REG_XCC_ID, 0, 4)
__device__ __forceinline__ unsigned xcc_id() { return (unsigned)__builtin_amdgcn_s_getreg(GETREG_XCC_ID) & 0xFu; }

// tab: this row tile's n_ct (<= 64) words, all `empty` before the launch.  Called by every thread of the
// workgroup BEFORE a __syncthreads() the caller already has; the result is read from *lds_flag after it.
__device__ __forceinline__ void xcd_agree(gu32* tab, unsigned empty, int n_ct, int ct, unsigned my_xcc, int tid,
                                          int* lds_flag) {
    if (tid >= 64) return;
    bool ok = tab != nullptr && n_ct <= 64;
    if (ok) {
        if (tid == 0) __hip_atomic_store(tab + ct, my_xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < n_ct) {
            const u64 t_start = __builtin_amdgcn_s_memrealtime();
            unsigned v = empty;
            for (unsigned spins = 0;; ++spins) {
                v = __hip_atomic_load(tab + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v != empty) break;
                __builtin_amdgcn_s_sleep(2);
                if ((spins & 63u) == 63u && __builtin_amdgcn_s_memrealtime() - t_start > TIMEOUT_TICKS / 100) break;  // 20 ms
            }
            ok = v == my_xcc + 1u;
        }
    }
    const bool all = __all(ok);
    if (tid == 0) *lds_flag = all ? 1 : 0;
}

// Host side: can `grid` workgroups of this kernel be resident at once?  (The persistent launches wait for each
// other inside the kernel.)  Occupancy as the runtime computes it for this kernel's registers / LDS, times the CU
// count; cached per kernel.  The in-kernel spins are bounded anyway — this keeps a foreseeable miss (fewer CUs
// than assumed, a build with more registers) from costing a 2 s timeout before the per-step fallback takes over.
template <auto Kernel>
bool grid_is_co_resident(unsigned grid, int threads, int cus) {
    static const int per_cu = [threads] {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, Kernel, threads, 0) != hipSuccess) n = 1;
        (void)hipGetLastError();
        return n;
    }();
    return (long long)per_cu * cus >= (long long)grid;
}

}  // namespace

// f-4: LiGRU baseline cell (LiGRULayer._ligru_cell, anns.py:449-462) as persistent kernels — the recurrent spiking
// backward's machine (reccell.hip: register-resident slice of the recurrent matrices as exact bf16 planes, the
// previous step's dense fp32 tiles handed over through the sentinel ring and split exactly by the consumer) with
// TWO recurrent matrices per cell:
//     z_t = sigmoid(xz_t + y_{t-1} Vz^T)      c_t = relu(xc_t + y_{t-1} V^T)      y_t = z_t y_{t-1} + (1 - z_t) c_t
// and the reverse pass
//     dy_t  = dropout'(g_t) + [dz_pre | dc_pre]_{t+1} [Vz ; V] + dy_{t+1} z_{t+1}
//     dz_pre = dy (y_{t-1} - c) z (1 - z)      dc_pre = dy (1 - z) [c > 0].
// Both matrices do not fit a 32-column slice each, so a workgroup owns 16 hidden units (64 column tiles at
// H = 1024; 4 row tiles per persistent launch, the row-tile groups run one after the other):
//   forward : the 32 MFMA columns of a workgroup are [z of its 16 units | c of its 16 units]; K = H.  Its y tile
//             is 32 rows x 16 units = HALF of a consumer k-group: producer ct fills k16-step ct & 1 of ring tile
//             ct >> 1 (same fragment order, same sentinel protocol);
//   backward: the product contracts over K = 2H — the stacked [dz_pre | dc_pre] of ALL units — into the
//             workgroup's 16 units: v_mfma_f32_16x16x32_bf16 (N = 16: nothing of the matrix pipe is spent on
//             padding columns), two 16-row blocks per workgroup; a producer's hand-off tile is its 32 rows x
//             (16 dz_pre + 16 dc_pre) = one 32-deep k step, stored in THAT MFMA's A-fragment order.  256 KiB of
//             tiles per workgroup and step go through the CU's L2 port (~7.4 k cycles at 70 GB/s): the bound
//             of this kernel, as the 128 KiB are of the spiking backward.
// GRU (GRULayer._gru_cell, anns.py:581-595) has three matrices and its reset gate inside the candidate's
// recurrent term,
//     z = sigmoid(xz + y Vz^T)   r = sigmoid(xr + y Vr^T)   c = tanh(xc + (r y) V^T)   y' = z y + (1 - z) c,
// so a step has TWO hand-offs: the same 16-units-per-workgroup machine runs the [z | r] product on the
// forward layout above, publishes q = r y (a second ring, tiles in the 16x16x32 A-fragment order), runs the
// candidate's product on v_mfma_f32_16x16x32_bf16 (16 columns) and publishes y'.  Backward likewise:
//     dy = g + [dz_pre | dr_pre]_{t+1} [Vz ; Vr] + (dq r + dy z)_{t+1}
//     dz_pre = dy (y - c) z (1 - z)     dc_pre = dy (1 - z) (1 - c^2)        -> publish dc_pre
//     dq = dc_pre V                     dr_pre = dq y r (1 - r)              -> publish [dz_pre | dr_pre]
// (K = H then K = 2H, both on the 16x16x32 MFMA).  The two hand-offs of a step need every workgroup of a row
// tile resident at once at ANY steps_per_launch, so there is no per-step degenerate form: where the grid
// cannot be co-resident (or after a timeout) the host takes the launch-per-step path of annstep.hip.
#include "rec_common.h"

#include <type_traits>

namespace {

constexpr int UT = 16;  // hidden units per workgroup

struct LigruArgs {
    int B, dirs, T, H, Bp;
    int n_ct;                 // workgroup column tiles = H / 16
    int n_kg;                 // forward: k-groups of 32 y values = H / 32
    int n_rt_total, rt_base, n_rt_launch;
    int s_begin, s_end;       // steps in processing order (forward t = s, backward t = T-1-s)
    const float* Wx; const float* sc; const float* sh;      // candidate projection (B,T,H) + folded BatchNorm
    const float* Wzx; const float* scz; const float* shz;   // update-gate projection
    const u32x4* vpack;       // B-operand fragments of the workgroup's slice (forward / backward layouts below)
    float p_drop, inv_keep; uint64_t seed;
    float* y_state; float* z_save; float* c_save; float* y_out;          // forward outputs
    const float* g_out;                                                  // backward input (B,T,H*dirs)
    float* dz_all; float* dc_all; float* yprev_all;                      // backward outputs (Bp,T,H), original time
    float* carry;             // (Bp,H) dy_t z_t between chunked launches
    char* ring; unsigned* status;
    // GRU: reset-gate projection, its saved activations, the candidate matrix' fragments and the second ring
    const float* Wrx; const float* scr; const float* shr;
    float* r_save; float* dr_all; float* ry_all;
    const u32x4* vpack2; char* ring2;
};

__device__ __forceinline__ float sigm(float v) { return 1.0f / (1.0f + expf(-v)); }
__device__ __forceinline__ f32x4 affine4(const float* W, const float* sc, const float* sh, size_t o, int h) {
    f32x4 v = ld4(W + o);
    if (sc) {
        const f32x4 s = ld4(sc + h), b = ld4(sh + h);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = bn_affine(v[e], s[e], b[e]);
    }
    return v;
}

// exact truncation split of 8 fp32 values (two 16-byte pieces) into three bf16 fragments
__device__ __forceinline__ void split_pieces(const u32x4& lo4, const u32x4& hi4, u32x4& p1, u32x4& p2, u32x4& p3) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const unsigned x0 = (q ? hi4 : lo4)[2 * pr], x1 = (q ? hi4 : lo4)[2 * pr + 1];
            const float r0 = __uint_as_float(x0) - __uint_as_float(x0 & 0xFFFF0000u);
            const float r1 = __uint_as_float(x1) - __uint_as_float(x1 & 0xFFFF0000u);
            const unsigned y0 = __float_as_uint(r0), y1 = __float_as_uint(r1);
            const float q0 = r0 - __uint_as_float(y0 & 0xFFFF0000u);
            const float q1 = r1 - __uint_as_float(y1 & 0xFFFF0000u);
            p1[2 * q + pr] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
            p2[2 * q + pr] = __builtin_amdgcn_perm(y1, y0, 0x07060302u);
            p3[2 * q + pr] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
        }
}

// ------------------------------------------------------------------------------ forward
// vpack[ct][kg][ks][p][lane] = 8 bf16 of plane p: rows k = kg*32 + 16*ks + 8*(lane>>5) + j of column
// n = lane & 31 of the slice, column n < 16 = Vz[ct*16 + n][k], n >= 16 = V[ct*16 + n - 16][k]  (y V^T).
template <int KGW, int NW>
__global__ __launch_bounds__(64 * NW, 1) void ligru_fwd_kernel(LigruArgs a) {
    __shared__ __attribute__((aligned(16))) float red[NW][RT * RED_LD];
    __shared__ __attribute__((aligned(16))) u32x4 vlo[NW][KGW][2][64];
    __shared__ int abort_flag[2];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hh = lane >> 5;
    const int rt = a.rt_base + (int)(blockIdx.x % a.n_rt_launch);
    const int ct = (int)(blockIdx.x / a.n_rt_launch);
    const int T = a.T, H = a.H, HO = a.H * a.dirs;

    // pointwise ownership: 128 threads, row r, 4 consecutive units
    const bool pw = tid < 128;
    const int r = (tid & 127) >> 2, uq = tid & 3;
    const int bp = rt * RT + r, unit = ct * UT + uq * 4;
    const bool valid = pw && bp < a.Bp && unit < H;
    const int bpc = min(bp, a.Bp - 1), uc = min(unit, H - 4);
    const int d = bpc / a.B, b = bpc - d * a.B;

    u32x4 vb[KGW][2][2];
#pragma unroll
    for (int kk = 0; kk < KGW; ++kk) {
        const int kg = wave + NW * kk;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const u32x4* src = a.vpack + ((((size_t)ct * (NW * KGW) + kg) * 2 + ks) * 3) * 64 + lane;
            vb[kk][ks][0] = src[0];
            vb[kk][ks][1] = src[64];
            vlo[wave][kk][ks][lane] = src[128];
        }
    }
    if (tid < 2) abort_flag[tid] = 0;
    __syncthreads();

    const unsigned slot_bytes = (unsigned)((size_t)a.n_rt_total * a.n_kg * TILE_BYTES);
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(a.ring, 0, (int)(RING * slot_bytes), 0x00020000);
    const unsigned rt_off = (unsigned)((size_t)rt * a.n_kg * TILE_BYTES);
    const bool drop = a.p_drop > 0.0f;
    const uint64_t seed = drop ? resolve_seed(a.seed) : 0;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    f32x4 yp = zero4;  // y_{t-1} of this thread's 4 units (zeros at t = 0, anns.py:452)
    if (pw && a.s_begin > 0) yp = ld4(a.y_state + ((size_t)bpc * T + (a.s_begin - 1)) * H + uc);
    auto load_x = [&](int t, f32x4& xz, f32x4& xc) {
        const int tt = d ? (T - 1 - t) : t;
        const size_t o = ((size_t)b * T + tt) * H + uc;
        xz = affine4(a.Wzx, a.scz, a.shz, o, uc);
        xc = affine4(a.Wx, a.sc, a.sh, o, uc);
    };
    f32x4 xz_n = zero4, xc_n = zero4;
    if (pw) load_x(a.s_begin, xz_n, xc_n);

    for (int s = a.s_begin; s < a.s_end; ++s) {
        const f32x4 xz = xz_n, xc = xc_n;
        const int par = s & 1;
        if (pw && s + 1 < a.s_end) load_x(s + 1, xz_n, xc_n);
        float rz[4] = {0.f, 0.f, 0.f, 0.f}, rc[4] = {0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
            const unsigned base = (unsigned)((s - 1) % RING) * slot_bytes + rt_off + (unsigned)lane * 16u;
            constexpr int AHEAD = KGW < REC_AHEAD ? KGW : REC_AHEAD;
            u32x4 raw[KGW][2][2];
#pragma unroll
            for (int kk = 0; kk < AHEAD; ++kk) issue_tile<NW>(raw[kk], rsrc, base, wave + NW * kk, a.n_kg);
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int kk = 0; kk < KGW; ++kk) {
                __builtin_amdgcn_sched_barrier(0);
                if (wave + NW * kk < a.n_kg)
                    settle_tile(raw[kk], rsrc, base + (unsigned)(wave + NW * kk) * TILE_BYTES, &abort_flag[par]);
                if (kk + AHEAD < KGW) issue_tile<NW>(raw[kk + AHEAD], rsrc, base, wave + NW * (kk + AHEAD), a.n_kg);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    u32x4 p1, p2, p3;
                    split_pieces(raw[kk][ks][0], raw[kk][ks][1], p1, p2, p3);
                    const u32x4 vl = vlo[wave][kk][ks][lane];
                    acc = mfma_bf16(p2, vb[kk][ks][1], acc);  // t2*mid
                    acc = mfma_bf16(p3, vb[kk][ks][0], acc);  // t3*hi
                    acc = mfma_bf16(p1, vl, acc);             // t1*lo
                    acc = mfma_bf16(p2, vb[kk][ks][0], acc);  // t2*hi
                    acc = mfma_bf16(p1, vb[kk][ks][1], acc);  // t1*mid
                    acc = mfma_bf16(p1, vb[kk][ks][0], acc);  // t1*hi
                }
            }
            float* rd = red[wave];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
                rd[row * RED_LD + li] = acc[i];
            }
        }
        lds_barrier();
        vm_settled();
        if (lds_flag_read(&abort_flag[par])) break;
        if (s > 0 && pw) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float sz = red[0][r * RED_LD + uq * 4 + e], sc_ = red[0][r * RED_LD + UT + uq * 4 + e];
#pragma unroll
                for (int w = 1; w < NW; ++w) {
                    sz = sz + red[w][r * RED_LD + uq * 4 + e];
                    sc_ = sc_ + red[w][r * RED_LD + UT + uq * 4 + e];
                }
                rz[e] = sz; rc[e] = sc_;
            }
        }
        // ---- gates (anns.py:457-459)
        const int t = s;
        const int tt = d ? (T - 1 - t) : t;
        const size_t o_out = ((size_t)b * T + tt) * HO + (size_t)d * H + uc;
        f32x4 z, c, y, yo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            z[e] = sigm(xz[e] + rz[e]);
            c[e] = fmaxf(xc[e] + rc[e], 0.0f);
            y[e] = z[e] * yp[e] + (1.0f - z[e]) * c[e];
            const float k = drop ? keep_scale(seed, o_out + e, a.p_drop, a.inv_keep) : 1.0f;
            yo[e] = y[e] * k;
            if (!valid) y[e] = 0.0f;
        }
        // ---- publish y_t: this thread's 4 units are one 16-byte piece of k16-step (ct & 1) of ring tile ct >> 1
        if (pw) {
            const int ksp = ct & 1, hq = uq >> 1, qq = uq & 1;
            const unsigned piece = (unsigned)((((ksp * 2 + qq) * 64) + hq * 32 + r) * 16);
            const unsigned tile_off = rt_off + (unsigned)(ct >> 1) * TILE_BYTES + piece;
            if (s + 1 < T) {
                u32x4 rawv;
#pragma unroll
                for (int e = 0; e < 4; ++e) rawv[e] = __float_as_uint(y[e]);
                __builtin_amdgcn_raw_buffer_store_b128(rawv, rsrc, (unsigned)(s % RING) * slot_bytes + tile_off, 0, REC_ST_AUX);
            }
            if (s >= 2) {
                const u32x4 sent = {SENTINEL, SENTINEL, SENTINEL, SENTINEL};
                __builtin_amdgcn_raw_buffer_store_b128(sent, rsrc, (unsigned)((s - 2) % RING) * slot_bytes + tile_off, 0, REC_ST_AUX);
            }
        }
        lds_barrier();
        if (valid) {
            const size_t o_st = ((size_t)bp * T + t) * H + unit;
            st4(a.y_state + o_st, y); st4(a.z_save + o_st, z); st4(a.c_save + o_st, c);
            st4(a.y_out + ((size_t)b * T + tt) * HO + (size_t)d * H + unit, yo);
        }
        yp = y;
    }
    if (tid == 0 && (lds_flag_read(&abort_flag[0]) | lds_flag_read(&abort_flag[1])))
        status_raise(a.status, SPARCH_STATUS_LIGRU_FWD, -1);
}

// ------------------------------------------------------------------------------ backward
// v_mfma_f32_16x16x32_bf16: A fragment = lane (row lane & 15, k-quarter lane >> 4) holds k = 8*(lane>>4) .. +7;
// B fragment = lane (column lane & 15, same k-quarter); C = 4 rows (4*(lane>>4) .. +3) of column lane & 15.
// A producer's tile: 32 rows x 32 k, k < 16 = dz_pre of its unit k, k >= 16 = dc_pre of unit k - 16, stored as
// 16-byte pieces ((mb*2 + half)*64 + kq*16 + row16): row = 16*mb + row16, k = 8*kq + 4*half .. +3 — the four
// 1 KiB wave-loads of a tile are (mb, half) = (0,0), (0,1), (1,0), (1,1).
// vpack[ct][kg][p][lane] = 8 bf16 of plane p: k = 8*(lane>>4) + j of k-group kg (= producer tile kg), column
// lane & 15 = unit ct*16 + (lane & 15):  k < 16: Vz[kg*16 + k][unit],  k >= 16: V[kg*16 + k - 16][unit].
__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
constexpr int RED16 = 17;

template <int KGW, int NW>
__global__ __launch_bounds__(64 * NW, 1) void ligru_bwd_kernel(LigruArgs a) {
    __shared__ __attribute__((aligned(16))) float red[NW][RT * RED16];
    __shared__ __attribute__((aligned(16))) u32x4 vlo[NW][KGW][64];
    __shared__ int abort_flag[2];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rt = a.rt_base + (int)(blockIdx.x % a.n_rt_launch);
    const int ct = (int)(blockIdx.x / a.n_rt_launch);
    const int T = a.T, H = a.H, HO = a.H * a.dirs;

    const bool pw = tid < 128;
    const int r = (tid & 127) >> 2, uq = tid & 3;
    const int bp = rt * RT + r, unit = ct * UT + uq * 4;
    const bool valid = pw && bp < a.Bp && unit < H;
    const int bpc = min(bp, a.Bp - 1), uc = min(unit, H - 4);
    const int d = bpc / a.B, b = bpc - d * a.B;

    u32x4 vb[KGW][2];
#pragma unroll
    for (int kk = 0; kk < KGW; ++kk) {
        const int kg = wave + NW * kk;
        const u32x4* src = a.vpack + (((size_t)ct * (NW * KGW) + kg) * 3) * 64 + lane;
        vb[kk][0] = src[0];
        vb[kk][1] = src[64];
        vlo[wave][kk][lane] = src[128];
    }
    if (tid < 2) abort_flag[tid] = 0;
    __syncthreads();

    const unsigned slot_bytes = (unsigned)((size_t)a.n_rt_total * a.n_ct * TILE_BYTES);
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(a.ring, 0, (int)(RING * slot_bytes), 0x00020000);
    const unsigned rt_off = (unsigned)((size_t)rt * a.n_ct * TILE_BYTES);
    const bool drop = a.p_drop > 0.0f;
    const uint64_t seed = drop ? resolve_seed(a.seed) : 0;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    f32x4 cdir = zero4;  // dy_{t+1} z_{t+1}
    if (pw && a.s_begin > 0) cdir = ld4(a.carry + (size_t)bpc * H + uc);
    auto load_step = [&](int s, f32x4& g, f32x4& z, f32x4& c, f32x4& ypv) {
        const int t = T - 1 - s;
        const int tt = d ? (T - 1 - t) : t;
        g = ld4(a.g_out + ((size_t)b * T + tt) * HO + (size_t)d * H + uc);
        const size_t o_st = ((size_t)bpc * T + t) * H + uc;
        z = ld4(a.z_save + o_st); c = ld4(a.c_save + o_st);
        ypv = t > 0 ? ld4(a.y_state + o_st - H) : zero4;
    };
    f32x4 g_n = zero4, z_n = zero4, c_n = zero4, yp_n = zero4;
    if (pw) load_step(a.s_begin, g_n, z_n, c_n, yp_n);

    for (int s = a.s_begin; s < a.s_end; ++s) {
        const f32x4 gv = g_n, zv = z_n, cv = c_n, ypv = yp_n;
        const int par = s & 1;
        if (pw && s + 1 < a.s_end) load_step(s + 1, g_n, z_n, c_n, yp_n);
        float cmv[4] = {0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
            const unsigned base = (unsigned)((s - 1) % RING) * slot_bytes + rt_off + (unsigned)lane * 16u;
            u32x4 raw[2][2][2];  // [parity of kk][mb][half]: one tile ahead
            auto issue = [&](int kk) {
                const int kg = wave + NW * kk;
                issue_tile<NW>(raw[kk & 1], rsrc, base, kg, a.n_ct);  // pieces (mb*2 + half)*1024: same offsets
            };
            issue(0);
            f32x4 acc[2] = {zero4, zero4};
#pragma unroll
            for (int kk = 0; kk < KGW; ++kk) {
                __builtin_amdgcn_sched_barrier(0);
                if (wave + NW * kk < a.n_ct)
                    settle_tile(raw[kk & 1], rsrc, base + (unsigned)(wave + NW * kk) * TILE_BYTES, &abort_flag[par]);
                if (kk + 1 < KGW) issue(kk + 1);
                __builtin_amdgcn_sched_barrier(0);
                const u32x4 vl = vlo[wave][kk][lane];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    u32x4 p1, p2, p3;
                    split_pieces(raw[kk & 1][mb][0], raw[kk & 1][mb][1], p1, p2, p3);
                    acc[mb] = mfma16(p2, vb[kk][1], acc[mb]);  // t2*mid
                    acc[mb] = mfma16(p3, vb[kk][0], acc[mb]);  // t3*hi
                    acc[mb] = mfma16(p1, vl, acc[mb]);         // t1*lo
                    acc[mb] = mfma16(p2, vb[kk][0], acc[mb]);  // t2*hi
                    acc[mb] = mfma16(p1, vb[kk][1], acc[mb]);  // t1*mid
                    acc[mb] = mfma16(p1, vb[kk][0], acc[mb]);  // t1*hi
                }
            }
            float* rd = red[wave];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    rd[(16 * mb + 4 * (lane >> 4) + i) * RED16 + (lane & 15)] = acc[mb][i];
        }
        lds_barrier();
        vm_settled();
        if (lds_flag_read(&abort_flag[par])) break;
        if (s > 0 && pw) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float sum = red[0][r * RED16 + uq * 4 + e];
#pragma unroll
                for (int w = 1; w < NW; ++w) sum = sum + red[w][r * RED16 + uq * 4 + e];
                cmv[e] = sum;
            }
        }
        // ---- gate gradients (annstep.hip mode 3)
        const int t = T - 1 - s;
        const int tt = d ? (T - 1 - t) : t;
        const size_t o_out = ((size_t)b * T + tt) * HO + (size_t)d * H + uc;
        f32x4 dzp, dcp, cdo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float k = drop ? keep_scale(seed, o_out + e, a.p_drop, a.inv_keep) : 1.0f;
            const float dy = gv[e] * k + cmv[e] + cdir[e];
            dzp[e] = (dy * (ypv[e] - cv[e])) * (zv[e] * (1.0f - zv[e]));
            dcp[e] = cv[e] > 0.0f ? dy * (1.0f - zv[e]) : 0.0f;
            cdo[e] = dy * zv[e];
            if (!valid) { dzp[e] = 0.0f; dcp[e] = 0.0f; }
        }
        // ---- publish [dz_pre | dc_pre]: two 16-byte pieces of this workgroup's tile, in A-fragment order
        if (pw) {
            const int mb = r >> 4, row16 = r & 15, half = uq & 1;
            const unsigned pz = (unsigned)((((mb * 2 + half) * 64) + (uq >> 1) * 16 + row16) * 16);
            const unsigned pc = (unsigned)((((mb * 2 + half) * 64) + (2 + (uq >> 1)) * 16 + row16) * 16);
            const unsigned tile_off = rt_off + (unsigned)ct * TILE_BYTES;
            if (s + 1 < T) {
                u32x4 rz, rc;
#pragma unroll
                for (int e = 0; e < 4; ++e) { rz[e] = __float_as_uint(dzp[e]); rc[e] = __float_as_uint(dcp[e]); }
                const unsigned so = (unsigned)(s % RING) * slot_bytes + tile_off;
                __builtin_amdgcn_raw_buffer_store_b128(rz, rsrc, so + pz, 0, REC_ST_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(rc, rsrc, so + pc, 0, REC_ST_AUX);
            }
            if (s >= 2) {
                const u32x4 sent = {SENTINEL, SENTINEL, SENTINEL, SENTINEL};
                const unsigned so = (unsigned)((s - 2) % RING) * slot_bytes + tile_off;
                __builtin_amdgcn_raw_buffer_store_b128(sent, rsrc, so + pz, 0, REC_ST_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(sent, rsrc, so + pc, 0, REC_ST_AUX);
            }
        }
        lds_barrier();
        if (valid) {
            const size_t o_or = ((size_t)bp * T + tt) * H + unit;
            st4(a.dz_all + o_or, dzp); st4(a.dc_all + o_or, dcp); st4(a.yprev_all + o_or, ypv);
        }
        cdir = cdo;
    }
    if (valid) st4(a.carry + (size_t)bp * H + unit, cdir);
    if (tid == 0 && (lds_flag_read(&abort_flag[0]) | lds_flag_read(&abort_flag[1])))
        status_raise(a.status, SPARCH_STATUS_LIGRU_BWD, -1);
}


// ============================================================================== GRU
// forward: vpack = ligru forward layout with [Vz | Vr] (32 columns = [z of 16 units | r of 16 units]);
// vpack2[ct][kg][p][lane] = 8 bf16 of plane p of V: column lane & 15 = unit ct*16 + (lane & 15),
// k = kg*32 + 8*(lane>>4) + j  (B fragment of the 16x16x32 MFMA for (r y) V^T).
// Ring 1 carries y (forward tile order), ring 2 carries q = r y: 32 rows x 32 k per tile in the 16x16x32
// A-fragment order (see the LiGRU backward), producer ct = k 16*(ct & 1) .. +15 of tile ct >> 1.
template <int KGW, int NW>
__global__ __launch_bounds__(64 * NW, 1) void gru_fwd_kernel(LigruArgs a) {
    __shared__ __attribute__((aligned(16))) float red[NW][RT * RED_LD];
    __shared__ __attribute__((aligned(16))) float red16[NW][RT * RED16];
    __shared__ __attribute__((aligned(16))) u32x4 vlo[NW][KGW][2][64];
    __shared__ __attribute__((aligned(16))) u32x4 vlo2[NW][KGW][64];
    __shared__ int abort_flag[2];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hh = lane >> 5;
    const int rt = a.rt_base + (int)(blockIdx.x % a.n_rt_launch);
    const int ct = (int)(blockIdx.x / a.n_rt_launch);
    const int T = a.T, H = a.H, HO = a.H * a.dirs;

    const bool pw = tid < 128;
    const int r = (tid & 127) >> 2, uq = tid & 3;
    const int bp = rt * RT + r, unit = ct * UT + uq * 4;
    const bool valid = pw && bp < a.Bp && unit < H;
    const int bpc = min(bp, a.Bp - 1), uc = min(unit, H - 4);
    const int d = bpc / a.B, b = bpc - d * a.B;

    u32x4 vb[KGW][2][2], vc[KGW][2];
#pragma unroll
    for (int kk = 0; kk < KGW; ++kk) {
        const int kg = wave + NW * kk;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const u32x4* src = a.vpack + ((((size_t)ct * (NW * KGW) + kg) * 2 + ks) * 3) * 64 + lane;
            vb[kk][ks][0] = src[0];
            vb[kk][ks][1] = src[64];
            vlo[wave][kk][ks][lane] = src[128];
        }
        const u32x4* src2 = a.vpack2 + (((size_t)ct * (NW * KGW) + kg) * 3) * 64 + lane;
        vc[kk][0] = src2[0];
        vc[kk][1] = src2[64];
        vlo2[wave][kk][lane] = src2[128];
    }
    if (tid < 2) abort_flag[tid] = 0;
    __syncthreads();

    const unsigned slot_bytes = (unsigned)((size_t)a.n_rt_total * a.n_kg * TILE_BYTES);  // both rings
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(a.ring, 0, (int)(RING * slot_bytes), 0x00020000);
    __amdgpu_buffer_rsrc_t rsrc2 = __builtin_amdgcn_make_buffer_rsrc(a.ring2, 0, (int)(RING * slot_bytes), 0x00020000);
    const unsigned rt_off = (unsigned)((size_t)rt * a.n_kg * TILE_BYTES);
    const bool drop = a.p_drop > 0.0f;
    const uint64_t seed = drop ? resolve_seed(a.seed) : 0;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const u32x4 sent = {SENTINEL, SENTINEL, SENTINEL, SENTINEL};

    f32x4 yp = zero4;  // y_{t-1} of this thread's 4 units (zeros at t = 0, anns.py:584)
    if (pw && a.s_begin > 0) yp = ld4(a.y_state + ((size_t)bpc * T + (a.s_begin - 1)) * H + uc);
    auto load_x = [&](int t, f32x4& xz, f32x4& xr, f32x4& xc) {
        const int tt = d ? (T - 1 - t) : t;
        const size_t o = ((size_t)b * T + tt) * H + uc;
        xz = affine4(a.Wzx, a.scz, a.shz, o, uc);
        xr = affine4(a.Wrx, a.scr, a.shr, o, uc);
        xc = affine4(a.Wx, a.sc, a.sh, o, uc);
    };
    f32x4 xz_n = zero4, xr_n = zero4, xc_n = zero4;
    if (pw) load_x(a.s_begin, xz_n, xr_n, xc_n);

    for (int s = a.s_begin; s < a.s_end; ++s) {
        const f32x4 xz = xz_n, xr = xr_n, xc = xc_n;
        const int par = s & 1;
        if (pw && s + 1 < a.s_end) load_x(s + 1, xz_n, xr_n, xc_n);
        // ---- [z | r] pre-activations: y_{t-1} [Vz | Vr]^T
        float rz[4] = {0.f, 0.f, 0.f, 0.f}, rr[4] = {0.f, 0.f, 0.f, 0.f}, rc[4] = {0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
            const unsigned base = (unsigned)((s - 1) % RING) * slot_bytes + rt_off + (unsigned)lane * 16u;
            constexpr int AHEAD = KGW < REC_AHEAD ? KGW : REC_AHEAD;
            u32x4 raw[KGW][2][2];
#pragma unroll
            for (int kk = 0; kk < AHEAD; ++kk) issue_tile<NW>(raw[kk], rsrc, base, wave + NW * kk, a.n_kg);
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int kk = 0; kk < KGW; ++kk) {
                __builtin_amdgcn_sched_barrier(0);
                if (wave + NW * kk < a.n_kg)
                    settle_tile(raw[kk], rsrc, base + (unsigned)(wave + NW * kk) * TILE_BYTES, &abort_flag[par]);
                if (kk + AHEAD < KGW) issue_tile<NW>(raw[kk + AHEAD], rsrc, base, wave + NW * (kk + AHEAD), a.n_kg);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    u32x4 p1, p2, p3;
                    split_pieces(raw[kk][ks][0], raw[kk][ks][1], p1, p2, p3);
                    const u32x4 vl = vlo[wave][kk][ks][lane];
                    acc = mfma_bf16(p2, vb[kk][ks][1], acc);  // t2*mid
                    acc = mfma_bf16(p3, vb[kk][ks][0], acc);  // t3*hi
                    acc = mfma_bf16(p1, vl, acc);             // t1*lo
                    acc = mfma_bf16(p2, vb[kk][ks][0], acc);  // t2*hi
                    acc = mfma_bf16(p1, vb[kk][ks][1], acc);  // t1*mid
                    acc = mfma_bf16(p1, vb[kk][ks][0], acc);  // t1*hi
                }
            }
            float* rd = red[wave];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
                rd[row * RED_LD + li] = acc[i];
            }
        }
        lds_barrier();
        vm_settled();
        if (lds_flag_read(&abort_flag[par])) break;
        if (s > 0 && pw) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float sz = red[0][r * RED_LD + uq * 4 + e], sr = red[0][r * RED_LD + UT + uq * 4 + e];
#pragma unroll
                for (int w = 1; w < NW; ++w) {
                    sz = sz + red[w][r * RED_LD + uq * 4 + e];
                    sr = sr + red[w][r * RED_LD + UT + uq * 4 + e];
                }
                rz[e] = sz; rr[e] = sr;
            }
        }
        // ---- gates (anns.py:589-590) and q = r y_{t-1}
        f32x4 z, rg, q;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            z[e] = sigm(xz[e] + rz[e]);
            rg[e] = sigm(xr[e] + rr[e]);
            q[e] = valid ? rg[e] * yp[e] : 0.0f;
        }
        // ---- publish q (steps > 0: y_{-1} = 0 makes step 0's product zero, nobody reads a step-0 tile)
        const int mbp = r >> 4, row16 = r & 15;
        const unsigned qpiece = (unsigned)((((mbp * 2 + (uq & 1)) * 64) + ((ct & 1) * 2 + (uq >> 1)) * 16 + row16) * 16);
        const unsigned qtile = rt_off + (unsigned)(ct >> 1) * TILE_BYTES + qpiece;
        if (pw) {
            if (s > 0) {
                u32x4 rawv;
#pragma unroll
                for (int e = 0; e < 4; ++e) rawv[e] = __float_as_uint(q[e]);
                __builtin_amdgcn_raw_buffer_store_b128(rawv, rsrc2, (unsigned)(s % RING) * slot_bytes + qtile, 0, REC_ST_AUX);
            }
            if (s >= 2) __builtin_amdgcn_raw_buffer_store_b128(sent, rsrc2, (unsigned)((s - 2) % RING) * slot_bytes + qtile, 0, REC_ST_AUX);
        }
        // ---- candidate pre-activation: q V^T on the 16x16x32 MFMA
        if (s > 0) {
            const unsigned base = (unsigned)(s % RING) * slot_bytes + rt_off + (unsigned)lane * 16u;
            u32x4 raw[2][2][2];  // [parity of kk][mb][half]: one tile ahead
            auto issue = [&](int kk) { issue_tile<NW>(raw[kk & 1], rsrc2, base, wave + NW * kk, a.n_kg); };
            issue(0);
            f32x4 acc[2] = {zero4, zero4};
#pragma unroll
            for (int kk = 0; kk < KGW; ++kk) {
                __builtin_amdgcn_sched_barrier(0);
                if (wave + NW * kk < a.n_kg)
                    settle_tile(raw[kk & 1], rsrc2, base + (unsigned)(wave + NW * kk) * TILE_BYTES, &abort_flag[par]);
                if (kk + 1 < KGW) issue(kk + 1);
                __builtin_amdgcn_sched_barrier(0);
                const u32x4 vl = vlo2[wave][kk][lane];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    u32x4 p1, p2, p3;
                    split_pieces(raw[kk & 1][mb][0], raw[kk & 1][mb][1], p1, p2, p3);
                    acc[mb] = mfma16(p2, vc[kk][1], acc[mb]);  // t2*mid
                    acc[mb] = mfma16(p3, vc[kk][0], acc[mb]);  // t3*hi
                    acc[mb] = mfma16(p1, vl, acc[mb]);         // t1*lo
                    acc[mb] = mfma16(p2, vc[kk][0], acc[mb]);  // t2*hi
                    acc[mb] = mfma16(p1, vc[kk][1], acc[mb]);  // t1*mid
                    acc[mb] = mfma16(p1, vc[kk][0], acc[mb]);  // t1*hi
                }
            }
            float* rd = red16[wave];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    rd[(16 * mb + 4 * (lane >> 4) + i) * RED16 + (lane & 15)] = acc[mb][i];
        }
        lds_barrier();
        vm_settled();
        if (lds_flag_read(&abort_flag[par])) break;
        if (s > 0 && pw) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float sum = red16[0][r * RED16 + uq * 4 + e];
#pragma unroll
                for (int w = 1; w < NW; ++w) sum = sum + red16[w][r * RED16 + uq * 4 + e];
                rc[e] = sum;
            }
        }
        // ---- candidate and state (anns.py:591-592)
        const int t = s;
        const int tt = d ? (T - 1 - t) : t;
        const size_t o_out = ((size_t)b * T + tt) * HO + (size_t)d * H + uc;
        f32x4 c, y, yo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            c[e] = tanhf(xc[e] + rc[e]);
            y[e] = z[e] * yp[e] + (1.0f - z[e]) * c[e];
            const float k = drop ? keep_scale(seed, o_out + e, a.p_drop, a.inv_keep) : 1.0f;
            yo[e] = y[e] * k;
            if (!valid) y[e] = 0.0f;
        }
        // ---- publish y_t (forward tile order, as the LiGRU)
        if (pw) {
            const int ksp = ct & 1, hq = uq >> 1, qq = uq & 1;
            const unsigned piece = (unsigned)((((ksp * 2 + qq) * 64) + hq * 32 + r) * 16);
            const unsigned tile_off = rt_off + (unsigned)(ct >> 1) * TILE_BYTES + piece;
            if (s + 1 < T) {
                u32x4 rawv;
#pragma unroll
                for (int e = 0; e < 4; ++e) rawv[e] = __float_as_uint(y[e]);
                __builtin_amdgcn_raw_buffer_store_b128(rawv, rsrc, (unsigned)(s % RING) * slot_bytes + tile_off, 0, REC_ST_AUX);
            }
            if (s >= 2) __builtin_amdgcn_raw_buffer_store_b128(sent, rsrc, (unsigned)((s - 2) % RING) * slot_bytes + tile_off, 0, REC_ST_AUX);
        }
        if (valid) {
            const size_t o_st = ((size_t)bp * T + t) * H + unit;
            st4(a.y_state + o_st, y); st4(a.z_save + o_st, z); st4(a.r_save + o_st, rg); st4(a.c_save + o_st, c);
            st4(a.y_out + ((size_t)b * T + tt) * HO + (size_t)d * H + unit, yo);
        }
        yp = y;
    }
    if (tid == 0 && (lds_flag_read(&abort_flag[0]) | lds_flag_read(&abort_flag[1])))
        status_raise(a.status, SPARCH_STATUS_GRU_FWD, -1);
}

// backward: vpack = ligru backward layout with [Vz ; Vr] (K = 2H: producer tile kg = 32 rows x (16 dz_pre +
// 16 dr_pre)); vpack2[ct][kg][p][lane]: column lane & 15 = unit ct*16 + (lane & 15), k = kg*32 + 8*(lane>>4) + j
// = SOURCE unit of dc_pre: V[k][unit]  (dq = dc_pre V).  Ring 1 carries [dz_pre | dr_pre] (one tile per
// producer), ring 2 dc_pre (producer ct = k 16*(ct & 1) .. +15 of tile ct >> 1).
template <int KGW, int NW>
__global__ __launch_bounds__(64 * NW, 1) void gru_bwd_kernel(LigruArgs a) {
    constexpr int KG1 = KGW > 1 ? KGW / 2 : 1;  // dc_pre tiles (32 source units) per wave
    __shared__ __attribute__((aligned(16))) float red[NW][RT * RED16];
    __shared__ __attribute__((aligned(16))) float red2[NW][RT * RED16];
    __shared__ __attribute__((aligned(16))) u32x4 vlo[NW][KGW][64];
    __shared__ __attribute__((aligned(16))) u32x4 vlo2[NW][KG1][64];
    __shared__ int abort_flag[2];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rt = a.rt_base + (int)(blockIdx.x % a.n_rt_launch);
    const int ct = (int)(blockIdx.x / a.n_rt_launch);
    const int T = a.T, H = a.H, HO = a.H * a.dirs;

    const bool pw = tid < 128;
    const int r = (tid & 127) >> 2, uq = tid & 3;
    const int bp = rt * RT + r, unit = ct * UT + uq * 4;
    const bool valid = pw && bp < a.Bp && unit < H;
    const int bpc = min(bp, a.Bp - 1), uc = min(unit, H - 4);
    const int d = bpc / a.B, b = bpc - d * a.B;

    u32x4 vb[KGW][2], vc[KG1][2];
#pragma unroll
    for (int kk = 0; kk < KGW; ++kk) {
        const u32x4* src = a.vpack + (((size_t)ct * (NW * KGW) + wave + NW * kk) * 3) * 64 + lane;
        vb[kk][0] = src[0];
        vb[kk][1] = src[64];
        vlo[wave][kk][lane] = src[128];
    }
#pragma unroll
    for (int kk = 0; kk < KG1; ++kk) {
        const u32x4* src = a.vpack2 + (((size_t)ct * (NW * KG1) + wave + NW * kk) * 3) * 64 + lane;
        vc[kk][0] = src[0];
        vc[kk][1] = src[64];
        vlo2[wave][kk][lane] = src[128];
    }
    if (tid < 2) abort_flag[tid] = 0;
    __syncthreads();

    const unsigned slot_bytes = (unsigned)((size_t)a.n_rt_total * a.n_ct * TILE_BYTES);    // ring 1: one tile per producer
    const unsigned slot2_bytes = (unsigned)((size_t)a.n_rt_total * a.n_kg * TILE_BYTES);   // ring 2: two producers per tile
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(a.ring, 0, (int)(RING * slot_bytes), 0x00020000);
    __amdgpu_buffer_rsrc_t rsrc2 = __builtin_amdgcn_make_buffer_rsrc(a.ring2, 0, (int)(RING * slot2_bytes), 0x00020000);
    const unsigned rt_off = (unsigned)((size_t)rt * a.n_ct * TILE_BYTES);
    const unsigned rt_off2 = (unsigned)((size_t)rt * a.n_kg * TILE_BYTES);
    const bool drop = a.p_drop > 0.0f;
    const uint64_t seed = drop ? resolve_seed(a.seed) : 0;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const u32x4 sent = {SENTINEL, SENTINEL, SENTINEL, SENTINEL};

    f32x4 cdir = zero4;  // (dq r + dy z)_{t+1}
    if (pw && a.s_begin > 0) cdir = ld4(a.carry + (size_t)bpc * H + uc);
    auto load_step = [&](int s, f32x4& g, f32x4& z, f32x4& rg, f32x4& c, f32x4& ypv) {
        const int t = T - 1 - s;
        const int tt = d ? (T - 1 - t) : t;
        g = ld4(a.g_out + ((size_t)b * T + tt) * HO + (size_t)d * H + uc);
        const size_t o_st = ((size_t)bpc * T + t) * H + uc;
        z = ld4(a.z_save + o_st); rg = ld4(a.r_save + o_st); c = ld4(a.c_save + o_st);
        ypv = t > 0 ? ld4(a.y_state + o_st - H) : zero4;
    };
    f32x4 g_n = zero4, z_n = zero4, r_n = zero4, c_n = zero4, yp_n = zero4;
    if (pw) load_step(a.s_begin, g_n, z_n, r_n, c_n, yp_n);

    // one 16-column product over `ntiles` tiles of 32 k from ring slot `base`: partial tiles -> rdst[wave]
    auto product16 = [&](auto& vreg, auto& vl_lds, auto ktiles, __amdgpu_buffer_rsrc_t rs, unsigned base, int ntiles,
                         float (*rdst)[RT * RED16], int par) __attribute__((always_inline)) {
        constexpr int KT = decltype(ktiles)::value;
        u32x4 raw[2][2][2];  // [parity of kk][mb][half]: one tile ahead
        auto issue = [&](int kk) { issue_tile<NW>(raw[kk & 1], rs, base, wave + NW * kk, ntiles); };
        issue(0);
        f32x4 acc[2] = {zero4, zero4};
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
            __builtin_amdgcn_sched_barrier(0);
            if (wave + NW * kk < ntiles)
                settle_tile(raw[kk & 1], rs, base + (unsigned)(wave + NW * kk) * TILE_BYTES, &abort_flag[par]);
            if (kk + 1 < KT) issue(kk + 1);
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 vl = vl_lds[wave][kk][lane];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                u32x4 p1, p2, p3;
                split_pieces(raw[kk & 1][mb][0], raw[kk & 1][mb][1], p1, p2, p3);
                acc[mb] = mfma16(p2, vreg[kk][1], acc[mb]);  // t2*mid
                acc[mb] = mfma16(p3, vreg[kk][0], acc[mb]);  // t3*hi
                acc[mb] = mfma16(p1, vl, acc[mb]);           // t1*lo
                acc[mb] = mfma16(p2, vreg[kk][0], acc[mb]);  // t2*hi
                acc[mb] = mfma16(p1, vreg[kk][1], acc[mb]);  // t1*mid
                acc[mb] = mfma16(p1, vreg[kk][0], acc[mb]);  // t1*hi
            }
        }
        float* rd = rdst[wave];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                rd[(16 * mb + 4 * (lane >> 4) + i) * RED16 + (lane & 15)] = acc[mb][i];
    };

    for (int s = a.s_begin; s < a.s_end; ++s) {
        const f32x4 gv = g_n, zv = z_n, rv = r_n, cv = c_n, ypv = yp_n;
        const int par = s & 1;
        if (pw && s + 1 < a.s_end) load_step(s + 1, g_n, z_n, r_n, c_n, yp_n);
        // ---- [dz_pre | dr_pre]_{t+1} [Vz ; Vr]
        float cmv[4] = {0.f, 0.f, 0.f, 0.f}, dq[4] = {0.f, 0.f, 0.f, 0.f};
        if (s > 0)
            product16(vb, vlo, std::integral_constant<int, KGW>{}, rsrc,
                      (unsigned)((s - 1) % RING) * slot_bytes + rt_off + (unsigned)lane * 16u, a.n_ct, red, par);
        lds_barrier();
        vm_settled();
        if (lds_flag_read(&abort_flag[par])) break;
        if (s > 0 && pw) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float sum = red[0][r * RED16 + uq * 4 + e];
#pragma unroll
                for (int w = 1; w < NW; ++w) sum = sum + red[w][r * RED16 + uq * 4 + e];
                cmv[e] = sum;
            }
        }
        // ---- gate gradients, first half (annstep.hip mode 4)
        const int t = T - 1 - s;
        const int tt = d ? (T - 1 - t) : t;
        const size_t o_out = ((size_t)b * T + tt) * HO + (size_t)d * H + uc;
        f32x4 dzp, dcp, cdo, ry;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float k = drop ? keep_scale(seed, o_out + e, a.p_drop, a.inv_keep) : 1.0f;
            const float dy = gv[e] * k + cmv[e] + cdir[e];
            dzp[e] = (dy * (ypv[e] - cv[e])) * (zv[e] * (1.0f - zv[e]));
            dcp[e] = (dy * (1.0f - zv[e])) * (1.0f - cv[e] * cv[e]);
            cdo[e] = dy * zv[e];
            ry[e] = rv[e] * ypv[e];
            if (!valid) { dzp[e] = 0.0f; dcp[e] = 0.0f; }
        }
        // ---- publish dc_pre, then dq = dc_pre V
        const int mbp = r >> 4, row16 = r & 15, half = uq & 1;
        if (pw) {
            const unsigned cpiece = (unsigned)((((mbp * 2 + half) * 64) + ((ct & 1) * 2 + (uq >> 1)) * 16 + row16) * 16);
            const unsigned ctile = rt_off2 + (unsigned)(ct >> 1) * TILE_BYTES + cpiece;
            u32x4 rawv;
#pragma unroll
            for (int e = 0; e < 4; ++e) rawv[e] = __float_as_uint(dcp[e]);
            __builtin_amdgcn_raw_buffer_store_b128(rawv, rsrc2, (unsigned)(s % RING) * slot2_bytes + ctile, 0, REC_ST_AUX);
            if (s >= 2) __builtin_amdgcn_raw_buffer_store_b128(sent, rsrc2, (unsigned)((s - 2) % RING) * slot2_bytes + ctile, 0, REC_ST_AUX);
        }
        product16(vc, vlo2, std::integral_constant<int, KG1>{}, rsrc2,
                  (unsigned)(s % RING) * slot2_bytes + rt_off2 + (unsigned)lane * 16u, a.n_kg, red2, par);
        lds_barrier();
        vm_settled();
        if (lds_flag_read(&abort_flag[par])) break;
        if (pw) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float sum = red2[0][r * RED16 + uq * 4 + e];
#pragma unroll
                for (int w = 1; w < NW; ++w) sum = sum + red2[w][r * RED16 + uq * 4 + e];
                dq[e] = sum;
            }
        }
        // ---- second half (annstep.hip mode 5)
        f32x4 drp;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            drp[e] = (dq[e] * ypv[e]) * (rv[e] * (1.0f - rv[e]));
            cdo[e] = cdo[e] + dq[e] * rv[e];
            if (!valid) drp[e] = 0.0f;
        }
        // ---- publish [dz_pre | dr_pre]: two 16-byte pieces of this workgroup's tile, in A-fragment order
        if (pw) {
            const unsigned pz = (unsigned)((((mbp * 2 + half) * 64) + (uq >> 1) * 16 + row16) * 16);
            const unsigned pr = (unsigned)((((mbp * 2 + half) * 64) + (2 + (uq >> 1)) * 16 + row16) * 16);
            const unsigned tile_off = rt_off + (unsigned)ct * TILE_BYTES;
            if (s + 1 < T) {
                u32x4 wz, wr;
#pragma unroll
                for (int e = 0; e < 4; ++e) { wz[e] = __float_as_uint(dzp[e]); wr[e] = __float_as_uint(drp[e]); }
                const unsigned so = (unsigned)(s % RING) * slot_bytes + tile_off;
                __builtin_amdgcn_raw_buffer_store_b128(wz, rsrc, so + pz, 0, REC_ST_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(wr, rsrc, so + pr, 0, REC_ST_AUX);
            }
            if (s >= 2) {
                const unsigned so = (unsigned)((s - 2) % RING) * slot_bytes + tile_off;
                __builtin_amdgcn_raw_buffer_store_b128(sent, rsrc, so + pz, 0, REC_ST_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(sent, rsrc, so + pr, 0, REC_ST_AUX);
            }
        }
        if (valid) {
            const size_t o_or = ((size_t)bp * T + tt) * H + unit;
            st4(a.dz_all + o_or, dzp); st4(a.dr_all + o_or, drp); st4(a.dc_all + o_or, dcp);
            st4(a.yprev_all + o_or, ypv); st4(a.ry_all + o_or, ry);
        }
        cdir = cdo;
    }
    if (valid) st4(a.carry + (size_t)bp * H + unit, cdir);
    if (tid == 0 && (lds_flag_read(&abort_flag[0]) | lds_flag_read(&abort_flag[1])))
        status_raise(a.status, SPARCH_STATUS_GRU_BWD, -1);
}

// fragments of ONE matrix for a 16-column product on the 16x16x32 MFMA: column lane & 15 = unit ct*16 + (lane & 15),
// k = kg*32 + 8*(lane>>4) + j;  transposed == 0: V[unit][k] (q V^T),  1: V[k][unit] (dc_pre V)
__global__ void gru_vpack16_kernel(int H, int n_ct, int nkg, int transposed, int rne, const float* __restrict__ V,
                                   u32x4* __restrict__ vpack) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)n_ct * nkg * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const int kg = (int)((idx >> 6) % nkg), ct = (int)((idx >> 6) / nkg);
    const int unit = ct * UT + (lane & 15);
    unsigned short pl[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = kg * 32 + 8 * (lane >> 4) + j;
        const float v = (k < H && unit < H) ? (transposed ? V[(size_t)k * H + unit] : V[(size_t)unit * H + k]) : 0.f;
        vsplit(v, rne, pl[0][j], pl[1][j], pl[2][j]);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = (unsigned)pl[p][2 * q] | ((unsigned)pl[p][2 * q + 1] << 16);
        vpack[(((size_t)ct * nkg + kg) * 3 + p) * 64 + lane] = o;
    }
}

// ------------------------------------------------------------------------------ prepack
__global__ void ligru_vpack_fwd_kernel(int H, int n_ct, int nkg, int rne, const float* __restrict__ Vz, const float* __restrict__ Vc,
                                       u32x4* __restrict__ vpack) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)n_ct * nkg * 2 * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63), ks = (int)((idx >> 6) & 1);
    const int kg = (int)((idx >> 7) % nkg), ct = (int)((idx >> 7) / nkg);
    const int n = lane & 31;
    const int unit = ct * UT + (n & 15);
    const float* V = n < UT ? Vz : Vc;
    unsigned short pl[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = kg * 32 + 16 * ks + 8 * (lane >> 5) + j;
        const float v = (k < H && unit < H) ? V[(size_t)unit * H + k] : 0.f;
        vsplit(v, rne, pl[0][j], pl[1][j], pl[2][j]);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = (unsigned)pl[p][2 * q] | ((unsigned)pl[p][2 * q + 1] << 16);
        vpack[((((size_t)ct * nkg + kg) * 2 + ks) * 3 + p) * 64 + lane] = o;
    }
}
__global__ void ligru_vpack_bwd_kernel(int H, int n_ct, int nkg, int rne, const float* __restrict__ Vz, const float* __restrict__ Vc,
                                       u32x4* __restrict__ vpack) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)n_ct * nkg * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const int kg = (int)((idx >> 6) % nkg), ct = (int)((idx >> 6) / nkg);
    const int unit = ct * UT + (lane & 15);
    unsigned short pl[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * (lane >> 4) + j;                 // 0..31 inside the k-group
        const int src_unit = kg * UT + (k & 15);           // row of Vz (k < 16) or V (k >= 16)
        const float* V = k < UT ? Vz : Vc;
        const float v = (src_unit < H && unit < H && kg < n_ct) ? V[(size_t)src_unit * H + unit] : 0.f;
        vsplit(v, rne, pl[0][j], pl[1][j], pl[2][j]);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = (unsigned)pl[p][2 * q] | ((unsigned)pl[p][2 * q + 1] << 16);
        vpack[(((size_t)ct * nkg + kg) * 3 + p) * 64 + lane] = o;
    }
}

int kgw_fwd(int H) {  // k-groups of 32 per wave, 8 waves
    const int need = cdiv(cdiv(H, 32), 8);
    for (int k : {1, 2, 4})
        if (need <= k) return k;
    return 0;
}
int kgw_bwd(int H) {  // producer tiles (16 units each) per wave, 8 waves
    const int need = cdiv(cdiv(H, UT), 8);
    for (int k : {1, 2, 4, 8})
        if (need <= k) return k;
    return 0;
}
size_t ring_bytes_fwd(int Bp, int H) { return (size_t)RING * cdiv(Bp, RT) * cdiv(H, 32) * TILE_BYTES; }
size_t ring_bytes_bwd(int Bp, int H) { return (size_t)RING * cdiv(Bp, RT) * cdiv(H, UT) * TILE_BYTES; }

template <bool BWD>
int run_ligru(LigruArgs& a, void* chan, size_t chan_bytes, int steps_per_launch, hipStream_t st) {
    const int kgw = BWD ? kgw_bwd(a.H) : kgw_fwd(a.H);
    if (kgw == 0) return SPARCH_EINVAL;
    a.n_ct = a.H / UT;
    a.n_kg = a.H / 32;
    a.n_rt_total = cdiv(a.Bp, RT);
    const size_t rb = BWD ? ring_bytes_bwd(a.Bp, a.H) : ring_bytes_fwd(a.Bp, a.H);
    if (!chan || chan_bytes < rb) return SPARCH_EWORKSPACE;
    if (rb >= ((size_t)1 << 31)) return SPARCH_EINVAL;  // 32-bit buffer offsets
    if (hipMemsetD32Async((hipDeviceptr_t)chan, (int)SENTINEL, rb / 4, st) != hipSuccess) return SPARCH_ELAUNCH;
    a.ring = reinterpret_cast<char*>(chan);
    int L = steps_per_launch;
    if (L < 1) L = 1;
    if (L > a.T) L = a.T;
    int cus = sparch_device_cus();
    if (cus <= 0) cus = 256;
    int rt_per_launch;
    if (L == 1) {
        rt_per_launch = a.n_rt_total;
    } else {
        rt_per_launch = cus / a.n_ct;  // one workgroup per CU must be co-resident
        if (rt_per_launch < 1) { L = 1; rt_per_launch = a.n_rt_total; }
    }
    for (int rt0 = 0; rt0 < a.n_rt_total; rt0 += rt_per_launch) {
        a.rt_base = rt0;
        a.n_rt_launch = min(rt_per_launch, a.n_rt_total - rt0);
        const unsigned grid = (unsigned)(a.n_ct * a.n_rt_launch);
        for (int s0 = 0; s0 < a.T; s0 += L) {
            a.s_begin = s0; a.s_end = min(a.T, s0 + L);
#define SP_LIGRU(K)                                                                                      \
    if (BWD) hipLaunchKernelGGL((ligru_bwd_kernel<K, 8>), dim3(grid), dim3(512), 0, st, a);              \
    else     hipLaunchKernelGGL((ligru_fwd_kernel<(K > 4 ? 4 : K), 8>), dim3(grid), dim3(512), 0, st, a);
            switch (kgw) {
                case 1: SP_LIGRU(1) break;
                case 2: SP_LIGRU(2) break;
                case 4: SP_LIGRU(4) break;
                case 8: SP_LIGRU(8) break;
                default: return SPARCH_EINVAL;
            }
#undef SP_LIGRU
            SPARCH_CHECK_LAUNCH();
        }
    }
    return SPARCH_OK;
}


// GRU launches: always co-resident (two hand-offs per step), the row-tile groups one after the other
template <bool BWD>
int run_gru(LigruArgs& a, void* chan, size_t chan_bytes, int steps_per_launch, hipStream_t st) {
    const int kgw = BWD ? kgw_bwd(a.H) : kgw_fwd(a.H);
    if (kgw == 0) return SPARCH_EINVAL;
    a.n_ct = a.H / UT;
    a.n_kg = a.H / 32;
    a.n_rt_total = cdiv(a.Bp, RT);
    const size_t r1 = BWD ? ring_bytes_bwd(a.Bp, a.H) : ring_bytes_fwd(a.Bp, a.H), r2 = ring_bytes_fwd(a.Bp, a.H);
    if (!chan || chan_bytes < r1 + r2) return SPARCH_EWORKSPACE;
    if (r1 >= ((size_t)1 << 31)) return SPARCH_EINVAL;  // 32-bit buffer offsets
    if (hipMemsetD32Async((hipDeviceptr_t)chan, (int)SENTINEL, (r1 + r2) / 4, st) != hipSuccess) return SPARCH_ELAUNCH;
    a.ring = reinterpret_cast<char*>(chan);
    a.ring2 = a.ring + r1;
    int L = steps_per_launch;
    if (L < 1) L = 1;
    if (L > a.T) L = a.T;
    int cus = sparch_device_cus();
    if (cus <= 0) cus = 256;
    const int rt_per_launch = cus / a.n_ct;
    if (rt_per_launch < 1) return SPARCH_EINVAL;  // a row tile's workgroups cannot all be resident: per-step path
    for (int rt0 = 0; rt0 < a.n_rt_total; rt0 += rt_per_launch) {
        a.rt_base = rt0;
        a.n_rt_launch = min(rt_per_launch, a.n_rt_total - rt0);
        const unsigned grid = (unsigned)(a.n_ct * a.n_rt_launch);
        for (int s0 = 0; s0 < a.T; s0 += L) {
            a.s_begin = s0; a.s_end = min(a.T, s0 + L);
#define SP_GRU(K)                                                                                        \
    if (BWD) hipLaunchKernelGGL((gru_bwd_kernel<K, 8>), dim3(grid), dim3(512), 0, st, a);                \
    else     hipLaunchKernelGGL((gru_fwd_kernel<(K > 4 ? 4 : K), 8>), dim3(grid), dim3(512), 0, st, a);
            switch (kgw) {
                case 1: SP_GRU(1) break;
                case 2: SP_GRU(2) break;
                case 4: SP_GRU(4) break;
                case 8: SP_GRU(8) break;
                default: return SPARCH_EINVAL;
            }
#undef SP_GRU
            SPARCH_CHECK_LAUNCH();
        }
    }
    return SPARCH_OK;
}

bool al16g(std::initializer_list<const void*> ps) {
    for (const void* p : ps)
        if (p && !aligned16(p)) return false;
    return true;
}

}  // namespace

extern "C" size_t sparch_ligru_vpack_bytes(int H, int backward) {
    if (H <= 0 || H % 32 != 0) return 0;
    const int kgw = backward ? kgw_bwd(H) : kgw_fwd(H);
    if (kgw == 0) return 0;
    const size_t n_ct = (size_t)H / UT;
    return backward ? n_ct * (8 * kgw) * 3 * 64 * sizeof(u32x4) : n_ct * (8 * kgw) * 2 * 3 * 64 * sizeof(u32x4);
}

extern "C" int sparch_ligru_vpack(int H, const float* Vz, const float* V, int backward, float* vpack, void* stream, int precision) {
    SPARCH_ENTER();
    PrecisionScope prec_scope_(precision);
    if (!prec_scope_.ok) return SPARCH_EINVAL;
    if (sparch_ligru_vpack_bytes(H, backward) == 0 || !Vz || !V || !vpack) return SPARCH_EINVAL;
    if (!aligned16(vpack)) return SPARCH_EALIGN;
    const int n_ct = H / UT;
    hipStream_t st = (hipStream_t)stream;
    if (!backward) {
        const int nkg = 8 * kgw_fwd(H);
        const size_t total = (size_t)n_ct * nkg * 2 * 64;
        hipLaunchKernelGGL(ligru_vpack_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, H, n_ct, nkg,
                           sparch_operand_bf16(), Vz, V, reinterpret_cast<u32x4*>(vpack));
    } else {
        const int nkg = 8 * kgw_bwd(H);
        const size_t total = (size_t)n_ct * nkg * 64;
        hipLaunchKernelGGL(ligru_vpack_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, H, n_ct, nkg,
                           sparch_operand_bf16(), Vz, V, reinterpret_cast<u32x4*>(vpack));
    }
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

extern "C" size_t sparch_ligru_chan_bytes(int Bp, int H) {
    if (Bp <= 0 || H <= 0) return 0;
    const size_t f = ring_bytes_fwd(Bp, H), b = ring_bytes_bwd(Bp, H);
    return f > b ? f : b;
}

extern "C" int sparch_ligru_fwd(int B, int dirs, int T, int H, const float* Wx, const float* sc, const float* sh,
                                const float* Wzx, const float* scz, const float* shz, const float* vpack,
                                float p_drop, uint64_t seed, float* y_out, float* y_state, float* z_save,
                                float* c_save, void* chan, size_t chan_bytes, uint32_t* status,
                                int steps_per_launch, void* stream) {
    SPARCH_ENTER();
    if (B <= 0 || T <= 0 || H <= 0 || H % 32 != 0 || (dirs != 1 && dirs != 2)) return SPARCH_EINVAL;
    if (!Wx || !Wzx || !vpack || !y_out || !y_state || !z_save || !c_save || !status) return SPARCH_EINVAL;
    if ((sc == nullptr) != (sh == nullptr) || (scz == nullptr) != (shz == nullptr)) return SPARCH_EINVAL;
    if (!(p_drop >= 0.0f && p_drop < 1.0f)) return SPARCH_EINVAL;
    if (!al16g({Wx, sc, sh, Wzx, scz, shz, vpack, y_out, y_state, z_save, c_save, chan})) return SPARCH_EALIGN;
    LigruArgs a{};
    a.B = B; a.dirs = dirs; a.T = T; a.H = H; a.Bp = B * dirs;
    a.Wx = Wx; a.sc = sc; a.sh = sh; a.Wzx = Wzx; a.scz = scz; a.shz = shz;
    a.vpack = reinterpret_cast<const u32x4*>(vpack);
    a.p_drop = p_drop; a.inv_keep = 1.0f / (1.0f - p_drop); a.seed = seed;
    a.y_out = y_out; a.y_state = y_state; a.z_save = z_save; a.c_save = c_save; a.status = status;
    return run_ligru<false>(a, chan, chan_bytes, steps_per_launch, (hipStream_t)stream);
}

extern "C" int sparch_ligru_bwd(int B, int dirs, int T, int H, const float* g_out, const float* y_state,
                                const float* z_save, const float* c_save, const float* vpack_b, float p_drop,
                                uint64_t seed, float* dz_all, float* dc_all, float* yprev_all, float* carry,
                                void* chan, size_t chan_bytes, uint32_t* status, int steps_per_launch,
                                void* stream) {
    SPARCH_ENTER();
    if (B <= 0 || T <= 0 || H <= 0 || H % 32 != 0 || (dirs != 1 && dirs != 2)) return SPARCH_EINVAL;
    if (!g_out || !y_state || !z_save || !c_save || !vpack_b || !dz_all || !dc_all || !yprev_all || !carry || !status)
        return SPARCH_EINVAL;
    if (!(p_drop >= 0.0f && p_drop < 1.0f)) return SPARCH_EINVAL;
    if (!al16g({g_out, y_state, z_save, c_save, vpack_b, dz_all, dc_all, yprev_all, carry, chan})) return SPARCH_EALIGN;
    LigruArgs a{};
    a.B = B; a.dirs = dirs; a.T = T; a.H = H; a.Bp = B * dirs;
    a.g_out = g_out; a.y_state = const_cast<float*>(y_state); a.z_save = const_cast<float*>(z_save);
    a.c_save = const_cast<float*>(c_save); a.vpack = reinterpret_cast<const u32x4*>(vpack_b);
    a.p_drop = p_drop; a.inv_keep = 1.0f / (1.0f - p_drop); a.seed = seed;
    a.dz_all = dz_all; a.dc_all = dc_all; a.yprev_all = yprev_all; a.carry = carry; a.status = status;
    return run_ligru<true>(a, chan, chan_bytes, steps_per_launch, (hipStream_t)stream);
}

// ---- GRU (sparch_amd/functional.py GatedLayerFn): two fragment buffers per direction of time
extern "C" size_t sparch_gru_vpack_bytes(int H, int backward, int which) {
    if (H <= 0 || H % 32 != 0) return 0;
    const int kgw = backward ? kgw_bwd(H) : kgw_fwd(H);
    if (kgw == 0) return 0;
    const size_t n_ct = (size_t)H / UT;
    if (which == 0) return sparch_ligru_vpack_bytes(H, backward);             // [Vz | Vr] / [Vz ; Vr]
    const int kg2 = backward ? (kgw > 1 ? kgw / 2 : 1) : kgw;                 // V: tiles of 32 k per wave
    return n_ct * (8 * kg2) * 3 * 64 * sizeof(u32x4);
}

extern "C" int sparch_gru_vpack(int H, const float* Vz, const float* Vr, const float* V, int backward, float* vpack_gate,
                                float* vpack_cand, void* stream, int precision) {
    SPARCH_ENTER();
    PrecisionScope prec_scope_(precision);
    if (!prec_scope_.ok) return SPARCH_EINVAL;
    if (sparch_gru_vpack_bytes(H, backward, 0) == 0 || !Vz || !Vr || !V || !vpack_gate || !vpack_cand) return SPARCH_EINVAL;
    if (!aligned16(vpack_gate) || !aligned16(vpack_cand)) return SPARCH_EALIGN;
    const int rc = sparch_ligru_vpack(H, Vz, Vr, backward, vpack_gate, stream, precision);
    if (rc != SPARCH_OK) return rc;
    const int n_ct = H / UT;
    const int kgw = backward ? kgw_bwd(H) : kgw_fwd(H);
    const int nkg = 8 * (backward ? (kgw > 1 ? kgw / 2 : 1) : kgw);
    const size_t total = (size_t)n_ct * nkg * 64;
    hipLaunchKernelGGL(gru_vpack16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, H,
                       n_ct, nkg, backward ? 1 : 0, sparch_operand_bf16(), V, reinterpret_cast<u32x4*>(vpack_cand));
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

extern "C" size_t sparch_gru_chan_bytes(int Bp, int H) {
    if (Bp <= 0 || H <= 0) return 0;
    return ring_bytes_bwd(Bp, H) + ring_bytes_fwd(Bp, H);  // the larger (backward) pair of rings
}

extern "C" int sparch_gru_fwd(int B, int dirs, int T, int H, const float* Wx, const float* sc, const float* sh,
                              const float* Wzx, const float* scz, const float* shz, const float* Wrx, const float* scr,
                              const float* shr, const float* vpack_gate, const float* vpack_cand, float p_drop,
                              uint64_t seed, float* y_out, float* y_state, float* z_save, float* r_save, float* c_save,
                              void* chan, size_t chan_bytes, uint32_t* status, int steps_per_launch, void* stream) {
    SPARCH_ENTER();
    if (B <= 0 || T <= 0 || H <= 0 || H % 32 != 0 || (dirs != 1 && dirs != 2)) return SPARCH_EINVAL;
    if (!Wx || !Wzx || !Wrx || !vpack_gate || !vpack_cand || !y_out || !y_state || !z_save || !r_save || !c_save || !status)
        return SPARCH_EINVAL;
    if ((sc == nullptr) != (sh == nullptr) || (scz == nullptr) != (shz == nullptr) || (scr == nullptr) != (shr == nullptr))
        return SPARCH_EINVAL;
    if (!(p_drop >= 0.0f && p_drop < 1.0f)) return SPARCH_EINVAL;
    if (!al16g({Wx, sc, sh, Wzx, scz, shz, Wrx, scr, shr, vpack_gate, vpack_cand, y_out, y_state, z_save, r_save, c_save, chan}))
        return SPARCH_EALIGN;
    LigruArgs a{};
    a.B = B; a.dirs = dirs; a.T = T; a.H = H; a.Bp = B * dirs;
    a.Wx = Wx; a.sc = sc; a.sh = sh; a.Wzx = Wzx; a.scz = scz; a.shz = shz; a.Wrx = Wrx; a.scr = scr; a.shr = shr;
    a.vpack = reinterpret_cast<const u32x4*>(vpack_gate); a.vpack2 = reinterpret_cast<const u32x4*>(vpack_cand);
    a.p_drop = p_drop; a.inv_keep = 1.0f / (1.0f - p_drop); a.seed = seed;
    a.y_out = y_out; a.y_state = y_state; a.z_save = z_save; a.r_save = r_save; a.c_save = c_save; a.status = status;
    return run_gru<false>(a, chan, chan_bytes, steps_per_launch, (hipStream_t)stream);
}

extern "C" int sparch_gru_bwd(int B, int dirs, int T, int H, const float* g_out, const float* y_state, const float* z_save,
                              const float* r_save, const float* c_save, const float* vpack_gate_b,
                              const float* vpack_cand_b, float p_drop, uint64_t seed, float* dz_all, float* dr_all,
                              float* dc_all, float* yprev_all, float* ry_all, float* carry, void* chan, size_t chan_bytes,
                              uint32_t* status, int steps_per_launch, void* stream) {
    SPARCH_ENTER();
    if (B <= 0 || T <= 0 || H <= 0 || H % 32 != 0 || (dirs != 1 && dirs != 2)) return SPARCH_EINVAL;
    if (!g_out || !y_state || !z_save || !r_save || !c_save || !vpack_gate_b || !vpack_cand_b || !dz_all || !dr_all ||
        !dc_all || !yprev_all || !ry_all || !carry || !status)
        return SPARCH_EINVAL;
    if (!(p_drop >= 0.0f && p_drop < 1.0f)) return SPARCH_EINVAL;
    if (!al16g({g_out, y_state, z_save, r_save, c_save, vpack_gate_b, vpack_cand_b, dz_all, dr_all, dc_all, yprev_all, ry_all,
                carry, chan}))
        return SPARCH_EALIGN;
    LigruArgs a{};
    a.B = B; a.dirs = dirs; a.T = T; a.H = H; a.Bp = B * dirs;
    a.g_out = g_out; a.y_state = const_cast<float*>(y_state); a.z_save = const_cast<float*>(z_save);
    a.r_save = const_cast<float*>(r_save); a.c_save = const_cast<float*>(c_save);
    a.vpack = reinterpret_cast<const u32x4*>(vpack_gate_b); a.vpack2 = reinterpret_cast<const u32x4*>(vpack_cand_b);
    a.p_drop = p_drop; a.inv_keep = 1.0f / (1.0f - p_drop); a.seed = seed;
    a.dz_all = dz_all; a.dr_all = dr_all; a.dc_all = dc_all; a.yprev_all = yprev_all; a.ry_all = ry_all;
    a.carry = carry; a.status = status;
    return run_gru<true>(a, chan, chan_bytes, steps_per_launch, (hipStream_t)stream);
}

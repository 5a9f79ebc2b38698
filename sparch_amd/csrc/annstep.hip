// f-4: the gated non-spiking baselines (LiGRULayer._ligru_cell anns.py:449-462, GRULayer._gru_cell 581-595),
// one time step per launch.
//
// These cells carry two or three recurrent matrices; their slices no longer fit the register-resident layout of
// the persistent recurrent kernels (reccell.hip), so this round runs them the way the reference does — a loop
// over time on the host — with this library's kernels inside the loop: the recurrent products
// y_{t-1} [Vz;V]^T (etc.) on the exact-split MFMA GEMMs and the gate arithmetic in the element-wise kernels
// below.  Launch-bound (4-8 launches per step), not a performance path; the arithmetic and the saved
// quantities are what a persistent version will reuse.
//
// Conventions as for the spiking cells: virtual rows bp = d*B + b (d = direction), row bp reads the
// projections of sample b at time tt = d ? T-1-t : t, `y_out` holds dropout(y) with the directions
// concatenated on features at the ORIGINAL time index; sequences saved for the weight-gradient GEMMs are
// stored at that index too.  BatchNorm is folded into per-matrix (scale, shift).
#include "common.h"

namespace {

struct GateArgs {
    int B, dirs, T, H, t;
    const float* Wx; const float* sc; const float* sh;      // candidate projection (B,T,H) + folded affine
    const float* Wzx; const float* scz; const float* shz;   // update gate
    const float* Wrx; const float* scr; const float* shr;   // reset gate (GRU)
    const float* rec;                                       // (Bp, 2H) or (Bp, H) recurrent product(s), NULL at t = 0
    float* y_state; float* z_save; float* r_save; float* c_save;  // (Bp,T,H), cell time order
    float* ry;                                              // (Bp,H) r * y_{t-1} (GRU, forward phase A out / B in)
    float* y_out; float p_drop, inv_keep; uint64_t seed;    // (B,T,H*dirs)
    // backward
    const float* g_out; const float* carry_mv; const float* carry_dir;  // (B,T,HO); (Bp,H) each, NULL at t = T-1
    float* carry_dir_out;                                   // (Bp,H)
    float* dgate;                                           // (Bp,2H) [dz_pre | dc_pre] (LiGRU) / [dz_pre | dr_pre] (GRU)
    float* dcp;                                             // (Bp,H) dc_pre of this step (GRU phase C out)
    const float* dry;                                       // (Bp,H) dc_pre V (GRU phase D in)
    float* dz_all; float* dr_all; float* dc_all; float* yprev_all; float* ry_all;  // (Bp,T,H), original time index
};

__device__ __forceinline__ float sigm(float v) { return 1.0f / (1.0f + expf(-v)); }

__device__ __forceinline__ f32x4 affine(const float* W, const float* sc, const float* sh, size_t o, int h) {
    f32x4 v = *reinterpret_cast<const f32x4*>(W + o);
    if (sc) {
        const f32x4 s = *reinterpret_cast<const f32x4*>(sc + h), b = *reinterpret_cast<const f32x4*>(sh + h);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = bn_affine(v[e], s[e], b[e]);
    }
    return v;
}
__device__ __forceinline__ f32x4 ld4g(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4g(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// MODE 0: LiGRU forward step.  1: GRU forward phase A (gates, r*y).  2: GRU forward phase B (candidate, y).
// 3: LiGRU backward step.  4: GRU backward phase C (dy, dz_pre, dc_pre).  5: GRU backward phase D (dr_pre).
template <int MODE>
__global__ __launch_bounds__(256) void gate_kernel(GateArgs a) {
    const int H = a.H, T = a.T, t = a.t, HO = a.H * a.dirs;
    const int q = H / 4;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.B * a.dirs * q) return;
    const int bp = i / q, h = (i - bp * q) * 4;
    const int d = bp / a.B, b = bp - d * a.B;
    const int tt = d ? (T - 1 - t) : t;
    const size_t o_in = ((size_t)b * T + tt) * H + h;          // projections of sample b at original time tt
    const size_t o_st = ((size_t)bp * T + t) * H + h;          // cell-time order
    const size_t o_or = ((size_t)bp * T + tt) * H + h;         // virtual row, original time index
    const size_t o_out = ((size_t)b * T + tt) * HO + (size_t)d * H + h;
    const size_t o_v = (size_t)bp * H + h;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const uint64_t seed = a.p_drop > 0.f ? resolve_seed(a.seed) : 0;
    const f32x4 yp = t > 0 ? ld4g(a.y_state + o_st - H) : zero;  // y_{t-1} (anns.py:452 / 584: zeros at t = 0)

    if constexpr (MODE == 0) {
        const f32x4 xz = affine(a.Wzx, a.scz, a.shz, o_in, h), xc = affine(a.Wx, a.sc, a.sh, o_in, h);
        const f32x4 rz = a.rec ? ld4g(a.rec + (size_t)bp * 2 * H + h) : zero;
        const f32x4 rc = a.rec ? ld4g(a.rec + (size_t)bp * 2 * H + H + h) : zero;
        f32x4 z, c, y, yo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            z[e] = sigm(xz[e] + rz[e]);                                   // anns.py:457
            c[e] = fmaxf(xc[e] + rc[e], 0.0f);                            // anns.py:458 (ReLU, line 388)
            y[e] = z[e] * yp[e] + (1.0f - z[e]) * c[e];                   // anns.py:459
            const float k = a.p_drop > 0.f ? keep_scale(seed, o_out + e, a.p_drop, a.inv_keep) : 1.0f;
            yo[e] = y[e] * k;
        }
        st4g(a.y_state + o_st, y); st4g(a.z_save + o_st, z); st4g(a.c_save + o_st, c); st4g(a.y_out + o_out, yo);
    } else if constexpr (MODE == 1) {
        const f32x4 xz = affine(a.Wzx, a.scz, a.shz, o_in, h), xr = affine(a.Wrx, a.scr, a.shr, o_in, h);
        const f32x4 rz = a.rec ? ld4g(a.rec + (size_t)bp * 2 * H + h) : zero;
        const f32x4 rr = a.rec ? ld4g(a.rec + (size_t)bp * 2 * H + H + h) : zero;
        f32x4 z, r, ry;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            z[e] = sigm(xz[e] + rz[e]);                                   // anns.py:589
            r[e] = sigm(xr[e] + rr[e]);                                   // anns.py:590
            ry[e] = r[e] * yp[e];
        }
        st4g(a.z_save + o_st, z); st4g(a.r_save + o_st, r); st4g(a.ry + o_v, ry);
    } else if constexpr (MODE == 2) {
        const f32x4 xc = affine(a.Wx, a.sc, a.sh, o_in, h);
        const f32x4 rc = a.rec ? ld4g(a.rec + o_v) : zero;
        const f32x4 z = ld4g(a.z_save + o_st);
        f32x4 c, y, yo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            c[e] = tanhf(xc[e] + rc[e]);                                  // anns.py:591 (Tanh, line 511)
            y[e] = z[e] * yp[e] + (1.0f - z[e]) * c[e];                   // anns.py:592
            const float k = a.p_drop > 0.f ? keep_scale(seed, o_out + e, a.p_drop, a.inv_keep) : 1.0f;
            yo[e] = y[e] * k;
        }
        st4g(a.y_state + o_st, y); st4g(a.c_save + o_st, c); st4g(a.y_out + o_out, yo);
    } else {
        // ---- backward: dy_t = dropout'(g_t) + what step t+1 sent back (matrix part + element-wise part)
        f32x4 dy = zero;
        if constexpr (MODE == 3 || MODE == 4) {
            const f32x4 g = ld4g(a.g_out + o_out);
            const f32x4 cm = a.carry_mv ? ld4g(a.carry_mv + o_v) : zero, cd = a.carry_dir ? ld4g(a.carry_dir + o_v) : zero;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float k = a.p_drop > 0.f ? keep_scale(seed, o_out + e, a.p_drop, a.inv_keep) : 1.0f;
                dy[e] = g[e] * k + cm[e] + cd[e];
            }
        }
        if constexpr (MODE == 3) {
            const f32x4 z = ld4g(a.z_save + o_st), c = ld4g(a.c_save + o_st);
            f32x4 dzp, dcp, cdo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dzp[e] = (dy[e] * (yp[e] - c[e])) * (z[e] * (1.0f - z[e]));
                dcp[e] = c[e] > 0.0f ? dy[e] * (1.0f - z[e]) : 0.0f;
                cdo[e] = dy[e] * z[e];
            }
            st4g(a.dgate + (size_t)bp * 2 * H + h, dzp); st4g(a.dgate + (size_t)bp * 2 * H + H + h, dcp);
            st4g(a.carry_dir_out + o_v, cdo);
            st4g(a.dz_all + o_or, dzp); st4g(a.dc_all + o_or, dcp); st4g(a.yprev_all + o_or, yp);
        } else if constexpr (MODE == 4) {
            const f32x4 z = ld4g(a.z_save + o_st), c = ld4g(a.c_save + o_st), r = ld4g(a.r_save + o_st);
            f32x4 dzp, dcp, cdo, ry;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dzp[e] = (dy[e] * (yp[e] - c[e])) * (z[e] * (1.0f - z[e]));
                dcp[e] = (dy[e] * (1.0f - z[e])) * (1.0f - c[e] * c[e]);
                cdo[e] = dy[e] * z[e];
                ry[e] = r[e] * yp[e];
            }
            st4g(a.dgate + (size_t)bp * 2 * H + h, dzp);   // dr_pre half is written by phase D
            st4g(a.dcp + o_v, dcp); st4g(a.carry_dir_out + o_v, cdo);
            st4g(a.dz_all + o_or, dzp); st4g(a.dc_all + o_or, dcp); st4g(a.yprev_all + o_or, yp);
            st4g(a.ry_all + o_or, ry);
        } else {  // MODE 5: dry = dc_pre V is the gradient of (r * y_{t-1})
            const f32x4 r = ld4g(a.r_save + o_st), dry = ld4g(a.dry + o_v);
            f32x4 drp, cdo = ld4g(a.carry_dir_out + o_v);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                drp[e] = (dry[e] * yp[e]) * (r[e] * (1.0f - r[e]));
                cdo[e] = cdo[e] + dry[e] * r[e];
            }
            st4g(a.dgate + (size_t)bp * 2 * H + H + h, drp);
            st4g(a.carry_dir_out + o_v, cdo);
            st4g(a.dr_all + o_or, drp);
        }
    }
}

template <int MODE>
int launch_gate(const GateArgs& a, hipStream_t st) {
    const int n = a.B * a.dirs * (a.H / 4);
    hipLaunchKernelGGL(gate_kernel<MODE>, dim3((n + 255) / 256), dim3(256), 0, st, a);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

bool base_ok(int B, int dirs, int T, int H, int t) {
    return B > 0 && T > 0 && H > 0 && H % 4 == 0 && (dirs == 1 || dirs == 2) && t >= 0 && t < T;
}

}  // namespace

extern "C" int sparch_gate_step(int mode, int B, int dirs, int T, int H, int t, const float* const* in,
                                float* const* out, float p_drop, uint64_t seed, void* stream) {
    SPARCH_ENTER();
    if (!base_ok(B, dirs, T, H, t) || !in || !out || mode < 0 || mode > 5) return SPARCH_EINVAL;
    if (!(p_drop >= 0.0f && p_drop < 1.0f)) return SPARCH_EINVAL;
    GateArgs a{};
    a.B = B; a.dirs = dirs; a.T = T; a.H = H; a.t = t;
    a.p_drop = p_drop; a.inv_keep = 1.0f / (1.0f - p_drop); a.seed = seed;
    // in[]:  0 Wx 1 sc 2 sh 3 Wzx 4 scz 5 shz 6 Wrx 7 scr 8 shr 9 rec 10 g_out 11 carry_mv 12 carry_dir 13 dry
    // out[]: 0 y_state 1 z_save 2 r_save 3 c_save 4 ry 5 y_out 6 carry_dir_out 7 dgate 8 dcp
    //        9 dz_all 10 dr_all 11 dc_all 12 yprev_all 13 ry_all
    a.Wx = in[0]; a.sc = in[1]; a.sh = in[2]; a.Wzx = in[3]; a.scz = in[4]; a.shz = in[5];
    a.Wrx = in[6]; a.scr = in[7]; a.shr = in[8]; a.rec = in[9]; a.g_out = in[10]; a.carry_mv = in[11];
    a.carry_dir = in[12]; a.dry = in[13];
    a.y_state = out[0]; a.z_save = out[1]; a.r_save = out[2]; a.c_save = out[3]; a.ry = out[4]; a.y_out = out[5];
    a.carry_dir_out = out[6]; a.dgate = out[7]; a.dcp = out[8]; a.dz_all = out[9]; a.dr_all = out[10];
    a.dc_all = out[11]; a.yprev_all = out[12]; a.ry_all = out[13];
    auto need = [](std::initializer_list<const void*> ps) { for (auto p : ps) if (!p) return false; return true; };
    hipStream_t st = (hipStream_t)stream;
    switch (mode) {
        case 0: if (!need({a.Wx, a.Wzx, a.y_state, a.z_save, a.c_save, a.y_out})) return SPARCH_EINVAL; return launch_gate<0>(a, st);
        case 1: if (!need({a.Wzx, a.Wrx, a.y_state, a.z_save, a.r_save, a.ry})) return SPARCH_EINVAL; return launch_gate<1>(a, st);
        case 2: if (!need({a.Wx, a.y_state, a.z_save, a.c_save, a.y_out})) return SPARCH_EINVAL; return launch_gate<2>(a, st);
        case 3: if (!need({a.g_out, a.y_state, a.z_save, a.c_save, a.carry_dir_out, a.dgate, a.dz_all, a.dc_all, a.yprev_all})) return SPARCH_EINVAL; return launch_gate<3>(a, st);
        case 4: if (!need({a.g_out, a.y_state, a.z_save, a.r_save, a.c_save, a.carry_dir_out, a.dgate, a.dcp, a.dz_all, a.dc_all, a.yprev_all, a.ry_all})) return SPARCH_EINVAL; return launch_gate<4>(a, st);
        default: if (!need({a.dry, a.y_state, a.r_save, a.carry_dir_out, a.dgate, a.dr_all})) return SPARCH_EINVAL; return launch_gate<5>(a, st);
    }
}

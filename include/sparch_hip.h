/*
 * sparch_hip.h — C ABI of libsparch_hip.so, the MI355X (gfx950) implementation of
 * sparch's recurrent spiking-layer training path.
 *
 * The reference (thebarnable/sparch) has NO native/FFI interface: its hot path is
 * eager PyTorch inside Python `for t in range(T)` loops.  Each entry point below
 * therefore replaces a *group of eager ops* of the reference, cited as
 * snns.py:LINE (= /root/reference/sparch/models/snns.py) or exp.py:LINE.  The
 * Python binding a maintainer adds is a ctypes stub; ours lives in
 * sparch_amd/_capi.py.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch allocator);
 *     the library never allocates, frees, copies to host or synchronises;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*);
 *   - tensors are fp32, dense, row-major; (B,T,H) means b-major, h contiguous;
 *   - return value: 0 = SPARCH_OK, negative = error (nothing was launched);
 *     sparch_strerror() maps codes to text.  Python maps them to ValueError /
 *     RuntimeError, the reference's error convention;
 *   - re-entrant per stream; no global mutable state.
 *
 * "kind" of a spiking cell: 0 LIF, 1 adLIF, 2 RLIF, 3 RadLIF
 *   bit0 = adaptive (w state, beta/a/b), bit1 = recurrent (V).
 */
#ifndef SPARCH_HIP_H
#define SPARCH_HIP_H

#include <stddef.h>
#include <stdint.h>

#define SPARCH_SEED_IN_MEMORY (1ull << 63)

#ifdef __cplusplus
extern "C" {
#endif

#define SPARCH_OK 0
#define SPARCH_EINVAL (-1)      /* bad shape / null pointer / unsupported size        */
#define SPARCH_EALIGN (-2)      /* pointer or leading dimension not 16-byte aligned   */
#define SPARCH_EWORKSPACE (-3)  /* workspace too small (see *_workspace_bytes)        */
#define SPARCH_ELAUNCH (-4)     /* HIP reported a launch error                        */
#define SPARCH_ETIMEOUT (-5)    /* reported through a status word: in-kernel wait gave up */

/* The status word of a device: FOUR uint32 the caller keeps zeroed.  [0] != 0: an in-kernel wait of a persistent
 * recurrent launch gave up (SPARCH_ETIMEOUT) and the results of that training step are invalid;
 * [1] the number of optimizer steps sparch_adam_step skipped while [0] was raised; [2] which kernel raised it
 * (SPARCH_STATUS_*); [3] the time step, in processing order, it gave up at (0xFFFFFFFF: not recorded). */
#define SPARCH_STATUS_WORDS 4
#define SPARCH_STATUS_REC_FWD 1
#define SPARCH_STATUS_REC_BWD 2
#define SPARCH_STATUS_ANN_REC_FWD 3
#define SPARCH_STATUS_ANN_REC_BWD 4
#define SPARCH_STATUS_LIGRU_FWD 5
#define SPARCH_STATUS_LIGRU_BWD 6
#define SPARCH_STATUS_GRU_FWD 7
#define SPARCH_STATUS_GRU_BWD 8

#define SPARCH_KIND_LIF 0
#define SPARCH_KIND_ADLIF 1
#define SPARCH_KIND_RLIF 2
#define SPARCH_KIND_RADLIF 3

int sparch_abi_version(void);
const char* sparch_strerror(int code);
/* Text of the HIP error behind the calling thread's most recent SPARCH_ELAUNCH. */
const char* sparch_last_hip_error(void);
/* Number of compute units / XCDs the library sizes its persistent grids for. */
int sparch_device_cus(void);

/* Operand precision of a matrix product: the `precision` argument — the last one — of every entry point that
 * multiplies (the split GEMMs below, the V packs, the recurrent cells' s_{t-1} @ V / dWx_{t+1} @ V^T, G3/G4).  The
 * reference has no such choice (it is fp32-only: snns.py:29 casts the spikes to float and snns.py:572 then requires
 * a float V); BASELINE.json configs[4] names a bf16 run.
 *   SPARCH_PRECISION_FP32_EXACT: fp32 operands split exactly into bf16 planes, exact products, fp32 accumulation —
 *       fp32 results.
 *   SPARCH_PRECISION_BF16: every fp32 operand is rounded ONCE to bf16 (nearest-even) as it is staged, one bf16
 *       MFMA per product, fp32 accumulation; states, statistics, outputs and parameter updates stay fp32.
 *       Spike operands (0 / 1) and bf16-representable weights lose nothing: a network whose weights are
 *       bf16-exact has a bit-identical forward pass in both modes.
 * Per call (ABI v5): the library keeps no precision state — rounds 1-2 had a process-wide
 * sparch_set_operand_precision().  A V pack must be consumed by cell calls of the precision it was made with; a
 * *_workspace_bytes query takes the precision of the call it sizes.  Unknown value: SPARCH_EINVAL (0 bytes). */
#define SPARCH_PRECISION_FP32_EXACT 0
#define SPARCH_PRECISION_BF16 1

/* ------------------------------------------------------------------------------------
 * G1  feed-forward projection  (replaces `self.W(x)` = nn.Linear, snns.py:261/398/533/675/796,
 *     and its autograd backward).  Hand-written fp32 MFMA (v_mfma_f32_32x32x2_f32) GEMMs.
 * ---------------------------------------------------------------------------------- */

/* C[M,N] = A[M,K] * B[N,K]^T (+ bias[N]).  Optional fused column statistics for
 * BatchNorm: if colstat_ws != NULL it receives per-row-tile partial sums
 * (2 * ceil(M/128) * N floats: [tile][N] sums then [tile][N] sums of squares) of the
 * values written to C, to be finished by sparch_bn_finalize().                       */
int sparch_gemm_nt(int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                   float* C, int ldc, const float* bias, float* colstat_ws, void* stream);

/* C[M,N] = A[M,K] * B[K,N]          (dX = dWx * W; rec0 = s0 * Vmasked)            */
int sparch_gemm_nn(int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                   float* C, int ldc, void* stream);

/* C[M,N] = A[K,M]^T * B[K,N], contraction over the long K axis (dW = dWx^T * x,
 * dV = s_prev^T * dWx).  Split-K over `splits` slabs held in `ws`
 * (sparch_gemm_tn_workspace_bytes), reduced in fixed order => bitwise reproducible.  */
size_t sparch_gemm_tn_workspace_bytes(int M, int N, int K);
int sparch_gemm_tn(int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                   float* C, int ldc, int zero_diag, int accumulate /* C += */, void* ws,
                   size_t ws_bytes, void* stream);

/* The same products when one operand is a spike tensor (entries 0 or one constant c, as produced
 * by the cell kernels: c = 1/(1-p_drop)): the spike operand is used as (x != 0) in bf16, the fp32
 * operand is split exactly into three bf16 planes, products are exact on the bf16 MFMA with fp32
 * accumulation, and `scale` (= c) is applied once in the epilogue.
 *   spike_nt: C[M,N] = scale * spike(A)[M,K] * B[N,K]^T (+bias) (+BatchNorm column statistics)
 *   spike_tn: C[M,N] (+)= scale * A[K,M]^T * B[K,N], spike_side 0: A is the spike operand, 1: B. */
int sparch_gemm_spike_nt(int M, int N, int K, const float* A_spk, int lda, float scale,
                         const float* B, int ldb, float* C, int ldc, const float* bias,
                         float* colstat_ws, void* stream, int precision);
size_t sparch_gemm_spike_tn_workspace_bytes(int M, int N, int K, int precision);

/* The same products with the spike operand given as a bf16 PLANE (uint16 bit patterns, entries 0 or
 * 0x3F80 = 1.0; any bf16 value is taken as is): the cell kernels write this plane next to their fp32
 * output (s16 below), and the GEMMs then pull half the operand bytes.  lda / ldb of the plane are in
 * elements.  spike16_tn: the operand named by spike_side is the uint16 plane, the other one fp32;
 * workspace as sparch_gemm_spike_tn_workspace_bytes.                                               */
int sparch_gemm_spike16_nt(int M, int N, int K, const uint16_t* A_spk16, int lda, float scale,
                           const float* B, int ldb, float* C, int ldc, const float* bias,
                           float* colstat_ws, void* stream, int precision);
int sparch_gemm_spike16_tn(int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                           int spike_side, float scale, float* C, int ldc, int zero_diag,
                           int accumulate, void* ws, size_t ws_bytes, void* stream, int precision);

/* Dense x dense products on the same exact-split machinery: BOTH fp32 operands are split into three
 * bf16 planes and the six largest cross terms are accumulated in fp32 (the rest measures ~1e-8 of
 * sum|a||b|, below an fp32 sgemm's own rounding): fp32-faithful results at 6/16 of the fp32-MFMA cost.  Same argument meaning as
 * sparch_gemm_nt / _nn / _tn (the tn workspace is sparch_gemm_spike_tn_workspace_bytes).          */
/* Split-K forms for small M*N with a long K (per-step recurrent products of the gated baselines): the
 * contraction is cut into slabs in `ws` (sparch_gemm6_splitk_workspace_bytes; 0 = not needed) and reduced in
 * fixed order.  No bias / statistics epilogue.                                                         */
size_t sparch_gemm6_splitk_workspace_bytes(int M, int N, int K, int precision);
int sparch_gemm6_nt_splitk(int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                           float* C, int ldc, void* ws, size_t ws_bytes, void* stream, int precision);
int sparch_gemm6_nn_splitk(int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                           float* C, int ldc, void* ws, size_t ws_bytes, void* stream, int precision);
int sparch_gemm6_nt(int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                    float* C, int ldc, const float* bias, float* colstat_ws, void* stream, int precision);
int sparch_gemm6_nn(int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                    float* C, int ldc, void* stream, int precision);
int sparch_gemm6_tn(int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                    float* C, int ldc, int zero_diag, int accumulate, void* ws, size_t ws_bytes,
                    void* stream, int precision);

/* Weight operands pre-split into their bf16 planes.  A spiking layer's W (H x K) is the B operand of its
 * projection (snns.py:261, x W^T) and of backward's dx = dWx W; both kernels would re-convert the same
 * tile of W in every workgroup that stages it (250 row tiles at B*T = 32000).  sparch_split3 writes the
 * exact truncation split x = p0 + p1 + p2 once: planes[p*n + i], bf16 bit patterns, n % 8 == 0.  The _wp
 * entries take B together with those planes (same layout and ldb; B_planes may be NULL) and return the
 * same bits as sparch_gemm_spike16_nt / sparch_gemm6_nn: where the pipelined kernel applies (K % 32 == 0,
 * ldb % 8 == 0, full tiles) it copies the planes into LDS, elsewhere it converts B as before.        */
int sparch_split3(size_t n, const float* x, uint16_t* planes, void* stream);
int sparch_gemm_spike16_nt_wp(int M, int N, int K, const uint16_t* A_spk16, int lda, float scale,
                              const float* B, const uint16_t* B_planes, int ldb, float* C, int ldc,
                              const float* bias, float* colstat_ws, void* stream, int precision);
int sparch_gemm6_nn_wp(int M, int N, int K, const float* A, int lda, const float* B,
                       const uint16_t* B_planes, int ldb, float* C, int ldc, void* stream, int precision);

/* First-layer input (snns.py:261 on the network input): SHD/SSC-style binned spike counts are small
 * integers, exactly representable in bf16, but the library cannot know that on the host.
 * sparch_flag_bf16_exact sets *flag (device uint32) to 1 iff every element of x is bf16-exact; the
 * _auto_ GEMMs then enqueue BOTH the single-plane exact kernel and the 6-term kernel, each gated on
 * the device by *flag, so exactly one runs — no host round trip.  nt: A is the flagged operand;
 * tn: B is (dW = dWx^T x).                                                                       */
int sparch_flag_bf16_exact(size_t n, const float* x, uint32_t* flag, void* stream);
int sparch_gemm_auto_nt(int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                        float* C, int ldc, const float* bias, float* colstat_ws,
                        const uint32_t* a_exact_flag, void* stream, int precision);
int sparch_gemm_auto_tn(int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                        float* C, int ldc, int zero_diag, int accumulate,
                        const uint32_t* b_exact_flag, void* ws, size_t ws_bytes, void* stream, int precision);
/* The same with the flagged operand's bf16 plane made by the check itself: sparch_plane_bf16_exact writes
 * plane[m][k] = upper 16 bits of x[m][k] (the exact value whenever the flag stays 1; rows of ldp >= K elements,
 * ldp % 8 == 0, columns K.. zero) in the pass that computes the flag, and the _auto16_ GEMMs read that plane
 * (2 bytes per element through the spike-plane kernels) when *flag == 1, the fp32 operand through the six-term
 * kernels otherwise.  One pass over the network input per step instead of three. */
int sparch_plane_bf16_exact(int M, int K, const float* x, int ldx, uint16_t* plane, int ldp, uint32_t* flag,
                            void* stream);
int sparch_gemm_auto16_nt(int M, int N, int K, const float* A, int lda, const uint16_t* A16, int lda16,
                          const float* B, int ldb, float* C, int ldc, const float* bias, float* colstat_ws,
                          const uint32_t* a_exact_flag, void* stream, int precision);
int sparch_gemm_auto16_tn(int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                          const uint16_t* B16, int ldb16, float* C, int ldc, int zero_diag, int accumulate,
                          const uint32_t* b_exact_flag, void* ws, size_t ws_bytes, void* stream, int precision);
int sparch_gemm_spike_tn(int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                         int spike_side, float scale, float* C, int ldc, int zero_diag,
                         int accumulate, void* ws, size_t ws_bytes, void* stream, int precision);

/* ------------------------------------------------------------------------------------
 * G2  normalisation on the (M = B*T, H) view  (replaces nn.BatchNorm1d(momentum=0.05) /
 *     nn.LayerNorm, snns.py:240-243, 264-266).  BatchNorm is folded into a per-column
 *     (scale, shift) that the cell kernels apply on load.
 * ---------------------------------------------------------------------------------- */

/* Train: finish the statistics from colstat_ws (n_tiles partial rows, count = M rows,
 * `dup` = 2 when a bidirectional layer sees every row twice: running_var uses
 * n = dup*M), write scale = gamma*invstd, shift = beta - mean*scale, save mean/invstd,
 * update running stats in place (momentum m, unbiased variance) — unless the device word
 * skip_if_nonzero (nullable; the recurrent kernels' status word) is non-zero.
 * num_batches_tracked (nullable, ABI v5): BatchNorm1d's int64 counter, incremented by the same launch under the
 * same guard (the reference's `norm(...)` call does it, snns.py:264; a separate host-side `+= 1` was one more
 * kernel per layer and advanced on skipped steps too).
 * Eval (training = 0): scale/shift from running stats; colstat_ws ignored.           */
int sparch_bn_finalize(int H, int M, int n_tiles, int dup, const float* colstat_ws,
                       const float* gamma, const float* beta, float* running_mean,
                       float* running_var, float momentum, float eps, int training,
                       float* scale, float* shift, float* save_mean, float* save_invstd,
                       const uint32_t* skip_if_nonzero, int64_t* num_batches_tracked, void* stream);

/* Column sums of dy and dy*xhat over M rows (xhat = (x-mean)*invstd), partials in ws
 * (2 * ceil(M/256) * H floats), finished into dgamma[H], dbeta[H].                   */
size_t sparch_bn_bwd_workspace_bytes(int M, int H);
int sparch_bn_bwd_reduce(int M, int H, const float* dy, const float* x, const float* mean,
                         const float* invstd, float* dgamma, float* dbeta, void* ws,
                         size_t ws_bytes, void* stream);
/* dx = scale * (dy - dbeta/M - xhat * dgamma/M), in place allowed (dx == dy).        */
int sparch_bn_bwd_apply(int M, int H, const float* dy, const float* x, const float* mean,
                        const float* invstd, const float* gamma, const float* dgamma,
                        const float* dbeta, float* dx, void* stream);

/* LayerNorm per row over the leading Hn of H columns (Hn == H normally; Hn < H: columns
 * Hn..H-1 are zero padding of a layer run at a padded width — y and dx are written 0 there
 * and they take no part in the statistics): y = (x-mu)*rstd*gamma + beta; saves mu, rstd
 * (M each).                                                                            */
int sparch_layernorm_fwd(int M, int H, int Hn, const float* x, const float* gamma, const float* beta,
                         float eps, float* y, float* mu, float* rstd, void* stream);
/* dx per row; dgamma/dbeta column sums via ws (2*ceil(M/256)*H floats).               */
int sparch_layernorm_bwd(int M, int H, int Hn, const float* dy, const float* x, const float* mu,
                         const float* rstd, const float* gamma, float* dx, float* dgamma,
                         float* dbeta, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * G3/G4/G6/G7  spiking cells, time loop inside the kernel  (replaces _lif_cell
 *     snns.py:282-303, _adlif_cell 419-445, _rlif_cell 554-578, _radlif_cell 696-727,
 *     SpikeFunctionBoxcar 20-36, the bidirectional flip/cat glue 252-254 / 272-275 and
 *     nn.Dropout 278, plus their autograd replay).
 *
 * Geometry: B = batch, dirs = 1 | 2 (bidirectional), Bp = B*dirs virtual rows.
 *   Wx      (B,T,H)  raw projection; the cell applies x*scale[h]+shift[h] when scale != NULL.
 *           Virtual row b' >= B reads Wx[b'-B, T-1-t] (time-flipped copy, never materialised).
 *   u0,w0,s0 (Bp,H)  random initial states drawn by the host in the reference's order.
 *   s_out   (B,T,H*dirs)  post-dropout output; direction d lands in features [d*H,(d+1)*H)
 *           at its ORIGINAL time index (un-flipped), as snns.py:272-275.  NULL => not written (ABI v5): a
 *           caller that feeds the next layer from s16_out alone saves the 4 bytes per element; at least
 *           one of s_out / s16_out must be given.
 *   s16_out (B,T,H*dirs)  the same spikes as a bf16 plane (uint16 bit patterns: 0x3F80 where
 *           s_out != 0, else 0) for sparch_gemm_spike16_*; NULL => not written.
 *   u_save, w_save (Bp,T,H) in cell time order; needed by the backward (NULL => not saved).
 *   spike_count (H*dirs) uint32: number of spikes surviving dropout per output feature;
 *           firing rate = count * keep_scale / (B*T)  (snns.py:174 after 278).
 *   Dropout: keep iff hash(seed, output element index) >= p; kept values scale by 1/(1-p).
 *   `seed` (every entry point that takes one): the 63-bit seed itself, or SPARCH_SEED_IN_MEMORY | the device
 *   address of a uint64 holding it — for a step captured in a HIP graph, whose arguments are frozen.
 * ---------------------------------------------------------------------------------- */

/* Non-recurrent kinds (LIF, adLIF): one thread per (row, 4 columns), T-loop in registers. */
int sparch_cell_fwd(int kind, int B, int dirs, int T, int H, const float* Wx,
                    const float* scale, const float* shift, const float* alpha,
                    const float* beta, const float* a, const float* b, const float* u0,
                    const float* w0, const float* s0, float theta, float p_drop,
                    uint64_t seed, float* s_out, uint16_t* s16_out, void* u_save, void* w_save,
                    int save_bf16, uint32_t* spike_count, void* stream);

/* save_bf16 (forward and backward alike): u_save / w_save are (Bp,T,H) bf16 instead of fp32 — half the bytes
 * of the two largest tensors a layer keeps for its backward pass.  The stored membrane potential is rounded
 * so that the backward's three discrete decisions (spike u-theta > 0, box-car edges) are EXACTLY the fp32
 * ones (csrc/common.h save_u16): recomputed spikes never flip, dWx / dW / dV are unchanged, only the
 * continuous uses of u and w (dalpha, dbeta, da) carry the 2^-9 relative rounding.  The recurrent kernels
 * accept it for whole-sequence launches only (steps_per_launch >= T).
 * g_out (B,T,H*dirs) upstream gradient of s_out; g_rate (H*dirs) upstream gradient of the
 * firing rates (NULL = none).  dWx (Bp,T,H): gradient w.r.t. the normalised projection,
 * virtual-row order but ORIGINAL time index (so rows b and b+B add elementwise).
 * dparam_ws: (4,Bp,H) per-row partials of (dalpha,dbeta,da,db); finish with
 * sparch_colsum_clamped().
 * BatchNorm backward folded in (SURVEY §8 b2 `bn_dsum` / `bn_dxhat`): with bn_x = the raw projection
 * (B,T,H), bn_mean / bn_invstd (H) non-NULL the kernel also leaves sum_t dWx and sum_t dWx*xhat,
 * xhat = (x - mean)*invstd, per (row, column) in planes 4 and 5 of dparam_ws ((6,Bp,H) then): their
 * column sums are BatchNorm's dbeta / dgamma — no separate pass over dy and x.            */
int sparch_cell_bwd(int kind, int B, int dirs, int T, int H, const float* g_out,
                    const float* g_rate, const void* u_save, const void* w_save, int save_bf16,
                    const float* alpha, const float* beta, const float* a, const float* b,
                    const float* u0, const float* w0, const float* s0, float theta,
                    float p_drop, uint64_t seed, float* dWx, float* dparam_ws, const float* bn_x,
                    const float* bn_mean, const float* bn_invstd, void* stream);

/* Recurrent kinds (RLIF, RadLIF).  V (H,H) = V.weight; the diagonal is masked inside
 * (snns.py:566/712).  The recurrent product s_{t-1}*V runs on MFMA with the V slice of
 * each workgroup resident in registers; workgroups of one 32-row batch tile exchange
 * spikes through `chan` (tagged 8-byte granules, agent-scope write-through).
 *   vpack   prepacked V from sparch_vpack (forward: transpose=0, backward: transpose=1)
 *   rec0    (Bp,H) = s0 * Vmasked, the t=0 recurrent drive (s0 is not binary)
 *   chan    sparch_rec_chan_bytes(Bp,T,H) bytes, zeroed by the call itself
 *   status  the device's status word (SPARCH_STATUS_WORDS uint32, see above): [0] set non-zero if an in-kernel
 *           wait gave up (SPARCH_ETIMEOUT), [2] / [3] which kernel and at which step
 *   steps_per_launch: T => one persistent launch per batch-tile group; 1 => one launch
 *           per time step (no inter-workgroup waiting at all; the safe fallback).     */
size_t sparch_vpack_bytes(int H);
/* transpose: bit 0 = pack V^T, bit 1 = keep the diagonal (dense cells of the ANN baselines) */
int sparch_vpack(int H, const float* V, int transpose, float* vpack, float* vmasked,
                 void* stream, int precision);
/* The forward fragments, the backward (transposed) fragments and the masked copy in one launch (a training step
 * needs all three; vmasked may be NULL).                                                                    */
int sparch_vpack_both(int H, const float* V, float* vpack_fwd, float* vpack_bwd, float* vmasked, void* stream, int precision);
/* vmasked (H,H) = V with its diagonal zeroed (snns.py:566/712), any H */
int sparch_vmask(int H, const float* V, float* vmasked, void* stream);
/* XCD-local hand-off stores of the spiking recurrent kernels (whole-sequence launches): the workgroups of a row
 * tile establish at run time (HW_REG_XCC_ID exchanged through agent-scope accesses) that they share an XCD and
 * then hand their tiles over with plain stores, which stay in that XCD's L2; anything else keeps the
 * write-through agent-scope stores.  Same results bit for bit; on by default (SPARCH_XCD_LOCAL=0 turns it off).
 * sparch_set_xcd_local(0 / 1) overrides, -1 returns to the environment's setting; returns the previous state.  */
int sparch_set_xcd_local(int on);
size_t sparch_rec_chan_bytes(int Bp, int T, int H);
int sparch_rec_cell_fwd(int kind, int B, int dirs, int T, int H, const float* Wx,
                        const float* scale, const float* shift, const float* alpha,
                        const float* beta, const float* a, const float* b,
                        const float* vpack, const float* rec0, const float* u0,
                        const float* w0, const float* s0, float theta, float p_drop,
                        uint64_t seed, float* s_out, uint16_t* s16_out, void* u_save,
                        void* w_save, int save_bf16, uint32_t* spike_count, void* chan,
                        size_t chan_bytes, uint32_t* status, int steps_per_launch, void* stream, int precision);
/* Backward: each step's 32x32 dWx tile is handed to the other workgroups through `chan`.
 * s_prev16 (Bp,T,H) receives s_{t-1} as a bf16 plane (binary for t >= 1, a zero row at
 * t = 0: the non-binary s0 term is added by the caller) for dV = s_prev^T * dWx
 * (sparch_gemm_spike16_tn, spike_side 0).  dparam_ws: (6,Bp,H) = per-row partials of
 * (dalpha,dbeta,da,db) + the du / dw carries of chunked launches; with bn_x / bn_mean / bn_invstd
 * non-NULL (see sparch_cell_bwd) planes 6 and 7 receive BatchNorm's sums ((8,Bp,H) then).   */
int sparch_rec_cell_bwd(int kind, int B, int dirs, int T, int H, const float* g_out,
                        const float* g_rate, const void* u_save, const void* w_save, int save_bf16,
                        const float* alpha, const float* beta, const float* a,
                        const float* b, const float* vpack_t, const float* u0,
                        const float* w0, const float* s0, float theta, float p_drop,
                        uint64_t seed, float* dWx, uint16_t* s_prev16, float* dparam_ws,
                        const float* bn_x, const float* bn_mean, const float* bn_invstd,
                        void* chan, size_t chan_bytes, uint32_t* status,
                        int steps_per_launch, void* stream, int precision);
/* ONE time step t of the same cells with the recurrent product supplied by the caller — the path for hidden
 * sizes whose V slice does not fit the persistent kernels' register-resident layout (H > 1024; the reference
 * accepts any nb_hiddens, snns.py:608-661): any grid size, nothing waits inside a launch.
 *   forward : rec (Bp,H) = s_{t-1} @ V  (t = 0: s0 @ V, V with its diagonal zeroed); the step's raw
 *             (pre-dropout) spikes also go to s_step16 (Bp,H) bf16 0/1, the operand of the next product;
 *             membrane / adaptation state is carried through u_save / w_save (row t-1 is read)
 *   backward: steps run t = T-1 ... 0; rec (Bp,H) = dWx_{t+1} @ V^T (ignored at t = T-1); the step's dWx
 *             also goes to dwx_step (Bp,H); du / dw carries and the parameter partial sums live in
 *             dparam_ws (6 planes of Bp*H) between calls.                                          */
int sparch_rec_cell_step_fwd(int kind, int B, int dirs, int T, int H, int t, const float* Wx,
                             const float* scale, const float* shift, const float* alpha,
                             const float* beta, const float* a, const float* b, const float* rec,
                             const float* u0, const float* w0, const float* s0, float theta,
                             float p_drop, uint64_t seed, float* s_out, uint16_t* s16_out,
                             float* u_save, float* w_save, uint32_t* spike_count,
                             uint16_t* s_step16, void* stream);
int sparch_rec_cell_step_bwd(int kind, int B, int dirs, int T, int H, int t, const float* g_out,
                             const float* g_rate, const float* u_save, const float* w_save,
                             const float* alpha, const float* beta, const float* a, const float* b,
                             const float* rec, const float* u0, const float* w0, const float* s0,
                             float theta, float p_drop, uint64_t seed, float* dWx,
                             uint16_t* s_prev16, float* dparam_ws, const float* bn_x,
                             const float* bn_mean, const float* bn_invstd, float* dwx_step, void* stream);

/* Finish per-row partials: out[j][h] = sum_r ws[j][r][h], zeroed where the raw parameter
 * lies outside [lo_j, hi_j] (torch.clamp's gradient gate).  n_params <= 8; raw[j]/lim may
 * be NULL for "no clamp".                                                             */
int sparch_colsum_clamped(int n_params, int rows, int H, const float* ws,
                          const float* const* raw, const float* lim_lo_hi, float* const* out,
                          void* stream);

/* Elementwise helpers of the layer backward: out = x[0:B] + x[B:2B] (bidirectional
 * halves of dWx share their projection rows), M*H elements.                           */
int sparch_add_halves(size_t n, const float* x, float* out, void* stream);
/* Column sums out[h] = sum_m x[m][h] (bias gradient), fixed order.                    */
int sparch_colsum(int M, int H, const float* x, float* out, void* ws, size_t ws_bytes,
                  void* stream);

/* ------------------------------------------------------------------------------------
 * G5  readout cell  (replaces _readout_cell, snns.py:808-825, and its autograd replay):
 *     u_t = alpha*u + (1-alpha)*(Wx_t*scale+shift);  out += softmax(u_t).
 *     One workgroup per batch row, classes on threads (C <= 256).
 *     Backward: dalpha_ws (1,B,C) per-row partials of dalpha; with bn_x (B,T,C) / bn_mean / bn_invstd
 *     non-NULL also sum_t dWx and sum_t dWx*xhat in planes 1 and 2 ((3,B,C) then), see sparch_cell_bwd.
 * ---------------------------------------------------------------------------------- */
int sparch_readout_fwd(int B, int T, int C, const float* Wx, const float* scale,
                       const float* shift, const float* alpha, const float* u0, float* out,
                       float* u_save, void* stream);
int sparch_readout_bwd(int B, int T, int C, const float* g_out, const float* bn_x,
                       const float* bn_mean, const float* bn_invstd, const float* u_save,
                       const float* alpha, const float* u0, float* dWx, float* dalpha_ws,
                       void* stream);

/* ------------------------------------------------------------------------------------
 * G9  mel filterbank front-end  (replaces torchaudio.compliance.kaldi.fbank(x,
 *     num_mel_bins=40) at nonspiking_datasets.py:96,194; third-party, parity unpinned).
 *     wave (n_clips, n_samples) fp32 in [-1,1], 16 kHz -> out (n_clips, frames, n_mels),
 *     frames = 1 + (n_samples-400)/160.                                              */
int sparch_fbank_frames(int n_samples);
int sparch_fbank_fwd(int n_clips, int n_samples, int n_mels, const float* wave, float* out,
                     void* stream);

/* ------------------------------------------------------------------------------------
 * f-3  SHD/SSC event lists -> dense binned spike counts (replaces SpikingDataset.__getitem__,
 *      spiking_datasets.py:66-78: np.digitize into np.linspace(0, max_time, nb_steps) edges, then a
 *      sparse->dense scatter in which duplicates add up).  Events of all samples are concatenated;
 *      sample_offsets (n_samples+1, device int64) delimits them.  out (n_samples, nb_steps, nb_units) is
 *      zeroed by the call.  Events the reference's sparse constructor would reject (t < 0, t >= max_time,
 *      unit out of range) are skipped and counted in *n_dropped (device uint32).                    */
int sparch_bin_events(long long n_events, const float* times, const int* units,
                      const long long* sample_offsets, int n_samples, int nb_steps, int nb_units,
                      double max_time, float* out, uint32_t* n_dropped, void* stream);

/* ---- f-4: non-spiking baselines (anns.py) --------------------------------------------------
 * Element-wise tail of MLPLayer.forward (anns.py:218-227): y = dropout(act(z * scale + shift)) over n
 * elements of an (n/H, H) tensor; scale/shift (H) = the folded BatchNorm affine, NULL for none.
 * Backward: dz = dy * keep * act'(z * scale + shift) — the gradient w.r.t. the NORMALISED
 * pre-activation (feed it to the norm's backward).  H % 4 == 0.                                   */
#define SPARCH_ACT_SIGMOID 0
#define SPARCH_ACT_RELU 1
#define SPARCH_ACT_TANH 2
int sparch_act_fwd(int kind, size_t n, int H, const float* z, const float* scale, const float* shift,
                   float p_drop, uint64_t seed, float* y, void* stream);
int sparch_act_bwd(int kind, size_t n, int H, const float* z, const float* scale, const float* shift,
                   const float* dy, float p_drop, uint64_t seed, float* dz, void* stream);
/* ReadoutLayerANN._readout_cell (anns.py:658-665): out[b,:] = sum_t softmax(x[b,t,:]) (softmax over the
 * K features, accumulated in time order); backward dx[b,t,:] = p_t * (g[b,:] - <p_t, g[b,:]>).
 * K % 4 == 0, K <= 4096.                                                                          */
int sparch_softmax_sum_fwd(int B, int T, int K, const float* x, float* out, void* stream);
int sparch_softmax_sum_bwd(int B, int T, int K, const float* x, const float* g, float* dx, void* stream);

/* Dense recurrent cell of the RNN baseline (RNNLayer._rnn_cell, anns.py:328-339):
 *     y_t = act(Wx_t * scale + shift + y_{t-1} V^T),  y_{-1} = 0          (forward)
 *     dpre_t = (g_t + dpre_{t+1} V) * act'(y_t)                            (backward)
 * The same persistent machine as sparch_rec_cell_bwd: V slice resident in registers, the previous step's
 * dense fp32 tiles handed over through `chan` (sparch_rec_chan_bytes) and contracted on the bf16 MFMA
 * with the exact six-term split.  vpack = sparch_vpack(H, V, transpose | 2, ...): bit 1 keeps the
 * diagonal; forward uses transpose = 1 (y V^T), backward transpose = 0 (dpre V).
 *   y_out   (B,T,H*dirs) dropout(y) with the directions concatenated on features (anns.py:319-324)
 *   y_state (Bp,T,H)     y in cell time order: the backward's input
 *   dpre    (Bp,T,H)     gradient w.r.t. the normalised projection, virtual rows, ORIGINAL time index
 *   y_prev  (Bp,T,H)     y_{t-1} at the same index (zero row at the first step): dV = dpre^T y_prev      */
int sparch_ann_rec_fwd(int act, int B, int dirs, int T, int H, const float* Wx, const float* scale,
                       const float* shift, const float* vpack, float p_drop, uint64_t seed,
                       float* y_out, float* y_state, void* chan, size_t chan_bytes, uint32_t* status,
                       int steps_per_launch, void* stream);
int sparch_ann_rec_bwd(int act, int B, int dirs, int T, int H, const float* g_out, const float* y_state,
                       const float* vpack, float p_drop, uint64_t seed, float* dpre, float* y_prev,
                       void* chan, size_t chan_bytes, uint32_t* status, int steps_per_launch,
                       void* stream);
/* ONE step of the same cell with the recurrent product supplied by the caller (hidden sizes > 1024, see
 * sparch_rec_cell_step_fwd).  `s` counts steps in processing order (forward t = s, backward t = T-1-s);
 * rec (Bp,H) = y_{t-1} V^T (forward) / dpre_{t+1} V (backward), ignored at s = 0; the step's y / dpre also
 * goes to the contiguous (Bp,H) buffer y_step / dpre_step.                                          */
int sparch_ann_rec_step_fwd(int act, int B, int dirs, int T, int H, int s, const float* Wx,
                            const float* scale, const float* shift, const float* rec, float p_drop,
                            uint64_t seed, float* y_out, float* y_state, float* y_step, void* stream);
int sparch_ann_rec_step_bwd(int act, int B, int dirs, int T, int H, int s, const float* g_out,
                            const float* y_state, const float* rec, float p_drop, uint64_t seed,
                            float* dpre, float* y_prev, float* dpre_step, void* stream);

/* LiGRU cell (LiGRULayer._ligru_cell, anns.py:449-462) as persistent kernels (gatedcell.hip): hidden sizes that are
 * multiples of 32 up to 1024; a workgroup owns 16 hidden units with its slices of BOTH recurrent matrices.
 *   z = sigmoid(Wzx*scz+shz + y_{t-1} Vz^T), c = relu(Wx*sc+sh + y_{t-1} V^T), y = z y_{t-1} + (1-z) c.
 * vpack = sparch_ligru_vpack(H, Vz, V, backward): the exact three-plane bf16 fragments of the slices.
 * forward outputs: y_out (B,T,H*dirs) = dropout(y), y_state / z_save / c_save (Bp,T,H) in cell time order;
 * backward outputs (Bp,T,H at the ORIGINAL time index): dz_all, dc_all = gradients w.r.t. the two normalised
 * projections, yprev_all = y_{t-1} (dVz = dz_all^T yprev_all, dV = dc_all^T yprev_all); carry (Bp,H): scratch
 * carried between chunked launches.  chan: sparch_ligru_chan_bytes(Bp, H) of scratch.                  */
size_t sparch_ligru_vpack_bytes(int H, int backward);
int sparch_ligru_vpack(int H, const float* Vz, const float* V, int backward, float* vpack, void* stream, int precision);
size_t sparch_ligru_chan_bytes(int Bp, int H);
int sparch_ligru_fwd(int B, int dirs, int T, int H, const float* Wx, const float* sc, const float* sh,
                     const float* Wzx, const float* scz, const float* shz, const float* vpack,
                     float p_drop, uint64_t seed, float* y_out, float* y_state, float* z_save,
                     float* c_save, void* chan, size_t chan_bytes, uint32_t* status,
                     int steps_per_launch, void* stream);
int sparch_ligru_bwd(int B, int dirs, int T, int H, const float* g_out, const float* y_state,
                     const float* z_save, const float* c_save, const float* vpack_b, float p_drop,
                     uint64_t seed, float* dz_all, float* dc_all, float* yprev_all, float* carry,
                     void* chan, size_t chan_bytes, uint32_t* status, int steps_per_launch,
                     void* stream);

/* Host routine (no device work): n uniform fp32 numbers in [0, 1) from an MT19937 state, exactly as torch.rand draws
 * them from its CPU generator — one 32-bit output y per element, x = (y & 0xFFFFFF) * 2^-24 — i.e. the reference's
 * initial-state draws (snns.py:286-287, 423-425, 558-559, 700-702, 812) at ~1 ns instead of ~5 ns per number.
 * key: the 624 state words, *pos: index of the next word (624 = regenerate first); both are advanced in place.  The
 * caller moves them out of / back into the generator's serialized state (sparch_amd/snns.py, which also verifies the
 * layout once against torch.rand and otherwise keeps drawing with torch.rand).                              */
int sparch_mt19937_uniform_f32(uint32_t* key, int* pos, size_t n, float* out);

/* GRU cell (GRULayer._gru_cell, anns.py:581-595) as persistent kernels: the same machine with a second hand-off
 * per step (the reset gate sits inside the candidate's recurrent term):
 *   z = sigmoid(xz + y Vz^T), r = sigmoid(xr + y Vr^T), c = tanh(xc + (r y) V^T), y' = z y + (1 - z) c.
 * Two fragment buffers per direction of time (sparch_gru_vpack_bytes(H, backward, which): which 0 = the gates'
 * [Vz | Vr], 1 = the candidate's V), filled by one sparch_gru_vpack call.  forward outputs as the LiGRU plus
 * r_save; backward outputs (Bp,T,H, original time index): dz_all, dr_all, dc_all, yprev_all = y_{t-1},
 * ry_all = r y_{t-1} (dVz = dz_all^T yprev_all, dVr = dr_all^T yprev_all, dV = dc_all^T ry_all).  Every workgroup of
 * a row tile must be resident at once (two hand-offs inside a step): SPARCH_EINVAL when H / 16 exceeds the CU
 * count — callers then use the launch-per-step path (sparch_gate_step).  chan: sparch_gru_chan_bytes(Bp, H).  */
size_t sparch_gru_vpack_bytes(int H, int backward, int which);
int sparch_gru_vpack(int H, const float* Vz, const float* Vr, const float* V, int backward, float* vpack_gate,
                     float* vpack_cand, void* stream, int precision);
size_t sparch_gru_chan_bytes(int Bp, int H);
int sparch_gru_fwd(int B, int dirs, int T, int H, const float* Wx, const float* sc, const float* sh,
                   const float* Wzx, const float* scz, const float* shz, const float* Wrx, const float* scr,
                   const float* shr, const float* vpack_gate, const float* vpack_cand, float p_drop,
                   uint64_t seed, float* y_out, float* y_state, float* z_save, float* r_save, float* c_save,
                   void* chan, size_t chan_bytes, uint32_t* status, int steps_per_launch, void* stream);
int sparch_gru_bwd(int B, int dirs, int T, int H, const float* g_out, const float* y_state, const float* z_save,
                   const float* r_save, const float* c_save, const float* vpack_gate_b,
                   const float* vpack_cand_b, float p_drop, uint64_t seed, float* dz_all, float* dr_all,
                   float* dc_all, float* yprev_all, float* ry_all, float* carry, void* chan, size_t chan_bytes,
                   uint32_t* status, int steps_per_launch, void* stream);

/* Gate arithmetic of ONE time step of the gated baselines (LiGRULayer._ligru_cell anns.py:449-462,
 * GRULayer._gru_cell anns.py:581-595); the recurrent products between the phases are GEMM calls: the
 * launch-per-step path (annstep.hip) for hidden sizes the persistent kernels above do not take.  mode: 0 LiGRU forward, 1 GRU forward gates
 * (z, r, r*y), 2 GRU forward candidate + state, 3 LiGRU backward, 4 GRU backward (dy, dz_pre, dc_pre),
 * 5 GRU backward (dr_pre).  `in` / `out` are HOST arrays of 14 device pointers each (unused slots NULL):
 *   in : Wx sc sh Wzx scz shz Wrx scr shr rec g_out carry_mv carry_dir dry
 *   out: y_state z_save r_save c_save ry y_out carry_dir_out dgate dcp dz_all dr_all dc_all yprev_all ry_all
 * Shapes: projections (B,T,H) with folded BatchNorm (H) each; rec (Bp,2H) or (Bp,H); saves (Bp,T,H) in cell
 * time order; *_all (Bp,T,H) at the original time index; y_out / g_out (B,T,H*dirs).  H % 4 == 0.      */
int sparch_gate_step(int mode, int B, int dirs, int T, int H, int t, const float* const* in,
                     float* const* out, float p_drop, uint64_t seed, void* stream);

/* ---- f-2: optimizer step on the device (replaces torch.optim.Adam.step, exp.py:89, 377) ----------
 * One launch for the whole parameter list; arithmetic identical, operation by operation, to
 * torch.optim.Adam's default path (see optim.hip).  `params`, `grads`, `exp_avg`, `exp_avg_sq` are HOST
 * arrays of n_tensors DEVICE pointers, `numel` a host array of element counts.
 * step_size = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t): formed by the caller in double precision.
 * scalars_dev (nullable): device float[2] = {step_size, bc2_sqrt} read instead of the two arguments — for a
 * step captured in a HIP graph, whose kernel arguments are frozen at capture time.
 * skip_if_nonzero (nullable): a device word, e.g. the recurrent kernels' status word — when it is non-zero
 * the step leaves parameters and moments untouched (a timed-out step must not be applied; no host sync) and
 * adds 1 to skip_if_nonzero[1] (hence not const: the caller learns how many steps to take back from its
 * bias-correction counter when it reads the word). */
int sparch_adam_step(int n_tensors, float* const* params, const float* const* grads,
                     float* const* exp_avg, float* const* exp_avg_sq, const int64_t* numel,
                     float step_size, float beta1, float beta2, float bc2_sqrt, float eps,
                     float weight_decay, const float* scalars_dev, uint32_t* skip_if_nonzero,
                     void* stream);

/* The per-step factors of a captured step, made on the device by ONE launch: t_dev (double, the step count) += 1;
 * scalars_dev[0] = lr_dev[0] / (1 - beta1^t), scalars_dev[1] = sqrt(1 - beta2^t), evaluated in double precision as the
 * host path does (replaces a dozen captured scalar torch ops per step, ~55 us of a 1 ms BASELINE configs[1] step). */
int sparch_adam_scalars(double* t_dev, const double* lr_dev, double beta1, double beta2, float* scalars_dev,
                        void* stream);

/* ---- G1/G2, round 3: the gradient of a BatchNorm'd projection as bf16 planes, made ONCE.
 * sparch_bn_bwd_apply_planes = sparch_bn_bwd_apply (dx = gamma*invstd*(dy - dbeta/M - xhat*dgamma/M)) writing dx as
 * its three exact bf16 planes (3 x M x H uint16, truncation split: dx = t1 + t2 + t3 — the split the dense GEMMs
 * otherwise redo in every workgroup that stages a tile of dx); H % 8 == 0.  dy2 (nullable): the second direction's
 * gradient of a bidirectional layer, added in the same pass (dy + dy2; replaces sparch_add_halves); dx (nullable):
 * the fp32 tensor as well.  Consumers: sparch_gemm6_nn_pp (dX = dx * W: both operands as planes) and
 * sparch_gemm_spike16_tn_ap (dW = dx^T * spikes: dense side as planes) — bit-identical to the fp32-operand entry
 * points; where the pipelined plane kernels do not apply (small or ragged shapes) they read the fp32 operand, which
 * may be NULL only if the caller has checked the shape (M, N >= one tile, K % 32 == 0, 16-byte rows). */
int sparch_bn_bwd_apply_planes(int M, int H, const float* dy, const float* dy2, const float* x, const float* mean,
                               const float* invstd, const float* gamma, const float* dgamma, const float* dbeta,
                               uint16_t* planes, float* dx, void* stream);
int sparch_gemm6_nn_pp(int M, int N, int K, const float* A, const uint16_t* A_planes, int lda, const float* B,
                       const uint16_t* B_planes, int ldb, float* C, int ldc, void* stream, int precision);
int sparch_gemm_spike16_tn_ap(int M, int N, int K, const float* A, const uint16_t* A_planes, int lda,
                              const uint16_t* B16, int ldb, float scale, float* C, int ldc, int zero_diag,
                              int accumulate, void* ws, size_t ws_bytes, void* stream, int precision);

/* ---- a11: the batch upload of the train step (exp.py:355-356 copies a dense fp32 batch to the device every step:
 * 179 MB at the headline shape).  counts (M,K) uint8 = the same binned spike counts (spiking_datasets.py:71-78) at
 * one byte per element; the call expands them into the bf16 plane the first layer's GEMMs read — (M, ldp) uint16
 * bit patterns, ldp >= K a multiple of 8, zeros behind column K: the layout and the values of
 * sparch_plane_bf16_exact on the float batch, whose flag would read 1 — and, if x != NULL, into the fp32 tensor
 * (M rows of ldx floats). */
int sparch_expand_counts_u8(long long M, int K, const uint8_t* counts, uint16_t* plane, int ldp, float* x, int ldx,
                            void* stream);

/* ---- a11: the train step's loss (exp.py:100, 362: nn.CrossEntropyLoss()(output, y), mean over the batch) and its
 * gradient with respect to the logits, one launch: loss[0] = mean_b(logsumexp(x_b) - x_b[y_b]),
 * dlogits = (softmax(x) - onehot(y)) / B.  logits (B,C) fp32, labels (B) int64 (a label outside [0,C) contributes
 * nothing), loss (1) fp32, dlogits (B,C) fp32. */
int sparch_ce_loss(int B, int C, const float* logits, const int64_t* labels, float* loss, float* dlogits,
                   void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPARCH_HIP_H */

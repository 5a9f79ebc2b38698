// G9: Kaldi-compatible log-mel filterbank front-end, fused into one kernel.
//
// Replaces `torchaudio.compliance.kaldi.fbank(x, num_mel_bins=40)` at
// nonspiking_datasets.py:96, 194.  The arithmetic lives in third-party torchaudio 0.12.0
// (requirements.txt:14), which is NOT in /root/reference and not installed: the algorithm
// below restates torchaudio's published defaults from memory —
//   16 kHz, 25 ms frames (400) every 10 ms (160), snip_edges (frames = 1 + (N-400)/160),
//   dither 0, per-frame DC removal, pre-emphasis 0.97 with replicated first sample,
//   povey window hann(400, periodic=False)^0.85, zero-pad to 512, |rFFT|^2,
//   triangular mel bins between 20 Hz and Nyquist on mel = 1127 ln(1 + f/700),
//   log(max(e, FLT_EPSILON)), no energy row, no mean subtraction.
// PARITY UNPINNED against torchaudio itself; pinned against an independent NumPy
// restatement and analytic known-answer signals in tests/.
//
// One 256-thread workgroup per FPW consecutive frames of a clip: the per-launch tables (FFT twiddles, povey
// window, mel value of every bin, first / last bin of every triangular filter) are built ONCE per workgroup
// in LDS; per frame: samples -> LDS, block mean, pre-emphasis + window, 512-point radix-2 FFT entirely in
// LDS (one butterfly per thread per stage), power spectrum, and the 40 triangular sums over each filter's
// own bin range only (ascending bin order: the same partial sums as a loop over all 256 bins, whose other
// terms are exact zeros).  (First version: one frame per workgroup, tables and window recomputed per frame —
// sincospif / powf / logf for every frame — and 40 threads x 256 bins for the mel stage: 0.45 ms for
// 256 x 16000 samples; this one: see DESIGN.md.)
#include "common.h"

namespace {

constexpr int FRAME = 400, SHIFT = 160, NFFT = 512, NBIN = NFFT / 2;  // 256 usable bins (+Nyquist, weight 0)
constexpr float SAMPLE_RATE = 16000.0f, LOW_HZ = 20.0f, PREEMPH = 0.97f;

__device__ __forceinline__ float mel_of(float hz) { return 1127.0f * logf(1.0f + hz / 700.0f); }

constexpr int FPW = 7;  // frames per workgroup (98 frames of a 1 s clip = 14 x 7)

__global__ __launch_bounds__(256) void fbank_kernel(int n_clips, int n_samples, int n_frames, int n_mels,
                                                    const float* __restrict__ wave, float* __restrict__ out) {
    __shared__ float re[NFFT], im[NFFT];
    __shared__ float tw_c[NBIN], tw_s[NBIN];
    __shared__ float melpt[NBIN];
    __shared__ float win[NFFT];
    __shared__ int f_lo[256], f_hi[256];
    __shared__ float part[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int chunks = (n_frames + FPW - 1) / FPW;
    const int clip = blockIdx.x / chunks, frame0 = (blockIdx.x % chunks) * FPW;

    // ---- tables, once per workgroup: twiddles e^{-2 pi i k / 512}, mel value of every FFT bin centre,
    //      povey window, and per mel filter the bins with a non-zero weight
    {
        float sn, c;
        sincospif(-2.0f * (float)tid / (float)NFFT, &sn, &c);
        tw_c[tid] = c; tw_s[tid] = sn;
        melpt[tid] = mel_of((float)tid * (SAMPLE_RATE / (float)NFFT));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + 256 * j;
            const float hann = 0.5f - 0.5f * cospif(2.0f * (float)i / (float)(FRAME - 1));
            win[i] = i < FRAME ? powf(hann, 0.85f) : 0.f;
        }
    }
    __syncthreads();
    const float mel_lo = mel_of(LOW_HZ), mel_hi = mel_of(0.5f * SAMPLE_RATE);
    const float delta = (mel_hi - mel_lo) / (float)(n_mels + 1);
    const float left = mel_lo + (float)tid * delta, center = left + delta, right = center + delta;
    if (tid < n_mels) {  // melpt is increasing: the bins strictly inside (left, right) form one range
        int lo = 0, hi = NBIN;  // lo = first bin with melpt > left (binary search, 8 steps)
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (melpt[mid] > left) hi = mid; else lo = mid + 1; }
        int a = lo, b = NBIN;   // a = first bin with melpt >= right
        while (a < b) { const int mid = (a + b) >> 1; if (melpt[mid] < right) a = mid + 1; else b = mid; }
        f_lo[tid] = lo; f_hi[tid] = a;
    }
    __syncthreads();

    for (int f = 0; f < FPW; ++f) {
        const int frame = frame0 + f;
        if (frame >= n_frames) break;  // uniform
        const float* src = wave + (size_t)clip * n_samples + (size_t)frame * SHIFT;
        // frame -> registers (2 samples per thread), block mean
        float x0 = (tid < FRAME) ? src[tid] : 0.f;
        float x1 = (tid + 256 < FRAME) ? src[tid + 256] : 0.f;
        float s = x0 + x1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) part[wv] = s;
        __syncthreads();
        const float mean = ((part[0] + part[1]) + (part[2] + part[3])) / (float)FRAME;
        if (tid < FRAME) re[tid] = x0 - mean;
        if (tid + 256 < FRAME) re[tid + 256] = x1 - mean;
        __syncthreads();
        // pre-emphasis (x[i] - 0.97 x[i-1], first sample replicated) and povey window
        float y[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + 256 * j;
            float v = 0.f;
            if (i < FRAME) {
                const float prev = re[i > 0 ? i - 1 : 0];
                v = (re[i] - PREEMPH * prev) * win[i];
            }
            y[j] = v;
        }
        __syncthreads();
        // bit-reversed scatter for the decimation-in-time FFT
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + 256 * j;
            const int rev = (int)(__brev((unsigned)i) >> (32 - 9));
            re[rev] = y[j];
            im[rev] = 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int stage = 0; stage < 9; ++stage) {
            const int half = 1 << stage;
            const int k = tid & (half - 1);
            const int i0 = ((tid >> stage) << (stage + 1)) + k, i1 = i0 + half;
            const int tw = k << (8 - stage);
            const float c = tw_c[tw], sn = tw_s[tw];
            const float ar = re[i0], ai = im[i0], br = re[i1], bi = im[i1];
            const float tr = br * c - bi * sn, ti = br * sn + bi * c;
            re[i0] = ar + tr; im[i0] = ai + ti;
            re[i1] = ar - tr; im[i1] = ai - ti;
            __syncthreads();
        }
        // power spectrum of bins 0..255 (the Nyquist bin carries zero mel weight)
        const float pw = re[tid] * re[tid] + im[tid] * im[tid];
        __syncthreads();
        re[tid] = pw;
        __syncthreads();
        if (tid < n_mels) {
            float e = 0.f;
            for (int i = f_lo[tid]; i < f_hi[tid]; ++i) {
                const float m = melpt[i];
                const float up = (m - left) / (center - left), down = (right - m) / (right - center);
                const float wgt = fmaxf(0.f, fminf(up, down));
                e += wgt * re[i];
            }
            out[((size_t)clip * n_frames + frame) * n_mels + tid] = logf(fmaxf(e, 1.1920928955078125e-07f));
        }
        __syncthreads();  // re / im / part are reused by the next frame
    }
}

}  // namespace

extern "C" int sparch_fbank_frames(int n_samples) {
    SPARCH_ENTER();
    return n_samples < FRAME ? 0 : 1 + (n_samples - FRAME) / SHIFT;
}

extern "C" int sparch_fbank_fwd(int n_clips, int n_samples, int n_mels, const float* wave, float* out,
                                void* stream) {
    SPARCH_ENTER();
    const int n_frames = sparch_fbank_frames(n_samples);
    if (n_clips <= 0 || n_frames <= 0 || n_mels <= 0 || n_mels > 256 || !wave || !out) return SPARCH_EINVAL;
    const int chunks = (n_frames + FPW - 1) / FPW;
    hipLaunchKernelGGL(fbank_kernel, dim3((unsigned)(n_clips * chunks)), dim3(256), 0, (hipStream_t)stream,
                       n_clips, n_samples, n_frames, n_mels, wave, out);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

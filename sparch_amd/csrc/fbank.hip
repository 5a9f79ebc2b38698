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
// One 256-thread workgroup per frame: samples -> LDS, block mean, pre-emphasis + window,
// 512-point radix-2 FFT entirely in LDS (one butterfly per thread per stage, twiddles
// from an LDS table built with sincospif), power spectrum, 40 sparse triangular sums.
#include "common.h"

namespace {

constexpr int FRAME = 400, SHIFT = 160, NFFT = 512, NBIN = NFFT / 2;  // 256 usable bins (+Nyquist, weight 0)
constexpr float SAMPLE_RATE = 16000.0f, LOW_HZ = 20.0f, PREEMPH = 0.97f;

__device__ __forceinline__ float mel_of(float hz) { return 1127.0f * logf(1.0f + hz / 700.0f); }

__global__ __launch_bounds__(256) void fbank_kernel(int n_clips, int n_samples, int n_frames, int n_mels,
                                                    const float* __restrict__ wave, float* __restrict__ out) {
    __shared__ float re[NFFT], im[NFFT];
    __shared__ float tw_c[NBIN], tw_s[NBIN];
    __shared__ float melpt[NBIN];
    __shared__ float part[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int frame = blockIdx.x % n_frames, clip = blockIdx.x / n_frames;
    const float* src = wave + (size_t)clip * n_samples + (size_t)frame * SHIFT;

    // twiddles e^{-2 pi i k / 512} and the mel value of every FFT bin centre
    {
        float s, c;
        sincospif(-2.0f * (float)tid / (float)NFFT, &s, &c);
        tw_c[tid] = c; tw_s[tid] = s;
        melpt[tid] = mel_of((float)tid * (SAMPLE_RATE / (float)NFFT));
    }
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
            const float hann = 0.5f - 0.5f * cospif(2.0f * (float)i / (float)(FRAME - 1));
            v = (re[i] - PREEMPH * prev) * powf(hann, 0.85f);
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
        const float mel_lo = mel_of(LOW_HZ), mel_hi = mel_of(0.5f * SAMPLE_RATE);
        const float delta = (mel_hi - mel_lo) / (float)(n_mels + 1);
        const float left = mel_lo + (float)tid * delta, center = left + delta, right = center + delta;
        float e = 0.f;
        for (int i = 0; i < NBIN; ++i) {
            const float m = melpt[i];
            const float up = (m - left) / (center - left), down = (right - m) / (right - center);
            const float wgt = fmaxf(0.f, fminf(up, down));
            e += wgt * re[i];
        }
        out[((size_t)clip * n_frames + frame) * n_mels + tid] = logf(fmaxf(e, 1.1920928955078125e-07f));
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
    hipLaunchKernelGGL(fbank_kernel, dim3((unsigned)(n_clips * n_frames)), dim3(256), 0, (hipStream_t)stream,
                       n_clips, n_samples, n_frames, n_mels, wave, out);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

"""
ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

NumPy restatement of `torchaudio.compliance.kaldi.fbank(waveform, num_mel_bins=40)` as the reference
calls it (nonspiking_datasets.py:96, 194).  PARITY UNPINNED: the algorithm lives in third-party
torchaudio==0.12.0 (requirements.txt:14), which is neither under /root/reference nor installed, and the
reference holds no fixture for it.  This file restates torchaudio's published defaults from memory
 independently of the HIP kernel (different FFT, float64 throughout); the tests also use
analytic known-answer signals.  Defaults: 16 kHz, 25 ms / 10 ms frames, snip_edges, dither 0,
remove_dc_offset, preemphasis 0.97 (first sample replicated), povey window, 512-point FFT, power
spectrum, 40 triangular mel bins from 20 Hz to Nyquist on mel = 1127 ln(1 + f/700), log with a
FLT_EPSILON floor, no energy, no mean subtraction.
"""
import numpy as np

SR, FRAME, SHIFT, NFFT = 16000, 400, 160, 512


def mel(f):
    return 1127.0 * np.log(1.0 + np.asarray(f, np.float64) / 700.0)


def mel_banks(n_mels=40, low=20.0, high=SR / 2):
    bins = NFFT // 2
    pts = mel(np.arange(bins) * (SR / NFFT))
    lo, hi = mel(low), mel(high)
    delta = (hi - lo) / (n_mels + 1)
    W = np.zeros((n_mels, bins + 1))
    for m in range(n_mels):
        left, center, right = lo + m * delta, lo + (m + 1) * delta, lo + (m + 2) * delta
        up = (pts - left) / (center - left)
        down = (right - pts) / (right - center)
        W[m, :bins] = np.maximum(0.0, np.minimum(up, down))
    return W


def fbank(wave, n_mels=40):
    """wave (n_samples,) float in [-1, 1] -> (frames, n_mels) log-mel energies."""
    wave = np.asarray(wave, np.float64)
    n = 1 + (len(wave) - FRAME) // SHIFT
    idx = np.arange(FRAME)[None, :] + SHIFT * np.arange(n)[:, None]
    fr = wave[idx]
    fr = fr - fr.mean(axis=1, keepdims=True)
    prev = np.concatenate([fr[:, :1], fr[:, :-1]], axis=1)
    fr = fr - 0.97 * prev
    win = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(FRAME) / (FRAME - 1))) ** 0.85
    spec = np.abs(np.fft.rfft(fr * win, NFFT, axis=1)) ** 2
    e = spec @ mel_banks(n_mels).T
    return np.log(np.maximum(e, np.finfo(np.float32).eps))

"""Oracle: Wav2Lip mel-spectrogram features, numpy fp64 on CPU.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates
  avatars/wav2lip/audio.py:20-23   preemphasis  (scipy.signal.lfilter([1,-k],[1],wav))
  avatars/wav2lip/audio.py:45-51   melspectrogram
  avatars/wav2lip/audio.py:57-61   _stft -> librosa.stft(n_fft=800, hop=200, win=800)
  avatars/wav2lip/audio.py:92-101  _linear_to_mel / _build_mel_basis -> librosa.filters.mel
  avatars/wav2lip/audio.py:103-105 _amp_to_db
  avatars/wav2lip/audio.py:110-122 _normalize (symmetric, clipped)
  avatars/wav2lip/hparams.py:33-73 constants
  avatars/audio_features/mel.py:34-67  MelASR.run_step window slicing

Third-party leaves (librosa is NOT vendored in the reference and not installed
here; requirements.txt:44 leaves it unpinned): `stft` and `mel_filterbank`
restate librosa's published algorithm (periodic Hann window, centre padding,
rFFT; Slaney mel scale with Slaney area normalisation, float32 basis).

Pinning: gen_golden.py (a) checks `mel_filterbank(16000,400,80)` against the
reference's only known-answer asset, avatars/musetalk/whisper/whisper/assets/
mel_filters.npz (= librosa.filters.mel(sr=16000,n_fft=400,n_mels=80) per
whisper/whisper/audio.py:80-85), (b) checks the whole chain against an
independent statement built from transformers.audio_utils, and (c) runs the
reference's own audio.melspectrogram / MelASR.run_step with these leaves
injected as the `librosa` module.  The centre-padding mode (constant vs reflect
across librosa versions) cannot change any column the render loop consumes
(SURVEY.md §8a-W3), which gen_golden.py also asserts.
"""
from __future__ import annotations

import numpy as np

# avatars/wav2lip/hparams.py:33-73
NUM_MELS = 80
N_FFT = 800
HOP = 200
WIN = 800
SR = 16000
PREEMPH = 0.97
MIN_LEVEL_DB = -100.0
REF_LEVEL_DB = 20.0
FMIN = 55.0
FMAX = 7600.0
MAX_ABS = 4.0


def preemphasis(wav: np.ndarray, k: float = PREEMPH) -> np.ndarray:
    """audio.py:20-23.  lfilter([1,-k],[1],x): y[n] = x[n] - k*x[n-1], y[0]=x[0];
    scipy returns float64 for float32 input."""
    x = np.asarray(wav, dtype=np.float64)
    y = x.copy()
    y[1:] -= k * x[:-1]
    return y


def hann_periodic(n: int) -> np.ndarray:
    """scipy.signal.get_window('hann', n, fftbins=True) as librosa.stft uses."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def stft(y: np.ndarray, n_fft: int = N_FFT, hop: int = HOP, win: int = WIN,
         pad_mode: str = "constant") -> np.ndarray:
    """librosa.stft(y, n_fft, hop_length, win_length, window='hann', center=True)
    -> complex (1+n_fft/2, 1+len(y)//hop)."""
    assert win == n_fft
    y = np.asarray(y, dtype=np.float64)
    ypad = np.pad(y, n_fft // 2, mode=pad_mode)
    n_frames = 1 + (len(ypad) - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    frames = ypad[idx] * hann_periodic(win)[None, :]
    return np.fft.rfft(frames, axis=1).T


def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filterbank(sr: float = SR, n_fft: int = N_FFT, n_mels: int = NUM_MELS,
                   fmin: float = FMIN, fmax: float = FMAX) -> np.ndarray:
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm='slaney')
    -> float32 (n_mels, 1+n_fft/2)."""
    fftfreqs = np.linspace(0.0, float(sr) / 2, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, None]
    return weights.astype(np.float32)


_MEL_BASIS = None


def mel_basis() -> np.ndarray:
    global _MEL_BASIS
    if _MEL_BASIS is None:
        _MEL_BASIS = mel_filterbank()
    return _MEL_BASIS


def amp_to_db(x: np.ndarray) -> np.ndarray:
    """audio.py:103-105."""
    min_level = np.exp(MIN_LEVEL_DB / 20 * np.log(10))
    return 20 * np.log10(np.maximum(min_level, x))


def normalize(S: np.ndarray) -> np.ndarray:
    """audio.py:110-115 (allow_clipping, symmetric)."""
    return np.clip((2 * MAX_ABS) * ((S - MIN_LEVEL_DB) / (-MIN_LEVEL_DB)) - MAX_ABS, -MAX_ABS, MAX_ABS)


def melspectrogram(wav: np.ndarray, pad_mode: str = "constant") -> np.ndarray:
    """audio.py:45-51 -> float64 (80, 1+len(wav)//200)."""
    D = stft(preemphasis(wav), pad_mode=pad_mode)
    S = amp_to_db(np.dot(mel_basis(), np.abs(D))) - REF_LEVEL_DB
    return normalize(S)


def window_starts(n_chunks: int, l: int, r: int, fps: int = 25) -> list:
    """mel.py:50-63: start column of each (80,16) window for a buffer of
    n_chunks 20-ms chunks with l/r context chunks."""
    left = max(0, l * 80 / 50)
    mult = 80.0 / fps
    out = []
    i = 0
    while i < (n_chunks - l - r) / 2:
        out.append(int(left + i * mult))
        i += 1
    return out


def mel_chunks(wav: np.ndarray, n_chunks: int, l: int = 10, r: int = 10, fps: int = 25,
               step: int = 16) -> list:
    """mel.py:47-63: the list MelASR.run_step puts on feat_queue."""
    mel = melspectrogram(wav)
    T = mel.shape[1]
    chunks = []
    for s in window_starts(n_chunks, l, r, fps):
        if s + step > T:
            chunks.append(mel[:, T - step:])
        else:
            chunks.append(mel[:, s:s + step])
    return chunks

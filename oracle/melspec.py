"""Oracle (TEST INFRASTRUCTURE): log-mel front end.  PARITY UNPINNED (see oracle/__init__.py).

Follows model/fp/melspec/melspectrogram.py:25-112 of the reference.  The
arithmetic itself lives in kapre==0.3.5 (`STFT`, `Magnitude`,
`ApplyFilterbank`, called at melspectrogram.py:82-98), tensorflow
(`tf.signal.stft`) and librosa 0.8.1 (`librosa.filters.mel`, reached through
kapre's `filterbank_mel`); those published algorithms are restated here:

  * kapre 0.3.5 `STFT(n_fft, hop_length, window_name=None, pad_begin=False,
    pad_end=False)` = `tf.signal.stft(x, frame_length=n_fft, frame_step=hop,
    fft_length=n_fft, window_fn=tf.signal.hann_window, pad_end=False)`:
    frame t = x[t*hop : t*hop+n_fft], n_frames = 1 + (L - n_fft)//hop,
    periodic Hann 0.5 - 0.5*cos(2*pi*n/n_fft), forward rfft (e^{-i...}).
  * kapre `Magnitude` = `tf.abs` (magnitude, not power).
  * kapre `ApplyFilterbank(type='mel')` = tensordot with
    `librosa.filters.mel(sr, n_fft=(n_freq-1)*2, n_mels, fmin, fmax,
    htk=False, norm='slaney').T` cast to float32.
"""
import math

import numpy as np


# ---------------------------------------------------------------------------
# librosa 0.8.1 Slaney mel scale (librosa/core/convert.py hz_to_mel/mel_to_hz,
# htk=False) and filter bank (librosa/filters.py mel()).
# ---------------------------------------------------------------------------
_F_SP = 200.0 / 3
_MIN_LOG_HZ = 1000.0
_MIN_LOG_MEL = _MIN_LOG_HZ / _F_SP
_LOGSTEP = math.log(6.4) / 27.0


def hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    mel = f / _F_SP
    log_t = f >= _MIN_LOG_HZ
    mel = np.where(log_t, _MIN_LOG_MEL + np.log(np.maximum(f, 1e-300) / _MIN_LOG_HZ) / _LOGSTEP, mel)
    return mel


def mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f = _F_SP * m
    log_t = m >= _MIN_LOG_MEL
    f = np.where(log_t, _MIN_LOG_HZ * np.exp(_LOGSTEP * (m - _MIN_LOG_MEL)), f)
    return f


def mel_filterbank(sr=8000, n_fft=1024, n_mels=256, fmin=300.0, fmax=4000.0):
    """(n_mels, 1+n_fft//2) float32 Slaney-normalised triangular filters.

    Arguments are the ones that reach librosa through kapre from
    melspectrogram.py:44-50 and config/default.yaml:39-46.
    """
    n_freq = 1 + n_fft // 2
    fftfreqs = np.linspace(0.0, float(sr) / 2, n_freq, endpoint=True)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    # librosa 0.8.1 allocates `weights` as float32: the triangles are rounded to
    # float32 on assignment, then `weights *= enorm[:, None]` multiplies in float64
    # and rounds to float32 again.  Both roundings are reproduced.
    weights = np.zeros((n_mels, n_freq), dtype=np.float32)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, None]
    return weights


def hann_periodic(n):
    """tf.signal.hann_window(n, periodic=True)."""
    k = np.arange(n, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)


def stft_magnitude(x, n_fft=1024, hop=256, dtype=np.float64):
    """x: (B, L) already padded.  Returns (B, T, n_fft//2+1) magnitudes."""
    x = np.asarray(x, dtype=dtype)
    B, L = x.shape
    n_frames = 1 + (L - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    frames = x[:, idx] * hann_periodic(n_fft).astype(dtype)[None, None, :]
    if dtype == np.float32:
        import scipy.fft
        spec = scipy.fft.rfft(frames, axis=-1)  # stays complex64
    else:
        spec = np.fft.rfft(frames, axis=-1)
    return np.abs(spec)


def melspec_layer(x, fs=8000, n_fft=1024, stft_hop=256, n_mels=256,
                  f_min=300.0, f_max=4000.0, amin=1e-10, dynamic_range=80.0,
                  segment_norm=False, group_size=None, dtype=np.float64,
                  return_raw=False):
    """Melspec_layer.call (melspectrogram.py:102-112) on x of shape (B,1,T).

    `group_size`: the reference subtracts `tf.reduce_max(x)` over the WHOLE
    device batch (melspectrogram.py:108); its batch is TS_BATCH_SZ consecutive
    segments.  With group_size=None the whole of `x` is
    one group (exactly the reference); otherwise rows are grouped in
    consecutive runs of `group_size` (last group ragged) so that a larger
    launch batch reproduces what the reference computes batch by batch.
    Returns (B, n_mels, n_frames, 1).
    """
    x = np.asarray(x)
    assert x.ndim == 3 and x.shape[1] == 1
    B = x.shape[0]
    pad = n_fft // 2                                   # melspectrogram.py:59-65
    xp = np.pad(x[:, 0, :].astype(dtype), ((0, 0), (pad, pad)))
    mag = stft_magnitude(xp, n_fft, stft_hop, dtype)   # (B,T,513) kapre STFT+Magnitude
    fb = mel_filterbank(fs, n_fft, n_mels, f_min, f_max).astype(dtype)  # (n_mels,513)
    mel = mag @ fb.T                                   # ApplyFilterbank, (B,T,n_mels)
    y = mel + dtype(0.06)                              # melspectrogram.py:104
    y = np.log(np.maximum(y, dtype(amin))) / dtype(math.log(10))   # :107
    raw = y
    if group_size is None:
        group_size = max(B, 1)
    out = np.empty_like(y)
    for g0 in range(0, B, group_size):
        g = y[g0:g0 + group_size]
        g = g - g.max()                                # :108 batch-global max
        g = np.maximum(g, dtype(-dynamic_range))       # :109
        if segment_norm:                               # :110-111
            mn = g.min()
            g = (g - mn / 2) / np.abs(mn / 2 + dtype(1e-10))
        out[g0:g0 + group_size] = g
    out = np.transpose(out, (0, 2, 1))[..., None]      # Permute((3,2,1)) -> (B,F,T,1)
    if return_raw:
        return out, np.transpose(raw, (0, 2, 1))
    return out

"""Oracle (TEST INFRASTRUCTURE): time-domain augmentation of the replicas.  PARITY UNPINNED for the
random streams only -- the arithmetic is the reference's own numpy (no third-party kernel involved).

Follows model/utils/audio_utils.py:10-137 (`max_normalize`, `background_mix`, `bg_mix_batch`,
`ir_aug_batch`, `log_scale_random_number_batch`) and the way model/utils/dataloader_keras.py:223-311,
316-482 calls them, with every random draw passed in explicitly (snr, amplitude ratio, offsets) so
that the HIP path can be fed the same draws.  float64 like the reference (`x / 2**15` makes float64
arrays, audio_utils.py:245-246; the batch is cast to float32 only at dataloader_keras.py:303-306).
"""
import numpy as np

MAX_IR_LENGTH = 600          # dataloader_keras.py:8


def max_normalize(x):
    """audio_utils.py:10-25."""
    m = np.max(np.abs(x))
    return x if m == 0 else x / m


def background_mix(x, x_bg, snr_db):
    """audio_utils.py:28-72 for equal lengths (the only case the loader produces)."""
    assert len(x) == len(x_bg)
    rmse_bg = np.sqrt(np.sum(x_bg ** 2 / len(x_bg)))
    x_bg = x_bg / rmse_bg
    rmse_x = np.sqrt(np.sum(x ** 2) / len(x))
    x = x / rmse_x
    magnitude = np.power(10, snr_db / 20.)
    return max_normalize(magnitude * x + x_bg)


def bg_mix_rows(event, bg, snrs, amps):
    """bg_mix_batch (audio_utils.py:82-117) with the SNRs (uniform in snr_range, :92-95) and the
    log-uniform amplitude ratios (:98-99) given."""
    out = np.zeros(event.shape)
    for i in range(len(event)):
        if np.max(np.abs(event[i])) == 0 or np.max(np.abs(bg[i])) == 0:
            out[i] = max_normalize(event[i] + bg[i])
        else:
            out[i] = background_mix(event[i], bg[i], snrs[i])
        out[i] = amps[i] * out[i]
    return out


def ir_aug_rows(x, ir):
    """ir_aug_batch (audio_utils.py:120-137): circular convolution of length max(len(x), len(ir))
    by FFT, first len(x) samples, max-normalised.  `ir` rows may be shorter than x (MAX_IR_LENGTH)."""
    out = np.zeros(x.shape)
    for i in range(len(x)):
        n = max(len(x[i]), len(ir[i]))
        y = np.fft.ifft(np.fft.fft(ir[i], n=n) * np.fft.fft(x[i], n=n))[:len(x[i])].real
        m = np.max(np.abs(y))
        out[i] = y if m == 0 else y / m
    return out


def ir_aug_rows_direct(x, ir):
    """The same circular convolution summed directly (independent formulation for the tests)."""
    out = np.zeros(x.shape)
    T = x.shape[1]
    for i in range(len(x)):
        y = np.zeros(T)
        for m, h in enumerate(ir[i]):
            if h != 0.0:
                y += h * np.roll(x[i], m)
        mx = np.max(np.abs(y))
        out[i] = y if mx == 0 else y / mx
    return out


def window(pcm_int16, start, T):
    """load_audio (audio_utils.py:221-264): frames [start, start+T) / 2**15, zero tail."""
    x = np.asarray(pcm_int16[start:start + T], dtype=np.float64) / 2 ** 15
    out = np.zeros(T)
    out[:len(x)] = x
    return out


def segment_offsets(n_frames, fs=8000, duration=1., hop=.5):
    """[(seg_idx, offset_min, offset_max)] of one file, mode 'all' (audio_utils.py:151-199)."""
    n_seg_frames, n_hop_frames = fs * duration, fs * hop
    if n_frames > n_seg_frames:
        n_segs = int((n_frames - n_seg_frames + n_hop_frames) // n_hop_frames)
    else:
        n_segs = 1
    residual = max(0, n_frames - ((n_segs - 1) * n_hop_frames + n_seg_frames))
    out = []
    for s in range(n_segs):
        lo, hi = int(-1 * n_hop_frames), n_hop_frames
        if s == 0:
            lo = 0
        if s == n_segs - 1:
            hi = residual
        out.append((s, lo, hi))
    return out

"""Oracle (TEST INFRASTRUCTURE): input contract of the hot path (segment enumeration).

Follows model/utils/audio_utils.py:140-218 (`get_fns_seg_list`, mode 'all'),
:221-264 (`load_audio`), and model/utils/dataloader_keras.py:303-306 (cast to
float32, shape (B,1,T)).  Pure stdlib `wave` + numpy, like the reference.
"""
import wave

import numpy as np


def n_segments(n_frames, fs=8000, duration=1.0, hop=0.5):
    """audio_utils.py:171-177."""
    seg = fs * duration
    hp = fs * hop
    if n_frames > seg:
        return int((n_frames - seg + hp) // hp)
    return 1


def enumerate_segments(filenames, fs=8000, duration=1.0, hop=0.5):
    """[(filename, seg_idx)] in file order then segment order (audio_utils.py:154-198)."""
    out = []
    for fn in filenames:
        with wave.open(fn, 'r') as w:
            if w.getframerate() != fs:
                raise ValueError('Sample rate should be {} but got {}'.format(fs, w.getframerate()))
            n = w.getnframes()
        for s in range(n_segments(n, fs, duration, hop)):
            out.append((fn, s))
    return out


def load_segment(filename, seg_idx, fs=8000, duration=1.0, hop=0.5):
    """load_audio(seg_start_sec=seg_idx*hop, seg_length_sec=duration) (audio_utils.py:221-264):
    int16 / 2**15 in float64, zero-padded at the tail to fs*duration samples."""
    start = int(np.floor(seg_idx * hop * fs))
    length = int(np.floor(duration * fs))
    with wave.open(filename, 'r') as w:
        w.setpos(start)
        raw = w.readframes(length)
    x = np.frombuffer(raw, dtype=np.int16) / 2 ** 15
    arr = np.zeros(int(duration * fs))
    arr[:len(x)] = x
    return arr


def load_batches(filenames, bsz, fs=8000, duration=1.0, hop=0.5):
    """Yield float32 (n,1,T) batches of consecutive segments, last one ragged
    (dataloader_keras.py:223-228, 303-306 with shuffle=False, drop_last=False)."""
    segs = enumerate_segments(filenames, fs, duration, hop)
    for i in range(0, len(segs), bsz):
        xs = [load_segment(fn, s, fs, duration, hop) for fn, s in segs[i:i + bsz]]
        yield np.expand_dims(np.stack(xs), 1).astype(np.float32)

"""Input contract of the hot path: WAV files -> fixed-length int16 segments.

Mirrors what the reference's loader delivers to `m_pre` during `generate`
(model/utils/audio_utils.py:140-218 `get_fns_seg_list` mode 'all', :221-264
`load_audio`; model/utils/dataloader_keras.py:223-228, 303-306 with shuffle=False and
no augmentation): segments of FS*DUR samples every FS*HOP samples, the tail zero-
padded, in file order then segment order.

Differences in mechanism, not in result: each file is opened ONCE and read whole
(the reference re-opens the file for every segment and grows each batch with
np.vstack row by row, dataloader_keras.py:389-397), and segments stay int16 -- the
HIP front end scales by 2^-15 itself, which is exactly `x / 2**15`
(audio_utils.py:245-246) in float32.
"""
import wave

import numpy as np


def n_segments(n_frames, fs=8000, duration=1., hop=.5):
    """audio_utils.py:171-177."""
    n_seg_frames, n_hop_frames = fs * duration, fs * hop
    if n_frames > n_seg_frames:
        return int((n_frames - n_seg_frames + n_hop_frames) // n_hop_frames)
    return 1


def wav_info(filename, fs):
    """Frame count of a 16-bit mono WAV; ValueError on a wrong sample rate
    (audio_utils.py:160-169)."""
    if filename[-3:] != 'wav':
        raise NotImplementedError(filename[-3:])
    with wave.open(filename, 'r') as w:
        if w.getframerate() != fs:
            raise ValueError('Sample rate should be {} but got {}'.format(str(fs), str(w.getframerate())))
        return w.getnframes()


def get_fns_seg_list(fns_list, fs=8000, duration=1., hop=None):
    """[[filename, seg_idx], ...] in file order then segment order (mode 'all')."""
    if hop is None:
        hop = duration
    out = []
    for fn in fns_list:
        for s in range(n_segments(wav_info(fn, fs), fs, duration, hop)):
            out.append([fn, s])
    return out


def read_wav_int16(filename):
    with wave.open(filename, 'r') as w:
        if w.getsampwidth() != 2 or w.getnchannels() != 1:
            raise ValueError(f'{filename}: expected 16-bit mono PCM')
        return np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16)


def file_segments(pcm, fs=8000, duration=1., hop=.5):
    """All segments of one file as an (n_segs, seg_len) int16 array, tail zero-padded
    (load_audio's `audio_arr[:len(x)] = x`, audio_utils.py:261-263)."""
    seg_len, hop_len = int(fs * duration), int(np.floor(hop * fs))
    n = n_segments(len(pcm), fs, duration, hop)
    need = (n - 1) * hop_len + seg_len
    if len(pcm) < need:
        pcm = np.concatenate([pcm, np.zeros(need - len(pcm), np.int16)])
    idx = np.arange(seg_len)[None, :] + hop_len * np.arange(n)[:, None]
    return pcm[idx]


class SegmentSource:
    """Ordered segments of a list of WAV files, served in consecutive batches.

    Counterpart of `genUnbalSequence(fns, bsz, n_anchor=bsz, shuffle=False,
    drop_the_last_non_full_batch=False)` as the generate path uses it
    (model/dataset.py:204-215): `.n_samples` segments, `len()` batches of `bsz`
    (the last one ragged), item i = segments [i*bsz, (i+1)*bsz) as int16 (n,1,T).
    """

    def __init__(self, fns_list, bsz, duration=1., hop=.5, fs=8000):
        self.fns, self.bsz, self.duration, self.hop, self.fs = list(fns_list), int(bsz), duration, hop, fs
        self.seg_len = int(fs * duration)
        counts = [n_segments(wav_info(fn, fs), fs, duration, hop) for fn in self.fns]
        self.file_first = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        self.n_samples = int(self.file_first[-1])

    def __len__(self):
        return (self.n_samples + self.bsz - 1) // self.bsz

    def read_rows(self, row0, row1):
        """int16 (row1-row0, 1, seg_len) for global segment rows [row0, row1)."""
        out = np.zeros((row1 - row0, 1, self.seg_len), np.int16)
        f = int(np.searchsorted(self.file_first, row0, side='right') - 1)
        r = row0
        while r < row1:
            first, nxt = int(self.file_first[f]), int(self.file_first[f + 1])
            segs = file_segments(read_wav_int16(self.fns[f]), self.fs, self.duration, self.hop)
            a, b = r - first, min(row1, nxt) - first
            out[r - row0:r - row0 + (b - a), 0] = segs[a:b]
            r += b - a
            f += 1
        return out

    def __getitem__(self, i):
        if i < 0 or i >= len(self):
            raise IndexError(i)
        return self.read_rows(i * self.bsz, min((i + 1) * self.bsz, self.n_samples)), None

    def iter_rows(self, row0, row1, rows_per_chunk):
        """Yield (start_row, int16 chunk) over [row0, row1), each file read once."""
        cache_f, cache = -1, None
        r = row0
        while r < row1:
            end = min(r + rows_per_chunk, row1)
            out = np.zeros((end - r, 1, self.seg_len), np.int16)
            q = r
            while q < end:
                f = int(np.searchsorted(self.file_first, q, side='right') - 1)
                if f != cache_f:
                    cache = file_segments(read_wav_int16(self.fns[f]), self.fs, self.duration, self.hop)
                    cache_f = f
                first, nxt = int(self.file_first[f]), int(self.file_first[f + 1])
                a, b = q - first, min(end, nxt) - first
                out[q - r:q - r + (b - a), 0] = cache[a:b]
                q += b - a
            yield r, out
            r = end

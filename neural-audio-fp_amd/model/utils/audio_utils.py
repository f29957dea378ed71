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
import os
import struct
import wave

import numpy as np


def n_segments(n_frames, fs=8000, duration=1., hop=.5):
    """audio_utils.py:171-177."""
    n_seg_frames, n_hop_frames = fs * duration, fs * hop
    if n_frames > n_seg_frames:
        return int((n_frames - n_seg_frames + n_hop_frames) // n_hop_frames)
    return 1


def riff_scan(filename):
    """Header scan of a RIFF/WAVE file WITHOUT reading the samples: (fs, channels, sample width in
    bytes, byte offset of the first sample, frame count).  What `wave.open(...).getnframes()` reports
    (audio_utils.py:160-169), plus where the data chunk starts, so that a sample range can be read
    straight into a staging buffer."""
    size = os.path.getsize(filename)
    with open(filename, 'rb') as f:
        head = f.read(12)
        if len(head) < 12 or head[:4] != b'RIFF' or head[8:12] != b'WAVE':
            raise ValueError(f'{filename}: not a RIFF/WAVE file')
        fmt = None
        while True:
            hdr = f.read(8)
            if len(hdr) < 8:
                raise ValueError(f'{filename}: no data chunk')
            cid, csz = hdr[:4], struct.unpack('<I', hdr[4:])[0]
            if cid == b'fmt ':
                body = f.read(csz + (csz & 1))
                tag, ch, rate, _, align, bits = struct.unpack('<HHIIHH', body[:16])
                fmt = (tag, ch, rate, align, bits)
            elif cid == b'data':
                if fmt is None:
                    raise ValueError(f'{filename}: data chunk before fmt chunk')
                off = f.tell()
                csz = min(csz, size - off)
                return fmt[2], fmt[1], fmt[4] // 8, off, csz // fmt[3]
            else:
                f.seek(csz + (csz & 1), 1)


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
        self.hop_len = int(np.floor(hop * fs))
        self.n_frames, self.data_offset = [], []
        for fn in self.fns:                      # ONE header scan per file (the reference re-opens per segment)
            if fn[-3:] != 'wav':
                raise NotImplementedError(fn[-3:])
            rate, ch, width, off, nfr = riff_scan(fn)
            if rate != fs:
                raise ValueError('Sample rate should be {} but got {}'.format(str(fs), str(rate)))
            if width != 2 or ch != 1:
                raise ValueError(f'{fn}: expected 16-bit mono PCM')
            self.n_frames.append(nfr); self.data_offset.append(off)
        counts = [n_segments(nfr, fs, duration, hop) for nfr in self.n_frames]
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

    def iter_windows(self, row0, row1, rows_per_chunk, alloc=None):
        """Yield (start_row, n_rows, arena, used, seg_offset, seg_valid) over rows [row0, row1):
        the PCM every row needs is read ONCE from each file (contiguous sample range, straight into
        `arena` = alloc(n_samples), e.g. pinned memory) and row i is the window
        arena[seg_offset[i] : seg_offset[i] + seg_len] with samples >= seg_valid[i] meaning zero --
        the same segments `iter_rows` materialises, without the 2x duplication of HOP = DUR/2."""
        seg_len, hop_len = self.seg_len, self.hop_len
        r = row0
        while r < row1:
            end = min(r + rows_per_chunk, row1)
            n = end - r
            pieces, total, q = [], 0, r
            while q < end:
                f = int(np.searchsorted(self.file_first, q, side='right') - 1)
                first, nxt = int(self.file_first[f]), int(self.file_first[f + 1])
                a, b = q - first, min(end, nxt) - first
                s0 = a * hop_len
                s1 = max(s0, min((b - 1) * hop_len + seg_len, self.n_frames[f]))
                pieces.append((f, a, b, s0, s1, total, q - r))
                total += (s1 - s0 + 7) // 8 * 8                      # 16-B aligned piece starts
                q += b - a
            arena = alloc(total) if alloc is not None else np.empty(max(total, 1), np.int16)
            seg_offset = np.empty(n, np.int64)
            seg_valid = np.empty(n, np.int32)
            for f, a, b, s0, s1, base, o in pieces:
                if s1 > s0:
                    with open(self.fns[f], 'rb', buffering=0) as fh:
                        fh.seek(self.data_offset[f] + 2 * s0)
                        dst = memoryview(arena[base:base + (s1 - s0)]).cast('B')
                        got = fh.readinto(dst)
                        if got != 2 * (s1 - s0):
                            raise IOError(f'{self.fns[f]}: short read')
                k = np.arange(a, b, dtype=np.int64)
                seg_offset[o:o + (b - a)] = base + (k - a) * hop_len
                seg_valid[o:o + (b - a)] = np.clip(self.n_frames[f] - k * hop_len, 0, seg_len)
            yield r, n, arena, total, seg_offset, seg_valid
            r = end

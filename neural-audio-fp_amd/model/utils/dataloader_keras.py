"""Training / validation batches: (anchors, augmented replicas) assembled ON THE DEVICE.

Host mirror of the reference's `genUnbalSequence` (model/utils/dataloader_keras.py:11-482) for its
training and validation use (model/dataset.py:118-186): same constructor arguments, same
enumeration (`fns_event_seg_list` = [file, seg_idx, offset_min, offset_max], audio_utils.py:140-218),
same batch composition (n_anchor anchors + n_pos_per_anchor replicas each, replicas in anchor order),
same selection of background / speech / impulse-response items per batch index
(dataloader_keras.py:223-311), same distributions of the random draws (anchor / replica offsets
within +-offset_margin, background offset, SNR uniform in snr_range, amplitude ratio log-uniform in
(0.1, 1)).  `__getitem__` returns float32 CUDA tensors (n_anchor, 1, T), (n_pos, 1, T).

What is different, on purpose:
  * every WAV of every list is read ONCE into one int16 arena resident in HBM (`PcmStore`; the whole
    10k x 30 s training set is 4.8 GB of 288 GB).  A batch is then only a table of windows
    (`nafp_aug_row`, 64 B per row) and ONE kernel launch (`nafp_augment_rows`): no file I/O, no
    np.vstack, no host FFT per step;
  * random draws come from numpy Generators (the reference's global `np.random` stream cannot be
    matched, and it re-seeds that global stream with the sample index for every anchor,
    dataloader_keras.py:327-331, which makes the anchor offsets a function of the index; here every
    batch draws from Generator([seed, epoch, batch index]), vectorised over the batch);
  * `randint(low, high)` with low >= high (single-segment files) raises in the reference; here it
    yields `low`.
Unsupported (NotImplementedError): experimental_mode, amp_mode other than 'normal', seg_mode other than 'all'.
"""
import numpy as np
import torch

from ... import _lib
from .audio_utils import n_segments, riff_scan

MAX_IR_LENGTH = 600          # dataloader_keras.py:8


def segment_table(n_frames, fs, duration, hop, mode='all'):
    """(seg_idx, offset_min, offset_max) rows of one file (audio_utils.py:151-218)."""
    n_seg_frames, n_hop_frames = fs * duration, fs * hop
    n_segs = n_segments(n_frames, fs, duration, hop)
    residual = int(max(0, n_frames - ((n_segs - 1) * n_hop_frames + n_seg_frames)))
    if mode == 'first':
        return [(0, 0, 0)]
    if mode != 'all':
        raise NotImplementedError(f'seg_mode={mode}')
    out = []
    for s in range(n_segs):
        lo, hi = int(-1 * n_hop_frames), int(n_hop_frames)
        if s == 0:
            lo = 0
        if s == n_segs - 1:
            hi = residual
        out.append((s, lo, hi))
    return out


class PcmStore:
    """All samples of a list of 16-bit mono WAV files in ONE int16 array: `start[f]` (first sample of
    file f, 8-sample aligned), `n_frames[f]`.  `.host()` materialises it in numpy (tests, small sets),
    `.device()` uploads it once through a pinned staging buffer."""

    def __init__(self, fns, fs, base=0):
        self.fns, self.fs = list(fns), fs
        self.n_frames, self.data_offset, self.start = [], [], []
        pos = int(base)
        for fn in self.fns:
            if fn[-3:] != 'wav':
                raise NotImplementedError(fn[-3:])
            rate, ch, width, off, nfr = riff_scan(fn)
            if rate != fs:
                raise ValueError('Sample rate should be {} but got {}'.format(str(fs), str(rate)))
            if width != 2 or ch != 1:
                raise ValueError(f'{fn}: expected 16-bit mono PCM')
            self.n_frames.append(nfr); self.data_offset.append(off); self.start.append(pos)
            pos += (nfr + 7) // 8 * 8
        self.base, self.end = int(base), pos
        self.n_frames = np.asarray(self.n_frames, np.int64)
        self.start = np.asarray(self.start, np.int64)

    def read_into(self, dst):
        """dst: int16 numpy view of [base, end) (zero-initialised by the caller)."""
        for f, fn in enumerate(self.fns):
            n = int(self.n_frames[f])
            if n:
                with open(fn, 'rb', buffering=0) as fh:
                    fh.seek(self.data_offset[f])
                    a = int(self.start[f] - self.base)
                    got = fh.readinto(memoryview(dst[a:a + n]).cast('B'))
                    if got != 2 * n:
                        raise IOError(f'{fn}: short read')


class PcmArena:
    """Several PcmStores back to back in one arena (event | bg | speech | ir)."""

    def __init__(self, stores):
        self.stores = stores
        self.total = max([s.end for s in stores] + [8])

    def host(self):
        a = np.zeros(self.total, np.int16)
        for s in self.stores:
            s.read_into(a[s.base:s.end])
        return a

    def device(self, device=None, piece=1 << 25):
        device = device or torch.device('cuda', torch.cuda.current_device())
        d = torch.zeros((self.total,), dtype=torch.int16, device=device)
        stage = torch.zeros((min(piece, self.total),), dtype=torch.int16).pin_memory()
        for s in self.stores:                      # per store, in pieces of whole files
            f = 0
            while f < len(s.fns):
                g, a0 = f, int(s.start[f])
                while g < len(s.fns) and int(s.start[g]) + (int(s.n_frames[g]) + 7) // 8 * 8 - a0 <= stage.shape[0]:
                    g += 1
                if g == f:
                    raise NotImplementedError(f'{s.fns[f]}: one file larger than the staging buffer')
                a1 = int(s.start[g - 1]) + (int(s.n_frames[g - 1]) + 7) // 8 * 8
                buf = stage.numpy()[:a1 - a0]
                buf[:] = 0
                sub = PcmStore.__new__(PcmStore)
                sub.fns, sub.n_frames, sub.data_offset, sub.start, sub.base = s.fns[f:g], s.n_frames[f:g], s.data_offset[f:g], s.start[f:g], a0
                sub.read_into(buf)
                d[a0:a1].copy_(stage[:a1 - a0], non_blocking=False)
                f = g
        return d


def _randint(rng, low, high, size=None):
    """np.random.randint(low, high) (exclusive high); low >= high -> low (see module docstring)."""
    if high <= low:
        return low if size is None else np.full(size, low, dtype=np.int64)
    return rng.integers(low, high, size=size)


class genUnbalSequence:
    def __init__(self, fns_event_list, bsz=120, n_anchor=60, duration=1, hop=.5, fs=8000, shuffle=False,
                 seg_mode='all', amp_mode='normal', random_offset_anchor=False, offset_margin_hop_rate=0.4,
                 bg_mix_parameter=[False], ir_mix_parameter=[False], speech_mix_parameter=[False], reduce_items_p=0,
                 reduce_batch_first_half=False, experimental_mode=False, drop_the_last_non_full_batch=True,
                 seed=0, device=None, resident=True, shard=(0, 1)):
        if experimental_mode:
            raise NotImplementedError('experimental_mode')
        self.reduce_batch_first_half = reduce_batch_first_half
        if amp_mode != 'normal':
            raise NotImplementedError(f'amp_mode={amp_mode}')
        if seg_mode != 'all':
            raise NotImplementedError('seg_mode={}'.format(seg_mode))
        self.bsz, self.n_anchor = bsz, n_anchor
        if bsz != n_anchor:
            self.n_pos_per_anchor = round((bsz - n_anchor) / n_anchor)
            self.n_pos_bsz = bsz - n_anchor
        else:
            self.n_pos_per_anchor = 0
            self.n_pos_bsz = 0
        self.duration, self.hop, self.fs, self.shuffle = duration, hop, fs, shuffle
        self.random_offset_anchor = random_offset_anchor
        self.offset_margin_frame = int(hop * offset_margin_hop_rate * fs)
        self.seg_len = int(duration * fs)
        self.bg_mix, self.ir_mix, self.speech_mix = bg_mix_parameter[0], ir_mix_parameter[0], speech_mix_parameter[0]
        self.rng = np.random.default_rng(seed)
        self.seed, self.epoch = int(seed), 0
        # ---- stores + segment tables -------------------------------------------------------
        stores, pos = [], 0

        def add_store(fns):
            nonlocal pos
            s = PcmStore(fns, fs, base=pos)
            pos = s.end
            stores.append(s)
            return s

        def seg_list(store, dur, hp, mode='all'):
            rows = []                                       # (file, seg_idx, offset_min, offset_max)
            for f in range(len(store.fns)):
                rows += [(f,) + t for t in segment_table(int(store.n_frames[f]), fs, dur, hp, mode)]
            return np.asarray(rows, np.int64).reshape(-1, 4)

        self.ev = add_store(fns_event_list)
        self.fns_event_seg_list = seg_list(self.ev, duration, hop)
        if drop_the_last_non_full_batch:
            self.n_samples = int((len(self.fns_event_seg_list) // n_anchor) * n_anchor)
        else:
            self.n_samples = len(self.fns_event_seg_list)
        self.index_event = self.rng.permutation(self.n_samples) if shuffle else np.arange(self.n_samples)
        if self.bg_mix:
            self.bg = add_store(bg_mix_parameter[1]); self.bg_snr_range = bg_mix_parameter[2]
            self.fns_bg_seg_list = seg_list(self.bg, duration, duration)
            self.n_bg_samples = len(self.fns_bg_seg_list)
            self.index_bg = self.rng.permutation(self.n_bg_samples) if shuffle else np.arange(self.n_bg_samples)
        if self.speech_mix:
            self.sp = add_store(speech_mix_parameter[1]); self.speech_snr_range = speech_mix_parameter[2]
            self.fns_speech_seg_list = seg_list(self.sp, duration, duration)
            self.n_speech_samples = len(self.fns_speech_seg_list)
            self.index_speech = self.rng.permutation(self.n_speech_samples) if shuffle else np.arange(self.n_speech_samples)
        if self.ir_mix:
            self.ir = add_store(ir_mix_parameter[1])
            self.fns_ir_seg_list = seg_list(self.ir, duration, duration, 'first')
            self.n_ir_samples = len(self.fns_ir_seg_list)
            self.index_ir = self.rng.permutation(self.n_ir_samples) if shuffle else np.arange(self.n_ir_samples)
        self.reduce_items_p = reduce_items_p
        assert reduce_items_p <= 100
        # data parallel: bsz / n_anchor are the GLOBAL batch, every rank builds the SAME permutation and the
        # same draws (same seed) and keeps rows [rank * n_anchor/world, (rank+1) * n_anchor/world) of each batch
        # with their replicas, so that len(ds), the epoch and the batches equal the single-process run
        self.rank, self.world = int(shard[0]), int(shard[1])
        if not 0 <= self.rank < self.world or n_anchor % self.world:
            raise ValueError(f'shard={shard}: n_anchor={n_anchor} must split evenly over the ranks')
        self.arena = PcmArena(stores)
        self.device = device
        self._pcm = None
        self._lib = None
        if not resident:
            raise NotImplementedError('streaming (non-resident) PCM arena')
        self.set_epoch(0)

    def __len__(self):
        """dataloader_keras.py:187-195."""
        if self.reduce_items_p != 0:
            return int(np.ceil(self.n_samples / float(self.n_anchor)) * (self.reduce_items_p / 100))
        return int(np.ceil(self.n_samples / float(self.n_anchor)))

    def on_epoch_end(self):
        """dataloader_keras.py:198-222."""
        self.set_epoch(self.epoch + 1)

    def set_epoch(self, epoch):
        """Permutations of epoch `epoch` (0-based): a function of (seed, epoch) only, so a restarted run that
        resumes at epoch e continues with the permutations and draws the uninterrupted run would have used."""
        self.epoch = int(epoch)
        if self.shuffle:
            rng = np.random.default_rng([self.seed, 0x5eed, self.epoch])
            self.index_event = rng.permutation(self.n_samples)
            if self.bg_mix:
                self.index_bg = rng.permutation(self.n_bg_samples)
            if self.ir_mix:
                self.index_ir = rng.permutation(self.n_ir_samples)
            if self.speech_mix:
                self.index_speech = rng.permutation(self.n_speech_samples)

    # ---- the batch as a table of windows (host only; unit-testable without a GPU) -----------
    def plan(self, idx):
        """nafp_aug_row table of batch `idx`: rows [0, n_anchor) anchors, then the replicas in anchor
        order (dataloader_keras.py:223-311, 316-398).  Vectorised over the batch."""
        anchors = self.index_event[idx * self.n_anchor:(idx + 1) * self.n_anchor]
        nA, npa, T = len(anchors), self.n_pos_per_anchor, self.seg_len
        nP = nA * npa
        rows = np.zeros(nA + nP, dtype=_lib.AUG_ROW_DTYPE)
        rows['nz_off'] = -1; rows['nz2_off'] = -1; rows['ir_off'] = -1; rows['amp'] = 1.0
        rng = np.random.default_rng([self.seed, self.epoch, int(idx)])
        tab = self.fns_event_seg_list[anchors]                       # (nA, 4): file, seg, offset_min, offset_max
        f, seg, off_min, off_max = tab[:, 0], tab[:, 1], tab[:, 2], tab[:, 3]
        m = self.offset_margin_frame
        if self.random_offset_anchor:
            lo, hi = np.maximum(off_min, -m), np.minimum(off_max, m)
            a_off = np.where(hi > lo, rng.integers(lo, np.maximum(hi, lo + 1)), lo)     # randint(low, high), exclusive high
        else:
            a_off = np.zeros(nA, np.int64)
        start_of = lambda off: np.floor((seg[:, None] * self.hop + off / self.fs) * self.fs).astype(np.int64)
        starts = start_of(a_off[:, None])                            # (nA, 1)
        if npa > 0:
            p_min, p_max = np.maximum(a_off - m, off_min), np.minimum(a_off + m, off_max)
            draw = rng.integers(p_min[:, None], np.maximum(p_max, p_min + 1)[:, None], size=(nA, npa))
            p_off = np.where((p_max > p_min)[:, None], draw, p_min[:, None])
            # p_min == p_max == 0 (dataloader_keras.py:361-366) is the constant case of the line above
            starts = np.concatenate([starts, start_of(p_off)], axis=1)    # (nA, 1 + npa)
        n_fr, base = self.ev.n_frames[f], self.ev.start[f]
        ev_off = base[:, None] + starts
        ev_valid = np.clip(n_fr[:, None] - starts, 0, T)
        rows['ev_off'][:nA] = ev_off[:, 0]; rows['ev_valid'][:nA] = ev_valid[:, 0]
        if nP > 0:
            rows['ev_off'][nA:] = ev_off[:, 1:].reshape(-1); rows['ev_valid'][nA:] = ev_valid[:, 1:].reshape(-1)
            sel = np.arange(idx * self.n_pos_bsz, idx * self.n_pos_bsz + nP)
            rep = slice(nA, nA + nP)

            def noise_windows(store, seg_list, index, n_items):
                """__bg_batch_load / __speech_batch_load (dataloader_keras.py:400-452)."""
                t = seg_list[index[sel % n_items] % n_items]
                rnd = rng.integers(0, max(int(self.duration * self.fs / 2), 1), size=nP)
                offset_sec = np.minimum(rnd / self.fs, t[:, 3] / self.fs)
                st = np.floor((t[:, 1] * self.duration + offset_sec) * self.fs).astype(np.int64)
                return store.start[t[:, 0]] + st, np.clip(store.n_frames[t[:, 0]] - st, 0, T)

            snr_range = None
            if self.bg_mix:
                rows['nz_off'][rep], rows['nz_valid'][rep] = noise_windows(self.bg, self.fns_bg_seg_list, self.index_bg, self.n_bg_samples)
                snr_range = self.bg_snr_range
            if self.speech_mix:
                key = 'nz2' if self.bg_mix else 'nz'
                rows[key + '_off'][rep], rows[key + '_valid'][rep] = noise_windows(self.sp, self.fns_speech_seg_list, self.index_speech,
                                                                                 self.n_speech_samples)
                # both: noise = bg + speech mixed at speech_snr_range (:241-259); speech alone uses bg_snr_range (:291-294)
                snr_range = self.speech_snr_range if self.bg_mix else getattr(self, 'bg_snr_range', self.speech_snr_range)
            if snr_range is not None:
                lo, hi = float(np.min(snr_range)), float(np.max(snr_range))
                rows['snr_db'][rep] = rng.random(nP) * (hi - lo) + lo                               # audio_utils.py:92-95
                rows['amp'][rep] = np.power(10.0, rng.random(nP) * (np.log10(1.0) - np.log10(0.1)) + np.log10(0.1))   # :75-79
                rows['mix'][rep] = 1
            if self.ir_mix:
                fi = self.fns_ir_seg_list[self.index_ir[sel % self.n_ir_samples] % self.n_ir_samples][:, 0]
                rows['ir_off'][rep] = self.ir.start[fi]
                rows['ir_len'][rep] = np.minimum(np.minimum(self.ir.n_frames[fi], MAX_IR_LENGTH), T)
        if self.world > 1:
            # this rank's anchors of the global batch and their replicas (a ragged last batch splits as evenly
            # as its anchors allow)
            a0, a1 = (nA * self.rank) // self.world, (nA * (self.rank + 1)) // self.world
            rows = np.concatenate([rows[a0:a1], rows[nA + a0 * npa:nA + a1 * npa]])
        return rows

    def n_local_anchors(self, idx):
        nA = min(self.n_anchor, self.n_samples - idx * self.n_anchor)
        return (nA * (self.rank + 1)) // self.world - (nA * self.rank) // self.world

    # ---- device ---------------------------------------------------------------------------
    def _resident(self):
        if self._pcm is None:
            self._pcm = self.arena.device(self.device)
            self._lib = _lib.load()
        return self._pcm

    def run(self, rows):
        """(n_rows, 1, T) float32 CUDA batch of a row table."""
        pcm = self._resident()
        d_rows = torch.from_numpy(rows.view(np.uint8).reshape(-1)).to(pcm.device)
        out = torch.empty((len(rows), 1, self.seg_len), dtype=torch.float32, device=pcm.device)
        with torch.cuda.device(pcm.device):
            _lib.check(self._lib.nafp_augment_rows(_lib.ptr(pcm), _lib.ptr(d_rows), len(rows), self.seg_len, _lib.ptr(out),
                                                   _lib.current_stream()), 'augment_rows')
        return out

    def __getitem__(self, idx):
        if idx < 0 or idx >= len(self):
            raise IndexError(idx)
        rows = self.plan(idx)
        nA = self.n_local_anchors(idx)
        if self.reduce_batch_first_half:           # synthesized queries only (dataloader_keras.py:308-309): the anchors are not built
            return self.run(rows[nA:]), []
        out = self.run(rows)
        return out[:nA], out[nA:]

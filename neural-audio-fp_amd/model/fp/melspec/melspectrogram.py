"""Host mirror of the reference's log-mel layer, backed by libnafp's HIP front end.

Mirrors `Melspec_layer` / `get_melspec_layer` of the reference
(model/fp/melspec/melspectrogram.py:10-141): same constructor arguments, same
call contract `(B,1,T) float32 -> (B, n_mels, n_frames, 1) float32`, same
`NotImplementedError` on an unknown FEAT.  Tensors are torch CUDA tensors; there
is no CPU path.

One addition the reference does not need: `group_size`.  The reference subtracts
the max over the whole device batch (melspectrogram.py:108), and its batches are
`TS_BATCH_SZ` consecutive segments; here a launch may hold several such batches,
so the grouping is a parameter (None = the whole input is one group, exactly the
reference's behaviour for one batch).
"""
import ctypes

import torch

from .... import _lib


class DeferredFeatures:
    """What `Melspec_layer(x, defer=True)` returns: the raw log10-mel tensor plus the per-group (max, min), i.e. the
    layer stopped before `x - reduce_max(x)` (melspectrogram.py:108).  `m_fp(deferred)` finishes the layer inside its
    first conv (nafp_encoder_forward_raw): same fingerprints, one kernel and one round trip of the feature tensor
    through HBM fewer.  `.finish()` materialises the ordinary (B, n_mels, n_frames, 1) tensor."""

    def __init__(self, layer, raw, gstat, group_size, segment_norm):
        self.layer, self.raw, self.gstat = layer, raw, gstat
        self.group_size, self.segment_norm = int(group_size), bool(segment_norm)
        self.shape, self.device = raw.shape, raw.device

    def finish(self):
        out = self.raw.clone()
        lay = self.layer
        with torch.cuda.device(out.device):
            _lib.check(lay._lib.nafp_melspec_finish(lay._h, _lib.ptr(out), _lib.ptr(self.gstat), out.shape[0], self.group_size,
                                                    int(self.segment_norm), _lib.current_stream()), 'melspec_finish')
        return out


class Melspec_layer:
    def __init__(self, input_shape=(1, 8000), segment_norm=False, n_fft=1024, stft_hop=256,
                 n_mels=256, fs=8000, dur=1., f_min=300., f_max=4000., amin=1e-10,
                 dynamic_range=80., name='Mel-spectrogram', trainable=False, **kwargs):
        if amin != 1e-10 or dynamic_range != 80.:
            # fixed by the reference's constructor defaults (melspectrogram.py:36-37);
            # get_melspec_layer never overrides them
            raise NotImplementedError('amin / dynamic_range other than 1e-10 / 80 dB')
        self.name = name
        self.trainable = False
        self.n_fft, self.stft_hop, self.n_mels = n_fft, stft_hop, n_mels
        self.amin, self.dynamic_range, self.segment_norm = amin, dynamic_range, segment_norm
        self.mel_fb_kwargs = {'sample_rate': fs, 'n_freq': n_fft // 2 + 1, 'n_mels': n_mels,
                              'f_min': f_min, 'f_max': f_max}
        self.pad_l = self.pad_r = n_fft // 2
        self.seg_len = int(input_shape[1])
        self.padded_input_shape = (1, int(fs * dur) + self.pad_l + self.pad_r)
        self.group_size = kwargs.pop('group_size', None)
        lib = _lib.load()
        h = ctypes.c_void_p()
        _lib.check(lib.nafp_melspec_create(ctypes.byref(h), int(fs), self.seg_len, int(n_fft),
                                           int(stft_hop), int(n_mels), float(f_min), float(f_max)),
                   'melspec_create')
        self._h, self._lib = h, lib
        self.n_frames = lib.nafp_melspec_n_frames(h)

    def __del__(self):
        h = getattr(self, '_h', None)
        if h:
            self._lib.nafp_melspec_destroy(h)
            self._h = None

    def __call__(self, x, group_size=None, defer=False):
        """`defer=True`: return `DeferredFeatures` (raw log-mel + group statistics) for `m_fp` to finish."""
        x = torch.as_tensor(x)
        if not x.is_cuda:
            x = x.cuda()
        if x.dim() != 3 or x.shape[1] != 1 or x.shape[2] != self.seg_len:
            raise ValueError(f'expected (B,1,{self.seg_len}), got {tuple(x.shape)}')
        if x.dtype not in (torch.float32, torch.int16):
            x = x.float()
        x = x.contiguous()
        B = x.shape[0]
        g = group_size if group_size is not None else self.group_size
        g = int(g) if g else 0
        n_groups = 1 if g <= 0 else (B + g - 1) // g
        feat = torch.empty((B, self.n_mels, self.n_frames, 1), dtype=torch.float32, device=x.device)
        gstat = torch.empty((2 * max(n_groups, 1),), dtype=torch.float32, device=x.device)
        fn = self._lib.nafp_melspec_forward_i16 if x.dtype == torch.int16 else self._lib.nafp_melspec_forward_f32
        flags = int(bool(self.segment_norm)) | (2 if defer else 0)            # NAFP_MELSPEC_DEFER
        with torch.cuda.device(x.device):
            _lib.check(fn(self._h, _lib.ptr(x), B, g, flags, _lib.ptr(feat),
                          _lib.ptr(gstat), _lib.current_stream()), 'melspec_forward')
        return DeferredFeatures(self, feat, gstat, g, self.segment_norm) if defer else feat


    def forward_windows(self, pcm, seg_offset, seg_valid, group_size=None, defer=False):
        """Same output as __call__ on the materialised segments: segment i = pcm[seg_offset[i] :
        seg_offset[i] + seg_len] with samples >= seg_valid[i] read as zero.  pcm int16 (n,),
        seg_offset int64 (B,), seg_valid int32 (B,), all CUDA."""
        for t, dt, nm in ((pcm, torch.int16, 'pcm'), (seg_offset, torch.int64, 'seg_offset'), (seg_valid, torch.int32, 'seg_valid')):
            _lib.require_cuda(t, nm)
            if t.dtype != dt or not t.is_contiguous() or t.dim() != 1:
                raise ValueError(f'{nm}: expected a contiguous 1-D {dt} tensor')
        B = seg_offset.shape[0]
        if seg_valid.shape[0] != B:
            raise ValueError('seg_offset and seg_valid differ in length')
        g = group_size if group_size is not None else self.group_size
        g = int(g) if g else 0
        n_groups = 1 if g <= 0 else (B + g - 1) // g
        feat = torch.empty((B, self.n_mels, self.n_frames, 1), dtype=torch.float32, device=pcm.device)
        gstat = torch.empty((2 * max(n_groups, 1),), dtype=torch.float32, device=pcm.device)
        flags = int(bool(self.segment_norm)) | (2 if defer else 0)
        with torch.cuda.device(pcm.device):
            _lib.check(self._lib.nafp_melspec_forward_windows_i16(
                self._h, _lib.ptr(pcm), _lib.ptr(seg_offset), _lib.ptr(seg_valid), B, g, flags,
                _lib.ptr(feat), _lib.ptr(gstat), _lib.current_stream()), 'melspec_forward_windows')
        return DeferredFeatures(self, feat, gstat, g, self.segment_norm) if defer else feat


def get_melspec_layer(cfg, trainable=False):
    """melspectrogram.py:115-141."""
    fs = cfg['MODEL']['FS']
    dur = cfg['MODEL']['DUR']
    if cfg['MODEL']['FEAT'] == 'melspec':
        segment_norm = False
    elif cfg['MODEL']['FEAT'] == 'melspec_maxnorm':
        segment_norm = True
    else:
        raise NotImplementedError(cfg['MODEL']['FEAT'])
    layer = Melspec_layer(input_shape=(1, int(fs * dur)), segment_norm=segment_norm,
                          n_fft=cfg['MODEL']['STFT_WIN'], stft_hop=cfg['MODEL']['STFT_HOP'],
                          n_mels=cfg['MODEL']['N_MELS'], fs=fs, dur=dur,
                          f_min=cfg['MODEL']['F_MIN'], f_max=cfg['MODEL']['F_MAX'])
    layer.trainable = trainable
    return layer

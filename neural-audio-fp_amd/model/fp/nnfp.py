"""Host mirror of the reference's fingerprinter, backed by libnafp's HIP encoder.

Mirrors `FingerPrinter` / `get_fingerprinter` of the reference
(model/fp/nnfp.py:159-258): `m_fp(feat) -> (B, emb_sz)`, `.front_conv(feat) ->
(B, 1024)`, `.div_enc(x) -> (B, emb_sz)`, attributes `front_hidden_ch`,
`front_strides`, `emb_sz`, `norm`, `use_L2layer`, `trainable`,
`trainable_variables` (used at trainer.py:42,47-48,73-75,87-88).

Parameters are torch CUDA tensors in the keras variable SHAPES, listed in the library's own fixed order
(include/nafp.h, "Parameter tensors": per conv kernel, bias, LN gamma, LN beta; the order of keras'
`trainable_variables` is not relied upon anywhere -- checkpoints are keyed by name, `tensor_names()`); they are pushed to the library (which
re-packs them for its kernels) lazily, whenever they were replaced or marked
dirty.  No CPU path.
"""
import ctypes
import math
import os

import torch

from ... import _lib

FRONT_HIDDEN_CH = [128, 128, 256, 256, 512, 512, 1024, 1024]            # nnfp.py:193
FRONT_STRIDES = [[(1, 2), (2, 1)], [(1, 2), (2, 1)], [(1, 2), (2, 1)], [(1, 2), (2, 1)],
                 [(1, 1), (2, 1)], [(1, 2), (2, 1)], [(1, 1), (2, 1)], [(1, 2), (2, 1)]]  # nnfp.py:194-197


NORM_KINDS = {'layer_norm2d': 0, 'layer_norm1d': 1}          # anything else is batch normalisation (nnfp.py:63-71): 2


def norm_kind(norm):
    """NAFP_NORM_* of a config MODEL.BN string: 'layer_norm1d', 'layer_norm2d', else ('batch_norm' or any other string, as the
    `else` of nnfp.py:69-71) batch normalisation."""
    return NORM_KINDS.get(norm, 2)


def tensor_names(norm='layer_norm2d'):
    """Checkpoint key of each parameter tensor, in library order (batch_norm: the 32 non-trainable moving statistics follow the
    68 trainable tensors)."""
    names = []
    for j in range(16):
        blk, kind = j // 2, ('conv2d_1x3' if j % 2 == 0 else 'conv2d_3x1')
        bn = 'BN_1x3' if j % 2 == 0 else 'BN_3x1'
        names += [f'front_conv.{blk}.{kind}.kernel', f'front_conv.{blk}.{kind}.bias',
                  f'front_conv.{blk}.{bn}.gamma', f'front_conv.{blk}.{bn}.beta']
    names += ['div_enc.fc1.kernel', 'div_enc.fc1.bias', 'div_enc.fc2.kernel', 'div_enc.fc2.bias']
    if norm_kind(norm) == 2:
        for j in range(16):
            blk, bn = j // 2, ('BN_1x3' if j % 2 == 0 else 'BN_3x1')
            names += [f'front_conv.{blk}.{bn}.moving_mean', f'front_conv.{blk}.{bn}.moving_variance']
    return names


class FingerPrinter:
    def __init__(self, input_shape=(256, 32, 1), front_hidden_ch=None, front_strides=None,
                 emb_sz=128, fc_unit_dim=(32, 1), norm='layer_norm2d', use_L2layer=True,
                 device=None, seed=None):
        front_hidden_ch = list(front_hidden_ch or FRONT_HIDDEN_CH)
        front_strides = [list(map(tuple, s)) for s in (front_strides or FRONT_STRIDES)]
        if front_hidden_ch != FRONT_HIDDEN_CH or front_strides != FRONT_STRIDES or \
                list(fc_unit_dim) != [32, 1] or input_shape[2] != 1:
            raise NotImplementedError('only the channel/stride tables of nnfp.py:193-197 are built')
        # norm (config MODEL.BN, nnfp.py:63-71, 250): 'layer_norm2d' (default), 'layer_norm1d', anything else = BatchNormalization --
        # which the reference only ever CALLS in inference mode (`m_fp(feat)` without `training=`: trainer.py:44, generate.py:88), so
        # it is the per-channel affine map of its moving statistics (initially 0 / 1, never updated) with trainable gamma / beta
        self.front_hidden_ch, self.front_strides = front_hidden_ch, front_strides
        self.emb_sz, self.norm, self.use_L2layer = emb_sz, norm, use_L2layer
        self.n_clayers = len(front_strides)
        self.input_shape = tuple(input_shape)
        self.trainable = False
        self.device = torch.device(device if device is not None else 'cuda')
        lib = _lib.load()
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.nafp_encoder_create_ex(ctypes.byref(h), int(input_shape[0]), int(input_shape[1]),
                                                  int(emb_sz), norm_kind(norm)), 'encoder_create_ex')
        self._h, self._lib = h, lib
        self.flat_dim = int(lib.nafp_encoder_flat_dim(h))
        self._shapes = []
        dims = (ctypes.c_int64 * 4)()
        for i in range(lib.nafp_encoder_n_tensors(h)):
            r = lib.nafp_encoder_tensor_shape(h, i, dims)
            self._shapes.append(tuple(int(dims[k]) for k in range(r)))
        self._names = tensor_names(norm)
        self._n_trainable = int(lib.nafp_encoder_n_trainable(h))
        self._vars = self._init_variables(seed)
        self._dirty = True
        self._weights_event, self._weights_stream, self._use_events = None, None, {}
        self._fuse0 = os.environ.get('NAFP_FUSE0', '') == '1'          # NAFP_OPT_FUSE_CONV0 (the library reads the same variable)
        self.split_arithmetic = 0      # NAFP_OPT_BF16X3 of the handle (what bench.py / the tests report it as)
        self._ws = {}          # one workspace per HIP stream: batches may be pipelined across streams
        # NAFP_BF16X3=1 | 2 (environment): the experimental split-bf16 products of the forward -- with 2 also of forward_train and the transposed convs (include/nafp.h NAFP_OPT_BF16X3;
        # 2 = the exact 3-way split: float32-equivalent, ~20 % faster than the fp32 MFMAs) for `run.py generate` without a code change
        if os.environ.get('NAFP_BF16X3', '') in ('1', '2'):
            self.set_option(3, int(os.environ['NAFP_BF16X3']))

    # ---- parameters -------------------------------------------------------
    def _init_variables(self, seed):
        """keras defaults: glorot-uniform kernels, zero biases, LN gamma=1 / beta=0
        (nnfp.py:48-59, 66-67, 135-137)."""
        g = torch.Generator().manual_seed(int(seed) if seed is not None else torch.seed() % (2 ** 31))
        out = []
        for i, shp in enumerate(self._shapes):
            if i < 64:
                kind = i % 4
                if kind == 0:
                    kh, kw, cin, cout = shp
                    lim = math.sqrt(6.0 / (kh * kw * cin + kh * kw * cout))
                    t = (torch.rand(shp, generator=g) * 2 - 1) * lim
                elif kind == 2:
                    t = torch.ones(shp)
                else:
                    t = torch.zeros(shp)
            elif i in (64, 66):
                fan_in, fan_out = shp[1], shp[2]
                lim = math.sqrt(6.0 / (fan_in + fan_out))
                t = (torch.rand(shp, generator=g) * 2 - 1) * lim
            elif i >= 68:                                   # batch_norm: moving_mean = 0, moving_variance = 1 (keras initialisers)
                t = torch.zeros(shp) if i % 2 == 0 else torch.ones(shp)
            else:
                t = torch.zeros(shp)
            out.append(t.to(self.device, torch.float32).contiguous())
        return out

    @property
    def trainable_variables(self):
        return list(self._vars[:self._n_trainable])

    @property
    def non_trainable_variables(self):
        """batch_norm: the moving statistics (keras keeps them in `model.non_trainable_variables`); else empty."""
        return list(self._vars[self._n_trainable:])

    def variable_lengths(self):
        """Length of one keras variable inside each tensor (LAMB's trust ratio is per keras
        variable): the 4 divide-and-encode tensors stack emb_sz variables each."""
        return [v.numel() if i < 64 else v.numel() // self.emb_sz for i, v in enumerate(self._vars[:self._n_trainable])]

    def mark_dirty(self):
        """Call after modifying a variable in place (e.g. an optimizer step)."""
        self._dirty = True

    def state_dict(self):
        return {n: v.detach().cpu().clone() for n, v in zip(self._names, self._vars)}

    def load_state_dict(self, sd):
        missing = [n for n in self._names if n not in sd]
        if missing:
            raise KeyError(f'missing checkpoint keys: {missing[:4]}...')
        for i, n in enumerate(self._names):
            t = torch.as_tensor(sd[n], dtype=torch.float32)
            if tuple(t.shape) != self._shapes[i]:
                raise ValueError(f'{n}: shape {tuple(t.shape)} != {self._shapes[i]}')
            self._vars[i].copy_(t)             # in place: optimizer slots and gradient views stay valid
        self._dirty = True

    def set_weights(self, arrays):
        """arrays: sequence of array-likes in library order (keras shapes): 68, or 100 with batch normalisation."""
        self.load_state_dict({n: a for n, a in zip(self._names, arrays)})

    def _sync(self):
        """Push replaced / modified variables to the library, ordered against every stream that uses
        the handle: the re-pack (copy, pack, G/Hb launches) writes the handle's shared packed-weight
        blob, so (i) it first waits for the forwards still running on OTHER streams, and (ii) every later
        pass on another stream waits for it INSIDE the library (`nafp_encoder_set_weights` records an event
        after its plain copies and one at its end; the passes wait for the one they need right before the
        first launch that needs it -- the training forward starts its first conv behind the copies only)."""
        if not self._dirty:
            return
        cur = torch.cuda.current_stream(self.device)
        self._wait_weights()               # a re-pack still in flight on another stream (prefetch_weights) writes the same blob: behind it
        for key, ev in list(self._use_events.items()):
            if key != cur.cuda_stream:
                cur.wait_event(ev)
        arr = (ctypes.c_void_p * len(self._vars))(*[v.data_ptr() for v in self._vars])
        with torch.cuda.device(self.device):
            _lib.check(self._lib.nafp_encoder_set_weights(self._h, arr, _lib.current_stream()),
                       'encoder_set_weights')
            self._weights_event = torch.cuda.Event()
            self._weights_event.record(cur)
        self._weights_stream = cur.cuda_stream
        self._use_events = {}
        self._dirty = False

    def prefetch_weights(self):
        """Start the re-pack of modified variables NOW, on a stream of the handle's own (default priority), instead of on the
        caller's stream at the next forward: `nafp_encoder_set_weights` (copies, re-layouts, the G / Hb images: ~0.3 ms) then
        runs next to whatever the caller enqueues before that forward -- in `train_step` the front end of the next batch
        (concatenation, log-mel, spec-augment).  Ordered like `_sync`: behind everything the current stream has enqueued
        (the optimizer step that modified the variables, the backward pass that read the packed weights), and every later
        forward waits for its completion event."""
        if not self._dirty:
            return
        cur = torch.cuda.current_stream(self.device)
        if getattr(self, '_prep_stream', None) is None:
            self._prep_stream = torch.cuda.Stream(device=self.device)
        self._prep_stream.wait_stream(cur)
        with torch.cuda.stream(self._prep_stream):
            self._sync()

    def _wait_weights(self):
        ev = self._weights_event
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            if cur.cuda_stream != self._weights_stream:
                cur.wait_event(ev)

    def _mark_use(self):
        """Record that the current stream has read the packed weights up to here (see `_sync`)."""
        cur = torch.cuda.current_stream(self.device)
        ev = self._use_events.get(cur.cuda_stream)
        if ev is None:
            ev = self._use_events[cur.cuda_stream] = torch.cuda.Event()
        ev.record(cur)

    def __del__(self):
        h = getattr(self, '_h', None)
        if h:
            self._lib.nafp_encoder_destroy(h)
            self._h = None

    # ---- forward ------------------------------------------------------------
    def _prep(self, x, shape_tail):
        x = torch.as_tensor(x)
        if not x.is_cuda:
            x = x.to(self.device)
        if tuple(x.shape[1:]) != tuple(shape_tail):
            raise ValueError(f'expected (B,{",".join(map(str, shape_tail))}), got {tuple(x.shape)}')
        return x.float().contiguous()

    def _workspace(self, n):
        need = int(self._lib.nafp_encoder_workspace_bytes(self._h, n))
        key = torch.cuda.current_stream(self.device).cuda_stream
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty((need,), dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        return ws, need

    def _forward(self, feat, want_flat, want_emb):
        deferred = None
        if hasattr(feat, 'raw') and hasattr(feat, 'gstat'):        # melspectrogram.DeferredFeatures
            deferred, feat = feat, feat.raw       # (the fused conv0 generator applies the layer's tail on load too: conv.hip)
        feat = self._prep(feat, self.input_shape)
        self._sync()
        B = feat.shape[0]
        flat = torch.empty((B, self.flat_dim), dtype=torch.float32, device=feat.device) if want_flat else None
        emb = torch.empty((B, self.emb_sz), dtype=torch.float32, device=feat.device) if want_emb else None
        ws, need = self._workspace(B)
        with torch.cuda.device(feat.device):
            if deferred is not None:
                _lib.check(self._lib.nafp_encoder_forward_raw(self._h, _lib.ptr(feat), _lib.ptr(deferred.gstat),
                                                              deferred.group_size, int(deferred.segment_norm), B,
                                                              _lib.ptr(ws), need, _lib.ptr(flat), _lib.ptr(emb),
                                                              int(bool(self.use_L2layer)), _lib.current_stream()),
                           'encoder_forward_raw')
            else:
                _lib.check(self._lib.nafp_encoder_forward(self._h, _lib.ptr(feat), B, _lib.ptr(ws), need,
                                                          _lib.ptr(flat), _lib.ptr(emb),
                                                          int(bool(self.use_L2layer)), _lib.current_stream()),
                           'encoder_forward')
            self._mark_use()
        return flat, emb

    # ---- training: forward that keeps activations + backward (trainer.py:41-47) ----------
    def forward_train(self, feat):
        """emb = m_fp(feat) keeping every activation for `backward`."""
        feat = self._prep(feat, self.input_shape)
        self._sync()
        B = feat.shape[0]
        need = int(self._lib.nafp_encoder_train_workspace_bytes(self._h, B))
        if getattr(self, '_train_ws', None) is None or self._train_ws.numel() < need:
            self._train_ws = torch.empty((need,), dtype=torch.uint8, device=self.device)
        emb = torch.empty((B, self.emb_sz), dtype=torch.float32, device=feat.device)
        with torch.cuda.device(feat.device):
            _lib.check(self._lib.nafp_encoder_forward_train(self._h, _lib.ptr(feat), B, _lib.ptr(self._train_ws), need,
                                                            _lib.ptr(emb), int(bool(self.use_L2layer)),
                                                            _lib.current_stream()), 'encoder_forward_train')
            self._mark_use()
        self._train_feat = feat
        return emb

    def backward(self, d_emb):
        """Gradients of the 68 parameter tensors (keras shapes, library order) for dL/d(emb) of the
        last forward_train call; the counterpart of tape.gradient(loss, m_fp.trainable_variables)."""
        feat = self._train_feat
        B = feat.shape[0]
        d_emb = _lib.require_cuda(torch.as_tensor(d_emb), 'd_emb').float().contiguous()
        if getattr(self, '_grads', None) is None:
            self._grads = [torch.empty_like(v) for v in self._vars[:self._n_trainable]]
        arr = (ctypes.c_void_p * len(self._grads))(*[g.data_ptr() for g in self._grads])
        need = int(self._lib.nafp_encoder_train_workspace_bytes(self._h, B))
        if getattr(self, '_train_ws', None) is None or need > self._train_ws.numel():
            raise RuntimeError('backward: the training workspace does not match the last forward_train '
                               '(an execution option was changed in between?)')
        with torch.cuda.device(feat.device):
            _lib.check(self._lib.nafp_encoder_backward(self._h, _lib.ptr(feat), _lib.ptr(d_emb), B,
                                                       _lib.ptr(self._train_ws), need, arr,
                                                       int(bool(self.use_L2layer)), _lib.current_stream()),
                       'encoder_backward')
            self._mark_use()
        return self._grads

    def grad_groups(self):
        """[(first_tensor, last_tensor)] of the library's gradient groups, in completion order of `backward`."""
        out = []
        a, b = ctypes.c_int(), ctypes.c_int()
        k = 0
        while self._lib.nafp_encoder_grad_group_range(self._h, k, ctypes.byref(a), ctypes.byref(b)) == 0:
            out.append((a.value, b.value))
            k += 1
        return out

    def grad_group_wait(self, group):
        """The CURRENT stream waits (on the device) until gradient group `group` of the last `backward` is complete."""
        with torch.cuda.device(self.device):
            _lib.check(self._lib.nafp_encoder_grad_group_wait(self._h, int(group), _lib.current_stream()),
                       'encoder_grad_group_wait')

    def set_option(self, option, value):
        """Execution options of the library handle (include/nafp.h NAFP_OPT_*); results do not change, except for the
        experimental option 3 (NAFP_OPT_BF16X3: split-bf16 products, off by default)."""
        if int(option) == 1:
            self._fuse0 = bool(value)
        if int(option) == 3:
            self.split_arithmetic = int(value)
        _lib.check(self._lib.nafp_encoder_set_option(self._h, int(option), int(value)), 'encoder_set_option')

    # ---- per-kernel HIP-event timing (bench.py roofline leg) ----------------
    def profile_enable(self, max_forwards, coarse=False):
        """HIP-event stamps for the next `max_forwards` forwards.  coarse: 4 stamps per forward (conv0 | the 15 GEMM
        convs as ONE span | tail) instead of 18 -- every stamp between two kernels idles the GPU for ~5 us."""
        with torch.cuda.device(self.device):
            _lib.check(self._lib.nafp_encoder_profile_coarse(self._h, int(coarse)), 'profile_coarse')     # 0 / 1 (True) / 2: only the GEMM span
            _lib.check(self._lib.nafp_encoder_profile_enable(self._h, int(max_forwards)), 'profile_enable')

    def profile_read(self):
        """-> list of 17-float lists (ms): conv0, 15 implicit-GEMM convs, tail; one per forward
        (coarse stamps: conv0, the span of all 15 GEMM convs, zeros, tail)."""
        out = []
        buf = (ctypes.c_float * 17)()
        for s in range(self._lib.nafp_encoder_profile_count(self._h)):
            _lib.check(self._lib.nafp_encoder_profile_read(self._h, s, buf), 'profile_read')
            out.append([float(v) for v in buf])
        return out

    def front_conv(self, feat):
        """(B,F,T,1) -> (B, F'*T'*C) (nnfp.py:210-218)."""
        return self._forward(feat, True, False)[0]

    def div_enc(self, x):
        """(B, D) -> (B, Q), no L2 (nnfp.py:141-156)."""
        x = self._prep(x, (self.flat_dim,))
        self._sync()
        out = torch.empty((x.shape[0], self.emb_sz), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(self._lib.nafp_encoder_div_enc(self._h, _lib.ptr(x), x.shape[0], _lib.ptr(out), 0,
                                                      _lib.current_stream()), 'encoder_div_enc')
            self._mark_use()
        return out

    def l2_normalize(self, x):
        """tf.math.l2_normalize(x, axis=1) (nnfp.py:229; trainer.py:74, 76)."""
        x = _lib.require_cuda(torch.as_tensor(x), 'x').float().contiguous()
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(self._lib.nafp_l2_normalize_rows(_lib.ptr(x), x.shape[0], x.shape[1], _lib.ptr(out),
                                                        _lib.current_stream()), 'l2_normalize_rows')
        return out

    def __call__(self, inputs):
        """nnfp.py:223-231."""
        return self._forward(inputs, False, True)[1]

    call = __call__


def get_fingerprinter(cfg, trainable=False):
    """nnfp.py:234-258 (input shape is hard-coded to (256,32,1) there, :248)."""
    m = FingerPrinter(input_shape=(256, 32, 1), emb_sz=cfg['MODEL']['EMB_SZ'], fc_unit_dim=[32, 1],
                      norm=cfg['MODEL']['BN'])
    m.trainable = trainable
    return m

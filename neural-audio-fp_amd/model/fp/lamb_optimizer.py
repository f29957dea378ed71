"""Host mirror of the reference's optimizers, backed by libnafp's multi-tensor HIP kernels.

Mirrors what the reference's train step uses (model/trainer.py:119-140):
  * `LAMB(learning_rate=...)` of model/fp/lamb_optimizer.py:20-158 (defaults beta 0.9/0.999,
    epsilon 1e-6, weight_decay_rate 1e-6, decay + layer adaptation on every variable);
  * `Adam(learning_rate=...)` = tf.keras.optimizers.Adam (epsilon 1e-7);
  * `CosineDecay(initial_learning_rate, decay_steps, alpha)`, `CosineDecayRestarts(initial_learning_rate,
    first_decay_steps, t_mul, m_mul, alpha)`.
`apply_gradients(zip(grads, variables))` updates the torch CUDA tensors in place; `iterations`
counts steps like keras.  `var_lens` maps a tensor to the length of one keras variable inside
it: the fingerprinter stacks its 128x4 divide-and-encode variables into 4 tensors, and LAMB's
trust ratio is per keras variable (576 of them), so `FingerPrinter.variable_lengths()` supplies
numel/128 for those.  No CPU path.
"""
import ctypes

import torch

from ... import _lib


class CosineDecay:
    def __init__(self, initial_learning_rate, decay_steps, alpha=0.0):
        self.initial_learning_rate, self.decay_steps, self.alpha = float(initial_learning_rate), int(decay_steps), float(alpha)
        self._lib = _lib.load()

    def __call__(self, step):
        return float(self._lib.nafp_cosine_decay_lr_host(self.initial_learning_rate, int(step), self.decay_steps,
                                                         self.alpha))


class CosineDecayRestarts:
    """tf.keras.experimental.CosineDecayRestarts (trainer.py:125-131)."""

    def __init__(self, initial_learning_rate, first_decay_steps, t_mul=2.0, m_mul=1.0, alpha=0.0):
        self.initial_learning_rate, self.first_decay_steps = float(initial_learning_rate), int(first_decay_steps)
        self.t_mul, self.m_mul, self.alpha = float(t_mul), float(m_mul), float(alpha)
        self._lib = _lib.load()

    def __call__(self, step):
        return float(self._lib.nafp_cosine_decay_restarts_lr_host(self.initial_learning_rate, int(step), self.first_decay_steps,
                                                                  self.t_mul, self.m_mul, self.alpha))


class _MultiTensorOptimizer:
    def __init__(self, learning_rate):
        self.learning_rate = learning_rate
        self.iterations = 0
        self._slots = {}
        self._lib = _lib.load()

    def _lr(self):
        lr = self.learning_rate
        return lr(self.iterations) if callable(lr) else float(lr)   # keras evaluates the schedule at `iterations`

    def state_dict(self, variables):
        """Slots in the order of `variables` (what tf.train.Checkpoint(optimizer=...) keeps:
        experiment_helper.py:100-107)."""
        zeros = lambda w: (torch.zeros_like(w), torch.zeros_like(w))
        mv = [self._slots.get(w.data_ptr()) or zeros(w) for w in variables]
        return {'iterations': int(self.iterations), 'm': [m.detach().cpu() for m, _ in mv],
                'v': [v.detach().cpu() for _, v in mv]}

    def load_state_dict(self, sd, variables):
        self.iterations = int(sd['iterations'])
        self._cache = None
        for w, m, v in zip(variables, sd['m'], sd['v']):
            self._slots[w.data_ptr()] = (m.to(w.device, torch.float32).contiguous().clone(),
                                         v.to(w.device, torch.float32).contiguous().clone())

    def _table(self, grads_and_vars, var_lens):
        gv = [(g, w) for g, w in grads_and_vars]
        key_all = tuple((g.data_ptr(), w.data_ptr(), g.dtype, g.is_contiguous()) for g, w in gv)
        cache = getattr(self, '_cache', None)
        if cache is not None and cache[0] == key_all and all(k[2] == torch.float32 and k[3] for k in key_all):
            return cache[1], cache[2], cache[3]          # same buffers as last step: reuse the table
        out = self._build_table(gv, var_lens)
        self._cache = (key_all,) + out
        return out

    def _build_table(self, gv, var_lens):
        arr = (_lib.OptTensor * len(gv))()
        keep = []
        for i, (g, w) in enumerate(gv):
            _lib.require_cuda(w, 'variable'); _lib.require_cuda(g, 'gradient')
            if not w.is_contiguous() or w.dtype != torch.float32:
                raise ValueError('variables must be contiguous float32 CUDA tensors')
            g = g.float().contiguous()
            key = w.data_ptr()
            if key not in self._slots:
                self._slots[key] = (torch.zeros_like(w), torch.zeros_like(w))
            m, v = self._slots[key]
            vl = w.numel() if var_lens is None else int(var_lens[i])
            arr[i] = _lib.OptTensor(w.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), w.numel(), vl)
            keep.append(g)
        return arr, keep, gv[0][1].device


class Adam(_MultiTensorOptimizer):
    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        super().__init__(learning_rate)
        self.beta_1, self.beta_2, self.epsilon = beta_1, beta_2, epsilon

    def apply_gradients(self, grads_and_vars, var_lens=None):
        arr, keep, dev = self._table(grads_and_vars, var_lens)
        lr = self._lr()
        with torch.cuda.device(dev):
            _lib.check(self._lib.nafp_adam_step(arr, len(arr), lr, self.beta_1, self.beta_2, self.epsilon,
                                                self.iterations + 1, _lib.current_stream()), 'adam_step')
        self.iterations += 1


class LAMB(_MultiTensorOptimizer):
    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-6, weight_decay_rate=1e-6,
                 exclude_from_weight_decay=None, exclude_from_layer_adaptation=None, name='LAMB'):
        if exclude_from_weight_decay or exclude_from_layer_adaptation:
            raise NotImplementedError('exclude lists (the reference never sets them: trainer.py:136)')
        super().__init__(learning_rate)
        self.beta_1, self.beta_2, self.epsilon, self.weight_decay_rate = beta_1, beta_2, epsilon, weight_decay_rate
        self._ws = None

    def apply_gradients(self, grads_and_vars, var_lens=None):
        arr, keep, dev = self._table(grads_and_vars, var_lens)
        need = int(self._lib.nafp_lamb_workspace_bytes(arr, len(arr)))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        lr = self._lr()
        with torch.cuda.device(dev):
            _lib.check(self._lib.nafp_lamb_step(arr, len(arr), lr, self.beta_1, self.beta_2, self.epsilon,
                                                self.weight_decay_rate, self.iterations + 1, _lib.ptr(self._ws),
                                                need, _lib.current_stream()), 'lamb_step')
        self.iterations += 1

"""Host mirror of the reference's NT-Xent loss, backed by libnafp's fused HIP kernels.

Mirrors `NTxentLoss` of the reference (model/fp/NTxent_loss_single_gpu.py:29-82):
`NTxentLoss(n_org, n_rep, tau).compute_loss(emb_org, emb_rep) -> (loss, sim_mtx
(N, 2N-1), labels (N, 2N-1))`.  Like the reference it requires n_org == n_rep
(the reference's drop_diag/labels only line up in that case).  Tensors are torch
CUDA tensors; no CPU path.

The loss is differentiable: when an input requires grad the forward call also runs
the backward kernels and `loss.backward()` hands out d(loss)/d(emb) (the reference
gets this from tf.GradientTape, trainer.py:43-48).

`sharded_loss` is the multi-GPU form (model/fp/NTxent_loss_tpu.py:90-137): local rows
against the all-gathered columns; the gradient w.r.t. the gathered arrays is summed
over ranks (what TF's all_reduce-in-the-forward yields in its backward).
"""
import torch

from ... import _lib


def _ntxent_call(lib, org_l, rep_l, org_all, rep_all, rank_offset, tau, want_sim, want_grad):
    n_l, d = org_l.shape
    n_g = org_all.shape[0]
    dev = org_l.device
    loss_sum = torch.empty((1,), dtype=torch.float32, device=dev)
    sim = torch.empty((n_l, 2 * n_g - 1), dtype=torch.float32, device=dev) if want_sim else None
    d_org = torch.empty((n_g, d), dtype=torch.float32, device=dev) if want_grad else None
    d_rep = torch.empty((n_g, d), dtype=torch.float32, device=dev) if want_grad else None
    need = int(lib.nafp_ntxent_workspace_bytes(n_l, n_g))
    ws = torch.empty((need,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.nafp_ntxent_forward(
            _lib.ptr(org_l), _lib.ptr(rep_l), _lib.ptr(org_all), _lib.ptr(rep_all),
            n_l, n_g, int(rank_offset), d, float(tau), _lib.ptr(loss_sum), _lib.ptr(sim),
            _lib.ptr(d_org), _lib.ptr(d_rep), _lib.ptr(ws), need, _lib.current_stream()), 'ntxent_forward')
    return loss_sum, sim, d_org, d_rep


class _NTxentFn(torch.autograd.Function):
    """loss = loss_sum / n_global; grads are those of exactly that scalar."""

    @staticmethod
    def forward(ctx, org_all, rep_all, n_local, rank_offset, tau, lib):
        lo = org_all[rank_offset:rank_offset + n_local].contiguous()
        lr = rep_all[rank_offset:rank_offset + n_local].contiguous()
        loss_sum, _, d_org, d_rep = _ntxent_call(lib, lo, lr, org_all, rep_all, rank_offset, tau, False, True)
        ctx.save_for_backward(d_org, d_rep)
        return loss_sum[0] / org_all.shape[0]

    @staticmethod
    def backward(ctx, g):
        d_org, d_rep = ctx.saved_tensors
        return g * d_org, g * d_rep, None, None, None, None


class NTxentLoss:
    def __init__(self, n_org=int(), n_rep=int(), tau=0.05, **kwargs):
        self.n_org, self.n_rep, self.tau = n_org, n_rep, tau
        self._labels = None
        self._lib = _lib.load()

    @property
    def labels(self):
        """tf.one_hot(tf.range(n_org), n_org*2-1) (NTxent_loss_single_gpu.py:42)."""
        if self._labels is None:
            lab = torch.zeros((self.n_org, 2 * self.n_org - 1), dtype=torch.float32, device='cuda')
            idx = torch.arange(self.n_org, device='cuda')
            lab[idx, idx] = 1.0
            self._labels = lab
        return self._labels

    def _check(self, emb_org, emb_rep):
        emb_org = _lib.require_cuda(torch.as_tensor(emb_org), 'emb_org')
        emb_rep = _lib.require_cuda(torch.as_tensor(emb_rep), 'emb_rep')
        n, d = emb_org.shape
        if emb_rep.shape != (n, d) or n != self.n_org or self.n_org != self.n_rep:
            raise ValueError(f'expected emb_org, emb_rep of shape ({self.n_org},{d}); '
                             f'got {tuple(emb_org.shape)}, {tuple(emb_rep.shape)}')
        return emb_org, emb_rep

    def compute_loss(self, emb_org, emb_rep, return_sim=True):
        emb_org, emb_rep = self._check(emb_org, emb_rep)
        a, b = emb_org.float().contiguous(), emb_rep.float().contiguous()
        n = a.shape[0]
        sim = None
        if return_sim:
            _, sim, _, _ = _ntxent_call(self._lib, a.detach(), b.detach(), a.detach(), b.detach(), 0, self.tau,
                                        True, False)
        if emb_org.requires_grad or emb_rep.requires_grad:
            loss = _NTxentFn.apply(a, b, n, 0, self.tau, self._lib)
        else:
            loss_sum, _, _, _ = _ntxent_call(self._lib, a, b, a, b, 0, self.tau, False, False)
            loss = loss_sum[0] / n          # mean-CE(a) + mean-CE(b): NTxent_loss_single_gpu.py:78-82
        return loss, sim, self.labels

    def loss_and_grad(self, emb_org, emb_rep):
        """(loss, d loss/d emb_org, d loss/d emb_rep) in one pass (no autograd graph)."""
        emb_org, emb_rep = self._check(emb_org, emb_rep)
        a, b = emb_org.detach().float().contiguous(), emb_rep.detach().float().contiguous()
        loss_sum, _, d_org, d_rep = _ntxent_call(self._lib, a, b, a, b, 0, self.tau, False, True)
        return loss_sum[0] / a.shape[0], d_org, d_rep


def sharded_loss(emb_org_local, emb_rep_local, tau=0.05, group=None):
    """One rank of the data-parallel NT-Xent (NTxent_loss_tpu.py:90-137): all-gather the
    L2-normalised embeddings over RCCL, score local rows against all columns, return this
    rank's share `loss_sum_local / n_global` (sum over ranks == the single-device loss).
    Differentiable: the backward all-reduces the gradient w.r.t. the gathered arrays and
    returns this rank's slice."""
    import torch.distributed as dist
    lib = _lib.load()
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_l = emb_org_local.shape[0]

    class _Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, lo, lr):
            ga = [torch.empty_like(lo) for _ in range(world)]
            gb = [torch.empty_like(lr) for _ in range(world)]
            dist.all_gather(ga, lo.contiguous(), group=group)
            dist.all_gather(gb, lr.contiguous(), group=group)
            org_all, rep_all = torch.cat(ga).contiguous(), torch.cat(gb).contiguous()
            loss_sum, _, d_org, d_rep = _ntxent_call(lib, lo.contiguous(), lr.contiguous(), org_all, rep_all,
                                                     rank * n_l, tau, False, True)
            ctx.save_for_backward(d_org, d_rep)
            return loss_sum[0] / org_all.shape[0]

        @staticmethod
        def backward(ctx, g):
            d_org, d_rep = ctx.saved_tensors
            d_org, d_rep = d_org.clone(), d_rep.clone()
            dist.all_reduce(d_org, group=group)
            dist.all_reduce(d_rep, group=group)
            sl = slice(rank * n_l, (rank + 1) * n_l)
            return g * d_org[sl], g * d_rep[sl]

    return _Fn.apply(emb_org_local.float(), emb_rep_local.float())

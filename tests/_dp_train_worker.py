"""Worker of tests/test_gpu_train_dp.py: one rank of a 2- or 4-process data-parallel train step.
Both ranks share cuda:0 and talk over gloo (the GPU box has one GPU; on a node the same code
runs one rank per GPU over RCCL).  With BACKEND = nccl and one process the same step runs its collectives on RCCL
itself (a group of one rank).  Usage: python -m torch.distributed.run ... _dp_train_worker.py OUT_DIR [N] [BACKEND]"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))


def features(rank, n):
    rng = np.random.default_rng(50 + rank)
    fa = -rng.uniform(0, 1.2, size=(n, 256, 32, 1))
    fp = fa + 0.05 * rng.normal(size=fa.shape)
    return fa.astype(np.float32), fp.astype(np.float32)


class Identity:
    bypass = False

    def __call__(self, x):
        return x


def main(out_dir, n=4, backend='gloo'):
    n = int(n)
    torch.cuda.set_device(0)
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
    else:
        dist.init_process_group(backend)
    rank = dist.get_rank()
    import neural_audio_fp_amd as nafp
    from neural_audio_fp_amd.model import trainer as T
    from neural_audio_fp_amd.model.fp.lamb_optimizer import LAMB
    import _inputs
    m_fp = nafp.FingerPrinter(seed=0)
    m_fp.set_weights(_inputs.weight_list(_inputs.weights(seed=31)))
    bucket = T.GradientBucket(m_fp)
    opt = LAMB(learning_rate=1e-3)
    fa, fp = features(rank, n)
    X = (torch.from_numpy(fa).cuda(), torch.from_numpy(fp).cuda())
    loss_obj = nafp.NTxentLoss(n_org=n, n_rep=n, tau=0.05)
    losses = []
    for step in range(2):
        loss, _ = T.train_step(X, Identity(), Identity(), m_fp, loss_obj, opt, bucket)
        losses.append(float(loss))
        if step == 0:
            g0 = bucket.flat.detach().cpu().clone()
    torch.save({'losses': losses, 'grad0': g0, 'params': [v.detach().cpu() for v in m_fp.trainable_variables],
                'backend': dist.get_backend(), 'world': dist.get_world_size()},
               os.path.join(out_dir, f'rank{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main(*sys.argv[1:4])

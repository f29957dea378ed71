"""How far ahead of the GPU does the host run in a train step?  Per step: host time inside `train_step` (enqueue only, no
synchronisation) for the plain step and for the same step through a 1-rank RCCL group; then the whole-loop wall time.
A host time close to the GPU step time means something in the step blocks the host until the GPU has caught up.
usage: python tools/host_ahead_probe.py [BSZ=640] [dist=0|1] [steps=30]"""
import os
import socket
import sys
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import neural_audio_fp_amd as nafp  # noqa: E402,F401
from neural_audio_fp_amd.model import trainer as T  # noqa: E402

bsz = int(sys.argv[1]) if len(sys.argv) > 1 else 640
use_dist = len(sys.argv) > 2 and sys.argv[2] == '1'
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
sys.stdout.flush(); fd = os.dup(1); os.dup2(2, 1)
torch.cuda.set_device(0)
if use_dist:
    import torch.distributed as dist
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=torch.device('cuda', 0))
cfg = yaml.safe_load(open(os.path.join(ROOT, 'config', 'default.yaml')))
cfg['BSZ']['TR_BATCH_SZ'], cfg['BSZ']['TR_N_ANCHOR'] = bsz, bsz // 2
cfg['TRAIN']['OPTIMIZER'], cfg['TRAIN']['LR'] = 'LAMB', 1e-4
m_pre, m_specaug, m_fp, opt, loss_obj, bucket = T.setup(cfg, 1000)
X = next(iter(T.synthetic_batches(cfg, 1)(1)))
for _ in range(4):
    T.train_step(X, m_pre, m_specaug, m_fp, loss_obj, opt, bucket)
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(steps):
    a = time.perf_counter()
    T.train_step(X, m_pre, m_specaug, m_fp, loss_obj, opt, bucket)
    host.append(time.perf_counter() - a)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
host.sort()
os.write(fd, (f'dist={int(use_dist)} BSZ {bsz}: wall {wall / steps * 1e3:.3f} ms/step; host inside train_step: median {host[len(host) // 2] * 1e3:.3f} ms, '
              f'min {host[0] * 1e3:.3f}, max {host[-1] * 1e3:.3f}; all {steps} steps enqueued after {t_enq * 1e3:.1f} ms of {wall * 1e3:.1f}\n').encode())
if use_dist:
    dist.destroy_process_group()

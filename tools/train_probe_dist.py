"""tools/train_probe.py's whole-step timing through a process group of ONE rank on RCCL (the `train_rank640` object of bench.py):
every collective of `train_step` executes.  usage: python tools/train_probe_dist.py [BSZ] [adam|lamb] [steps]"""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import neural_audio_fp_amd as nafp  # noqa: E402,F401
from neural_audio_fp_amd.model import trainer as T  # noqa: E402

bsz = int(sys.argv[1]) if len(sys.argv) > 1 else 640
which = sys.argv[2] if len(sys.argv) > 2 else 'lamb'
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
sys.stdout.flush(); fd = os.dup(1); os.dup2(2, 1)                    # RCCL's banner goes to stderr
with socket.socket() as sk:
    sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]
torch.cuda.set_device(0)
dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=torch.device('cuda', 0))
cfg = yaml.safe_load(open(os.path.join(ROOT, 'config', 'default.yaml')))
cfg['BSZ']['TR_BATCH_SZ'], cfg['BSZ']['TR_N_ANCHOR'] = bsz, bsz // 2
cfg['TRAIN']['OPTIMIZER'], cfg['TRAIN']['LR'] = ('LAMB' if which == 'lamb' else 'Adam'), 1e-4
m_pre, m_specaug, m_fp, opt, loss_obj, bucket = T.setup(cfg, 1000)
X = next(iter(T.synthetic_batches(cfg, 1)(1)))
for _ in range(4):
    T.train_step(X, m_pre, m_specaug, m_fp, loss_obj, opt, bucket)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    loss, _ = T.train_step(X, m_pre, m_specaug, m_fp, loss_obj, opt, bucket)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
os.write(fd, f'train_step wall {dt * 1e3:.3f} ms/step through a 1-rank RCCL group (BSZ {bsz}, {which}); loss {float(loss):.4f}\n'.encode())
dist.destroy_process_group()

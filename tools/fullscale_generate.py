"""SURVEY.md 8(d) config 5, scaled by argument: every rank generates its contiguous slice of ONE
`dummy_db.mm` from on-the-fly seeded audio (no 443 GB dataset exists here), rank 0 writes
`dummy_db_shape.npy`, and the result is opened exactly like eval/eval_faiss.py:47-59 does and searched
with the exact index.  Full scale = 100 M rows over 8 ranks (12.5 M rows, 6.4 GB per rank).

    python tools/fullscale_generate.py [total_rows=2000000] [out_dir=/tmp/nafp_full]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/fullscale_generate.py 100000000 OUT
"""
import os
import sys
import time

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

total = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
out_dir = (sys.argv[2] if len(sys.argv) > 2 else '/tmp/nafp_full').rstrip('/') + '/'
world, rank, local = int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0))
torch.cuda.set_device(local)
dist = None
if world > 1:
    import torch.distributed as dist
    dist.init_process_group('nccl', device_id=torch.device('cuda', local))
import neural_audio_fp_amd as nafp  # noqa: E402
from neural_audio_fp_amd.model.generate import shard_rows  # noqa: E402
from neural_audio_fp_amd.eval.eval_faiss import FlatL2Index, load_memmap_data  # noqa: E402

cfg = yaml.safe_load(open(os.path.join(ROOT, 'config', 'default.yaml')))
GROUP, LAUNCH = cfg['BSZ']['TS_BATCH_SZ'], 5 * cfg['BSZ']['TS_BATCH_SZ']          # 625 rows per launch = 5 max-norm groups
m_pre, m_fp = nafp.get_melspec_layer(cfg), nafp.FingerPrinter(seed=0)
os.makedirs(out_dir, exist_ok=True)
path = out_dir + 'dummy_db.mm'
if rank == 0:
    arr = np.memmap(path, dtype='float32', mode='w+', shape=(total, 128))
    np.save(out_dir + 'dummy_db_shape.npy', (total, 128))
if dist:
    dist.barrier()
if rank != 0:
    arr = np.memmap(path, dtype='float32', mode='r+', shape=(total, 128))
r0, r1 = shard_rows(total, GROUP, rank, world)
t = torch.arange(8000, device='cuda', dtype=torch.float32) / 8000.0


def audio(row0, n):
    """segment `row` = seeded noise + one tone whose frequency depends on the row (deterministic per launch)."""
    g = torch.Generator(device='cuda').manual_seed(row0)
    f = 300.0 + (torch.arange(row0, row0 + n, device='cuda') % 3500).float()
    return 0.1 * torch.randn((n, 1, 8000), generator=g, device='cuda') + 0.2 * torch.sin(2 * torch.pi * f[:, None, None] * t)


streams = [torch.cuda.Stream() for _ in range(4)]
host = [torch.empty((LAUNCH, 128), dtype=torch.float32).pin_memory() for _ in range(4)]
pend = [None] * 4
torch.cuda.synchronize(); t0 = time.perf_counter()
for k, a in enumerate(range(r0, r1, LAUNCH)):
    s = k % 4
    if pend[s] is not None:
        ev, pa, pn = pend[s]; ev.synchronize(); arr[pa:pa + pn] = host[s][:pn].numpy()
    n = min(LAUNCH, r1 - a)
    with torch.cuda.stream(streams[s]):
        emb = m_fp(m_pre(audio(a, n), group_size=GROUP))
        host[s][:n].copy_(emb, non_blocking=True)
        ev = torch.cuda.Event(); ev.record()
    pend[s] = (ev, a, n)
for s in range(4):
    if pend[s] is not None:
        ev, pa, pn = pend[s]; ev.synchronize(); arr[pa:pa + pn] = host[s][:pn].numpy()
arr.flush()
dt = time.perf_counter() - t0
if dist:
    tt = torch.tensor([dt], device='cuda', dtype=torch.float64); dist.all_reduce(tt, op=dist.ReduceOp.MAX); dt = float(tt[0]); dist.barrier()
if rank == 0:
    print(f'{total} rows ({total * 512 / 1e9:.2f} GB) over {world} rank(s) in {dt:.1f} s = {total / dt:.0f} segments/s incl. audio synthesis, D2H and memmap writes')
    data, shape = load_memmap_data(out_dir, 'dummy_db')
    assert tuple(shape) == (total, 128) and np.isfinite(data[:1000]).all() and abs(float(np.linalg.norm(data[total - 1])) - 1) < 1e-4
    # consumer check: exact search of one regenerated launch (625 rows) against the first <= 2 M rows finds each row at its own id
    n_idx = min(total, 2_000_000)
    idx = FlatL2Index(128, capacity=n_idx); idx.add(data[:n_idx])
    probe = LAUNCH                                              # the second launch of rank 0, regenerated from its seed
    q = m_fp(m_pre(audio(probe, LAUNCH), group_size=GROUP))
    D, I = idx.search_device(q, 1)
    assert (I[:, 0].cpu() == torch.arange(probe, probe + LAUNCH)).all() and float(D.max()) < 1e-5
    print(f'opened like eval_faiss.load_memmap_data; exact search over the first {n_idx} rows returns every probed row at its own id')
if dist:
    dist.destroy_process_group()

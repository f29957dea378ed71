"""Worker of tests/test_gpu_configs.py::test_config4_*: one rank of the scaled full-scale generate (BASELINE.json
configs[4]).  Both ranks share cuda:0 and rendezvous over gloo (the GPU box has one GPU; on a node the same code runs
one rank per GPU).  Rows come from seeded on-device synthesis: no dataset of that size exists on the box.
Usage: python -m torch.distributed.run ... _fullscale_worker.py OUT_DIR N_ROWS"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

WEIGHT_SEED = 11
GROUP = 125                     # TS_BATCH_SZ of config/default.yaml


def synth_rows(row0, n, device):
    """rows [row0, row0 + n): seeded noise + one tone whose frequency depends on the row; a function of the row's
    GROUP only (seed = first row of its group), so any launch composition regenerates the same audio."""
    t = torch.arange(8000, device=device, dtype=torch.float32) / 8000.0
    out = torch.empty((n, 1, 8000), dtype=torch.float32, device=device)
    g = torch.Generator(device=device)
    for a in range(row0 - row0 % GROUP, row0 + n, GROUP):
        g.manual_seed(a)
        blk = 0.1 * torch.randn((GROUP, 1, 8000), generator=g, device=device)
        f = 300.0 + (torch.arange(a, a + GROUP, device=device) % 3500).float()
        blk = blk + 0.2 * torch.sin(2 * torch.pi * f[:, None, None] * t)
        lo, hi = max(a, row0), min(a + GROUP, row0 + n)
        out[lo - row0:hi - row0] = blk[lo - a:hi - a]
    return out


def main(out_dir, n_rows):
    import torch.distributed as dist
    import yaml
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    import neural_audio_fp_amd as nafp
    from neural_audio_fp_amd.model import generate as G
    cfg = yaml.safe_load(open(os.path.join(ROOT, 'config', 'default.yaml')))
    assert cfg['BSZ']['TS_BATCH_SZ'] == GROUP
    m_pre, m_fp = nafp.get_melspec_layer(cfg), nafp.FingerPrinter(seed=WEIGHT_SEED)
    out_dir = out_dir.rstrip('/') + '/'
    path = out_dir + 'dummy_db.mm'
    if rank == 0:                                                   # generate.py:157-161
        arr = np.memmap(path, dtype='float32', mode='w+', shape=(n_rows, 128))
        np.save(out_dir + 'dummy_db_shape.npy', (n_rows, 128))
    dist.barrier()
    if rank != 0:
        arr = np.memmap(path, dtype='float32', mode='r+', shape=(n_rows, 128))
    r0, r1 = G.write_fingerprints_from_device_rows(lambda a, n: synth_rows(a, n, 'cuda'), n_rows, m_pre, m_fp, arr, GROUP,
                                                   rank, world)
    arr.flush()
    ranges = [None] * world
    dist.all_gather_object(ranges, (int(r0), int(r1)))
    dist.barrier()
    if rank == 0:
        json.dump({str(r): ranges[r] for r in range(world)}, open(out_dir + 'ranks.json', 'w'))
    dist.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]))

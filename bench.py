#!/usr/bin/env python
"""Benchmark of the hot path: fingerprint generation, 1-s segments per second.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one pass of  audio -> log-mel -> encoder -> L2-normalised fingerprint
over one batch of 640 synthetic 1-s segments that are already resident in HBM
(BASELINE.json configs[1]: "Fingerprint generate, d=128 encoder, BSZ=640, 1xMI355X").
With N > 1 every rank (one per GPU, torch.distributed over RCCL) owns its own batches:
generation shards by segment with no data-path collective, so scaling is weak and `value`
is the whole-job segments/s.  `python bench.py --gpus N` with no WORLD_SIZE in the
environment starts the N ranks ITSELF (`python -m torch.distributed.run --nproc-per-node N
bench.py ...` as a child process, before anything in this process touches the GPU), relays
rank 0's JSON line and exits with the child's return code; under torch.distributed.run
(WORLD_SIZE set) it is one of the ranks.

The timed region (EXACTLY --steps steps between barrier + synchronize) is repeated
--repeats times; `ms_per_step` / `value` are the MEDIAN region, `spread` holds all of them.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     the dominant kernel (the fp32-MFMA implicit-GEMM convs): algorithmic
               FLOPs per launch / mean launch duration from HIP events recorded on the
               launch stream inside the timed region, vs the 157.3 TFLOP/s fp32 matrix peak
  cpu_baseline the oracle's torch-CPU restatement of the same graph timed on ALL host
               cores of this box on a bounded sample (kind "port": TensorFlow is absent)
  e2e_generate disk -> .mm through `write_fingerprints` on the 100-clip config-1 set and a
               600-clip set (second figure), with the ingest / launch split
  train, train_1280   contrastive train steps/s at global BSZ 5120 / LAMB (configs[3]) and
               BSZ 1280 / Adam (configs[2])
  train_rank640       the 8-GPU operating point of configs[3] on ONE GPU: per-rank batch 640, LAMB, through a
               1-rank RCCL group (every collective executes, none crosses xGMI): a one-GPU compute bound, not scaling
  fullscale_generate  configs[4] through the product's writer on seeded on-device audio, --fullscale-rows rows
               (default 1.25 M = a tenth of one rank's share; the full 12.5 M-row share is a pytest -m gpu test)
stdout carries the JSON line only: fd 1 is pointed at stderr for the run (RCCL prints a banner there), the line is
written to the saved descriptor.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BSZ = 640                      # BASELINE.json configs[1]
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense (never the 2:1-sparsity figure)


def conv_effective_macs(input_shape=(256, 32, 1)):
    """Per-conv multiply-accumulates per segment, zero-padding taps excluded
    (SURVEY.md appendix A).  Host arithmetic on the geometry only."""
    hidden = [128, 128, 256, 256, 512, 512, 1024, 1024]
    stride_t = [2, 2, 2, 2, 1, 2, 1, 2]
    F, T, C = input_shape
    out = []

    def same(n, k, s):
        n_out = -(-n // s)
        tot = max((n_out - 1) * s + k - n, 0)
        return n_out, tot // 2

    def live(n_in, n_out, s, pad):
        return sum(1 for o in range(n_out) for t in range(3) if 0 <= o * s - pad + t < n_in)

    for i in range(8):
        To, pb = same(T, 3, stride_t[i])
        out.append(F * live(T, To, stride_t[i], pb) * C * hidden[i])
        T, C = To, hidden[i]
        Fo, pb = same(F, 3, 2)
        out.append(T * live(F, Fo, 2, pb) * C * hidden[i])
        F = Fo
    return out


def make_audio(n, seed, torch):
    """white noise sigma=0.1 + a 440 Hz tone."""
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(8000, dtype=torch.float32) / 8000.0
    x = 0.1 * torch.randn((n, 1, 8000), generator=g) + 0.2 * torch.sin(2 * torch.pi * 440.0 * t)
    return x.float()


def _cpu_worker(args):
    """One process of the CPU baseline: `threads` intra-op threads, batches of 125 for ~target_s seconds."""
    threads, target_s, max_batches, seed = args
    import torch
    from oracle import nnfp as o_nnfp, torch_ref
    torch.set_num_threads(threads)
    tf = torch_ref.TorchFingerprinter(o_nnfp.init_weights(seed=0))
    x = make_audio(125, seed, torch)
    with torch.no_grad():
        tf(torch_ref.melspec_layer(x))                  # warm-up (oneDNN primitive caches)
        n, t0 = 0, time.perf_counter()
        while True:
            tf(torch_ref.melspec_layer(x))
            n += 125
            el = time.perf_counter() - t0
            if el >= target_s or n >= 125 * max_batches:
                break
    return n, el


def cpu_baseline(target_s=10.0, max_batches=40):
    """The oracle's torch-CPU restatement of melspec + encoder on the HOST CORES OF THIS BOX (kind "port"): first the
    fastest intra-op thread count of one process is found (all cores is not the fastest at batch 125), then as many
    such processes as fit the box run side by side, each on its own batches; the value is their summed throughput."""
    import multiprocessing as mp
    import torch
    from oracle import nnfp as o_nnfp, torch_ref
    cores = os.cpu_count() or 1
    w = o_nnfp.init_weights(seed=0)
    tf = torch_ref.TorchFingerprinter(w)
    x = make_audio(125, 7, torch)                       # TS_BATCH_SZ of config/default.yaml
    with torch.no_grad():
        best_t, best_n, sweep = None, 1, {}
        for nt in sorted({min(cores, c) for c in (8, 16, 32, 64)}):
            torch.set_num_threads(nt)
            tf(torch_ref.melspec_layer(x))
            t0 = time.perf_counter()
            tf(torch_ref.melspec_layer(x))
            dt = time.perf_counter() - t0
            sweep[str(nt)] = round(dt, 3)
            # efficiency per core decides: the processes below fill the box
            if best_t is None or dt * nt < best_t * best_n:
                best_t, best_n = dt, nt
            if dt > 8.0:
                break
    # Every logical core gets a thread (procs x threads == host cores), and -- because on an SMT / bandwidth-bound host the
    # half-filled box can be the faster one -- the half-filled arrangement is timed as well: `value` is the better of the
    # two, `process_sweep` shows both.
    ctx = mp.get_context('spawn')
    full = max(1, cores // best_n)
    runs = {}
    for procs in sorted({max(1, full // 2), full}):
        threads = best_n if procs < full else max(best_n, cores // procs)
        with ctx.Pool(procs) as pool:
            res = pool.map(_cpu_worker, [(threads, target_s, max_batches, 7 + k) for k in range(procs)])
        runs[procs] = {'processes': procs, 'threads_per_process': threads, 'cores': procs * threads,
                       'value': round(sum(r[0] / r[1] for r in res), 2), 'segments': sum(r[0] for r in res),
                       'seconds': round(max(r[1] for r in res), 1)}
    best = max(runs.values(), key=lambda r: r['value'])
    return {'value': best['value'], 'unit': 'segments/s', 'cores': best['cores'], 'kind': 'port', 'host_cores': cores,
            'processes': best['processes'], 'threads_per_process': best['threads_per_process'],
            'thread_sweep_s_per_batch': sweep, 'process_sweep': list(runs.values()),
            'sample': f"{best['segments']} segments in batches of 125 (TS_BATCH_SZ), torch-CPU fp32 restatement of melspec+encoder "
                      f"(oracle/torch_ref.py), {best['processes']} processes x {best['threads_per_process']} threads side by side "
                      f"for {best['seconds']} s; the host was timed half-filled and filled ({cores} logical cores), the faster is reported"}


def train_region(cfg, world, rank, dist, global_bsz, steps, torch, warmup=2, optimizer='LAMB', repeats=3, arith=None):
    """Second half of BASELINE.json's metric: contrastive-train steps/s at a GLOBAL batch of 5120
    (configs[3]: config/640_lamb.yaml scaled, LAMB, tau 0.05), strong scaling: every rank takes
    5120/N segments (anchors + replicas), all-gathers the embeddings, all-reduces the gradients.
    A step = melspec + spec-augment + forward + NT-Xent + backward + (collectives) + optimizer.
    Also used for configs[2] (BSZ 1280, Adam: `train_1280`).  Like the headline: `repeats` timed regions of exactly
    `steps` steps each (barrier + synchronize on both sides, MAX over ranks), the MEDIAN region is reported, `spread`
    lists all of them."""
    import copy
    from neural_audio_fp_amd.model import trainer as T
    c = copy.deepcopy(cfg)
    c['BSZ']['TR_BATCH_SZ'], c['BSZ']['TR_N_ANCHOR'] = global_bsz, global_bsz // 2
    c['TRAIN']['OPTIMIZER'], c['TRAIN']['LR'] = optimizer, 1e-4
    if (global_bsz // 2) % world:
        return {'skipped': f'global batch {global_bsz} does not split over {world} ranks'}
    # arith = 'x6': forward_train and the transposed convs on the exact 3-way bf16 split (NAFP_OPT_BF16X3 = 2; weight gradients stay f32)
    prev_env = os.environ.get('NAFP_BF16X3')
    if arith == 'x6':
        os.environ['NAFP_BF16X3'] = '2'
    try:
        m_pre, m_specaug, m_fp, opt, loss_obj, bucket = T.setup(c, 1000)
    finally:
        if arith == 'x6':
            if prev_env is None:
                os.environ.pop('NAFP_BF16X3', None)
            else:
                os.environ['NAFP_BF16X3'] = prev_env
    assert getattr(m_fp, 'split_arithmetic', 0) == (2 if arith == 'x6' else (int(prev_env) if prev_env in ('1', '2') else 0))
    batches = list(T.synthetic_batches(c, 2)(1))
    for i in range(warmup):
        T.train_step(batches[i % 2], m_pre, m_specaug, m_fp, loss_obj, opt, bucket)
    regions = []
    for _ in range(max(1, repeats)):
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            loss, _ = T.train_step(batches[i % 2], m_pre, m_specaug, m_fp, loss_obj, opt, bucket)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        el = time.perf_counter() - t0
        if dist:
            t = torch.tensor([el], dtype=torch.float64, device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t[0])
        regions.append(el)
    el = sorted(regions)[len(regions) // 2]
    coll = 'none'
    if dist:
        # one more step, outside the timed region, with device events around every collective (rank 0's view):
        # the embedding all-gather and reduce-scatter on the compute stream, the gradient pieces on the
        # communication stream, where they overlap the backward pass
        timers, bucket.timed = [], True
        T.train_step(batches[0], m_pre, m_specaug, m_fp, loss_obj, opt, bucket, timers=timers)
        torch.cuda.synchronize()
        bucket.timed = False
        pieces = bucket.read_timings() or []
        coll = {'backend': dist.get_backend(), 'world_size': dist.get_world_size(),
                'all_gather(emb)_ms': None, 'reduce_scatter(d emb)_ms': None,
                'all_reduce(grad pieces)_ms': [round(x, 4) for x in pieces],
                'grad_piece_MB': [round(p.numel() * 4 / 1e6, 2) for p in bucket.pieces],
                'overlap': 'gradient pieces run on a communication stream behind per-group events of the backward pass; '
                           'piece 0 (convs 12-15 + divide-and-encode) is ready after the first 4 of 16 layers',
                'bytes': {'all_gather': global_bsz * 128 * 4, 'reduce_scatter_send_per_rank': global_bsz * 128 * 4,
                          'all_reduce': int(bucket.flat.numel() * 4)}}
        for name, a, b in timers:
            coll[name + '_ms'] = round(a.elapsed_time(b), 4)
    exposed = None
    if dist and world > 1:
        # the same per-rank step with NO process group (own models, per-rank batch as its whole batch, an n x n loss instead of
        # the rank's n x N share), timed the same way in this run: step - that = what the collectives leave exposed
        T.NO_DIST = True
        try:
            plain = train_region(cfg, 1, 0, None, global_bsz // world, steps, torch, warmup=warmup, optimizer=optimizer, repeats=repeats, arith=arith)
        finally:
            T.NO_DIST = False
        dist.barrier()
        exposed = {'exposed_comm_ms': round(el / steps * 1e3 - plain['ms_per_step'], 3),
                   'no_process_group_ms_per_step': plain['ms_per_step'], 'no_process_group_spread': plain['spread']['ms_per_step_all'],
                   'note': "rank 0's compute-only step (per-rank batch, no collectives, n x n loss) in the same run; the difference also "
                           "holds the larger n x N loss share of the distributed step"}
    enc = 2.0 * (sum(conv_effective_macs()) + 36864)               # forward FLOPs per segment
    flops = 3.0 * enc * global_bsz + 3.0 * 2.0 * global_bsz * global_bsz * 128   # fwd + dgrad + wgrad, NT-Xent x3
    tf = flops / (el / steps) / 1e12
    return {'metric': 'contrastive train steps/s', 'value': round(steps / el, 4), 'unit': 'steps/s',
            'global_batch': global_bsz, 'per_gpu_batch': global_bsz // world, 'n_gpus': world, 'steps': steps,
            'warmup': warmup, 'ms_per_step': round(el / steps * 1e3, 3), 'repeats': len(regions),
            'spread': {'ms_per_step_all': [round(r / steps * 1e3, 3) for r in regions], 'min': round(min(regions) / steps * 1e3, 3),
                       'max': round(max(regions) / steps * 1e3, 3),
                       'note': f'{len(regions)} timed regions of {steps} steps each; value / ms_per_step = the median region'},
            'scaling': 'strong', 'optimizer': optimizer,
            'segments_per_s': round(global_bsz * steps / el, 1), 'loss': round(float(loss), 4),
            'algorithmic_TFLOP_per_step': round(flops / 1e12, 3), 'achieved_TFLOP/s': round(tf, 2),
            'mfma_frac_of_peak': round(tf / (FP32_MFMA_PEAK_TFLOPS * world), 4),
            'collectives': coll, **(exposed or {}),
            'data': 'synthetic (seeded noise anchors, replicas = anchors + noise at 5 dB SNR), resident in HBM'}


def train_rank640(cfg, torch, steps=20, warmup=4, repeats=3):
    """The operating point of the metric's 8-GPU entry, on ONE GPU: a rank's share of the global batch of 5120 is 640
    segments (320 anchors + 320 replicas), LAMB.  The step runs through a process group of ONE rank on RCCL (legal for
    RCCL), so every collective of `train_step` executes -- all-gather of the embeddings, reduce-scatter of their
    gradient, the 4 gradient pieces on the communication stream -- but each moves data only inside this GPU:
    a ONE-GPU COMPUTE BOUND of the 8-rank step, NOT a scaling measurement (no xGMI traffic, a 640 x 640 loss instead of
    one rank's 640 x 5120 share).  `stage_ms` = the same step's pieces between device events (the split of
    tools/train_probe.py), taken without the process group."""
    import copy
    import socket
    import torch.distributed as dist
    from neural_audio_fp_amd.model import trainer as T
    import neural_audio_fp_amd as nafp
    bsz = 640
    # ---- stage split (no process group) ----
    c = copy.deepcopy(cfg)
    c['BSZ']['TR_BATCH_SZ'], c['BSZ']['TR_N_ANCHOR'] = bsz, bsz // 2
    c['TRAIN']['OPTIMIZER'], c['TRAIN']['LR'] = 'LAMB', 1e-4
    m_pre, m_specaug, m_fp, opt, loss_obj, bucket = T.setup(c, 1000)
    X = next(iter(T.synthetic_batches(c, 1)(1)))
    names = ['melspec+aug', 'forward_train', 'ntxent', 'backward', 'optimizer', 'set_weights(next fwd)']
    tot = [0.0] * len(names)

    def ev():
        e = torch.cuda.Event(enable_timing=True); e.record(); return e
    n_split = 8
    for it in range(n_split + 3):
        e = [ev()]
        feat = m_specaug(m_pre(torch.cat(X, 0))); e.append(ev())
        emb = m_fp.forward_train(feat); e.append(ev())
        _, da, db = loss_obj.loss_and_grad(emb[:bsz // 2], emb[bsz // 2:]); e.append(ev())
        grads = m_fp.backward(torch.cat([da, db])); e.append(ev())
        opt.apply_gradients(zip(grads, m_fp.trainable_variables), var_lens=m_fp.variable_lengths()); m_fp.mark_dirty(); e.append(ev())
        m_fp._sync(); e.append(ev())
        torch.cuda.synchronize()
        if it >= 3:
            for k in range(len(names)):
                tot[k] += e[k].elapsed_time(e[k + 1])
    stage = {n: round(t / n_split, 4) for n, t in zip(names, tot)}
    del m_pre, m_specaug, m_fp, opt, loss_obj, bucket
    # ---- the same step WITHOUT a process group, timed like the one below: the difference is what the six collectives and
    # their stream hand-overs add to the step when none of them moves data off the GPU (`exposed_comm_ms`) ----
    plain = train_region(cfg, 1, 0, None, bsz, steps, torch, warmup=warmup, optimizer='LAMB', repeats=repeats)
    # ---- the step itself, through a 1-rank RCCL group ----
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1,
                            device_id=torch.device('cuda', torch.cuda.current_device()))
    try:
        r = train_region(cfg, 1, 0, dist, bsz, steps, torch, warmup=warmup, optimizer='LAMB', repeats=repeats)
    finally:
        dist.destroy_process_group()
    r['stage_ms'] = stage
    r['no_process_group_ms_per_step'] = plain['ms_per_step']
    r['no_process_group_spread'] = plain['spread']['ms_per_step_all']
    r['exposed_comm_ms'] = round(r['ms_per_step'] - plain['ms_per_step'], 3)
    r['exposed_comm_note'] = ('ms_per_step through the 1-rank RCCL group minus the same step with no process group, both the median of '
                              f'{repeats} regions of {steps} steps in this run: what the all-gather, the reduce-scatter and the 4 gradient pieces cost '
                              'when nothing crosses xGMI (stream hand-overs and launch overhead of the collectives; the wire time of a real '
                              '8-rank step comes on top where the backward pass does not hide it -- DESIGN.md section 6)')
    r['what'] = ('one rank\'s share (640 of 5120 segments) of the 8-GPU train step on ONE GPU through a 1-rank RCCL group: every '
                 'collective of the step executes, none crosses xGMI -- a one-GPU compute bound of the 8-rank step, not a scaling '
                 'measurement')
    r['scaling'] = 'none (single GPU)'
    return r


def e2e_generate(cfg, torch, n_small=100, n_large=600):
    """SURVEY.md 8d config 2, second figure: DISK -> .mm through the product's own `write_fingerprints` (whole-file
    upload into pinned arenas, device-side windows, 4 HIP streams, pinned download, memmap store), on the config-1 set
    (100 clips x 30 s, 16-bit 8 kHz mono WAV: `default_rng(1000+k)` noise + 3 tones, TS_BATCH_SZ 125 -> 5,900 segments)
    and on a 600-clip set (35,400 segments).  The files were written a moment ago, so "disk" is the page cache of this
    box.  Split: `ingest_only` = the host side alone (header scan excluded; file reads into the pinned arenas + window
    plan, no GPU work), `device_only` = the headline single-stream figure of this run."""
    import shutil
    import tempfile
    import wave
    import numpy as np
    from neural_audio_fp_amd.model import generate as g
    from neural_audio_fp_amd.model.utils.audio_utils import SegmentSource
    d = tempfile.mkdtemp(prefix='nafp_e2e_')
    try:
        t = np.arange(240000) / 8000.0
        for k in range(n_large):
            rng = np.random.default_rng(1000 + k)
            x = rng.integers(-8192, 8192, size=240000).astype(np.float64)
            for f in rng.uniform(300, 3900, size=3):
                x += 4000 * np.sin(2 * np.pi * f * t)
            with wave.open(os.path.join(d, f'{k:05d}.wav'), 'w') as w:
                w.setnchannels(1); w.setsampwidth(2); w.setframerate(8000)
                w.writeframes(np.clip(x, -32768, 32767).astype('<i2').tobytes())
        paths = sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith('.wav'))
        m_pre, m_fp = g.build_fp(cfg)
        group = int(cfg['BSZ']['TS_BATCH_SZ'])
        out = {'unit': 'segments/s', 'group_size': group, 'launch_segments': g.LAUNCH_SEGMENTS, 'streams': g.N_STREAMS,
               'source': '16-bit 8 kHz mono WAV, 30 s each, written by this run (page cache), read by riff_scan + whole-file '
                         'upload; output = np.memmap float32 (n,128) + flush'}
        for name, n in (('clips_100', n_small), ('clips_600', n_large)):
            t0 = time.perf_counter()
            src = SegmentSource(paths[:n], bsz=group)
            scan_s = time.perf_counter() - t0
            arr = np.memmap(os.path.join(d, f'{name}.mm'), dtype='float32', mode='w+', shape=(src.n_samples, 128))
            emb = g.StreamedEmbedder(m_pre, m_fp)
            runs = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                g.write_fingerprints(src, emb, arr, group)
                arr.flush()
                runs.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            for _ in src.iter_windows(0, src.n_samples, -(-g.LAUNCH_SEGMENTS // group) * group, alloc=emb.alloc):
                pass
            ingest_s = time.perf_counter() - t0
            best = min(runs)
            assert bool(np.isfinite(arr[-1]).all()) and abs(float(np.linalg.norm(arr[src.n_samples // 2])) - 1.0) < 1e-4
            out[name] = {'segments': int(src.n_samples), 'value': round(src.n_samples / best, 1),
                         'seconds_runs': [round(r, 4) for r in runs], 'header_scan_s': round(scan_s, 4),
                         'ingest_only_s': round(ingest_s, 4), 'ingest_only_segments_per_s': round(src.n_samples / ingest_s, 1)}
            del arr
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def fullscale_generate(cfg, torch, rows):
    """BASELINE.json configs[4] in the bench line, scaled: `rows` (default 1,250,000 = a tenth of ONE rank's
    12,500,000-row share of the 100 M-row / 8-rank job; --fullscale-rows 12500000 runs the whole share) through the
    product's writer `write_fingerprints_from_device_rows` -- seeded on-device audio (no 443 GB dataset exists on the box)
    -> log-mel -> encoder -> pinned D2H on 4 HIP streams -> np.memmap stores -> flush.  The whole share at full size is
    tests/test_gpu_configs.py::test_config4_one_rank_full_share_12_5_million_rows (oracle-checked groups, search self
    hits); its record of a run is kept in profiles/."""
    import shutil
    import tempfile
    import numpy as np
    from neural_audio_fp_amd.model import generate as g
    group = int(cfg['BSZ']['TS_BATCH_SZ'])
    rows = rows // group * group
    d = tempfile.mkdtemp(prefix='nafp_full_')
    try:
        if shutil.disk_usage(d).free < rows * 512 + (1 << 30):
            return {'skipped': f'not enough room under {d} for {rows * 512 / 1e9:.2f} GB'}
        m_pre, m_fp = g.build_fp(cfg)
        t_ax = torch.arange(8000, device='cuda', dtype=torch.float32) / 8000.0
        gen = torch.Generator(device='cuda')

        def synth(row0, n):                      # a launch of whole groups: seeded noise + one tone per row
            gen.manual_seed(row0)
            f = 300.0 + (torch.arange(row0, row0 + n, device='cuda') % 3500).float()
            return 0.1 * torch.randn((n, 1, 8000), generator=gen, device='cuda') + 0.2 * torch.sin(2 * torch.pi * f[:, None, None] * t_ax)
        m_fp(m_pre(synth(0, 5 * group), group_size=group, defer=True))
        arr = np.memmap(os.path.join(d, 'dummy_db.mm'), dtype='float32', mode='w+', shape=(rows, 128))
        np.save(os.path.join(d, 'dummy_db_shape.npy'), (rows, 128))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.write_fingerprints_from_device_rows(synth, rows, m_pre, m_fp, arr, group)
        t1 = time.perf_counter()
        arr.flush()
        t2 = time.perf_counter()
        ok = bool(np.isfinite(arr[-1]).all()) and abs(float(np.linalg.norm(arr[rows // 2])) - 1.0) < 1e-4
        del arr
        return {'rows': rows, 'bytes': rows * 512, 'value': round(rows / (t2 - t0), 1), 'unit': 'segments/s incl. audio synthesis, D2H, memmap stores and flush',
                'seconds': round(t2 - t0, 3), 'flush_seconds': round(t2 - t1, 3), 'launch_rows': 5 * group, 'streams': g.N_STREAMS,
                'share_of_one_rank': round(rows / 12_500_000, 4), 'rows_ok': ok}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def spawn_ranks(n):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as ONE child process tree
    (`python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>`), relay rank 0's JSON line, return
    the child's exit code.  This parent never initialises the GPU (no HIP call, no torch.cuda.is_available()), so
    nothing that touched the GPU is replaced or forked."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = []
    for line in child.stdout:
        if line.startswith('{'):
            lines.append(line)
        else:
            sys.stderr.write(line)
    rc = child.wait()
    for line in lines:
        sys.stdout.write(line)
    sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--streams', type=int, default=1,
                    help='streams of the timed region (1: every kernel runs alone, so HIP-event launch '
                         'durations are the kernels own).  With 1, a second region with 4 streams is '
                         'timed afterwards and reported as "pipelined".')
    ap.add_argument('--repeats', type=int, default=5, help='timed regions of --steps steps each; the median is reported')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-e2e', action='store_true', help='skip the disk -> .mm figures (reported as "e2e_generate")')
    ap.add_argument('--no-train', action='store_true', help='skip the contrastive-train region (reported as "train")')
    ap.add_argument('--train-steps', type=int, default=10, help='steps per timed region of the BSZ-5120 train object')
    ap.add_argument('--train-repeats', type=int, default=3, help='timed regions of the train objects; the median is reported')
    ap.add_argument('--train-bsz', type=int, default=5120, help='GLOBAL train batch (BASELINE.json configs[3])')
    ap.add_argument('--fullscale-rows', type=int, default=1_250_000,
                    help='rows of the configs[4] stand-in (reported as "fullscale_generate"); 12500000 = one rank\'s whole share; 0 = skip')
    ap.add_argument('--no-pipelined', action='store_true',
                    help='skip the extra 4-stream region (used under rocprofv3 so that its per-kernel '
                         'averages cover the single-stream launches only)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args.gpus))                # nothing above has touched the GPU

    # stdout carries ONE JSON line and nothing else: RCCL prints a version banner to fd 1 when a communicator is created
    # (rank 0, every init), libraries chat -- all of that is sent to stderr for the rest of the run, the line is written to
    # the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import yaml
    import __graft_entry__
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != max(args.gpus, 1):
        sys.exit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}')
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if rank == 0:
        __graft_entry__.build()
    dist = None
    if world > 1:
        import torch.distributed as dist
        # NAFP_BENCH_BACKEND=gloo + NAFP_BENCH_ONE_GPU=1: exercise this code path with several
        # ranks on ONE device (RCCL refuses two ranks per GPU); never used by the driver.
        backend = os.environ.get('NAFP_BENCH_BACKEND', 'nccl')
        if os.environ.get('NAFP_BENCH_ONE_GPU'):
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
        dist.barrier()
    else:
        torch.cuda.set_device(0)
    import neural_audio_fp_amd as nafp
    dev = torch.device('cuda', local_rank if world > 1 else 0)

    with open(os.path.join(ROOT, 'config', 'default.yaml')) as f:
        cfg = yaml.safe_load(f)
    m_pre = nafp.get_melspec_layer(cfg)
    m_fp = nafp.FingerPrinter(emb_sz=cfg['MODEL']['EMB_SZ'], norm=cfg['MODEL']['BN'], seed=0, device=dev)
    n_pool = 4
    pool = [make_audio(BSZ, 1000 * rank + i, torch).to(dev) for i in range(n_pool)]

    def step(i):
        return m_fp(m_pre(pool[i % n_pool], group_size=BSZ, defer=True))

    # Every step is one complete pass over its own batch.  Consecutive steps are issued
    # round-robin on `--streams` HIP streams so that the low-occupancy late convs of one batch
    # overlap the large convs of the next (same pipelining the generate driver uses).
    n_str = max(1, args.streams)
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_str)]

    def step_on(i, ev=None):
        with torch.cuda.stream(streams[i % n_str]):
            x = pool[i % n_pool]
            if ev:
                ev[2 * i].record()
            feat = m_pre(x, group_size=BSZ, defer=True)      # the layer's max subtraction is finished inside conv0
            if ev:
                ev[2 * i + 1].record()
            return m_fp(feat)

    for i in range(max(args.warmup, n_str)):
        step_on(i)
    torch.cuda.synchronize()
    # HIP events of the timed region: 2 per forward, attached by the library to the dispatch packets of conv1 (start) and
    # of the last GEMM-conv kernel (stop) on the launch stream -- the 15 GEMM-conv launches are timed as ONE span per step,
    # average launch = span / 15.  An event RECORDED between two kernels idles the GPU (~20 us of un-overlapped dispatch
    # set-up each), so everything else -- the per-conv split, conv0, the front end, the tail -- is taken from a second,
    # untimed pass below.
    reps = max(1, args.repeats)
    m_fp.profile_enable(args.steps * reps, coarse=2)
    ev = None
    regions = []
    for _ in range(reps):
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            emb = step_on(i, ev)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        el = time.perf_counter() - t0
        if dist:
            t = torch.tensor([el], dtype=torch.float64, device=dev if dist.get_backend() == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t[0])
        regions.append(el)
    el = sorted(regions)[len(regions) // 2]             # the median region is the one reported
    assert emb.shape == (BSZ, 128) and bool(torch.isfinite(emb).all())

    prof = m_fp.profile_read()
    # per-stage split: the same steps once more on one stream with a stamp after every launch (outside the timed region)
    n_fine = min(args.steps, 8)
    m_fp.profile_enable(n_fine)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * n_fine)]
    for i in range(n_fine):
        with torch.cuda.stream(streams[0]):
            ev[2 * i].record()
            feat = m_pre(pool[i % n_pool], group_size=BSZ, defer=True)
            ev[2 * i + 1].record()
            m_fp(feat)
    torch.cuda.synchronize()
    prof_fine = m_fp.profile_read()
    mel_ms = sum(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(n_fine)) / n_fine
    # Outside the timed region: the same steps on ONE stream, so that each kernel's duration is
    # its own (in the pipelined region kernels of different batches share the chip and the
    # per-launch durations include that sharing).
    # Same K steps again, pipelined over 4 streams (what the generate driver does): the low-
    # occupancy late convs of one batch overlap the large convs of the next.  Reported
    # separately; `value` stays the single-stream figure that `roofline` is consistent with.
    pipelined = None
    if n_str == 1 and not args.no_pipelined:
        m_fp.profile_enable(0)
        ps = [torch.cuda.Stream(device=dev) for _ in range(4)]

        def pstep(i):
            with torch.cuda.stream(ps[i % 4]):
                return m_fp(m_pre(pool[i % n_pool], group_size=BSZ, defer=True))
        for i in range(4):
            pstep(i)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        tp0 = time.perf_counter()
        for i in range(args.steps):
            pstep(i)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        pel = time.perf_counter() - tp0
        if dist:
            t = torch.tensor([pel], dtype=torch.float64, device=dev if dist.get_backend() == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            pel = float(t[0])
        pipelined = {'streams': 4, 'value': round(world * BSZ * args.steps / pel, 1), 'unit': 'segments/s',
                     'ms_per_step': round(pel / args.steps * 1e3, 4)}
    # EXPERIMENTAL, reported separately and never as `value`: the unsplit GEMM convs with split-bf16 products (f32 operands
    # split into hi + lo bf16 in registers, hi*hi + hi*lo + lo*hi on the bf16 matrix pipe, f32 accumulation:
    # NAFP_OPT_BF16X3).  Narrower arithmetic than the reference's f32, so its error against the f32 path is measured here.
    bf16x3, bf16x6 = None, None
    if n_str == 1 and not args.no_pipelined:
        with torch.cuda.stream(streams[0]):
            ref_emb = m_fp(m_pre(pool[0], group_size=BSZ, defer=True)).clone()
            m_fp.set_option(3, 1)
            got_emb = m_fp(m_pre(pool[0], group_size=BSZ, defer=True)).clone()
            for i in range(3):
                m_fp(m_pre(pool[i % n_pool], group_size=BSZ, defer=True))
        torch.cuda.synchronize()
        tb0 = time.perf_counter()
        with torch.cuda.stream(streams[0]):
            for i in range(args.steps):
                m_fp(m_pre(pool[i % n_pool], group_size=BSZ, defer=True))
        torch.cuda.synchronize()
        bel = time.perf_counter() - tb0
        # ... and the exact 3-way split with six products (option value 2): float32-equivalent arithmetic on the bf16 pipe
        with torch.cuda.stream(streams[0]):
            m_fp.set_option(3, 2)
            got6_emb = m_fp(m_pre(pool[0], group_size=BSZ, defer=True)).clone()
            for i in range(3):
                m_fp(m_pre(pool[i % n_pool], group_size=BSZ, defer=True))
        torch.cuda.synchronize()
        tb0 = time.perf_counter()
        with torch.cuda.stream(streams[0]):
            for i in range(args.steps):
                m_fp(m_pre(pool[i % n_pool], group_size=BSZ, defer=True))
        torch.cuda.synchronize()
        bel6 = time.perf_counter() - tb0
        # its own roofline: the span of the GEMM convs from the library's dispatch-attached stamps (one span per step, as for `roofline`),
        # and a per-conv pass for the table of profiles/r06_summary.md
        m_fp.profile_enable(args.steps, coarse=2)
        with torch.cuda.stream(streams[0]):
            for i in range(args.steps):
                m_fp(m_pre(pool[i % n_pool], group_size=BSZ, defer=True))
        torch.cuda.synchronize()
        prof6 = m_fp.profile_read()
        m_fp.profile_enable(min(args.steps, 8))
        with torch.cuda.stream(streams[0]):
            for i in range(min(args.steps, 8)):
                m_fp(m_pre(pool[i % n_pool], group_size=BSZ, defer=True))
        torch.cuda.synchronize()
        prof6_fine = m_fp.profile_read()
        m_fp.profile_enable(0)
        # the same steps pipelined over 4 streams, as `pipelined` does for the f32 path (front end and forwards of other batches next
        # to the split kernels: allowed since the library holds no packed-f32 instruction, include/nafp.h) -- with a bit-equality
        # check of every pipelined result against the single-stream result of the same batch
        solo6 = []
        with torch.cuda.stream(streams[0]):
            for i in range(n_pool):
                solo6.append(m_fp(m_pre(pool[i], group_size=BSZ, defer=True)).clone())
        ps6 = [torch.cuda.Stream(device=dev) for _ in range(4)]

        def pstep6(i):
            with torch.cuda.stream(ps6[i % 4]):
                return m_fp(m_pre(pool[i % n_pool], group_size=BSZ, defer=True))
        torch.cuda.synchronize()
        keep6 = [pstep6(i) for i in range(8)]             # (fresh output tensors; compared after the synchronize below)
        torch.cuda.synchronize()
        pipe6_equal = all(torch.equal(keep6[i], solo6[i % n_pool]) for i in range(8))
        tb0 = time.perf_counter()
        for i in range(args.steps):
            pstep6(i)
        torch.cuda.synchronize()
        pel6 = time.perf_counter() - tb0
        m_fp.set_option(3, 0)
        bf16x6 = {'value': round(world * BSZ * args.steps / bel6, 1), 'unit': 'segments/s', 'ms_per_step': round(bel6 / args.steps * 1e3, 4),
                  'dtype': 'exact 3-way bf16 split x = h + m + l, 6 products (relative weight >= 2^-16), f32 accumulation, f32 storage',
                  'max_abs_diff_vs_f32_path': float((got6_emb - ref_emb).abs().max()),
                  'min_cosine_vs_f32_path': float((got6_emb * ref_emb).sum(1).min()),
                  'note': 'experimental option NAFP_OPT_BF16X3 = 2: all 15 GEMM convs on the exact split, conv0 generated inside conv1; '
                          'float32-equivalent (error vs the float64 oracle = the f32 path\'s own: tests/test_gpu_parity_forward.py, '
                          'tests/test_gpu_exact_split_adversarial.py); `value` on a single stream like the headline; a separate object, not part of `value`',
                  'pipelined': {'streams': 4, 'value': round(world * BSZ * args.steps / pel6, 1), 'unit': 'segments/s',
                                'ms_per_step': round(pel6 / args.steps * 1e3, 4), 'bit_identical_to_single_stream': bool(pipe6_equal)}}
        macs6 = conv_effective_macs()
        alg6 = 2.0 * sum(macs6[1:]) * BSZ                                   # algorithmic float32 FLOPs of the 15 GEMM convs per step
        gemm6_ms = sum(sum(p[1:16]) for p in prof6) / len(prof6)
        ex6 = 6.0 * alg6 / (gemm6_ms * 1e-3) / 1e12                         # executed bf16 TFLOP/s: six products per float32 product
        bf16x6['roofline'] = {
            'bound': 'mfma', 'kernel': 'conv_gemm_*_bf16x6 (15 launches/step; conv1 with conv0 generated in-kernel), v_mfma_f32_32x32x16_bf16',
            'achieved': round(ex6, 1), 'peak': BF16_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s executed on the bf16 matrix pipe',
            'frac': round(ex6 / BF16_MFMA_PEAK_TFLOPS, 4),
            'f32_equivalent_TFLOP/s': round(alg6 / (gemm6_ms * 1e-3) / 1e12, 1),
            'executed_flops_per_step': 6.0 * alg6, 'algorithmic_f32_flops_per_step': alg6, 'gemm_span_ms_per_step': round(gemm6_ms, 4),
            'per_conv_ms': [round(sum(p[k] for p in prof6_fine) / len(prof6_fine), 4) for k in range(17)],
            'note': 'executed FLOPs = 6 x the algorithmic float32 FLOPs (every product is formed as six bf16 products) / the span from the first '
                    'GEMM-conv launch to the end of the last, against the dense bf16 matrix peak; never a fraction of the fp32 peak.  '
                    'per_conv_ms[0] is the statistics pass of conv0 (its activation is generated inside conv1)'}
        bf16x3 = {'value': round(world * BSZ * args.steps / bel, 1), 'unit': 'segments/s', 'ms_per_step': round(bel / args.steps * 1e3, 4),
                  'dtype': 'bf16 x 3 products (hi*hi + hi*lo + lo*hi), f32 accumulation, f32 storage',
                  'max_abs_diff_vs_f32_path': float((got_emb - ref_emb).abs().max()),
                  'min_cosine_vs_f32_path': float((got_emb * ref_emb).sum(1).min()),
                  'note': 'experimental option NAFP_OPT_BF16X3 on the unsplit GEMM convs (convs 1-6, 8 at BSZ 640); NOT the '
                          "reference's arithmetic, not part of `value`"}
    train, train_1280, train_r640, train_x6 = None, None, None, None
    if not args.no_train:
        train = train_region(cfg, world, rank, dist, args.train_bsz, args.train_steps, torch, repeats=args.train_repeats)
        if world == 1:
            # EXPERIMENTAL, a separate object, never `train`: the same step with the GEMM products of forward_train and of the transposed
            # convs on the exact 3-way bf16 split (gradient parity: tests/test_gpu_backward.py, tests/test_gpu_configs.py)
            try:
                train_x6 = train_region(cfg, world, rank, dist, args.train_bsz, args.train_steps, torch, repeats=args.train_repeats, arith='x6')
                train_x6['dtype'] = ('forward_train and transposed convs: exact 3-way bf16 split, 6 products, f32 accumulation (float32-equivalent); '
                                     'weight gradients, LayerNorm backward, loss, optimizer: f32')
                train_x6['vs_f32_step'] = round(train['ms_per_step'] / train_x6['ms_per_step'], 4)
                # (its work rate is float32-EQUIVALENT work per time on two pipes: never a fraction of the fp32 matrix peak)
                train_x6['f32_equivalent_TFLOP/s'] = train_x6.pop('achieved_TFLOP/s')
                train_x6.pop('mfma_frac_of_peak', None)
                # ... and at the 8-GPU operating point (per-rank batch 640, no process group: compare `train_rank640.no_process_group_ms_per_step`)
                r6 = train_region(cfg, 1, 0, None, 640, 20, torch, warmup=4, optimizer='LAMB', repeats=args.train_repeats, arith='x6')
                train_x6['rank640'] = {k: r6[k] for k in ('value', 'unit', 'ms_per_step', 'spread', 'global_batch', 'optimizer')}
                train_x6['rank640']['what'] = 'one rank\'s share (640 of 5120 segments) on ONE GPU, no process group: a one-GPU compute bound of the 8-rank step'
            except Exception as ex:      # a secondary object must not take the headline line down
                train_x6 = {'error': f'{type(ex).__name__}: {ex}'}
        if world == 1:                                  # BSZ 1280, Adam, one GPU
            train_1280 = train_region(cfg, world, rank, dist, min(1280, args.train_bsz), max(args.train_steps, 12), torch,
                                      warmup=3, optimizer='Adam', repeats=args.train_repeats)
            # the 8-GPU operating point (per-rank batch 640) as a one-GPU compute bound.  Secondary object: if the 1-rank RCCL
            # group cannot be created on this box the headline line must still come out -- the failure is reported, not hidden
            try:
                train_r640 = train_rank640(cfg, torch, repeats=args.train_repeats)
            except Exception as ex:                      # noqa: BLE001
                train_r640 = {'error': f'{type(ex).__name__}: {ex}'[:400]}
    e2e, fullscale = None, None
    if world == 1 and not args.no_e2e:
        try:                                            # (secondary objects: a full disk must not cost the headline line)
            e2e = e2e_generate(cfg, torch)
        except OSError as ex:
            e2e = {'error': f'{type(ex).__name__}: {ex}'[:400]}
        if args.fullscale_rows > 0:
            try:
                fullscale = fullscale_generate(cfg, torch, args.fullscale_rows)
            except OSError as ex:
                fullscale = {'error': f'{type(ex).__name__}: {ex}'[:400]}
    iso = None
    if n_str > 1:
        m_fp.profile_enable(6)
        for i in range(6):
            with torch.cuda.stream(streams[0]):
                m_fp(m_pre(pool[i % n_pool], group_size=BSZ, defer=True))
        torch.cuda.synchronize()
        iso = m_fp.profile_read()
    if rank == 0:
        macs = conv_effective_macs()
        gemm_flops_per_step = 2.0 * sum(macs[1:]) * BSZ            # the 15 implicit-GEMM launches
        gemm_ms = sum(sum(p[1:16]) for p in prof) / len(prof)      # per step, all 15 launches
        ach = gemm_flops_per_step / (gemm_ms * 1e-3) / 1e12
        conv0_ms = sum(p[0] for p in prof_fine) / len(prof_fine)
        tail_ms = sum(p[16] for p in prof_fine) / len(prof_fine)
        value = world * BSZ * args.steps / el
        traffic, traffic_src, fe_traffic, fe_kernel_us = None, None, None, None
        tp = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tp):          # PMC passes cannot run inside this process: profiles/ holds them
            tj = json.load(open(tp))
            traffic = tj.get('per_launch_bytes')
            fe_traffic = (tj.get('frontend') or {}).get('bytes_per_launch')
            fe_k = ((tj.get('frontend') or {}).get('kernels') or {})
            fe_k = fe_k.get('melspec_r16_kernel') or fe_k.get('melspec_kernel') or {}
            fe_kernel_us = fe_k.get('mean_us_under_pmc')
            traffic_src = f"profiles/traffic.json ({tj.get('tag')}: rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)"
        out = {
            'metric': 'fingerprint generation throughput (1-s segments/s)',
            'value': round(value, 1), 'unit': 'segments/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(el / args.steps * 1e3, 4),
            'repeats': reps,
            'spread': {'ms_per_step_all': [round(r / args.steps * 1e3, 4) for r in regions],
                       'min': round(min(regions) / args.steps * 1e3, 4), 'max': round(max(regions) / args.steps * 1e3, 4),
                       'note': f'{reps} timed regions of {args.steps} steps each; ms_per_step / value = the median region'},
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': 'generate: f32 audio (640,1,8000) resident in HBM -> log-mel -> '
                                   '16-conv encoder -> div-enc -> L2, d=128, BSZ=640 per GPU per step '
                                   '(BASELINE.json configs[1]); seeded glorot weights',
                       'segments_per_step_per_gpu': BSZ, 'parallelism': f'segment-sharded x{world}, no collective'},
            'roofline': {
                'bound': 'mfma', 'kernel': 'conv_gemm_* (15 launches/step: tile variants m256k16s3 / k16s3 / k16s3_fuse0 / n64k16s3 '
                                                   'of one template, fp32 v_mfma_f32_32x32x2_f32)',
                'achieved': round(ach, 2), 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(ach / FP32_MFMA_PEAK_TFLOPS, 4), 'traffic': traffic, 'traffic_unit': 'bytes/launch',
                'traffic_source': traffic_src,
                'flops_per_launch_avg': gemm_flops_per_step / 15, 'ms_per_launch_avg': round(gemm_ms / 15, 5),
                'note': 'HIP events recorded by the library on the launch stream inside the timed region: one span per step '
                        'from the first GEMM-conv launch to the end of the last (split-K finish kernels and inter-launch gaps '
                        'included), average launch = span / 15' + ('' if n_str == 1 else f'; {n_str} batches are in flight on separate '
                        'streams there, so launches of different batches share the chip; "isolated" = the same '
                        'launches alone on one stream (un-timed pass after the region)')},
            'stage_ms_per_step': {'melspec(2 kernels)': round(mel_ms, 4), 'conv0': round(conv0_ms, 4),
                                  'conv_gemm x15': round(gemm_ms, 4), 'tail': round(tail_ms, 4),
                                  'per_conv': [round(sum(p[k] for p in prof_fine) / len(prof_fine), 4) for k in range(17)],
                                  'per_conv_note': '"conv_gemm x15" is the span timed inside the timed region; melspec, conv0, tail and '
                                                   'per_conv (conv0, the 15 GEMM convs, tail) come from a second, untimed pass: each GEMM '
                                                   'conv from the start of its first kernel to the end of its last (dispatch-attached time '
                                                   'stamps, split-K finish kernel included), conv0 / tail / melspec between recorded '
                                                   'events.  Stamping every launch still costs: these figures sum to ~10 % more than the '
                                                   'timed span; un-stamped kernel durations are in profiles/*_summary.md'},
            'frontend_hbm': {'bound': 'hbm', 'kernel': 'melspec_r16_kernel (STFT + mel + log; the max subtraction is applied by conv0 on load)',
                             'algorithmic_bytes_per_segment': 32000 + 32768,
                             'achieved': round(BSZ * (32000 + 32768) / (mel_ms * 1e-3) / 1e9, 2), 'peak': 8000.0, 'unit': 'GB/s',
                             'frac': round(BSZ * (32000 + 32768) / (mel_ms * 1e-3) / 1e9 / 8000.0, 4),
                             'kernel_us_rocprof': fe_kernel_us,
                             'frac_at_kernel_time': round(BSZ * (32000 + 32768) / (fe_kernel_us * 1e-6) / 1e9 / 8000.0, 4) if fe_kernel_us else None,
                             'note': '`achieved` divides by the time between two events around the whole stage (melspec_init_stats + the '
                                     'kernel + launch gaps, measured in this run); kernel_us_rocprof is the kernel alone under rocprofv3 '
                                     '(profiles/traffic.json)',
                             'traffic': fe_traffic, 'traffic_unit': 'bytes/launch (640 segments)',
                             'traffic_ratio_to_algorithmic': round(fe_traffic / (BSZ * (32000 + 32768)), 3) if fe_traffic else None},
        }
        out['config']['streams'] = n_str
        if pipelined:
            out['pipelined'] = pipelined
        if bf16x3:
            out['bf16x3_experimental'] = bf16x3
        if bf16x6:
            out['bf16x6_f32_equivalent_experimental'] = bf16x6
        if train_x6:
            out['train_x6_experimental'] = train_x6
        if train:
            out['train'] = train
        if train_1280:
            out['train_1280'] = train_1280
        if train_r640:
            out['train_rank640'] = train_r640
        if e2e:
            out['e2e_generate'] = e2e
        if fullscale:
            out['fullscale_generate'] = fullscale
        if iso:
            iso_ms = sum(sum(p[1:16]) for p in iso) / len(iso)
            iso_ach = gemm_flops_per_step / (iso_ms * 1e-3) / 1e12
            out['roofline']['isolated'] = {
                'achieved': round(iso_ach, 2), 'frac': round(iso_ach / FP32_MFMA_PEAK_TFLOPS, 4),
                'ms_per_launch_avg': round(iso_ms / 15, 5),
                'per_conv_ms': [round(sum(p[k] for p in iso) / len(iso), 4) for k in range(17)]}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

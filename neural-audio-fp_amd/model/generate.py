"""Fingerprint generation driver: checkpoint -> sources -> <key>.mm + <key>_shape.npy.

Mirrors `generate_fingerprint` of the reference (model/generate.py:91-194): same
arguments, same output directory layout `<OUTPUT_ROOT_DIR>/<NAME>/<IDX>/`, same
files (`np.memmap` float32 (n_items, EMB_SZ) C-order + `np.save(shape)`), same row
order, same keys (`dummy_db`, `query`, `db`, or `custom_source`), the same
overwrite prompt, FileNotFoundError and size warning.

What is different underneath:
  * the device work is the HIP path (`m_fp(m_pre(X))`, generate.py:83-88); every WAV is read once into a
    pinned arena and the 1-s windows are indexed on the device (`nafp_melspec_forward_windows_i16`);
  * one launch carries several TS_BATCH_SZ batches; the log-mel max-normalisation
    group stays TS_BATCH_SZ (melspectrogram.py:108), so every fingerprint equals what
    the reference computes batch by batch;
  * with torch.distributed initialised (one process per GPU) the rows are split
    across ranks on group boundaries and every rank writes its own slice of the
    same memmap -- no collective on the data path, one barrier at the end.
Checkpoints are torch files `<LOG_ROOT_DIR>/checkpoint/<NAME>/ckpt-<IDX>.pt` holding
{'model': state_dict} with the key names of `nnfp.tensor_names()`.
"""
import glob
import os
import re
import sys

import numpy as np
import torch

from .dataset import Dataset
from .fp.melspec.melspectrogram import get_melspec_layer
from .fp.nnfp import get_fingerprinter

LAUNCH_SEGMENTS = 640       # target segments per launch (whole groups)
# consecutive launches are pipelined round-robin over this many HIP streams (NAFP_GEN_STREAMS overrides; round 6, same box: the device-only loop
# gains 2 - 3 % from 6 streams, the disk -> .mm and full-scale drivers nothing: 4 / 6 / 8 -> 183 / 180 / 183 k and 179 / 177 / 181 k segments/s)
N_STREAMS = int(os.environ.get('NAFP_GEN_STREAMS', '4'))


def build_fp(cfg):
    """generate.py:16-23."""
    return get_melspec_layer(cfg, trainable=False), get_fingerprinter(cfg, trainable=False)


def load_checkpoint(checkpoint_root_dir, checkpoint_name, checkpoint_index, m_fp, optimizer=None):
    """generate.py:26-52: latest index when none is given; FileNotFoundError if absent.  In order of preference:
    `ckpt-N.pt` (this build); `ckpt-N.npz` (weights converted inside the reference's TensorFlow environment by
    tools/convert_tf_checkpoint.py); `ckpt-N.index` + `ckpt-N.data-00000-of-00001` = a checkpoint WRITTEN BY THE
    REFERENCE itself, read directly (utils/tf_checkpoint.py: every checksum of the format verified; see its caveat)."""
    checkpoint_dir = checkpoint_root_dir + f'/{checkpoint_name}/'
    if checkpoint_index is None:
        print("\x1b[1;32mArgument 'checkpoint_index' was not specified.\x1b[0m")
        print('\x1b[1;32mSearching for the latest checkpoint...\x1b[0m')
        idx = [int(m.group(1)) for f in glob.glob(checkpoint_dir + 'ckpt-*')
               for m in [re.search(r'ckpt-(\d+)\.(pt|npz|index)$', f)] if m]
        if not idx:
            raise FileNotFoundError(f'Cannot find checkpoint in {checkpoint_dir}')
        checkpoint_index = max(idx)
    fpath = checkpoint_dir + 'ckpt-' + str(checkpoint_index) + '.pt'
    if os.path.exists(fpath):
        ck = torch.load(fpath, map_location='cpu', weights_only=True)
        m_fp.load_state_dict(ck['model'] if 'model' in ck else ck)
        if optimizer is not None and 'optimizer' in ck:
            optimizer.load_state_dict(ck['optimizer'], m_fp.trainable_variables)
    elif os.path.exists(fpath[:-3] + '.npz'):
        fpath = fpath[:-3] + '.npz'
        with np.load(fpath) as z:
            m_fp.load_state_dict({k: z[k] for k in z.files})
    elif os.path.exists(fpath[:-3] + '.index'):
        from .utils import tf_checkpoint as tfc
        from .fp.nnfp import tensor_names
        fpath = fpath[:-3]
        variables = list(m_fp.trainable_variables) + list(getattr(m_fp, 'non_trainable_variables', []))      # (batch_norm: + the moving statistics)
        m_fp.load_state_dict(tfc.state_dict_from_tf_checkpoint(fpath, tensor_names(getattr(m_fp, 'norm', 'layer_norm2d')),
                                                               [tuple(v.shape) for v in variables], m_fp.emb_sz))
        fpath += '.index'
    else:
        raise FileNotFoundError(f'Cannot find checkpoint {fpath}')
    print(f'---Restored from {fpath}---')
    return int(checkpoint_index)


def save_checkpoint(checkpoint_root_dir, checkpoint_name, checkpoint_index, m_fp, extra=None):
    d = checkpoint_root_dir + f'/{checkpoint_name}/'
    os.makedirs(d, exist_ok=True)
    payload = {'model': m_fp.state_dict()}
    if extra:
        payload.update(extra)
    torch.save(payload, d + f'ckpt-{int(checkpoint_index)}.pt')
    return d + f'ckpt-{int(checkpoint_index)}.pt'


def prune_checkpoints(checkpoint_root_dir, checkpoint_name, max_to_keep=3, keep_every_n_hours=None):
    """What tf.train.CheckpointManager(max_to_keep=3, keep_checkpoint_every_n_hours=N) does for the reference
    (experiment_helper.py:100-107): the 3 newest checkpoints stay; an older one is deleted unless at least N hours lie
    between it and the previously preserved one (file modification times)."""
    d = checkpoint_root_dir + f'/{checkpoint_name}/'
    found = sorted((int(m.group(1)), f) for f in glob.glob(d + 'ckpt-*.pt') for m in [re.search(r'ckpt-(\d+)\.pt$', f)] if m)
    old = found[:-max_to_keep] if max_to_keep else []
    last_kept = None
    removed = []
    for _, f in old:
        t = os.path.getmtime(f)
        if keep_every_n_hours and (last_kept is None or t - last_kept >= 3600.0 * keep_every_n_hours):
            last_kept = t
            continue
        os.remove(f)
        removed.append(f)
    return removed


def overwrite_refused(key, target_path):
    """The question of generate.py:55-58, answered without exiting: 0 = go ahead, 1 = the user declined, 2 = nobody
    could be asked.  The reference calls `input()`: a terminal or a piped answer is read the same way, and an exhausted
    stdin (batch job, nohup, torchrun) raises EOFError there.  Here that case refuses -- the expensive dummy_db.mm is
    never overwritten silently -- unless NAFP_OVERWRITE=1 says so explicitly."""
    if (key == 'dummy_db') & os.path.exists(target_path):
        if os.environ.get('NAFP_OVERWRITE', '').lower() in ('1', 'y', 'yes', 'true'):
            print(f'{target_path} exists; NAFP_OVERWRITE is set: the file will be overwritten.')
            return 0
        try:
            answer = input(f'{target_path} exists. Will you overwrite (y/N)?')
        except (EOFError, OSError):
            print(f'\n{target_path} exists and there is nobody to ask (stdin is closed): NOT overwriting.  Set '
                  'NAFP_OVERWRITE=1 to overwrite without the prompt, or pass --skip_dummy.', file=sys.stderr)
            return 2
        return 0 if answer.lower() in ['y', 'yes'] else 1
    return 0


def _leave(refused):
    """generate.py:58 exits quietly when the user says no; an unanswerable prompt is an error exit."""
    sys.exit(None if refused == 1 else f'generate: refused to overwrite (code {refused})')


def prevent_overwrite(key, target_path):
    """generate.py:55-58."""
    refused = overwrite_refused(key, target_path)
    if refused:
        _leave(refused)


def get_data_source(cfg, source_root_dir, skip_dummy):
    """generate.py:61-80."""
    dataset = Dataset(cfg)
    ds = dict()
    if source_root_dir:
        ds['custom_source'] = dataset.get_custom_db_ds(source_root_dir)
    else:
        if skip_dummy:
            print("Excluding \033[33m'dummy_db'\033[0m from source.")
        else:
            ds['dummy_db'] = dataset.get_test_dummy_db_ds()
        if dataset.datasel_test_query_db in ['unseen_icassp', 'unseen_syn']:
            ds['query'], ds['db'] = dataset.get_test_query_db_ds()
        else:
            raise ValueError(dataset.datasel_test_query_db)
    print(f'\x1b[1;32mData source: {ds.keys()}\x1b[0m', f'{dataset.datasel_test_query_db}')
    return ds


def test_step(X, m_pre, m_fp, group_size=None):
    """generate.py:83-88: m_fp(m_pre(X)) (the layer's max subtraction is finished inside the encoder's first conv)."""
    m_fp.trainable = False
    return m_fp(m_pre(X, group_size=group_size, defer=True))


def shard_rows(n_items, group, rank, world):
    """Contiguous row range of `rank`: whole max-normalisation groups, balanced to
    within one group.  Every row belongs to exactly one rank."""
    n_groups = (n_items + group - 1) // group
    g0 = (n_groups * rank) // world
    g1 = (n_groups * (rank + 1)) // world
    return min(g0 * group, n_items), min(g1 * group, n_items)


def write_fingerprints(source, embed_fn, arr, group, rank=0, world=1, launch_rows=None):
    """Fill arr[rows of this rank] = embed_fn(int16 chunk (n,1,T), group) in row order."""
    if hasattr(source, 'plan'):                      # a device-side augmenting sequence ('unseen_syn' queries)
        return _write_from_sequence(source, embed_fn, arr, rank, world)
    r0, r1 = shard_rows(source.n_samples, group, rank, world)
    k = max(1, -(-LAUNCH_SEGMENTS // group))
    launch_rows = launch_rows or k * group
    # embed_fn may return the array itself or a handle with .result() (device work still in
    # flight); up to `depth` launches are kept pending so that host I/O, copies and the kernels
    # of consecutive launches overlap.
    pending, depth = [], getattr(embed_fn, 'depth', 1)

    def drain(limit):
        while len(pending) > limit:
            st, n, fut = pending.pop(0)
            arr[st:st + n, :] = fut.result() if hasattr(fut, 'result') else fut

    if getattr(embed_fn, 'windows', False) and hasattr(source, 'iter_windows'):
        # whole-file upload + on-device windowing (no segment assembly on the host)
        it = source.iter_windows(r0, r1, launch_rows, alloc=embed_fn.alloc)
        if getattr(embed_fn, 'prefetch', 0) > 0:
            it = _prefetch(it, embed_fn.prefetch)          # file reads of the next launches on a reader thread
        for start, n, arena, used, off, valid in it:
            pending.append((start, n, embed_fn.embed_windows(arena, used, off, valid, group)))
            drain(depth - 1)
    else:
        for start, chunk in source.iter_rows(r0, r1, launch_rows):
            pending.append((start, len(chunk), embed_fn(chunk, group)))
            drain(depth - 1)
    drain(0)
    return r0, r1


def _write_from_sequence(seq, embed_fn, arr, rank=0, world=1):
    """Rows of a `genUnbalSequence` built with reduce_batch_first_half=True: batch i holds the synthesised replicas of
    rows [i*n_anchor, (i+1)*n_anchor); one batch = one max-normalisation group, like generate.py:176-181 feeds m_pre."""
    n_b = len(seq)
    b0, b1 = (n_b * rank) // world, (n_b * (rank + 1)) // world
    for i in range(b0, b1):
        X, _ = seq[i]
        emb = embed_fn.embed_device(X)
        arr[i * seq.n_anchor:i * seq.n_anchor + len(X), :] = emb.cpu().numpy()
    return b0 * seq.n_anchor, min(b1 * seq.n_anchor, seq.n_samples)


def write_fingerprints_from_device_rows(row_fn, n_items, m_pre, m_fp, arr, group, rank=0, world=1, launch_groups=5,
                                        n_streams=N_STREAMS):
    """Fill arr[rows of this rank] from a DEVICE-side source: row_fn(row0, n) -> (n, 1, T) CUDA batch of rows
    [row0, row0 + n) (audio synthesised or already resident in HBM -- the full-scale stand-in of SURVEY.md 8d
    config 5, where no 443 GB dataset exists).  Same sharding (`shard_rows`, whole max-normalisation groups), same
    launch pipelining (round-robin over `n_streams` HIP streams, pinned download buffers) and same row placement as
    `write_fingerprints`.  Returns the row range written."""
    r0, r1 = shard_rows(n_items, group, rank, world)
    launch = launch_groups * group
    n_streams = streams_for(m_fp, n_streams)
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    host = [torch.empty((launch, m_fp.emb_sz), dtype=torch.float32).pin_memory() for _ in range(n_streams)]
    pend = [None] * n_streams

    def drain(s):
        if pend[s] is not None:
            ev, a, n = pend[s]
            ev.synchronize()
            arr[a:a + n, :] = host[s][:n].numpy()
            pend[s] = None

    m_fp.trainable = False
    m_fp._sync()
    torch.cuda.current_stream().synchronize()
    for k, a in enumerate(range(r0, r1, launch)):
        s = k % n_streams
        drain(s)
        n = min(launch, r1 - a)
        with torch.cuda.stream(streams[s]):
            emb = m_fp(m_pre(row_fn(a, n), group_size=group, defer=True))
            host[s][:n].copy_(emb, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        pend[s] = (ev, a, n)
    for s in range(n_streams):
        drain(s)
    return r0, r1


def _prefetch(gen, ahead):
    """Run generator `gen` on a reader thread, at most `ahead` items ahead of the consumer."""
    import queue
    import threading
    q = queue.Queue(maxsize=max(1, ahead - 1))
    end = object()

    def work():
        try:
            for item in gen:
                q.put(item)
            q.put(end)
        except BaseException as e:              # re-raised in the consumer
            q.put(e)

    threading.Thread(target=work, daemon=True).start()
    while True:
        item = q.get()
        if item is end:
            return
        if isinstance(item, BaseException):
            raise item
        yield item


class _Pending:
    def __init__(self, host, event):
        self.host, self.event = host, event

    def result(self):
        self.event.synchronize()
        return self.host.numpy()


def streams_for(m_fp, n_streams=N_STREAMS):
    """Streams a pipelined consumer of `m_fp` may use: all of them, whatever the arithmetic.  (For most of round 6 a model on the
    split-bf16 arithmetic, NAFP_OPT_BF16X3 != 0, was kept on ONE stream here: its GEMM kernels left the front end and the 64-column
    f32 GEMM kernels of other streams changed in a few rows.  Cause found since -- packed-f32 instructions with op_sel next to the
    128-bit-operand matrix instructions, include/nafp.h -- and removed at the root: the library holds no packed-f32 instruction.)"""
    return n_streams


class StreamedEmbedder:
    """m_fp(m_pre(X)) for consecutive launches, round-robin over HIP streams, with pinned
    staging buffers for the int16 upload and the float32 download (the reference does a
    synchronous `emb.numpy()` per batch, generate.py:180)."""

    def __init__(self, m_pre, m_fp, n_streams=N_STREAMS, windows=True):
        self.m_pre, self.m_fp = m_pre, m_fp
        n_streams = streams_for(m_fp, n_streams)
        self.streams = [torch.cuda.Stream() for _ in range(n_streams)]
        self.depth = n_streams
        self.i = 0
        self.h_in = [None] * n_streams
        self.h_out = [None] * n_streams
        self.windows = windows
        self.prefetch = 0                      # launches a reader thread may run ahead (measured: no gain, the
                                               # path is GPU-bound with the reads on the launching thread)
        # pinned PCM arenas of the window path: one per launch in flight + one per launch being read ahead
        # (an arena is reused n_streams + prefetch launches later, when its upload has been waited for)
        self.h_pcm = [None] * (n_streams + self.prefetch)
        self.a = 0
        self.h_idx = [None] * n_streams

    def __call__(self, chunk_i16, group):
        k = self.i % len(self.streams)
        self.i += 1
        n = chunk_i16.shape[0]
        if self.h_in[k] is None or self.h_in[k].shape[0] < n:
            self.h_in[k] = torch.empty((n,) + chunk_i16.shape[1:], dtype=torch.int16).pin_memory()
            self.h_out[k] = torch.empty((n, self.m_fp.emb_sz), dtype=torch.float32).pin_memory()
        self.h_in[k][:n].copy_(torch.from_numpy(chunk_i16))
        with torch.cuda.stream(self.streams[k]):
            x = self.h_in[k][:n].cuda(non_blocking=True)
            emb = test_step(x, self.m_pre, self.m_fp, group_size=group)
            out = self.h_out[k][:n]
            out.copy_(emb, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        return _Pending(out, ev)


    def embed_device(self, X):
        """(n,1,T) CUDA batch -> fingerprints; the batch is one max-normalisation group."""
        self.m_fp.trainable = False
        return self.m_fp(self.m_pre(X, group_size=len(X), defer=True))

    # ---- window path: the next launch's PCM is read straight into this slot's pinned arena ----
    def alloc(self, n_samples):
        k = self.a % len(self.h_pcm)
        self.a += 1
        if self.h_pcm[k] is None or self.h_pcm[k].shape[0] < n_samples:
            self.h_pcm[k] = torch.empty((max(n_samples, 1) * 5 // 4,), dtype=torch.int16).pin_memory()
        return self.h_pcm[k].numpy()

    def embed_windows(self, arena, used, seg_offset, seg_valid, group):
        k = self.i % len(self.streams)
        self.i += 1
        n = len(seg_offset)
        if self.h_idx[k] is None or self.h_idx[k].shape[1] < n:
            self.h_idx[k] = torch.empty((2, n), dtype=torch.int64).pin_memory()
        if self.h_out[k] is None or self.h_out[k].shape[0] < n:
            self.h_out[k] = torch.empty((n, self.m_fp.emb_sz), dtype=torch.float32).pin_memory()
        idx = self.h_idx[k]
        idx[0, :n].copy_(torch.from_numpy(seg_offset))
        idx[1, :n].copy_(torch.from_numpy(seg_valid))
        pcm_host = next((t for t in self.h_pcm if t is not None and t.data_ptr() == arena.ctypes.data), None)
        if pcm_host is None:
            pcm_host = torch.from_numpy(arena)              # caller-provided arena (not pinned): still correct
        with torch.cuda.stream(self.streams[k]):
            pcm = pcm_host[:max(used, 1)].cuda(non_blocking=True)
            d_idx = idx[:, :n].cuda(non_blocking=True)
            self.m_fp.trainable = False
            feat = self.m_pre.forward_windows(pcm, d_idx[0].contiguous(), d_idx[1].to(torch.int32), group_size=group,
                                              defer=True)
            emb = self.m_fp(feat)
            out = self.h_out[k][:n]
            out.copy_(emb, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        return _Pending(out, ev)


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def generate_fingerprint(cfg, checkpoint_name, checkpoint_index, source_root_dir, output_root_dir,
                         skip_dummy):
    """generate.py:91-194."""
    dist, rank, world = _dist()
    m_pre, m_fp = build_fp(cfg)
    checkpoint_root_dir = cfg['DIR']['LOG_ROOT_DIR'] + 'checkpoint/'
    checkpoint_index = load_checkpoint(checkpoint_root_dir, checkpoint_name, checkpoint_index, m_fp)

    ds = get_data_source(cfg, source_root_dir, skip_dummy)

    if output_root_dir:
        output_root_dir = output_root_dir + f'/{checkpoint_name}/{checkpoint_index}/'
    else:
        output_root_dir = cfg['DIR']['OUTPUT_ROOT_DIR'] + f'/{checkpoint_name}/{checkpoint_index}/'
    os.makedirs(output_root_dir, exist_ok=True)
    if not skip_dummy:
        # rank 0 asks; every rank learns the answer, so that all of them leave together (a lone sys.exit on
        # rank 0 would strand the others in the barrier below)
        refused = [overwrite_refused('dummy_db', f'{output_root_dir}/dummy_db.mm') if rank == 0 else 0]
        if dist:
            dist.broadcast_object_list(refused, src=0)
        if refused[0]:
            _leave(refused[0])

    embed = StreamedEmbedder(m_pre, m_fp)

    sz_check = dict()
    for key in ds.keys():
        bsz = int(cfg['BSZ']['TS_BATCH_SZ'])
        n_items = ds[key].n_samples
        dim = cfg['MODEL']['EMB_SZ']
        assert n_items > 0
        arr_shape = (n_items, dim)
        path = f'{output_root_dir}/{key}.mm'
        if rank == 0:
            arr = np.memmap(path, dtype='float32', mode='w+', shape=arr_shape)
            np.save(f'{output_root_dir}/{key}_shape.npy', arr_shape)
        if dist:
            dist.barrier()
        if rank != 0:
            arr = np.memmap(path, dtype='float32', mode='r+', shape=arr_shape)
        print(f"=== Generating fingerprint from \x1b[1;32m'{key}'\x1b[0m bsz={bsz}, {n_items} items, d={dim} ===")
        write_fingerprints(ds[key], embed, arr, bsz, rank, world)
        arr.flush()
        if dist:
            dist.barrier()
        print(f'=== Succesfully stored {arr_shape[0]} fingerprint to {output_root_dir} ===')
        sz_check[key] = len(arr)
        del arr

    if 'custom_source' in ds.keys():
        pass
    elif sz_check['db'] != sz_check['query']:
        print("\033[93mWarning: 'db' and 'qeury' size does not match. This can cause a problem in evaluataion stage.\033[0m")
    return

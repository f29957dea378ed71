"""Training driver on the HIP path.

Mirrors the reference's `build_fp`, `train_step`, `val_step`, `test_step` and `trainer`
(model/trainer.py:19-77, 111-230): same roles and call order -- X = concat(Xa, Xp); feat =
m_specaug(m_pre(X)) outside the differentiated region; emb = m_fp(feat); NT-Xent on
(emb[:nA], emb[nA:]); gradients w.r.t. m_fp.trainable_variables; optimizer step (Adam or LAMB
with a cosine schedule).  tf.GradientTape is replaced by the library's explicit backward pass
(`FingerPrinter.forward_train` / `.backward`) and the fused NT-Xent backward.

Data parallel (one process per GPU, torch.distributed over RCCL): every rank holds n_a anchors
and their replicas; the L2-normalised embeddings are all-gathered before the loss
(NTxent_loss_tpu.py:57-87, 110-126), each rank scores its local rows against all columns, the
gradient w.r.t. the gathered embeddings is summed over ranks (what TF's all_reduce-in-the-forward
yields in its backward) by a reduce-scatter, so each rank receives just its slice and back-propagates
it; parameter gradients live in one flat buffer that is all-reduced (sum) in NAFP_GRAD_GROUPS pieces on a
communication stream, each piece as soon as the backward pass has finished its layers (layers complete
last to first; the first piece -- convs 12-15, 65 % of the parameters -- is ready after a few per cent of
the backward pass), so every rank applies the gradient of the GLOBAL mean loss and the replicas stay
bit-identical.

The training set comes from `Dataset(cfg).get_train_ds()` (model/dataset.py, utils/dataloader_keras.py:
PCM resident in HBM, time-domain augmentation in one kernel) unless the caller passes its own
iterable of (Xa, Xp) batches; the validation loss (trainer.py:200-213) is printed per epoch when the
validation directory exists, followed by the mini search test (trainer.py:226-230) when TRAIN.MINI_TEST_IN_TRAIN
is set.  What is NOT here: TensorBoard.
"""
import torch

from .fp.melspec.melspectrogram import get_melspec_layer
from .fp.nnfp import get_fingerprinter
from .fp.NTxent_loss_single_gpu import NTxentLoss, _ntxent_call
from .fp.online_triplet_loss import OnlineTripletLoss
from .fp.lamb_optimizer import LAMB, Adam, CosineDecay, CosineDecayRestarts
from .fp.specaug_chain.specaug_chain import SpecAugChainer, get_specaug_chain_layer
from . import generate as _gen


_PREFETCH_WEIGHTS = __import__('os').environ.get('NAFP_PREFETCH_WEIGHTS', '1') != '0'


def build_fp(cfg):
    """trainer.py:19-30."""
    m_pre = get_melspec_layer(cfg, trainable=False)
    m_specaug = get_specaug_chain_layer(cfg, trainable=False)
    assert m_specaug.bypass is False
    m_fp = get_fingerprinter(cfg, trainable=False)
    return m_pre, m_specaug, m_fp


NO_DIST = False       # bench.py sets this around its compute-only reference step (`exposed_comm_ms`): the step of ONE rank, no collectives


def _dist():
    """torch.distributed when a process group exists -- also a group of ONE rank (legal for RCCL): the step then runs
    the same collectives as on a node (tests/test_gpu_train_dp.py drives that on the 1-GPU box)."""
    import torch.distributed as dist
    if NO_DIST:
        return None
    if dist.is_available() and dist.is_initialized():
        return dist
    return None


class GradientBucket:
    """All parameter gradients as views into ONE flat buffer (67.8 MB for the 1-s model), reduced in
    NAFP_GRAD_GROUPS contiguous pieces instead of 68 tensors: `pieces[k]` = the slice of gradient group k
    of the library (include/nafp.h: groups in completion order of the backward pass)."""

    def __init__(self, m_fp):
        vs = m_fp.trainable_variables
        self.flat = torch.zeros(sum(v.numel() for v in vs), dtype=torch.float32, device=vs[0].device)
        self.views, offs, o = [], [], 0
        for v in vs:
            self.views.append(self.flat[o:o + v.numel()].view_as(v))
            offs.append(o)
            o += v.numel()
        offs.append(o)
        m_fp._grads = self.views          # FingerPrinter.backward writes straight into the views
        self.m_fp = m_fp
        self.pieces = [self.flat[offs[a]:offs[b + 1]] for a, b in m_fp.grad_groups()]
        self.comm_stream = None
        self.timings = None               # optional: {'allreduce_ms': [...]} filled when `timed` is set
        self.timed = False
        self._ev = None

    def all_reduce(self, dist):
        """Sum the gradients over the ranks, overlapped with the backward pass still running on the current
        stream: piece k is reduced on the communication stream as soon as the library's event for group k has
        fired.  Returns after making the current stream wait for all pieces (no host block)."""
        cur = torch.cuda.current_stream(self.flat.device)
        if self.comm_stream is None:
            self.comm_stream = torch.cuda.Stream(device=self.flat.device)
        cs = self.comm_stream
        evs = []
        with torch.cuda.stream(cs):
            for k, piece in enumerate(self.pieces):
                self.m_fp.grad_group_wait(k)               # device-side wait of `cs` on the backward pass
                if self.timed:
                    e0 = torch.cuda.Event(enable_timing=True); e0.record(cs)
                dist.all_reduce(piece)
                if self.timed:
                    e1 = torch.cuda.Event(enable_timing=True); e1.record(cs)
                    evs.append((e0, e1))
        cur.wait_stream(cs)
        self._ev = evs or None

    def read_timings(self):
        """ms of each piece's all-reduce of the last `all_reduce` call (after a device synchronisation)."""
        if not self._ev:
            return None
        return [a.elapsed_time(b) for a, b in self._ev]


def train_step(X, m_pre, m_specaug, m_fp, loss_obj, opt, bucket=None, timers=None):
    """trainer.py:33-50.  Returns (loss, None): loss is the GLOBAL batch loss (identical on every
    rank).  `bucket` (GradientBucket) is required for data-parallel runs.  `timers` (a list, bench.py): receives
    (name, start_event, end_event) of the two embedding collectives; the gradient pieces are timed by the bucket."""
    Xa, Xp = X
    n_anchors = len(Xa)
    X = torch.cat([torch.as_tensor(Xa), torch.as_tensor(Xp)], dim=0)
    feat = m_pre(X)                                 # outside the tape (trainer.py:41)
    feat = m_specaug(feat, inplace=True) if isinstance(m_specaug, SpecAugChainer) else m_specaug(feat)   # (the front end's own fresh tensor)
    m_fp.trainable = True
    emb = m_fp.forward_train(feat)
    ha, hb = emb[:n_anchors].contiguous(), emb[n_anchors:].contiguous()
    dist = _dist()
    if dist is None:
        loss, d_a, d_b = loss_obj.loss_and_grad(ha, hb)
        d_emb = torch.cat([d_a, d_b], dim=0)
    else:
        # 2 small collectives: all-gather of [ha; hb], then ONE reduce-scatter of [d/d a; d/d b; loss] laid out
        # per destination rank (every rank needs only the gradient of its own rows, NTxent_loss_tpu.py:57-87)
        world, rank = dist.get_world_size(), dist.get_rank()
        d = emb.shape[1]
        gathered = torch.empty((world * 2 * n_anchors, d), dtype=emb.dtype, device=emb.device)
        t0 = _mark(timers)
        dist.all_gather_into_tensor(gathered, emb.contiguous())
        _mark(timers, 'all_gather(emb)', t0)
        g4 = gathered.view(world, 2, n_anchors, d)
        a_all, b_all = g4[:, 0].reshape(-1, d), g4[:, 1].reshape(-1, d)
        n_g = world * n_anchors
        loss_sum, _, d_a_all, d_b_all = _ntxent_call(loss_obj._lib, ha, hb, a_all, b_all, rank * n_anchors,
                                                     loss_obj.tau, False, True)
        t0 = _mark(timers)
        loss, d_emb = scatter_embedding_gradients(dist, loss_sum, d_a_all, d_b_all, n_anchors, loss_scale=1.0 / n_g,
                                                  lib=loss_obj._lib)
        _mark(timers, 'reduce_scatter(d emb)', t0)
    grads = m_fp.backward(d_emb)
    if dist is not None:
        if bucket is None:
            raise ValueError('data-parallel train_step needs a GradientBucket')
        bucket.all_reduce(dist)
    opt.apply_gradients(zip(grads, m_fp.trainable_variables), var_lens=m_fp.variable_lengths())
    m_fp.mark_dirty()
    if _PREFETCH_WEIGHTS and hasattr(m_fp, 'prefetch_weights'):
        m_fp.prefetch_weights()                 # the re-pack for the next forward starts now, next to the next batch's front end
    return loss, None


def _mark(timers, name=None, start=None):
    if timers is None:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    if name is not None:
        timers.append((name, start, e))
    return e


def scatter_embedding_gradients(dist, loss_local, d_a_all, d_b_all, n_anchors, loss_scale=1.0, lib=None):
    """The backward of the embedding all-gather.  Every rank holds its contribution to the gradient w.r.t. ALL gathered
    embeddings (d_a_all, d_b_all: (world * n_anchors, d)) and its share `loss_local` of the global loss; it needs the sum
    over ranks of only ITS rows.  ONE reduce-scatter of [d/d a | d/d b | loss] laid out per destination rank; the loss
    rides in a 4-float tail of every chunk, so each rank also receives the global loss.
    Returns (loss, d_emb (2 * n_anchors, d) = [d/d a_local ; d/d b_local])."""
    world, d = dist.get_world_size(), d_a_all.shape[1]
    chunk = 2 * n_anchors * d
    send = torch.empty((world, chunk + 4), dtype=torch.float32, device=d_a_all.device)
    if lib is not None and d_a_all.is_cuda:
        # one library launch packs [d/d a | d/d b | loss * loss_scale] per destination rank (three strided torch copies, a
        # fill and a division before: five small launches on the latency-bound per-rank-640 step)
        from .. import _lib
        loss_t = loss_local if torch.is_tensor(loss_local) else torch.tensor([float(loss_local)], device=d_a_all.device)
        with torch.cuda.device(d_a_all.device):
            _lib.check(lib.nafp_pack_embedding_grads(_lib.ptr(d_a_all.contiguous()), _lib.ptr(d_b_all.contiguous()),
                                                     _lib.ptr(loss_t.reshape(-1).float().contiguous()), float(loss_scale), world,
                                                     n_anchors, d, _lib.ptr(send), _lib.current_stream()), 'pack_embedding_grads')
    else:                                           # (the CPU-side gloo harness of the tests)
        send[:, :n_anchors * d] = d_a_all.reshape(world, n_anchors * d)
        send[:, n_anchors * d:chunk] = d_b_all.reshape(world, n_anchors * d)
        send[:, chunk:] = loss_local * loss_scale
    recv = torch.empty((chunk + 4,), dtype=torch.float32, device=d_a_all.device)
    _reduce_scatter(dist, recv, send)
    return recv[chunk], recv[:chunk].view(2 * n_anchors, d)


def _reduce_scatter(dist, recv, send):
    """recv = sum over ranks of send[rank].  The path is chosen ONCE from the backend, never from an exception:
    RCCL ('nccl') runs `reduce_scatter_tensor` and every error it raises propagates (a rank that swallowed one would
    issue a different collective than its peers); gloo -- the CPU-side test harness, which has no tensor
    reduce-scatter -- sums the whole send buffer and keeps this rank's chunk."""
    if dist.get_backend() == 'nccl':
        dist.reduce_scatter_tensor(recv, send.view(-1))
        return
    full = send.clone()
    dist.all_reduce(full)
    recv.copy_(full[dist.get_rank()])


def val_step(X, m_pre, m_fp, loss_obj):
    """trainer.py:53-64 (no augmentation, no gradient)."""
    Xa, Xp = X
    n_anchors = len(Xa)
    X = torch.cat([torch.as_tensor(Xa), torch.as_tensor(Xp)], dim=0)
    m_fp.trainable = False
    emb = m_fp(m_pre(X))
    loss, sim_mtx, _ = loss_obj.compute_loss(emb[:n_anchors], emb[n_anchors:])
    return loss, sim_mtx


def test_step(X, m_pre, m_fp):
    """trainer.py:67-77: f(.), L2(f(.)), L2(g(f(.)))."""
    X = torch.cat([torch.as_tensor(x) for x in X], dim=0)
    m_fp.trainable = False
    emb_f = m_fp.front_conv(m_pre(X))
    emb_f_postL2 = m_fp.l2_normalize(emb_f)
    emb_gf = m_fp.l2_normalize(m_fp.div_enc(emb_f))
    return emb_f, emb_f_postL2, emb_gf


def make_optimizer(cfg, total_nsteps):
    """trainer.py:119-140."""
    sched = cfg['TRAIN']['LR_SCHEDULE'].upper()
    if sched == 'COS':
        lr = CosineDecay(float(cfg['TRAIN']['LR']), total_nsteps, alpha=1e-06)
    elif sched == 'COS-RESTART':
        # trainer.py:125-131.  The reference also passes `num_periods=0.5`, which tf.keras' CosineDecayRestarts
        # does not accept (its constructor raises TypeError there); the schedule below is the one its other
        # arguments describe (t_mul / m_mul at the keras defaults).
        lr = CosineDecayRestarts(float(cfg['TRAIN']['LR']), int(total_nsteps * 0.1), alpha=2e-06)
    else:
        lr = float(cfg['TRAIN']['LR'])
    name = cfg['TRAIN']['OPTIMIZER'].upper()
    if name == 'LAMB':
        return LAMB(learning_rate=lr)
    elif name == 'ADAM':
        return Adam(learning_rate=lr)
    raise NotImplementedError(cfg['TRAIN']['OPTIMIZER'])


def synthetic_batches(cfg, steps_per_epoch, device=None, snr_db=5.0):
    """SURVEY.md section 8(d) config 3/4 input: Xa = seeded noise segments, Xp = Xa + noise at
    5 dB SNR, float32 (n_local, 1, T) on the device; seeds differ per rank and per step."""
    td = _dist()
    world = td.get_world_size() if td is not None else 1
    rank = td.get_rank() if world > 1 else 0
    n_a = cfg['BSZ']['TR_N_ANCHOR']
    if n_a % world or cfg['BSZ']['TR_BATCH_SZ'] != 2 * n_a:
        raise ValueError('TR_BATCH_SZ must be 2*TR_N_ANCHOR and TR_N_ANCHOR divisible by the world size')
    n_l = n_a // world
    T = int(cfg['MODEL']['DUR'] * cfg['MODEL']['FS'])
    device = device or torch.device('cuda', torch.cuda.current_device())
    amp = 10.0 ** (-snr_db / 20.0)

    def epoch(ep):
        g = torch.Generator(device=device)
        for i in range(steps_per_epoch):
            g.manual_seed(1_000_003 * (100 + rank) + 7919 * ep + i)
            xa = 0.1 * torch.randn((n_l, 1, T), generator=g, device=device)
            xp = xa + 0.1 * amp * torch.randn((n_l, 1, T), generator=g, device=device)
            yield xa, xp
    return epoch


def setup(cfg, total_nsteps):
    """Models, optimizer, loss object and gradient bucket of one rank (trainer.py:113-176).
    TR_BATCH_SZ / TR_N_ANCHOR are the GLOBAL batch; each rank takes 1/world of it."""
    loss_mode = cfg['LOSS']['LOSS_MODE'].upper()
    if loss_mode not in ('NTXENT', 'ONLINE-TRIPLET'):
        raise NotImplementedError(cfg['LOSS']['LOSS_MODE'])
    m_pre, m_specaug, m_fp = build_fp(cfg)
    opt = make_optimizer(cfg, total_nsteps)
    dist = _dist()
    world = dist.get_world_size() if dist is not None else 1
    n_a = cfg['BSZ']['TR_N_ANCHOR'] // world
    if loss_mode == 'NTXENT':
        loss_obj = NTxentLoss(n_org=n_a, n_rep=(cfg['BSZ']['TR_BATCH_SZ'] - cfg['BSZ']['TR_N_ANCHOR']) // world,
                              tau=cfg['LOSS']['TAU'])
    else:                                           # trainer.py:159-164 (single device only: no cross-replica form exists)
        if world > 1:
            raise NotImplementedError('Online-Triplet loss with more than one rank')
        loss_obj = OnlineTripletLoss(bsz=cfg['BSZ']['TR_BATCH_SZ'], n_anchor=cfg['BSZ']['TR_N_ANCHOR'], mode='semi-hard',
                                     margin=cfg['LOSS']['MARGIN'])
    bucket = GradientBucket(m_fp)
    sync_replicas(m_fp)
    return m_pre, m_specaug, m_fp, opt, loss_obj, bucket


def sync_replicas(m_fp):
    """Replicas start from rank 0's variables."""
    dist = _dist()
    if dist is not None:
        for v in m_fp.trainable_variables:
            dist.broadcast(v, src=0)
        m_fp.mark_dirty()


def trainer(cfg, checkpoint_name, train_batches=None, steps_per_epoch=None, max_epoch=None):
    """trainer.py:111-230.  `train_batches`: optional callable epoch -> iterable of (Xa, Xp) CUDA (or host)
    batches of shape (n, 1, T); default = the reference's training set (cfg DIR / DATA_SEL / TD_AUG) through the
    device-side loader (all ranks share one permutation; rank r keeps rows r*n_a/world ... of each global batch)."""
    own_dataset = train_batches is None
    if train_batches is None:
        # the reference's loader (trainer.py:113, 181-197): anchors + augmented replicas, assembled on the device
        from .dataset import Dataset
        dist = _dist()
        world = dist.get_world_size() if dist is not None else 1
        rank = dist.get_rank() if dist is not None else 0
        # ONE permutation and one set of draws shared by all ranks (same seed); rank r takes its rows of each
        # global batch, so an epoch is n_samples / TR_N_ANCHOR steps whatever the world size
        ds = Dataset(cfg).get_train_ds(cfg['DATA_SEL']['REDUCE_ITEMS_P'], seed=1000, shard=(rank, world))
        if ds.n_pos_per_anchor != 1 and cfg['LOSS']['LOSS_MODE'].upper() == 'NTXENT':
            raise NotImplementedError('NT-Xent trains with one replica per anchor (TR_BATCH_SZ = 2 * TR_N_ANCHOR)')

        def train_batches(ep, ds=ds):
            # epoch `ep` (1-based) always sees the same permutation and draws, also after a restart: the loader
            # state is a function of (seed, epoch), rebuilt here rather than stored in the checkpoint
            ds.set_epoch(ep - 1)
            for i in range(len(ds)):
                yield ds[i]
        steps_per_epoch = steps_per_epoch or len(ds)
    max_epoch = max_epoch or cfg['TRAIN']['MAX_EPOCH']
    if steps_per_epoch is None:
        raise ValueError('steps_per_epoch is required when train_batches is supplied (the LR schedule needs the '
                         'total number of steps)')
    m_pre, m_specaug, m_fp, opt, loss_obj, bucket = setup(cfg, max_epoch * steps_per_epoch)
    ck_root = cfg['DIR']['LOG_ROOT_DIR'] + 'checkpoint/'
    start = 1
    # resume after the latest checkpoint (experiment_helper.py:125-136; the reference re-enters the epoch whose
    # number equals the latest checkpoint index, i.e. repeats one epoch on every restart -- not mirrored)
    try:
        start = _gen.load_checkpoint(ck_root, checkpoint_name, None, m_fp, optimizer=opt) + 1
    except FileNotFoundError:
        pass
    dist = _dist()
    sync_replicas(m_fp)
    # validation set (trainer.py:200-213): only rank 0 evaluates it, only if the directory exists
    val_ds, loss_obj_val = None, None
    if own_dataset and (dist is None or dist.get_rank() == 0):
        from .dataset import Dataset
        try:
            val_ds = Dataset(cfg).get_val_ds(max_song=250)
            if val_ds.n_samples == 0 or (val_ds.n_pos_per_anchor != 1 and cfg['LOSS']['LOSS_MODE'].upper() == 'NTXENT'):
                val_ds = None
        except (ValueError, IndexError):
            val_ds = None
        if val_ds is not None and cfg['LOSS']['LOSS_MODE'].upper() == 'NTXENT':
            loss_obj_val = NTxentLoss(n_org=cfg['BSZ']['VAL_N_ANCHOR'], n_rep=cfg['BSZ']['VAL_BATCH_SZ'] - cfg['BSZ']['VAL_N_ANCHOR'],
                                      tau=cfg['LOSS']['TAU'])
        elif val_ds is not None:                                    # trainer.py:165-169
            loss_obj_val = OnlineTripletLoss(bsz=cfg['BSZ']['VAL_BATCH_SZ'], n_anchor=cfg['BSZ']['VAL_N_ANCHOR'], mode='all', margin=0.)
    history = []
    for ep in range(start, max_epoch + 1):
        tot, n = None, 0
        for X in train_batches(ep):
            loss, _ = train_step(X, m_pre, m_specaug, m_fp, loss_obj, opt, bucket)
            tot = loss.detach().clone() if tot is None else tot + loss       # stays on the device: one read per epoch
            n += 1
        history.append(float(tot) / max(n, 1) if n else 0.0)
        msg = f'epoch {ep}: tr_loss:{history[-1]:.4f}'
        if val_ds is not None:
            vt, vn = 0.0, 0
            for i in range(len(val_ds)):
                Xv = val_ds[i]
                if len(Xv[0]) != cfg['BSZ']['VAL_N_ANCHOR']:
                    continue
                vloss, _ = val_step(Xv, m_pre, m_fp, loss_obj_val)
                vt += float(vloss); vn += 1
            if vn:
                msg += f', val_loss:{vt / vn:.4f}'
        print(msg)
        if val_ds is not None and cfg['TRAIN'].get('MINI_TEST_IN_TRAIN') and val_ds.n_pos_per_anchor == 1:
            from .utils.mini_search_subroutines import mini_search_validation          # trainer.py:226-230
            mini_search_validation(val_ds, m_pre, m_fp)
        if dist is None or dist.get_rank() == 0:
            _gen.save_checkpoint(ck_root, checkpoint_name, ep, m_fp, extra={'optimizer': opt.state_dict(m_fp.trainable_variables)})
            _gen.prune_checkpoints(ck_root, checkpoint_name, 3, cfg['TRAIN'].get('CHECKPOINT_KEEP_N_HOUR'))
    return history

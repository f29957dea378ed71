"""GPU parity: the HIP path (through the C ABI) vs the CPU oracle on the same seeded
inputs.  Floating point; tolerances are written next to each check.  The contract
(BASELINE.json north_star) is cosine(embedding, oracle) >= 1 - 1e-3 per segment."""
import numpy as np
import pytest
import torch

from oracle import melspec as o_mel, nnfp as o_nnfp

pytestmark = pytest.mark.gpu


def _audio(B, seed=0, T=8000):
    rng = np.random.default_rng(seed)
    t = np.arange(T) / 8000.0
    x = 0.1 * rng.normal(size=(B, 1, T))
    for b in range(B):
        f = rng.uniform(300, 3900, size=3)
        x[b, 0] += sum(0.2 * np.sin(2 * np.pi * fi * t + rng.uniform(0, 6.28)) for fi in f)
    return x.astype(np.float32)


def test_melspec_matches_oracle(nafp, cfg):
    x = _audio(7, seed=1)
    m_pre = nafp.get_melspec_layer(cfg)
    got = m_pre(torch.from_numpy(x).cuda()).cpu().numpy()
    want = o_mel.melspec_layer(x, dtype=np.float64)
    assert got.shape == want.shape == (7, 256, 32, 1)
    # log10 of fp32 FFT magnitudes: abs tolerance 2e-5 (values span [-1.3, 0])
    assert np.abs(got - want).max() < 2e-5
    assert got.max() == 0.0          # the group max is subtracted exactly


def test_melspec_groups_and_int16(nafp, cfg):
    x = _audio(10, seed=2)
    xi = np.clip(np.round(x * 32768), -32768, 32767).astype(np.int16)
    xf = (xi / 2 ** 15).astype(np.float32)          # audio_utils.py:245-246
    m_pre = nafp.get_melspec_layer(cfg)
    got_i = m_pre(torch.from_numpy(xi).cuda(), group_size=4).cpu().numpy()
    got_f = m_pre(torch.from_numpy(xf).cuda(), group_size=4).cpu().numpy()
    want = o_mel.melspec_layer(xf, group_size=4, dtype=np.float64)
    assert np.array_equal(got_i, got_f)              # int16 path == float path, bit for bit
    assert np.abs(got_f - want).max() < 2e-5
    # each group (4,4,2 segments) has its own zero maximum
    assert got_f[:4].max() == 0.0 and got_f[4:8].max() == 0.0 and got_f[8:].max() == 0.0


def test_melspec_maxnorm(nafp, cfg):
    import copy
    c = copy.deepcopy(cfg)
    c['MODEL']['FEAT'] = 'melspec_maxnorm'
    x = _audio(3, seed=3)
    got = nafp.get_melspec_layer(c)(torch.from_numpy(x).cuda()).cpu().numpy()
    want = o_mel.melspec_layer(x, segment_norm=True, dtype=np.float64)
    assert np.abs(got - want).max() < 1e-4


def _load_oracle_weights(m_fp, w):
    arrays = []
    for j in range(16):
        arrays += [w[f'conv{j}.kernel'], w[f'conv{j}.bias'], w[f'ln{j}.gamma'], w[f'ln{j}.beta']]
    arrays += [w['div.w1'], w['div.b1'], w['div.w2'], w['div.b2']]
    m_fp.set_weights(arrays)


@pytest.mark.parametrize('B', [1, 5])
def test_encoder_matches_oracle(nafp, cfg, B, observe, arith):
    rng = np.random.default_rng(10 + B)
    feat = (-rng.uniform(0, 1.2, size=(B, 256, 32, 1))).astype(np.float32)
    w = o_nnfp.init_weights(seed=3, randomize_affine=True)
    m_fp = nafp.get_fingerprinter(cfg)
    _load_oracle_weights(m_fp, w)
    f_t = torch.from_numpy(feat).cuda()
    flat = m_fp.front_conv(f_t).cpu().numpy()
    emb = m_fp(f_t).cpu().numpy()
    want_flat = o_nnfp.front_conv(feat, w, dtype=np.float64)
    want_emb = o_nnfp.fingerprinter(feat, w, dtype=np.float64)
    assert flat.shape == (B, 1024) and emb.shape == (B, 128)
    # fp32 through 16 conv+LN layers vs float64 oracle: abs 5e-5 on O(1) activations
    observe('|d flat|', np.abs(flat - want_flat).max(), 5e-5)
    observe('|d emb|', np.abs(emb - want_emb).max(), 5e-6)
    cos = (emb * want_emb).sum(1) / np.linalg.norm(emb, axis=1) / np.linalg.norm(want_emb, axis=1)
    observe('1 - cos', (1 - cos).max(), 1e-6)      # contract: 1e-3
    # div_enc alone (trainer.py:73-76 calls the halves separately), no L2
    de = m_fp.div_enc(torch.from_numpy(want_flat.astype(np.float32)).cuda()).cpu().numpy()
    assert np.abs(de - o_nnfp.div_enc(want_flat, w)).max() < 1e-5


def test_end_to_end_audio_to_fingerprint(nafp, cfg, arith):
    x = _audio(6, seed=5)
    w = o_nnfp.init_weights(seed=4, randomize_affine=True)
    m_pre, m_fp = nafp.get_melspec_layer(cfg), nafp.get_fingerprinter(cfg)
    _load_oracle_weights(m_fp, w)
    emb = m_fp(m_pre(torch.from_numpy(x).cuda())).cpu().numpy()       # generate.py:83-88
    want = o_nnfp.fingerprinter(o_mel.melspec_layer(x), w)
    cos = (emb * want).sum(1)
    assert (1 - cos).max() < 1e-5
    assert np.abs(np.linalg.norm(emb, axis=1) - 1).max() < 1e-5


def test_two_second_input_geometry(nafp, arith):
    """nnfp.py:266-268 builds FingerPrinter on (256,63,1) as well (19,224,576 params, nnfp.py:271):
    odd frame counts exercise symmetric SAME padding (1/1) and the ragged position tiles."""
    rng = np.random.default_rng(42)
    feat = (-rng.uniform(0, 1.2, size=(3, 256, 63, 1))).astype(np.float32)
    w = o_nnfp.init_weights(seed=6, input_shape=(256, 63, 1), randomize_affine=True)
    m_fp = nafp.FingerPrinter(input_shape=(256, 63, 1))
    assert sum(v.numel() for v in m_fp.trainable_variables) == 19224576
    _load_oracle_weights(m_fp, w)
    emb = m_fp(torch.from_numpy(feat).cuda()).cpu().numpy()
    want = o_nnfp.fingerprinter(feat, w, dtype=np.float64)
    assert np.abs(emb - want).max() < 2e-5
    assert (1 - (emb * want).sum(1)).max() < 1e-6


def test_melspec_two_second_segments(nafp, cfg):
    import copy
    c = copy.deepcopy(cfg)
    c['MODEL']['DUR'] = 2.
    x = _audio(3, seed=9, T=16000)
    got = nafp.get_melspec_layer(c)(torch.from_numpy(x).cuda()).cpu().numpy()
    want = o_mel.melspec_layer(x, dtype=np.float64)
    assert got.shape == want.shape == (3, 256, 63, 1)
    assert np.abs(got - want).max() < 2e-5


@pytest.mark.parametrize('feat_kind', ['melspec', 'melspec_maxnorm'])
def test_deferred_log_mel_tail_is_bit_identical(nafp, cfg, feat_kind):
    """`x - reduce_max(x)`, the clamp and the optional segment normalisation
    (melspectrogram.py:108-111) applied by conv0 as it loads (nafp_encoder_forward_raw) instead of by a separate pass
    over the feature tensor: same float operations in the same order -> the fingerprints must not move by one bit."""
    import copy
    c = copy.deepcopy(cfg); c['MODEL']['FEAT'] = feat_kind
    m_pre = nafp.get_melspec_layer(c)
    m_fp = nafp.FingerPrinter(seed=3)
    x = torch.from_numpy(_audio(253, seed=77)).cuda()
    xi = (x * 32768.0).clamp(-32768, 32767).to(torch.int16)
    for inp in (x, xi):
        for g in (None, 125, 7):
            ref_feat = m_pre(inp, group_size=g)
            d = m_pre(inp, group_size=g, defer=True)
            assert torch.equal(d.finish(), ref_feat)
            ref = m_fp(ref_feat)
            got = m_fp(d)
            # the per-sample LayerNorm statistics are double atomics whose order varies run to run: 1e-6, like
            # two runs of the same path (test_batch_independence_and_ragged_sizes)
            assert float((got - ref).abs().max()) < 1e-6
            flat_ref, flat = m_fp.front_conv(ref_feat), m_fp.front_conv(d)
            assert float((flat - flat_ref).abs().max()) < 1e-5


def test_experimental_bf16x3_option_stays_within_the_contract(nafp):
    """NAFP_OPT_BF16X3 (experimental, off by default): split-bf16 products on the unsplit GEMM convs.  Not the reference's
    arithmetic -- the fingerprints move, but by far less than the 1e-3 cosine contract; switching it off restores the f32
    path exactly."""
    rng = np.random.default_rng(12)
    feat = torch.from_numpy((-rng.uniform(0, 1.2, size=(640, 256, 32, 1))).astype(np.float32)).cuda()
    m_fp = nafp.FingerPrinter(seed=3)
    ref = m_fp(feat).clone()
    m_fp.set_option(3, 1)
    got = m_fp(feat).clone()
    m_fp.set_option(3, 0)
    back = m_fp(feat).clone()
    assert float((got - ref).abs().max()) > 0.0                      # it does change the arithmetic
    assert float((got - ref).abs().max()) < 1e-4
    assert float((1 - (got * ref).sum(1)).max()) < 1e-6
    assert float((back - ref).abs().max()) < 1e-6


def test_exact_split_option_is_float32_equivalent(nafp, observe):
    """NAFP_OPT_BF16X3 = 2 (experimental, off by default): the EXACT 3-way bf16 split x = h + m + l with the six products of weight
    >= 2^-16 and f32 accumulation -- float32-equivalent arithmetic on the bf16 matrix pipe.  Its error against the
    float64 oracle is recorded next to the f32 path's own on the same inputs; the two must be of the same size, and the two paths
    agree with each other far inside the f32 path's own error."""
    from oracle import nnfp as o_nnfp
    import _inputs
    rng = np.random.default_rng(13)
    B = 640
    feat = (-rng.uniform(0, 1.2, size=(B, 256, 32, 1))).astype(np.float32)
    w = _inputs.weights(seed=9)
    m_fp = nafp.FingerPrinter(seed=0)
    m_fp.set_weights(_inputs.weight_list(w))
    ft = torch.from_numpy(feat).cuda()
    ref = m_fp(ft).clone()
    m_fp.set_option(3, 2)
    got = m_fp(ft).clone()
    m_fp.set_option(3, 1)
    two_term = m_fp(ft).clone()
    m_fp.set_option(3, 0)
    assert torch.equal(m_fp(ft), ref)
    sel = np.linspace(0, B - 1, 6).astype(int)
    want = o_nnfp.fingerprinter(feat[sel], w)
    e_f32 = np.abs(ref.cpu().numpy()[sel] - want).max()
    e_x6 = np.abs(got.cpu().numpy()[sel] - want).max()
    e_x3 = np.abs(two_term.cpu().numpy()[sel] - want).max()
    observe('f32 MFMA path vs float64 oracle, fingerprint component', e_f32, 5e-6)
    observe('exact 3-way bf16 split (6 products) vs float64 oracle', e_x6, 5e-6)
    observe('hi / lo bf16 split (3 products) vs float64 oracle', e_x3, 1e-4)
    observe('exact split vs f32 path, all 640 rows', float((got - ref).abs().max()), 5e-6)
    assert e_x6 < 2.0 * e_f32 + 2e-7


def test_exact_split_option_across_streams_and_weight_updates(nafp):
    """The pre-split weights of NAFP_OPT_BF16X3 = 2 follow set_weights and are ordered for passes on other streams: switching the
    option on, then launching on four streams at once, then replacing the weights -- every result equals the single-stream one."""
    import _inputs
    rng = np.random.default_rng(3)
    feat = torch.from_numpy((-rng.uniform(0, 1.2, size=(130, 256, 32, 1))).astype(np.float32)).cuda()
    m_fp = nafp.FingerPrinter(seed=0)
    m_fp.set_weights(_inputs.weight_list(_inputs.weights(seed=5)))
    m_fp.set_option(3, 2)
    streams = [torch.cuda.Stream() for _ in range(4)]
    outs = []
    torch.cuda.synchronize()
    for s in streams:                       # the first launch after the switch splits the weights; the others must wait for it
        with torch.cuda.stream(s):
            outs.append(m_fp(feat))
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    ref = m_fp(feat).clone()
    assert torch.equal(ref, outs[0])
    m_fp.set_weights(_inputs.weight_list(_inputs.weights(seed=6)))      # new parameters: new split, on whatever stream comes first
    with torch.cuda.stream(streams[2]):
        a = m_fp(feat)
    with torch.cuda.stream(streams[1]):
        b = m_fp(feat)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and float((a - ref).abs().max()) > 1e-3
    # like the f32 path, the exact-split path gives a segment the same bytes whatever shares its launch
    assert torch.equal(m_fp(feat[:37]), a[:37]) and torch.equal(m_fp(feat[60:61]), a[60:61])
    m_fp.set_option(3, 0)
    f32 = m_fp(feat)
    assert float((a - f32).abs().max()) < 5e-6


@pytest.mark.parametrize('seed', [21, 22])
def test_exact_split_error_statistics(nafp, observe, seed):
    """More of the same evidence: 10 rows per weight set, keras-default initialisation (seed 21) and perturbed affine terms (22): the
    exact split's maximum AND root-mean-square error against the float64 oracle are no larger than the fp32 MFMA path's (x 1.25)."""
    from oracle import nnfp as o_nnfp
    rng = np.random.default_rng(seed)
    B = 130
    feat = (-rng.uniform(0, 1.2, size=(B, 256, 32, 1))).astype(np.float32)
    w = o_nnfp.init_weights(seed=seed, randomize_affine=(seed % 2 == 0))
    m_fp = nafp.FingerPrinter(seed=0)
    m_fp.set_weights(__import__('_inputs').weight_list(w))
    ft = torch.from_numpy(feat).cuda()
    ref = m_fp(ft).cpu().numpy()
    m_fp.set_option(3, 2)
    got = m_fp(ft).cpu().numpy()
    m_fp.set_option(3, 0)
    sel = np.linspace(0, B - 1, 10).astype(int)
    want = o_nnfp.fingerprinter(feat[sel], w)
    e32, e6 = ref[sel] - want, got[sel] - want
    observe(f'seed {seed}: f32 path max |err|', np.abs(e32).max(), 5e-6)
    observe(f'seed {seed}: exact split max |err|', np.abs(e6).max(), 5e-6)
    observe(f'seed {seed}: f32 path rms err', float(np.sqrt((e32 ** 2).mean())), 1e-6)
    observe(f'seed {seed}: exact split rms err', float(np.sqrt((e6 ** 2).mean())), 1e-6)
    assert np.abs(e6).max() <= 1.25 * np.abs(e32).max() + 1e-7
    assert np.sqrt((e6 ** 2).mean()) <= 1.25 * np.sqrt((e32 ** 2).mean()) + 2e-8


@pytest.mark.parametrize('norm', ['layer_norm2d', 'layer_norm1d', 'batch_norm'])
@pytest.mark.parametrize('shape', [(256, 32, 1), (256, 63, 1)])
def test_exact_split_on_ragged_launches_geometries_and_norms(nafp, shape, norm, observe):
    """NAFP_OPT_BF16X3 = 2 away from the bench shape: B = 1 / 9 / 257 (ragged sample groups and position tiles), the 63-frame input, every
    MODEL.BN -- always within the fp32 path's own error of the fp32 path."""
    from oracle import nnfp as o_nnfp
    import _inputs
    rng = np.random.default_rng(7)
    w = o_nnfp.init_weights(seed=8, input_shape=shape, randomize_affine=True)
    if norm != 'layer_norm2d':
        w = o_nnfp.convert_norm(w, norm, seed=4)
    m_fp = nafp.FingerPrinter(input_shape=shape, norm=norm)
    m_fp.set_weights(_inputs.weight_list(w))
    worst = 0.0
    for B in (1, 9, 257):
        feat = torch.from_numpy((-rng.uniform(0, 1.2, size=(B,) + shape)).astype(np.float32)).cuda()
        ref = m_fp(feat).clone()
        m_fp.set_option(3, 2)
        got = m_fp(feat).clone()
        m_fp.set_option(3, 0)
        assert bool(torch.isfinite(got).all())
        worst = max(worst, float((got - ref).abs().max()))
    observe(f'exact split vs f32 path, {shape[1]} frames, {norm}', worst, 5e-6)

"""GPU: the exact-split forward (NAFP_OPT_BF16X3 = 2) next to the library's OWN kernels on other streams.

Round 6 found that on gfx950 a packed-f32 vector instruction with an op_sel modifier comes out wrong in a wave that shares a compute
unit with waves issuing 128-bit-operand matrix instructions next to vector work (tools/probes/pk_opsel_hazard_probe.hip; include/nafp.h;
profiles/r06_experiments.md section 5).  Before the fix these arrangements failed in 14 - 17 of 20 launches (front end next to the
split kernels) and in 60 of 120 (f32 64-column GEMM kernels next to them); the library is now built without packed-f32 instructions
(tests/test_abi.py holds that on the disassembly) and every result must be BIT-IDENTICAL to the result of the same input computed
alone."""
import numpy as np
import pytest
import torch

import _inputs

pytestmark = pytest.mark.gpu


def _models(nafp, cfg):
    m_pre = nafp.get_melspec_layer(cfg)
    w = _inputs.weight_list(_inputs.weights(seed=5))
    m6, m32 = nafp.get_fingerprinter(cfg), nafp.get_fingerprinter(cfg)
    m6.set_weights(w)
    m32.set_weights(w)
    m6.set_option(3, 2)
    m32.set_option(3, 0)
    return m_pre, m6, m32


@pytest.mark.parametrize('rows', [125, 640])
def test_front_end_and_split_forward_pipelined_on_four_streams(nafp, cfg, rows):
    """What the generate driver does: launch k on stream k % 4, front end (deferred form) and forward back to back -- the front end of
    one launch runs next to the split GEMM kernels of the previous ones."""
    m_pre, m6, _ = _models(nafp, cfg)
    g = torch.Generator(device='cuda').manual_seed(3)
    n_l = 20 if rows == 125 else 8
    xs = [0.1 * torch.randn((rows, 1, 8000), generator=g, device='cuda') for _ in range(n_l)]
    ref_f = [m_pre(x, group_size=125, defer=True) for x in xs]
    ref_raw = [f.raw.clone() for f in ref_f]
    refs = [m6(f).clone() for f in ref_f]
    torch.cuda.synchronize()
    for rep in range(4):
        streams = [torch.cuda.Stream() for _ in range(4)]
        outs, raws = [], []
        for i in range(n_l):
            with torch.cuda.stream(streams[i % 4]):
                f = m_pre(xs[i], group_size=125, defer=True)
                raws.append(f.raw)
                outs.append(m6(f))
        torch.cuda.synchronize()
        bad_f = [i for i in range(n_l) if not torch.equal(raws[i], ref_raw[i])]
        bad_e = [i for i in range(n_l) if not torch.equal(outs[i], refs[i])]
        assert not bad_f, f'rep {rep}: the front end of launches {bad_f} differs from its solo result'
        assert not bad_e, f'rep {rep}: the fingerprints of launches {bad_e} differ from their solo result'


def test_f32_forward_next_to_split_forward(nafp, cfg):
    """Two handles, one per arithmetic, each on its own stream: the f32 GEMM kernels (their 64-column tile's epilogue held packed-f32
    op_sel instructions) run next to the split kernels of the other handle -- both come out as they do alone."""
    m_pre, m6, m32 = _models(nafp, cfg)
    g = torch.Generator(device='cuda').manual_seed(4)
    feats = [m_pre(0.1 * torch.randn((250, 1, 8000), generator=g, device='cuda'), group_size=125) for _ in range(6)]
    ref6 = [m6(f).clone() for f in feats]
    ref32 = [m32(f).clone() for f in feats]
    torch.cuda.synchronize()
    s6, s32 = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(5):
        o6, o32 = [], []
        for i in range(len(feats)):
            with torch.cuda.stream(s6):
                o6.append(m6(feats[i]))
            with torch.cuda.stream(s32):
                o32.append(m32(feats[(i + 3) % len(feats)]))
        torch.cuda.synchronize()
        assert all(torch.equal(o6[i], ref6[i]) for i in range(len(feats))), f'rep {rep}: split-arithmetic results moved'
        assert all(torch.equal(o32[i], ref32[(i + 3) % len(feats)]) for i in range(len(feats))), f'rep {rep}: f32 results next to the split kernels moved'


def test_train_step_next_to_split_forward(nafp, cfg):
    """A whole f32 train pass (forward_train + backward: LayerNorm backward, weight gradients, transposed convs) on one stream while a
    split forward of another handle runs on a second one: the embeddings bit-identical to the pass alone, the gradients within the
    noise of their float atomics (1e-5 of each tensor's largest entry)."""
    m_pre, m6, m32 = _models(nafp, cfg)
    g = torch.Generator(device='cuda').manual_seed(6)
    feat = m_pre(0.1 * torch.randn((128, 1, 8000), generator=g, device='cuda'), group_size=128)
    d_emb = torch.randn((128, m32.emb_sz), generator=g, device='cuda') * 1e-2
    big = m_pre(0.1 * torch.randn((640, 1, 8000), generator=g, device='cuda'), group_size=128)

    def train_pass():
        emb = m32.forward_train(feat)
        grads = m32.backward(d_emb)
        return emb.clone(), [t.clone() for t in grads]
    emb0, g0 = train_pass()
    torch.cuda.synchronize()
    s_t, s_f = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(4):
        with torch.cuda.stream(s_f):
            for _ in range(4):
                m6(big)
        with torch.cuda.stream(s_t):
            emb1, g1 = train_pass()
        torch.cuda.synchronize()
        assert torch.equal(emb1, emb0), f'rep {rep}: forward_train moved'
        # the gradients meet in float atomics (dgamma / dbeta / dbias, the weight-gradient chunks): their order follows the timing, so
        # the last bits may move with ANY neighbour; the effect this file guards against moved single values by 1e-3 .. 1e-2 relative
        for k in range(len(g0)):
            scale = float(g0[k].abs().max()) + 1e-30
            assert float((g1[k] - g0[k]).abs().max()) <= 1e-5 * scale, (rep, k, float((g1[k] - g0[k]).abs().max()) / scale)

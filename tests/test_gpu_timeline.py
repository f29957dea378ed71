"""The diagnostic phase timeline of the GEMM conv (include/nafp.h: nafp_conv_timeline): every wave of every tile of the
selected conv stamps the shader clock at its phase boundaries; the stamps must be there, ordered, and carry plausible
hardware ids -- and recording must not change the result."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_conv_timeline_stamps_are_complete_and_ordered(nafp):
    from neural_audio_fp_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(3)
    feat = torch.from_numpy((-rng.uniform(0, 1.2, size=(64, 256, 32, 1))).astype(np.float32)).cuda()
    m_fp = nafp.FingerPrinter(seed=5)
    ref = m_fp(feat)
    cap = 1 << 22
    buf = torch.zeros(cap, dtype=torch.int64, device='cuda')
    # conv1 of the encoder: 128 -> 128 channels, 128 x 16 = 2048 output positions per segment
    assert lib.nafp_conv_timeline(ctypes.c_void_p(buf.data_ptr()), cap, 128, 128, 2048) == 0
    got = m_fp(feat)
    torch.cuda.synchronize()
    assert lib.nafp_conv_timeline(None, 0, 0, 0, 0) == 0
    g = (ctypes.c_int * 5)()
    assert lib.nafp_conv_timeline_grid(g) == 0
    gx, gy, gz, bm, bn = list(g)
    assert bm in (128, 256) and bn in (64, 128) and gx * gy * gz > 0
    n_wg, nw = gx * gy * gz, bm // 32
    t = buf[:n_wg * 64].cpu().numpy().reshape(n_wg, 8, 8)[:, :nw, :]
    assert (t[..., 1:] > 0).all()                                   # every wave of every workgroup stamped every phase
    assert (np.diff(t[..., 1:], axis=-1) >= 0).all()                # phase boundaries in program order
    simd = (t[..., 0] >> 4) & 3
    assert set(np.unique(simd)) <= {0, 1, 2, 3}
    assert len(np.unique((t[..., 0] >> 32) & 0xf)) <= 8             # XCC ids
    assert torch.equal(got, ref) or float((got - ref).abs().max()) < 1e-6
    # switched off again: the buffer stays untouched
    buf.zero_()
    m_fp(feat)
    torch.cuda.synchronize()
    assert int(buf.abs().sum()) == 0

"""CPU: the melspec oracle against independent formulations and known facts
(melspectrogram.py:25-112)."""
import numpy as np
import pytest
import torch

from oracle import melspec as o_mel, torch_ref
import _inputs


def test_mel_bank_known_facts():
    fb = o_mel.mel_filterbank()                       # args of melspectrogram.py:44-50 + default.yaml:39-46
    assert fb.shape == (256, 513) and fb.dtype == np.float32
    assert int((fb != 0).sum()) == 941                # 0.72 % dense
    taps = (fb != 0).sum(1)
    assert taps.min() == 2 and taps.max() == 8        # never an empty filter
    active = np.nonzero(fb.sum(0))[0]
    assert active[0] == 39 and active[-1] == 511
    assert np.isclose(fb.max(), 0.12558146, atol=1e-8)
    # band edges start at 300 Hz with 7.954 Hz spacing in the linear region
    edges = o_mel.mel_to_hz(np.linspace(o_mel.hz_to_mel(300.), o_mel.hz_to_mel(4000.), 258))
    assert np.allclose(edges[:4], [300.0, 307.954, 315.909, 323.863], atol=2e-3)
    assert np.isclose(edges[-1], 4000.0)


def test_mel_scale_roundtrip_and_break():
    f = np.array([0., 200., 999.9, 1000., 1000.1, 2500., 4000.])
    assert np.allclose(o_mel.mel_to_hz(o_mel.hz_to_mel(f)), f, atol=1e-9)
    assert np.isclose(o_mel.hz_to_mel(1000.), 15.0)   # Slaney: 1 kHz = 15 mel


def test_hann_is_periodic():
    w = o_mel.hann_periodic(1024)
    assert w[0] == 0 and np.isclose(w[512], 1.0) and not np.isclose(w[-1], 0.0)
    assert np.allclose(w, torch.hann_window(1024, periodic=True, dtype=torch.float64).numpy())


def test_stft_against_torch_and_frame_count():
    x = _inputs.audio(3, seed=5)[:, 0]
    xp = np.pad(x, ((0, 0), (512, 512)))
    mag = o_mel.stft_magnitude(xp)                    # (3,32,513)
    assert mag.shape == (3, 32, 513)                  # 1 + (9024-1024)//256 = 32 frames (nnfp.py:248)
    ref = torch.stft(torch.from_numpy(xp).double(), 1024, 256, 1024,
                     torch.hann_window(1024, periodic=True, dtype=torch.float64), center=False,
                     return_complex=True).abs().numpy().transpose(0, 2, 1)
    assert np.abs(mag - ref).max() < 1e-10


def test_melspec_layer_vs_torch_formulation():
    x = _inputs.audio(5, seed=6)
    a = o_mel.melspec_layer(x)
    b = torch_ref.melspec_layer(x).numpy()
    assert a.shape == (5, 256, 32, 1)
    assert np.abs(a - b).max() < 5e-6
    assert a.max() == 0.0 and a.min() >= np.log10(0.06) - a.max() - 2   # clamp at -80 never active
    a32 = o_mel.melspec_layer(x, dtype=np.float32)
    assert np.abs(a - a32).max() < 5e-5


def test_melspec_group_semantics_and_maxnorm():
    x = _inputs.audio(6, seed=7)
    x[3:] *= 0.01                                      # quieter second half -> different maxima
    whole = o_mel.melspec_layer(x)
    g3 = o_mel.melspec_layer(x, group_size=3)
    assert np.allclose(g3[:3], o_mel.melspec_layer(x[:3]))     # a group == the reference's batch
    assert np.allclose(g3[3:], o_mel.melspec_layer(x[3:]))
    assert not np.allclose(g3[3:], whole[3:])
    assert g3[3:].max() == 0.0
    mn = o_mel.melspec_layer(x, segment_norm=True)
    # (x - min/2)/|min/2| maps [min, 0] onto [-1, 1]   (melspectrogram.py:110-111)
    assert np.isclose(mn.min(), -1.0, atol=1e-6) and np.isclose(mn.max(), 1.0, atol=1e-6)
    assert np.abs(mn - torch_ref.melspec_layer(x, segment_norm=True).numpy()).max() < 2e-5


def test_golden_mel(golden):
    x = _inputs.audio(4, seed=11)
    assert np.abs(o_mel.melspec_layer(x) - golden['mel_seed11']).max() < 1e-6
    assert np.abs(o_mel.melspec_layer(x, group_size=2) - golden['mel_seed11_group2']).max() < 1e-6
    fb = o_mel.mel_filterbank()
    assert int(golden['melbank_nnz'][0]) == 941
    assert np.array_equal(fb.sum(1), golden['melbank_rowsum'])


def test_mel_filterbank_matches_independent_slaney_implementation():
    """Pin against a published third-party implementation present in this image:
    transformers.audio_utils.mel_filter_bank(norm='slaney', mel_scale='slaney') is written to
    reproduce librosa.filters.mel, which is what kapre's ApplyFilterbank calls
    (melspectrogram.py:93-98).  Same 941 non-zero taps, values equal to float32 rounding."""
    tau = pytest.importorskip('transformers.audio_utils')
    fb = tau.mel_filter_bank(num_frequency_bins=513, num_mel_filters=256, min_frequency=300.0, max_frequency=4000.0,
                             sampling_rate=8000, norm='slaney', mel_scale='slaney')
    ours = o_mel.mel_filterbank()
    assert fb.shape == (513, 256) and ours.shape == (256, 513)
    assert np.array_equal(fb.T > 0, ours > 0) and int((ours > 0).sum()) == 941
    assert np.abs(fb.T - ours).max() < 2e-8


def test_logmel_matches_independent_spectrogram_implementation():
    """The whole front end (framing of the 512+512 padded segment, periodic Hann, |rfft|, Slaney mel,
    +0.06, log10, batch-max subtraction: melspectrogram.py:82-112) against transformers.audio_utils
    (window_function / spectrogram / mel_filter_bank), an implementation this repo did not write."""
    tau = pytest.importorskip('transformers.audio_utils')
    x = _inputs.audio(3, seed=4)
    fb = tau.mel_filter_bank(513, 256, 300.0, 4000.0, 8000, norm='slaney', mel_scale='slaney')
    win = tau.window_function(1024, 'hann', periodic=True)
    s = np.stack([tau.spectrogram(np.pad(x[b, 0].astype(np.float64), (512, 512)), win, frame_length=1024, hop_length=256,
                                  fft_length=1024, power=1.0, center=False, mel_filters=fb, mel_floor=0.0)
                  for b in range(3)])
    y = np.log10(np.maximum(s + 0.06, 1e-10))
    y = y - y.max()
    ours = o_mel.melspec_layer(x)[..., 0]
    assert ours.shape == y.shape == (3, 256, 32)
    assert np.abs(ours - y).max() < 1e-6

"""CPU: the C-ABI library loads and exports every symbol include/nafp.h declares
(no compute calls: there is no GPU here), and its host-side entry points agree with
the oracle."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import melspec as o_mel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, 'include', 'nafp.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(nafp_[a-z0-9_]+)\s*\(', txt)))


def test_header_symbols_are_bound_and_exported(nafp):
    lib = nafp._lib.load()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert n in nafp._lib.PROTOTYPES, f'{n} declared in nafp.h but not bound in _lib.py'
        assert getattr(lib, n) is not None
    for n in nafp._lib.PROTOTYPES:
        assert n in names, f'{n} bound but not declared in nafp.h'
    assert lib.nafp_abi_version() == 1


def test_status_strings(nafp):
    lib = nafp._lib.load()
    assert lib.nafp_status_string(0) == b'ok'
    assert b'invalid' in lib.nafp_status_string(1)
    assert b'unknown' in lib.nafp_status_string(99)


def test_host_mel_bank_bit_exact_with_oracle(nafp):
    lib = nafp._lib.load()
    for (fs, n_mels, fmin, fmax) in [(8000, 256, 300., 4000.), (8000, 128, 0., 4000.), (16000, 64, 50., 7000.)]:
        out = np.zeros((n_mels, 513), np.float32)
        assert lib.nafp_mel_filterbank_host(fs, 1024, n_mels, fmin, fmax, out.ctypes.data_as(ctypes.c_void_p)) == 0
        assert np.array_equal(out, o_mel.mel_filterbank(fs, 1024, n_mels, fmin, fmax))


def test_argument_checking_without_gpu(nafp):
    lib = nafp._lib.load()
    assert lib.nafp_mel_filterbank_host(8000, 1024, 256, 300., 4000., None) == 1          # INVALID_ARG
    assert lib.nafp_mel_filterbank_host(8000, 1024, 256, 4000., 300., None) == 1
    assert lib.nafp_melspec_n_frames(None) == -1
    assert lib.nafp_encoder_n_tensors(None) == -1
    assert lib.nafp_encoder_workspace_bytes(None, 4) == -1
    assert lib.nafp_ntxent_workspace_bytes(60, 60) > 0
    h = ctypes.c_void_p()
    assert lib.nafp_melspec_create(ctypes.byref(h), 8000, 8000, 512, 256, 256, 300., 4000.) == 2   # UNSUPPORTED n_fft
    assert lib.nafp_melspec_create(ctypes.byref(h), 8000, 8000, 1024, 256, 100, 300., 4000.) == 2  # n_mels % 64
    assert lib.nafp_encoder_create(None, 256, 32, 128) == 1
    # entry points added with the training / ingest / search rows: argument validation precedes any HIP call
    assert lib.nafp_melspec_forward_windows_i16(None, None, None, None, 4, 0, 0, None, None, None) == 1
    assert lib.nafp_search_index_aux_floats(1000) == 1088 and lib.nafp_search_index_aux_floats(-1) == -1
    assert lib.nafp_search_index_prepare(None, 10, 128, None, None) == 1
    assert lib.nafp_search_workspace_bytes(10, 1000, 20) > 0
    assert lib.nafp_search_workspace_bytes(10, 1000, 33) == -1 and lib.nafp_search_workspace_bytes(10, 0, 20) == -1
    assert lib.nafp_search_topk_l2(None, 1, None, None, 10, 128, 20, None, None, None, 0, None) == 1
    assert lib.nafp_search_seq_scores(None, None, 10, 128, None, None, 1, None, 4, None, None) == 1
    assert lib.nafp_augment_rows(None, None, 4, 8000, None, None) == 1
    assert lib.nafp_triplet_workspace_bytes(64, 256) > 0 and lib.nafp_triplet_workspace_bytes(0, 4) == -1
    assert lib.nafp_triplet_forward(None, None, 4, 8, 128, 0, 0.5, None, None, None, None, None, 0, None) == 1
    assert lib.nafp_encoder_train_workspace_bytes(None, 4) == -1
    assert lib.nafp_encoder_backward(None, None, None, 4, None, 0, None, 1, None) == 1
    assert lib.nafp_adam_step(None, 0, 1e-3, 0.9, 0.999, 1e-7, 1, None) in (0, 1)
    assert abs(lib.nafp_cosine_decay_lr_host(1e-4, 0, 100, 1e-6) - 1e-4) < 1e-11          # float32 return
    assert abs(lib.nafp_cosine_decay_lr_host(1e-4, 100, 100, 1e-6) - 1e-10) < 1e-15


def test_missing_library_fails_loudly(nafp, monkeypatch):
    monkeypatch.setattr(nafp._lib, '_lib', None)
    monkeypatch.setattr(nafp._lib, 'LIB_PATH', '/nonexistent/libnafp.so')
    with pytest.raises(nafp._lib.NafpError, match='no CPU fallback'):
        nafp._lib.load()


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'neural-audio-fp_amd')
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), os.path.join(d, f)
    for f in ('run.py',):
        assert 'oracle' not in open(os.path.join(ROOT, f)).read()

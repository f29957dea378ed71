"""CPU: oracle and host mirror against golden vectors PRODUCED BY THE REFERENCE'S OWN CODE
(tests/golden/audio_utils_v1.npz, written by tests/gen_golden_audio_utils.py from
/root/reference/model/utils/audio_utils.py, which is pure numpy and runs without TensorFlow).
This pins the input contract (segment enumeration, load_audio) and the time-domain augmentation
arithmetic (max_normalize, background_mix, bg_mix_batch, ir_aug_batch) to the reference itself."""
import os
import wave

import numpy as np
import pytest

from oracle import augment as A, segments as S

G = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'audio_utils_v1.npz')))
FS, DUR, HOP = int(G['fs']), float(G['dur']), float(G['hop'])
T = int(FS * DUR)


@pytest.fixture(scope='module')
def wavs(tmp_path_factory):
    d = tmp_path_factory.mktemp('gold')
    fns = []
    for i in range(len(G['wav_lens'])):
        p = str(d / f'{i}.wav')
        with wave.open(p, 'w') as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(FS)
            w.writeframes(G[f'pcm{i}'].astype('<i2').tobytes())
        fns.append(p)
    return fns


def test_segment_enumeration_matches_reference(wavs):
    from neural_audio_fp_amd.model.utils.audio_utils import get_fns_seg_list
    from neural_audio_fp_amd.model.utils.dataloader_keras import segment_table
    want = G['seglist_all_hop']
    got = S.enumerate_segments(wavs, FS, DUR, HOP)
    assert [(wavs.index(f), s) for f, s in got] == [(int(r[0]), int(r[1])) for r in want]
    assert [[wavs.index(f), s] for f, s in get_fns_seg_list(wavs, FS, DUR, HOP)] == [[int(r[0]), int(r[1])] for r in want]
    for mode, key, hp in (('all', 'seglist_all_hop', HOP), ('all', 'seglist_all_nohop', DUR), ('first', 'seglist_first_nohop', DUR)):
        rows = []
        for f, n in enumerate(G['wav_lens']):
            rows += [[f, s, lo, hi] for s, lo, hi in segment_table(int(n), FS, DUR, hp, mode)]
        assert np.array_equal(np.asarray(rows), G[key]), key
        if mode == 'all':
            o = [[f, s, lo, hi] for f, n in enumerate(G['wav_lens']) for s, lo, hi in A.segment_offsets(int(n), FS, DUR, hp)]
            assert np.array_equal(np.asarray(o), G[key]), key


def test_load_audio_matches_reference(wavs):
    for (f, st, off), want in zip(G['load_cases'], G['load_out']):
        start = int(np.floor((st + off) * FS))
        got = A.window(G[f'pcm{int(f)}'], start, T)
        assert np.array_equal(got, want)
    # the whole-segment loader of the generate path is the offset-free case
    for f in (5, 6):
        for s in range(S.n_segments(int(G['wav_lens'][f]), FS, DUR, HOP)):
            assert np.array_equal(S.load_segment(wavs[f], s, FS, DUR, HOP), A.window(G[f'pcm{f}'], int(s * HOP * FS), T))


def test_augmentation_arithmetic_matches_reference():
    ev, bg = G['ev'], G['bg']
    assert np.array_equal(np.stack([A.max_normalize(ev[0]), A.max_normalize(ev[4])]), G['max_normalize'])
    assert np.abs(A.background_mix(ev[1], bg[1], 7.5) - G['background_mix']).max() < 1e-15
    got = A.bg_mix_rows(ev, bg, G['snrs'], G['amps'])
    assert np.abs(got - G['bg_mix_batch']).max() < 1e-15          # incl. the silent-event and silent-background rows
    irs = [r[:np.max(np.nonzero(r)[0]) + 1] if r.any() else r[:20] for r in G['ir']]
    for impl in (A.ir_aug_rows, A.ir_aug_rows_direct):
        assert np.abs(impl(ev, irs) - G['ir_aug_batch']).max() < 1e-12
    assert np.abs(A.ir_aug_rows(ev, list(G['ir'])) - G['ir_aug_batch']).max() < 1e-12      # zero-padded taps change nothing


def test_host_windows_equal_reference_load_audio(wavs):
    """the arena + window table of the device-side loader reproduce the reference's load_audio windows"""
    from neural_audio_fp_amd.model.utils.dataloader_keras import PcmStore, PcmArena
    st = PcmStore(wavs, FS)
    arena = PcmArena([st]).host()
    for (f, s, off), want in zip(G['load_cases'], G['load_out']):
        f = int(f)
        start = int(np.floor((s + off) * FS))
        valid = int(np.clip(st.n_frames[f] - start, 0, T))
        got = np.zeros(T)
        got[:valid] = arena[st.start[f] + start: st.start[f] + start + valid] / 2 ** 15
        assert np.array_equal(got, want)

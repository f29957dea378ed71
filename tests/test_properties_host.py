"""CPU: property tests (hypothesis) of the host logic the multi-rank generate path and the window ingest rest on."""
import wave

import numpy as np
from hypothesis import given, settings, strategies as st


@settings(max_examples=300, deadline=None)
@given(n=st.integers(0, 200000), group=st.integers(1, 700), world=st.integers(1, 16))
def test_shard_rows_partitions_every_row_exactly_once_on_group_boundaries(n, group, world):
    from neural_audio_fp_amd.model.generate import shard_rows
    rs = [shard_rows(n, group, r, world) for r in range(world)]
    assert rs[0][0] == 0 and rs[-1][1] == n
    for (a0, a1), (b0, b1) in zip(rs, rs[1:]):
        assert a1 == b0 and a0 <= a1
    for a0, a1 in rs:
        assert a0 % group == 0 or a0 == n                      # a rank starts on a max-normalisation group boundary
    sizes = [(a1 - a0 + group - 1) // group for a0, a1 in rs]
    assert max(sizes) - min(sizes) <= 1                        # balanced to within one group


@settings(max_examples=25, deadline=None)
@given(lens=st.lists(st.integers(0, 30000), min_size=1, max_size=5), chunk=st.integers(1, 40), seed=st.integers(0, 99))
def test_windows_equal_rows_for_arbitrary_file_lengths(tmp_path_factory, lens, chunk, seed):
    from neural_audio_fp_amd.model.utils.audio_utils import SegmentSource
    d = tmp_path_factory.mktemp('w')
    rng = np.random.default_rng(seed)
    paths = []
    for i, n in enumerate(lens):
        p = str(d / f'{i}.wav')
        with wave.open(p, 'w') as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(8000)
            w.writeframes(rng.integers(-30000, 30000, size=n).astype('<i2').tobytes())
        paths.append(p)
    src = SegmentSource(paths, bsz=7)
    want = np.concatenate([c for _, c in src.iter_rows(0, src.n_samples, 11)])[:, 0]
    got = np.zeros_like(want)
    for start, n, arena, used, off, valid in src.iter_windows(0, src.n_samples, chunk):
        for i in range(n):
            got[start + i, :valid[i]] = arena[off[i]:off[i] + valid[i]]
    assert np.array_equal(got, want)


@settings(max_examples=200, deadline=None)
@given(n_frames=st.integers(0, 400000), fs=st.sampled_from([8000, 16000]), hop=st.sampled_from([0.5, 1.0, 0.25]))
def test_segment_table_is_consistent(n_frames, fs, hop):
    from neural_audio_fp_amd.model.utils.dataloader_keras import segment_table
    from neural_audio_fp_amd.model.utils.audio_utils import n_segments
    tab = segment_table(n_frames, fs, 1.0, hop)
    assert len(tab) == n_segments(n_frames, fs, 1.0, hop) >= 1
    assert tab[0][1] == 0 and all(lo <= 0 <= hi for _, lo, hi in tab)
    last_start = tab[-1][0] * hop * fs
    assert last_start + tab[-1][2] + fs <= max(n_frames, fs) + 1e-9      # the latest offset still ends inside the file

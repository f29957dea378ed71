"""Generates tests/golden/audio_utils_v1.npz by RUNNING THE REFERENCE's own code: model/utils/audio_utils.py
of /root/reference is pure numpy + stdlib `wave` (no TensorFlow), so it imports in this container.  The
fixture holds inputs and the reference's outputs only (no reference source); tests/test_golden_audio_utils.py
checks the oracle (oracle/segments.py, oracle/augment.py) and the host mirror against it.

    python tests/gen_golden_audio_utils.py        # needs /root/reference; not needed to run the tests
"""
import importlib.util
import os
import tempfile
import wave

import numpy as np

REF = '/root/reference/model/utils/audio_utils.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'audio_utils_v1.npz')


def main():
    spec = importlib.util.spec_from_file_location('ref_audio_utils', REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    rng = np.random.default_rng(20260101)
    fs, dur, hop = 1000, 1.0, 0.5                         # the functions are generic in fs: small fixtures
    T = int(fs * dur)
    out = {'fs': fs, 'dur': dur, 'hop': hop}
    # ---- WAV files: enumeration and segment loading -------------------------------------------
    lens = [350, 1000, 1001, 1499, 1500, 2750, 5003]
    out['wav_lens'] = np.asarray(lens)
    with tempfile.TemporaryDirectory() as d:
        fns = []
        for i, n in enumerate(lens):
            pcm = rng.integers(-20000, 20000, size=n).astype('<i2')
            out[f'pcm{i}'] = pcm
            p = os.path.join(d, f'{i}.wav')
            with wave.open(p, 'w') as w:
                w.setnchannels(1); w.setsampwidth(2); w.setframerate(fs)
                w.writeframes(pcm.tobytes())
            fns.append(p)
        for mode, hp in (('all', hop), ('all', None), ('first', None)):
            lst = ref.get_fns_seg_list(fns, mode, fs, dur, hop=hp)
            key = f'seglist_{mode}_{"hop" if hp else "nohop"}'
            out[key] = np.asarray([[fns.index(f), s, lo, hi] for f, s, lo, hi in lst], dtype=np.int64)
        # load_audio at assorted starts/offsets (incl. windows that run past the end of the file)
        cases = [(6, 0.0, 0.0), (6, 2.0, 0.123), (6, 4.5, 0.0), (5, 2.0, -0.25), (0, 0.0, 0.0), (3, 1.0, 0.4), (2, 0.5, 0.001)]
        out['load_cases'] = np.asarray(cases, dtype=np.float64)
        out['load_out'] = np.stack([ref.load_audio(fns[int(f)], seg_start_sec=st, offset_sec=off, seg_length_sec=dur, fs=fs)
                                    for f, st, off in cases])
        # (load_audio_multi_start hard-codes fs=8000 in its inner call, audio_utils.py:276-279: not usable at fs=1000)
    # ---- augmentation arithmetic --------------------------------------------------------------
    ev = rng.normal(size=(6, T)) * rng.uniform(0.01, 0.3, size=(6, 1))
    bg = rng.normal(size=(6, T)) * rng.uniform(0.01, 0.5, size=(6, 1))
    ev[4] = 0.0                                            # silent event row
    bg[5] = 0.0                                            # silent background row
    out['ev'], out['bg'] = ev, bg
    out['max_normalize'] = np.stack([ref.max_normalize(ev[0].copy()), ref.max_normalize(ev[4].copy())])
    out['background_mix'] = ref.background_mix(ev[1].copy(), bg[1].copy(), fs, snr_db=7.5)
    np.random.seed(1234)                                   # bg_mix_batch draws snrs, then the amplitude ratios
    out['bg_mix_batch'] = ref.bg_mix_batch(ev.copy(), bg.copy(), fs, snr_range=(0, 10))
    np.random.seed(1234)                                   # the same draws, replayed in the same order
    snrs = np.random.rand(6) * (10 - 0) + 0
    amps = ref.log_scale_random_number_batch(bsz=6, amp_range=(0.1, 1))
    out['snrs'], out['amps'] = snrs, amps
    irs = np.zeros((6, 60))
    for i in range(6):
        irs[i, :20 + 8 * i] = rng.normal(size=20 + 8 * i) * np.exp(-np.arange(20 + 8 * i) / 9.0)
    irs[3] = 0.0                                           # silent impulse response
    out['ir'] = irs
    out['ir_aug_batch'] = ref.ir_aug_batch(ev.copy(), irs.copy())
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **out)
    print(OUT, os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
    main()

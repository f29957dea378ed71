"""Synthesise a FIXED augmented query set as WAV files (the reference's extras/dataset2wav.py:1-121): every source
clip is cut into consecutive INTERVAL-second pieces, each piece gets its own random background (at SNR), impulse
response and +-0.2*INTERVAL s offset, and the pieces are written back as one WAV per source clip.

    python tools/dataset2wav.py [-c CONFIG] [--source_dir val-query-db-500-30s/db] [--out ../aug_output/val_10dB]
                                [--snr 10 10] [--interval 1] [--split test]

The augmentation runs on the device (`genUnbalSequence(..., reduce_batch_first_half=True)`).  The reference writes
through `wavio.write(..., sampwidth=2)`; here the float signal in [-1, 1] is scaled by 32767 to int16.
"""
import argparse
import glob
import os
import sys
import wave

import numpy as np
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def synthesize(cfg, source_dir, out_root, snr=(10, 10), interval=1, clip_sec=30, split='ts', seed=0):
    from neural_audio_fp_amd.model.utils.dataloader_keras import genUnbalSequence
    fs = cfg['MODEL']['FS']
    src = sorted(glob.glob(cfg['DIR']['SOURCE_ROOT_DIR'] + source_dir + '/**/*.wav', recursive=True))
    bg = sorted(glob.glob(cfg['DIR']['BG_ROOT_DIR'] + split + '/**/*.wav', recursive=True))
    ir = sorted(glob.glob(cfg['DIR']['IR_ROOT_DIR'] + split + '/**/*.wav', recursive=True))
    assert clip_sec / interval == int(clip_sec / interval)
    n_anchor = int(clip_sec / interval)                        # one batch = one source clip
    ds = genUnbalSequence(src, bsz=2 * n_anchor, n_anchor=n_anchor, duration=interval, hop=interval, fs=fs, shuffle=False,
                          random_offset_anchor=False, offset_margin_hop_rate=0.2, bg_mix_parameter=[True, bg, snr],
                          ir_mix_parameter=[True, ir], speech_mix_parameter=[False], reduce_batch_first_half=True, seed=seed)
    written = []
    for i in range(len(ds)):
        X, _ = ds[i]
        x = X.reshape(-1).cpu().numpy()
        f = int(ds.fns_event_seg_list[n_anchor * i][0])
        sub_dir, fname = ds.ev.fns[f].split('/')[-2:]
        os.makedirs(f'{out_root}/{sub_dir}', exist_ok=True)
        with wave.open(f'{out_root}/{sub_dir}/{fname}', 'w') as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(fs)
            w.writeframes(np.clip(np.round(x * 32767.0), -32768, 32767).astype('<i2').tobytes())
        written.append(f'{out_root}/{sub_dir}/{fname}')
    return written


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('-c', '--config', default='640_lamb')
    ap.add_argument('--source_dir', default='val-query-db-500-30s/db')
    ap.add_argument('--out', default='../aug_output/val_10dB')
    ap.add_argument('--snr', nargs=2, type=float, default=(10, 10))
    ap.add_argument('--interval', type=int, default=1)
    ap.add_argument('--split', default='ts')
    a = ap.parse_args()
    cfg = yaml.safe_load(open(f'./config/{a.config}.yaml'))
    files = synthesize(cfg, a.source_dir, a.out, tuple(a.snr), a.interval, split=a.split)
    print(f'{len(files)} files under {a.out}')

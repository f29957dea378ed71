"""Convergence sanity of the whole training path on a synthetic dataset tree (80 x 30-s clips, bg / ir sets):
`python tools/train_sanity.py [config=640_lamb] [epochs=6]` -> per-epoch training loss.  Measured on one MI355X:
default.yaml (Adam, BSZ 120), 40 epochs = 1,160 steps in 24 s, loss 5.71 -> 0.13."""
import os, sys, wave, time, copy, yaml, numpy as np, torch
ROOT='/root/repo'; sys.path.insert(0, ROOT)
from neural_audio_fp_amd.model import trainer as T
rng=np.random.default_rng(0); d='/tmp/ts/'
def wav(p,x):
    os.makedirs(os.path.dirname(p),exist_ok=True)
    with wave.open(p,'w') as w: w.setnchannels(1); w.setsampwidth(2); w.setframerate(8000); w.writeframes(np.clip(x,-32768,32767).astype('<i2').tobytes())
t=np.arange(240000)/8000.0
for i in range(80):
    x=rng.integers(-800,800,size=240000).astype(float)
    for f in rng.uniform(200,3800,size=5): x+=3000*np.sin(2*np.pi*f*t*(1+0.02*np.sin(2*np.pi*rng.uniform(0.1,1)*t))+rng.uniform(0,6))*(0.6+0.4*np.sin(2*np.pi*rng.uniform(0.2,3)*t))
    wav(f'{d}music/train-10k-30s/a/{i}.wav',x)
for i in range(10):
    wav(f'{d}aug/bg/tr/{i}.wav',rng.integers(-4000,4000,size=80000)); wav(f'{d}aug/ir/tr/{i}.wav',12000*rng.normal(size=1000)*np.exp(-np.arange(1000)/60.0))
cfg=yaml.safe_load(open(ROOT+'/config/'+(sys.argv[1] if len(sys.argv)>1 else '640_lamb')+'.yaml'))
cfg['DIR'].update({'SOURCE_ROOT_DIR':d+'music/','BG_ROOT_DIR':d+'aug/bg/','IR_ROOT_DIR':d+'aug/ir/','LOG_ROOT_DIR':'/tmp/ts/logs/'})
cfg['TRAIN']['MAX_EPOCH']=int(sys.argv[2]) if len(sys.argv)>2 else 6
t0=time.perf_counter(); h=T.trainer(cfg,'sanity'); print('history',[round(x,3) for x in h], 'time %.1fs'%(time.perf_counter()-t0))

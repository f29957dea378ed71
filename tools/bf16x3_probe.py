import os, sys, time, torch, yaml, numpy as np
ROOT='/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd(); sys.path.insert(0, ROOT)
import neural_audio_fp_amd as nafp, bench
cfg = yaml.safe_load(open(os.path.join(ROOT, 'config', 'default.yaml')))
pre = nafp.get_melspec_layer(cfg); fp = nafp.FingerPrinter(seed=0)
x = bench.make_audio(640, 0, torch).cuda()
feat = pre(x, group_size=640)
ref = fp(feat).clone()
fp.set_option(3, 1)
got = fp(feat).clone()
print('max |d emb|', float((got-ref).abs().max()), 'min cos', float((got*ref).sum(1).min()))
for opt in (0, 1, 0, 1):
    fp.set_option(3, opt)
    for _ in range(3): fp(pre(x, group_size=640, defer=True))
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(20): fp(pre(x, group_size=640, defer=True))
    torch.cuda.synchronize(); el=time.perf_counter()-t0
    print('bf16x3' if opt else 'f32   ', '%.1f seg/s  %.3f ms/step' % (640*20/el, el/20*1e3))

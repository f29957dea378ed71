"""Which ingredient of an exact-split GEMM K-loop disturbs the front end when the two overlap?  The culprit here is the STANDALONE probe kernel
(tools/probes/x6_gemm_probe.hip, built as a library) in several reduced forms, the victim the library's front end on a second stream."""
import ctypes, os, sys
import torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import neural_audio_fp_amd as nafp
lib = ctypes.CDLL(os.path.join(ROOT, 'tools', 'probes', 'libx6_gemm_probe.so'))
lib.x6_probe_launch.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p]
cfg = yaml.safe_load(open(os.path.join(ROOT, 'config', 'default.yaml')))
m_pre = nafp.get_melspec_layer(cfg)
g = torch.Generator(device='cuda').manual_seed(1)
xs = [0.1 * torch.randn((125, 1, 8000), generator=g, device='cuda') for _ in range(8)]
refs = [m_pre(x, group_size=125, defer=True).raw.clone() for x in xs]
M = 160 * 2048                                             # a quarter of conv1's rows: ~0.18 ms per launch
A = torch.randn((640 * 256 * 16 * 128,), generator=g, device='cuda')
Bhm = torch.randint(0, 2 ** 15, (128 * 384 * 2,), generator=g, device='cuda', dtype=torch.int16)
Bl = torch.randint(0, 2 ** 15, (128 * 384,), generator=g, device='cuda', dtype=torch.int16)
C = torch.empty((M * 128,), device='cuda')
s_c, s_v = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
for name, v, m_rows, n_launch in (('no DMA inside the loop, 1280 workgroups', 1, M, 6), ('the same, 16 workgroups x 300 launches', 1, 4096, 300), ('the same, 64 workgroups x 150 launches', 1, 16384, 150),
                                  ('the same, 256 workgroups x 40 launches', 1, 65536, 40), ('nothing', -1, M, 0)):
    bad = 0
    for rep in range(10):
        if v >= 0:
            for _ in range(n_launch):
                lib.x6_probe_launch(v, A.data_ptr(), Bhm.data_ptr(), Bl.data_ptr(), C.data_ptr(), m_rows, s_c.cuda_stream)
        res = []
        with torch.cuda.stream(s_v):
            for x in xs:
                res.append(m_pre(x, group_size=125, defer=True).raw)
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(r, refs[i]) else 1 for i, r in enumerate(res))
    print(f'culprit = probe kernel, {name}: {bad} of 80 front-end results differ from the solo run', flush=True)

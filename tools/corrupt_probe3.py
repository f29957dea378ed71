"""A self-checking victim (no shared state: every thread recomputes a fixed chain of float operations and compares with its own first
result) next to the standalone exact-split probe kernel: which kind of instruction comes out different?"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vic = ctypes.CDLL(os.path.join(ROOT, 'tools', 'probes', 'libcorrupt_probe.so'))
vic.selfcheck.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
cul = ctypes.CDLL(os.path.join(ROOT, 'tools', 'probes', 'libx6_gemm_probe.so'))
cul.x6_probe_launch.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p]
g = torch.Generator(device='cuda').manual_seed(1)
M = 160 * 2048
A = torch.randn((640 * 256 * 16 * 128,), generator=g, device='cuda')
Bhm = torch.randint(0, 2 ** 15, (128 * 384 * 2,), generator=g, device='cuda', dtype=torch.int16)
Bl = torch.randint(0, 2 ** 15, (128 * 384,), generator=g, device='cuda', dtype=torch.int16)
C = torch.empty((M * 128,), device='cuda')
s_c, s_v = torch.cuda.Stream(), torch.cuda.Stream()
names = ['plain FMAs', 'v_sqrt_f32', 'v_exp_f32', 'v_log_f32', 'cross-lane (shfl)']
for culprit in (1, -1):
    counts = torch.zeros(16, dtype=torch.int32, device='cuda')
    for rep in range(10):
        if culprit >= 0:
            for _ in range(8):
                cul.x6_probe_launch(culprit, A.data_ptr(), Bhm.data_ptr(), Bl.data_ptr(), C.data_ptr(), M, s_c.cuda_stream)
        for which in range(5):
            vic.selfcheck(counts.data_ptr(), 1024, 40, which, s_v.cuda_stream)
        torch.cuda.synchronize()
    c = counts.tolist()
    print(('next to the exact-split probe kernel' if culprit >= 0 else 'alone') + ': ' + ', '.join(f'{names[k]} {c[4 + k]}' for k in range(5)) + f'  ({c[2]} workgroups)', flush=True)

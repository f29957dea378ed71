"""What does a co-resident kernel lose next to the exact-split GEMM kernels: LDS contents or register contents?  (tools/probes/corrupt_probe.hip)
    OPT=2 NAFP_X6_FUSE0=0 NAFP_X6_LAYERS=0x3e python tools/corrupt_probe.py"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import neural_audio_fp_amd as nafp
lib = ctypes.CDLL(os.path.join(ROOT, 'tools', 'probes', 'libcorrupt_probe.so'))
lib.victim.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
opt = int(os.environ.get('OPT', '2'))
m = nafp.FingerPrinter(seed=0)
m.set_option(3, opt)
g = torch.Generator(device='cuda').manual_seed(1)
feat = -1.2 * torch.rand((250, 256, 32, 1), generator=g, device='cuda')
m(feat); torch.cuda.synchronize()
s_enc, s_vic = torch.cuda.Stream(), torch.cuda.Stream()
for lds_kb, ticks in ((72, 3000), (72, -3000), (72, -1003000), (40, -1003000), (16, -1003000)):
    counts = torch.zeros(4, dtype=torch.int32, device='cuda')
    for rep in range(20):
        with torch.cuda.stream(s_enc):
            for _ in range(3):
                m(feat)
        for _ in range(6):
            lib.victim(counts.data_ptr(), 512, lds_kb * 1024, ticks, s_vic.cuda_stream)      # 3000 ticks of the 100 MHz clock = 30 us; negative: the active form
        torch.cuda.synchronize()
    c = counts.tolist()
    print(f'opt {opt}, {"active" if ticks < 0 else "passive"} victim with {lds_kb} KB of LDS + 96 registers per lane: {c[2]} workgroups, LDS words changed {c[0]}, register values changed {c[1]}, arithmetic checks failed {c[3]}', flush=True)

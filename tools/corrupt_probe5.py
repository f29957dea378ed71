"""Packed-f32 self-checking victims (tools/probes/corrupt_probe.hip: pkcheck) next to the standalone exact-split culprit: is it the PACKED float
instructions of a co-resident wave that come out wrong?  Controls: the same chains on scalar v_fma_f32, and every chain alone.
One (culprit, victim) pair at a time; the first mismatches are printed with expected / found bits."""
import ctypes, os, struct, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vic = ctypes.CDLL(os.path.join(ROOT, 'tools', 'probes', 'libcorrupt_probe.so'))
vic.pkcheck.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
cul = ctypes.CDLL(os.path.join(ROOT, 'tools', 'probes', 'libx6_gemm_probe.so'))
cul.x6_probe_launch.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p]
g = torch.Generator(device='cuda').manual_seed(1)
M = 160 * 2048
A = torch.randn((640 * 256 * 16 * 128,), generator=g, device='cuda')
Bhm = torch.randint(0, 2 ** 15, (128 * 384 * 2,), generator=g, device='cuda', dtype=torch.int16)
Bl = torch.randint(0, 2 ** 15, (128 * 384,), generator=g, device='cuda', dtype=torch.int16)
C = torch.empty((M * 128,), device='cuda')
s_c, s_v = torch.cuda.Stream(), torch.cuda.Stream()
names = ['v_pk_fma_f32', 'v_pk_mul_f32 + v_pk_add_f32', 'v_pk_fma_f32 with op_sel / neg', 'v_fma_f32 (scalar control)', 'v_pk_mul / v_pk_add with op_sel / neg']
cnames = {0: 'whole K-loop', 1: 'LDS reads + split + MFMAs', 3: 'MFMAs on static registers', 5: 'LDS reads + split, no MFMAs', 6: 'split + MFMAs on computed operands', -1: 'alone'}
f = lambda u: struct.unpack('<f', struct.pack('<I', u & 0xffffffff))[0]
shown = 0
for culprit in (1, 6, 0, 5, 3, -1):
    line = []
    for which in range(5):
        counts = torch.zeros(16 + 64 * 4, dtype=torch.int32, device='cuda')
        per_rep = []
        for rep in range(6):
            before = int(counts[10 + which])
            if culprit >= 0:
                for _ in range(30):
                    cul.x6_probe_launch(culprit, A.data_ptr(), Bhm.data_ptr(), Bl.data_ptr(), C.data_ptr(), M, s_c.cuda_stream)
            for _ in range(4):
                vic.pkcheck(counts.data_ptr(), 1024, 200, which, s_v.cuda_stream)
            torch.cuda.synchronize()
            per_rep.append(int(counts[10 + which]) - before)
        line.append(f'{names[which]}: {sum(per_rep)} {per_rep}')
        c = counts.tolist()
        if c[15] and shown < 4:
            shown += 1
            print(f'  first mismatches, victim "{names[which]}" next to "{cnames[culprit]}":')
            for k in range(min(c[15], 10)):
                e, b, w, blk = c[16 + 4 * k: 20 + 4 * k]
                print(f'    workgroup {blk} round {w >> 16} chain {(w >> 8) & 255} half {(w >> 7) & 1} lane {w & 63}: expected {f(e):+.8e} ({e & 0xffffffff:08x}) found {f(b):+.8e} ({b & 0xffffffff:08x})', flush=True)
    print(f'{cnames[culprit]}:\n    ' + '\n    '.join(line), flush=True)

"""Two questions about the overlap effect (profiles/r06_experiments.md section 5), culprit = the standalone probe kernel in its worst form
(split + bf16 MFMAs on operands computed in registers, no LDS reads, no memory traffic inside the loop):

 A. WHAT is wrong in a disturbed front-end result: how many elements, where, and how the wrong values relate to the right ones;
 B. are kernels this repository did NOT write disturbed too (torch elementwise, rocBLAS / hipBLASLt f32 GEMM, rocFFT, softmax, layer_norm,
    cumsum, sort): every one is run next to the culprit and compared bit for bit with its own solo result, with a solo-vs-solo control."""
import ctypes, os, sys
import torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import neural_audio_fp_amd as nafp
lib = ctypes.CDLL(os.path.join(ROOT, 'tools', 'probes', 'libx6_gemm_probe.so'))
lib.x6_probe_launch.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p]
VARIANT = int(os.environ.get('CULPRIT', '6'))
cfg = yaml.safe_load(open(os.path.join(ROOT, 'config', 'default.yaml')))
m_pre = nafp.get_melspec_layer(cfg)
g = torch.Generator(device='cuda').manual_seed(1)
M = 160 * 2048
A = torch.randn((640 * 256 * 16 * 128,), generator=g, device='cuda')
Bhm = torch.randint(0, 2 ** 15, (128 * 384 * 2,), generator=g, device='cuda', dtype=torch.int16)
Bl = torch.randint(0, 2 ** 15, (128 * 384,), generator=g, device='cuda', dtype=torch.int16)
C = torch.empty((M * 128,), device='cuda')
s_c, s_v = torch.cuda.Stream(), torch.cuda.Stream()


def culprit(n=6):
    for _ in range(n):
        lib.x6_probe_launch(VARIANT, A.data_ptr(), Bhm.data_ptr(), Bl.data_ptr(), C.data_ptr(), M, s_c.cuda_stream)


# ---- A
xs = [0.1 * torch.randn((125, 1, 8000), generator=g, device='cuda') for _ in range(8)]
refs = [m_pre(x, group_size=125, defer=True).raw.clone() for x in xs]
torch.cuda.synchronize()
print('front-end result', tuple(refs[0].shape), refs[0].dtype, flush=True)
shown = 0
n_bad = 0
for rep in range(10):
    culprit()
    with torch.cuda.stream(s_v):
        res = [m_pre(x, group_size=125, defer=True).raw for x in xs]
    torch.cuda.synchronize()
    for i, r in enumerate(res):
        if torch.equal(r, refs[i]):
            continue
        n_bad += 1
        if shown >= 6:
            continue
        shown += 1
        ne = (r != refs[i]) | (r.isnan() != refs[i].isnan())
        idx = ne.nonzero()
        print(f'  rep {rep} launch {i}: {idx.shape[0]} elements differ; per-dimension ranges ' + ', '.join(f'[{int(idx[:, d].min())}..{int(idx[:, d].max())}]' for d in range(idx.shape[1])), flush=True)
        for row in idx[:12].tolist():
            a, b = float(refs[i][tuple(row)]), float(r[tuple(row)])
            ai = refs[i][tuple(row)].view(torch.int32).item() & 0xffffffff
            bi = r[tuple(row)].view(torch.int32).item() & 0xffffffff
            print(f'    {row}: solo {a:+.7e} ({ai:08x})  next to the culprit {b:+.7e} ({bi:08x})  xor {ai ^ bi:08x}', flush=True)
        # is the wrong value the right value of ANOTHER element (a moved lane / a stale read)?
        wrong = r[ne]
        pool = refs[i].flatten()
        hits = sum(int((pool == w).any()) for w in wrong[:200])
        print(f'    of the first {min(200, wrong.numel())} wrong values, {hits} are the solo value of some other element of the same result', flush=True)
print(f'A: {n_bad} of 80 front-end results differ', flush=True)

# ---- B
xe = torch.randn((32 * 1024 * 1024,), generator=g, device='cuda')
xm1, xm2 = torch.randn((4096, 4096), generator=g, device='cuda'), torch.randn((4096, 4096), generator=g, device='cuda')
xf = torch.randn((16000, 1024), generator=g, device='cuda')
xl = torch.randn((65536, 512), generator=g, device='cuda')
foreign = {
    'elementwise sin/fma (32 M)': lambda: torch.sin(xe) * 1.5 + xe,
    'f32 GEMM 4096^3 (torch.mm)': lambda: torch.mm(xm1, xm2),
    'rfft 16000 x 1024': lambda: torch.view_as_real(torch.fft.rfft(xf)),
    'softmax 65536 x 512': lambda: torch.softmax(xl, dim=1),
    'layer_norm 65536 x 512': lambda: torch.nn.functional.layer_norm(xl, (512,)),
    'cumsum 65536 x 512': lambda: torch.cumsum(xl, dim=1),
    'sort 65536 x 512': lambda: torch.sort(xl, dim=1).values,
    'exp/log chain (32 M)': lambda: torch.log1p(torch.exp(xe * 0.1)) * xe,
}
for name, f in foreign.items():
    with torch.cuda.stream(s_v):
        ref = f().clone()
    torch.cuda.synchronize()
    out = {}
    for mode in ('alone', 'next to the culprit'):
        bad = 0
        for rep in range(10):
            if mode != 'alone':
                culprit(60)
            with torch.cuda.stream(s_v):
                rs = [f() for _ in range(4)]
            torch.cuda.synchronize()
            bad += sum(0 if torch.equal(r, ref) else 1 for r in rs)
        out[mode] = bad
    print(f'B: {name}: differ from the solo result: alone {out["alone"]} of 40, next to the culprit {out["next to the culprit"]} of 40', flush=True)

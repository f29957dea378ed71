"""Probe: throughput with consecutive batches issued round-robin on S HIP streams."""
import os, sys, time, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import neural_audio_fp_amd as nafp, bench
cfg = yaml.safe_load(open(os.path.join(ROOT, 'config', 'default.yaml')))
for S in (1, 2, 3, 4, 6, 8):
    streams = [torch.cuda.Stream() for _ in range(S)]
    pres = [nafp.get_melspec_layer(cfg) for _ in range(S)]
    fps = [nafp.FingerPrinter(seed=0) for _ in range(S)]
    pool = [bench.make_audio(640, i, torch).cuda() for i in range(4)]
    def run(n):
        for i in range(n):
            k = i % S
            with torch.cuda.stream(streams[k]):
                emb = fps[k](pres[k](pool[i % 4], group_size=640, defer=True))
        return emb
    run(2 * S); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(48); torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f'streams {S}: {640 * 48 / el:.0f} seg/s  ({el / 48 * 1e3:.3f} ms/step)')

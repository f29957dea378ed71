"""Does a producer -> consumer hand-off through memory run faster when the working set fits the 256 MiB Infinity Cache?
Plain torch kernels: `y = x * 2` (write y) followed by `y += 1` in place (read + write y), per working-set size; and a
read-only pass over a buffer that the previous kernel has just written."""
import torch


def t(f, reps=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for mb in (8, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048):
    n = mb * (1 << 20) // 4
    x = torch.randn(n, device='cuda'); y = torch.empty_like(x)

    def chain():
        torch.mul(x, 2.0, out=y)      # read x, write y
        y.add_(1.0)                   # read y, write y (the hand-off)
    ms = t(chain)
    ms_sum = t(lambda: (y.add_(1.0), y.sum()))
    print(f'{mb:5d} MB: mul+add_ {ms * 1e3:8.1f} us = {4 * n * 4 / ms / 1e9:6.2f} TB/s over 4 streams;  add_+sum {ms_sum * 1e3:8.1f} us = {3 * n * 4 / ms_sum / 1e9:6.2f} TB/s over 3 streams')

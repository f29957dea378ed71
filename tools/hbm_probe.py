"""HBM ceilings on this box with plain torch kernels: write-only (fill), read+write (copy), read-only (sum)."""
import torch
n = 1_342_177_280 // 4          # 1.34 GB of float32: conv0's output at B = 640
x = torch.empty(n, device='cuda'); y = torch.empty(n, device='cuda')
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: x.fill_(1.5));  print(f'fill   {ms:.3f} ms  {n * 4 / ms / 1e9:.2f} TB/s write')
ms = t(lambda: y.copy_(x));    print(f'copy   {ms:.3f} ms  {2 * n * 4 / ms / 1e9:.2f} TB/s read+write')
ms = t(lambda: x.sum());       print(f'sum    {ms:.3f} ms  {n * 4 / ms / 1e9:.2f} TB/s read')
ms = t(lambda: torch.add(x, y, out=y)); print(f'add    {ms:.3f} ms  {3 * n * 4 / ms / 1e9:.2f} TB/s 2 reads + 1 write')

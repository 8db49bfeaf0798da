"""Achievable HBM bandwidth on this MI355X: device copy (read + write) and read-only reduction, 2 GiB buffers."""
import torch
dev = "cuda:0"
n = 512 * 1024 * 1024          # floats = 2 GiB
x = torch.randn(n, device=dev); y = torch.empty_like(x)
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e-3
tc = t(lambda: y.copy_(x)); print(f"copy   : {2 * 4 * n / tc / 1e12:.2f} TB/s (read + write)")
tr = t(lambda: x.sum());    print(f"reduce : {4 * n / tr / 1e12:.2f} TB/s (read only)")
tw = t(lambda: y.zero_());  print(f"memset : {4 * n / tw / 1e12:.2f} TB/s (write only)")

"""achievable HBM write / read / copy rates with torch kernels (what bounds the activation stores of the training forward)"""
import torch
n = 704 * 1024 * 1024 // 4
a = torch.empty(n, device="cuda"); b = torch.empty(n, device="cuda")
def t(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: a.fill_(1.0)); print(f"write-only fill 0.70 GB: {ms * 1e3:7.1f} us = {n * 4 / ms / 1e9:.2f} TB/s")
ms = t(lambda: a.zero_()); print(f"memset     zero 0.70 GB: {ms * 1e3:7.1f} us = {n * 4 / ms / 1e9:.2f} TB/s")
ms = t(lambda: b.copy_(a)); print(f"copy 0.70 GB -> 0.70 GB: {ms * 1e3:7.1f} us = {2 * n * 4 / ms / 1e9:.2f} TB/s (read + write)")
ms = t(lambda: a.sum()); print(f"read-only sum  0.70 GB: {ms * 1e3:7.1f} us = {n * 4 / ms / 1e9:.2f} TB/s")

#!/usr/bin/env python3
"""Measured HBM ceilings on this box (torch device copy / fill / read), to put next to the 8 TB/s spec peak."""
import torch
n = 1 << 28                                  # 2 GiB of int64
a = torch.empty(n, dtype=torch.int64, device="cuda")
b = torch.ones(n, dtype=torch.int64, device="cuda")
def timed(f, reps=10):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
t = timed(lambda: a.copy_(b)); print("copy  : %.2f ms  %.0f GB/s (read+write)" % (t, 2 * n * 8 / t / 1e6))
t = timed(lambda: a.fill_(3)); print("fill  : %.2f ms  %.0f GB/s (write)" % (t, n * 8 / t / 1e6))
t = timed(lambda: b.sum());    print("read  : %.2f ms  %.0f GB/s (read)" % (t, n * 8 / t / 1e6))

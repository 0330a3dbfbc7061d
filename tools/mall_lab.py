#!/usr/bin/env python3
"""Does a working set that fits the 256 MiB Infinity Cache stream faster than one that does not?  In-place scale (read + write of the same lines) and copy
(two arrays) of fp64 arrays from 16 MB to 2 GB, repeated back to back: GB/s of algorithmic bytes.  tools/mall_lab.py"""
import torch

e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for mb in (16, 32, 64, 96, 128, 192, 256, 512, 1024, 2048):
    n = mb * (1 << 20) // 8
    a = torch.randn(n, dtype=torch.float64, device="cuda")
    b = torch.empty_like(a)
    res = []
    for name, fn, nbytes in (("in-place x*=c", lambda: a.mul_(1.0000001), 2 * n * 8), ("copy b<-a", lambda: b.copy_(a), 2 * n * 8), ("read sum(a)", lambda: a.sum(), n * 8)):
        for _ in range(3):
            fn()
        reps = max(5, min(200, (8 << 30) // nbytes))
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(f"{name}: {nbytes * reps / e0.elapsed_time(e1) / 1e6:6.0f} GB/s")
    print(f"{mb:5d} MB   " + "   ".join(res), flush=True)
    del a, b

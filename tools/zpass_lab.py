#!/usr/bin/env python3
"""The fused z pass alone (ins_dbg_zsolve) at a fixed byte count with different plane strides: tools/zpass_lab.py [label:OPT=V,...] ...
nz = 512; boxes x lines = 135168 lines in total (the 512^3 half spectrum: 2.2 GB); plane stride = lines per box x 16 B."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins
from ins_amd import _lib
lib = _lib.load()
lib.ins_dbg_zsolve.restype = C.c_int
lib.ins_dbg_zsolve.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
nz, total = 512, 135168
buf = torch.randn(2 * nz * total, dtype=torch.float64, device="cuda")
variants = [("default", {})]
for a in sys.argv[1:]:
    label, _, spec = a.partition(":")
    variants.append((label, {k: int(v) for k, _, v in (kv.partition("=") for kv in filter(None, spec.split(",")))}))
allkeys = sorted({k for _, o in variants for k in o})
base = {k: _lib.get_option(k) for k in allkeys}
for label, opts in variants:
    for k in allkeys:
        _lib.set_option(k, opts.get(k, base[k]))
    for nbox in (1, 4, 16, 64):
        nl = total // nbox
        kxs = 264 if nl % 264 == 0 else 8
        ms = C.c_float()
        rc = lib.ins_dbg_zsolve(buf.data_ptr(), nz, nl, kxs - 7, kxs, nbox, 5, C.byref(ms))
        assert rc == 0, lib.ins_last_error()
        print(f"{label:24s} boxes {nbox:3d}  plane stride {nl * 16 / 1024:8.0f} KB   {ms.value:.4f} ms   {2 * 16.0 * nz * total / ms.value / 1e6:6.0f} GB/s", flush=True)

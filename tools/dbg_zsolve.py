import numpy as np, torch, sys, os
sys.path.insert(0,'.')
import ins_amd as ins
from oracle import ins_oracle as o
from tests import fixtures as fx
for n in [(32,32,16),(64,16,16)]:
    x = tuple(np.linspace(0.0, 1.0, ni + 1) for ni in n)
    so = o.make_setup(x, Re=800.0)
    sp = ins.Setup(x=x, Re=800.0)
    f = fx.randn_field(so.grid.N, 21)
    ip = tuple(slice(lo, hi) for lo, hi in so.grid.Ip)
    f[ip] -= f[ip].mean()
    want = o.poisson(o.psolver_spectral(so), f)
    ps = ins.psolver_spectral(sp)
    got = ins.to_numpy(ins.poisson(ps, ins.from_numpy(sp, f)))
    d = (got - want)[ip]
    print(n, "relerr %.3e" % (np.linalg.norm(d)/np.linalg.norm(want[ip])))
    if np.linalg.norm(d) > 1e-8:
        print(" err by x:", np.round(np.sqrt((d**2).sum(axis=(1,2)))[:66:4],3))
        print(" err by y:", np.round(np.sqrt((d**2).sum(axis=(0,2))),3))
        print(" err by z:", np.round(np.sqrt((d**2).sum(axis=(0,1))),3))
        # is `got` the solution for a different right-hand side? try: solve with the oracle for f scaled / shifted
        dh = np.fft.rfftn(d, axes=(2,1,0)); wh = np.fft.rfftn(want[ip], axes=(2,1,0))
        big = np.argwhere(np.abs(dh) > 1e-6*np.abs(wh).max())
        print(" modes with error:", len(big), "of", dh.size, "first:", big[:12].tolist())
        # ghost cells untouched?
        mask = np.ones(so.grid.N, bool); mask[ip] = False
        print(" ghosts untouched:", np.array_equal(got[mask], f[mask]))

import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins
from ins_amd import _lib
Dr, Sy, Pe = ins.DirichletBC, ins.SymmetricBC, ins.PeriodicBC
def run(name, x, bc):
    sp = ins.Setup(x=x, Re=1000.0, boundary_conditions=bc)
    rng = np.random.default_rng(5)
    u = ins.apply_bc_u(ins.from_numpy(sp, np.asfortranarray(rng.standard_normal(sp.grid.N + (3,)))), 0.0, sp)
    m = ins.smagorinsky_closure(sp)
    with _lib.options(INS_DISABLE_SMAGFORCE_GEN=1):
        three = ins.to_numpy(m(u, 0.17)).copy()
    one = ins.to_numpy(m(u, 0.17)).copy()
    d = np.abs(one - three)
    print(name, "N", sp.grid.N, "max|three|", np.abs(three).max())
    for c in range(3):
        dc = d[..., c]
        print(" comp", c, "max err", dc.max(), "at", np.unravel_index(dc.argmax(), dc.shape), "nbad", int((dc > 1e-10 * np.abs(three).max()).sum()), "of", int((three[..., c] != 0).sum()))
        bad = np.argwhere(dc > 1e-10 * np.abs(three).max())
        if len(bad):
            for ax in range(3):
                vals, cnt = np.unique(bad[:, ax], return_counts=True)
                print("   axis", ax, "bad idx:", dict(zip(vals.tolist()[:12], cnt.tolist()[:12])), "..." if len(vals) > 12 else "")
U = lambda n: np.linspace(0, 1, n + 1)
run("uniform walls", (U(8), U(6), U(7)), ((Dr(), Dr()),) * 3)
run("perx walls yz", (U(8), U(6), U(7)), ((Pe(), Pe()), (Dr(), Dr()), (Dr(), Dr())))
run("pery", (U(8), U(6), U(7)), ((Dr(), Dr()), (Pe(), Pe()), (Dr(), Dr())))
run("perz", (U(8), U(6), U(7)), ((Dr(), Dr()), (Dr(), Dr()), (Pe(), Pe())))
run("stretched periodic", (ins.tanh_grid(0, 1, 8, 1.2), ins.tanh_grid(0, 1, 6, 1.2), ins.tanh_grid(0, 1, 7, 1.2)), ((Pe(), Pe()),) * 3)
run("sym all", (U(8), U(6), U(7)), ((Sy(), Sy()),) * 3)
run("mixed sides", (ins.tanh_grid(0, 1, 8, 1.2), U(6), ins.cosine_grid(0, 1, 7)), ((Sy(), Dr()), (Dr(), Sy()), (Sy(), Dr())))
# oracle check of the Symmetric-x case
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ins_oracle as o
x = (U(8), U(6), U(7))
so = o.make_setup(x, ((o.SymmetricBC(), o.SymmetricBC()),) * 3, Re=1000.0)
sp = ins.Setup(x=x, Re=1000.0, boundary_conditions=((Sy(), Sy()),) * 3)
rng = np.random.default_rng(5)
u_h = o.apply_bc_u(np.asfortranarray(rng.standard_normal(so.grid.N + (3,))), 0.0, so)
want = o.smagorinsky_closure(so)(u_h, 0.17)
m = ins.smagorinsky_closure(sp)
for name, opts in (("one", {}), ("three rows", dict(INS_DISABLE_SMAGFORCE_GEN=1)), ("three plain", dict(INS_DISABLE_SMAGFORCE_GEN=1, INS_FIELDS_ROWS=-1))):
    with _lib.options(**opts):
        got = ins.to_numpy(m(ins.from_numpy(sp, u_h), 0.17))
    print(name, "vs oracle:", [float(np.abs(got[..., c] - want[..., c]).max()) for c in range(3)])

"""Debug helper: one seed of tests/test_gpu_random_geometries.py under run-time options: tools/dbg_seed.py SEED OPT=V ..."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ins_amd as ins  # noqa: E402
from ins_amd import _lib  # noqa: E402
from oracle import ins_oracle as o  # noqa: E402
from tests import fixtures as fx  # noqa: E402
from tests.test_gpu_parity import mirror  # noqa: E402
from tests.test_gpu_random_geometries import random_setup  # noqa: E402

seed = int(sys.argv[1])
for kv in sys.argv[2:]:
    k, _, v = kv.partition("=")
    _lib.set_option(k, int(v))
rng = np.random.default_rng(1000 + seed)
so = random_setup(o, rng)
sp = mirror(ins, so, o)
g = so.grid
print("N", g.N, "bcs", [[type(b).__name__ for b in side] for side in so.boundary_conditions], flush=True)
D = g.D
u_raw = fx.randn_field(g.N + (D,), seed)
u_h = o.apply_bc_u(u_raw, 0.0, so)
u_d = ins.apply_bc_u(ins.from_numpy(sp, u_raw), 0.0, sp)
ps_h, ps_d = o.default_psolver(so), ins.default_psolver(sp)
print("fold mask", _lib.load().ins_dbg_fdm_fold_mask(ps_d.handle))
q_h = o.project(u_h, so, ps_h)
q_d = ins.project(u_d, sp, ps_d)
print("project err", np.abs(ins.to_numpy(q_d) - q_h).max() / np.abs(q_h).max())
o.apply_bc_u_(q_h, 0.0, so)
ref = o.solve_unsteady(so, (0.0, 1e-3), 0.05 * q_h, psolver=ps_h, dt=1e-3)
(v, _, _), _ = ins.solve_unsteady(setup=sp, tlims=(0.0, 1e-3), ustart=ins.from_numpy(sp, 0.05 * q_h), psolver=ps_d, Δt=1e-3)
d = np.abs(ins.to_numpy(v) - ref["u"])
print("step err", d.max() / np.abs(ref["u"]).max(), "at", np.unravel_index(d.argmax(), d.shape))

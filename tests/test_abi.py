"""CPU-side checks of the drop-in boundary: libinship.so loads and exports every symbol that
include/ins_hip.h declares; argument validation errors come back as codes, not crashes."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ins_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ins_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    import ins_amd

    lib = ins_amd._lib.load()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in ins_hip.h but not exported"
    assert sorted(ins_amd._lib.SIGNATURES) == names, "ctypes table and header disagree"


def test_error_codes_without_gpu():
    import ins_amd

    lib = ins_amd._lib.load()
    assert lib.ins_version() >= 100
    h = C.c_void_p()
    assert lib.ins_grid_create(None, C.byref(h)) == -1
    assert b"null" in lib.ins_last_error()
    d = ins_amd._lib.GridDesc()
    d.D = 4
    assert lib.ins_grid_create(C.byref(d), C.byref(h)) == -1
    assert lib.ins_divergence_f64(None, None, None, None) == -1
    assert lib.ins_grid_destroy(None) == 0 and lib.ins_poisson_destroy(None) == 0 and lib.ins_rk_destroy(None) == 0


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "incompressiblenavierstokes.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                for bad in ("import oracle", "from oracle", "ins_oracle", "oracle/"):
                    assert bad not in text, f"{f} references the oracle ({bad!r})"


def test_host_grid_matches_oracle(oracle):
    """The product's host-side Grid (grid.jl mirror) agrees with the oracle's on the reference fixtures."""
    import numpy as np

    import ins_amd
    from tests import fixtures as fx

    for mk in (fx.setup2d, fx.setup3d, fx.setup_mixed):
        so = mk(oracle)
        cls = {"PeriodicBC": ins_amd.PeriodicBC, "DirichletBC": ins_amd.DirichletBC, "SymmetricBC": ins_amd.SymmetricBC,
               "PressureBC": ins_amd.PressureBC}
        bcs = tuple(tuple(cls[type(b).__name__]() for b in side) for side in so.boundary_conditions)
        D = so.grid.D
        x = [so.grid.x[a] for a in range(D)]
        # strip the ghosts the oracle added to recover the physical coordinates
        xin = []
        for a in range(D):
            lo = 2 if isinstance(so.boundary_conditions[a][0], oracle.PressureBC) else 1
            xin.append(x[a][lo:-1])
        g = ins_amd.Grid(xin, bcs)
        assert g.N == so.grid.N and g.Ip == so.grid.Ip and g.Iu == so.grid.Iu
        for a in range(D):
            assert np.array_equal(g.Δ[a], so.grid.dx[a]) and np.array_equal(g.Δu[a], so.grid.dxu[a])
            for b in range(D):
                assert np.array_equal(g.A[a][b][0], so.grid.A[a][b][0]) and np.array_equal(g.A[a][b][1], so.grid.A[a][b][1])


def test_setup_without_gpu_fails_loudly():
    import torch

    import ins_amd

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ins_amd.INSHipError):
        ins_amd.Setup(x=([0.0, 0.5, 1.0], [0.0, 0.5, 1.0]))


def test_rk_tableaux_match_oracle(oracle):
    import numpy as np

    import ins_amd

    for name in ("RK44", "Wray3", "SSP33", "FE11"):
        a, b = getattr(ins_amd.RKMethods, name)(), getattr(oracle, name)()
        assert np.allclose(a.A, b.A) and np.allclose(a.c, b.c) and np.allclose(a.b, b.b)


def test_every_run_time_option_is_documented():
    """DESIGN.md §5 lists every switch of csrc/ins_options.hip (name spelled out, or as `_SUFFIX` next to a sibling with the same prefix)."""
    import ctypes
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = ctypes.CDLL(os.path.join(root, "incompressiblenavierstokes.jl_amd", "libinship.so"))
    lib.ins_option_name.restype = ctypes.c_char_p
    names = [lib.ins_option_name(i).decode() for i in range(lib.ins_option_count())]
    assert len(names) == len(set(names)) and all(n.startswith("INS_") for n in names)
    doc = open(os.path.join(root, "DESIGN.md"), encoding="utf-8").read()
    missing = []
    for n in names:
        if n in doc:
            continue
        prefix, suffix = n.rsplit("_", 1)
        if not re.search(r"`%s_\w+`(?: / `_\w+`)* / `_%s`" % (re.escape(prefix), re.escape(suffix)), doc):
            missing.append(n)
    assert not missing, missing

"""The example scripts (examples/*.py: the physical settings of the reference's own examples on this framework's API) run end to end on
the GPU at reduced sizes and give physically sensible answers."""
import importlib.util
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
EX = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples")


def load(name):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, EX)
    spec = importlib.util.spec_from_file_location(name, os.path.join(EX, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_taylor_green_vortex_3d():
    r = load("TaylorGreenVortex3D").main(n=32, tend=0.02, dt=1e-3, verbose=False)
    E = [e for _, e in r["energy"]]
    assert all(b < a for a, b in zip(E, E[1:])) and r["maxdiv"] < 1e-12
    assert r["κ"][np.argmax(r["ehat"])] in (1, 2, 3)  # energy sits at the vortex scale, |k| = 2√3 shells


def test_decaying_turbulence_3d(tmp_path):
    r = load("DecayingTurbulence3D").main(n=32, tend=0.01, kp=5, vtk=str(tmp_path), verbose=False)
    assert r["E1"] < r["E0"] and r["maxdiv"] < 1e-11 and r["t"] == pytest.approx(0.01)
    assert abs(int(r["κ"][np.argmax(r["ehat0"])]) - 5) <= 2
    assert any(f.endswith(".pvd") for f in os.listdir(tmp_path))


def test_lid_driven_cavity_3d():
    r = load("LidDrivenCavity3D").main(n=16, tend=0.05, dt=2.5e-3, verbose=False)
    assert r["psolver"] == "psolver_direct" and r["maxdiv"] < 1e-10 and 0.05 < r["umax"] <= 1.0 and r["E"] > 0


def test_rayleigh_benard_2d():
    r = load("RayleighBenard2D").main(n=16, tend=0.5, dt=1e-2, Ra=1e5, verbose=False)
    t, lo, hi = r["nusselt"][-1]
    # conduction profile develops from T = 1/2: heat enters at the bottom plate and leaves at the top one
    assert lo > 0 and hi > 0 and -1e-6 <= r["Tmin"] and r["Tmax"] <= 1 + 1e-6 and r["maxdiv"] < 1e-10
    assert r["Re"] == pytest.approx((1e5 / 0.71) ** 0.5)


def test_actuator_2d():
    r = load("Actuator2D").main(n=12, tend=1.0, dt=0.05, verbose=False)
    assert r["maxdiv"] < 1e-10 and r["wake"] < r["free"]  # the disk slows the flow behind it
    assert all(np.isfinite(f).all() for f in r["fields"].values())


def test_kolmogorov_2d():
    r = load("Kolmogorov2D").main(n=64, tend=0.1, dt=1e-3, verbose=False)
    E = [e for _, e in r["energy"]]
    assert E[-1] > E[0] and r["forced_mode"] > 0.05 and r["maxdiv"] < 1e-11  # the body force feeds the sin(8πy) mode


def test_shear_layer_2d():
    r = load("ShearLayer2D").main(n=64, tend=0.3, dt=0.01, verbose=False)
    # ½∫|u|² over the (2π)² box with |u| ≈ 1 outside the layers; layers of thickness π/15: |ω| ~ 1/d ≈ 4.8
    assert r["maxdiv"] < 1e-11 and 0.3 < r["E"] / (4 * np.pi**2) < 0.55 and r["hist"][-1][1] > 2.0


def test_planar_mixing_2d():
    r = load("PlanarMixing2D").main(n=16, tend=2.0, verbose=False)
    assert r["t"] == pytest.approx(2.0) and r["maxdiv"] < 1e-9 and r["ulo"] < 1.0 < r["uhi"] and np.isfinite(r["vmax"])


def test_turbulent_channel():
    r = load("TurbulentChannel").main(n=16, tend=0.02, verbose=False)
    p = r["profile"]
    assert r["psolver"] == "psolver_direct" and r["maxdiv"] < 1e-10 and r["t"] == pytest.approx(0.02)
    assert p[len(p) // 2] > 0.8 and p[0] < 0.5 * p[len(p) // 2] and p[-1] < 0.5 * p[len(p) // 2]  # parabola-like: fast core, slow walls

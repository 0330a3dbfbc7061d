"""INS_STEP_GRAPH=1: ins_rk_steps_f64 replays one captured step as a hipGraph (csrc/ins_rk.hip; opt-in — it frees the host thread, the device span does
not shrink, see the comment there).  A replay issues the same kernels with the same arguments in the same order as the plain loop, so the two must agree
BITWISE; the tests also prove the graph really ran (replay counter of the stepper cache) and that the default and the off switch keep the plain loop."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ins():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ins_amd

    return ins_amd


def _replays(cache):
    from ins_amd import _lib

    fn = _lib.load().ins_dbg_rk_graph_replays
    fn.restype, fn.argtypes = C.c_longlong, [C.c_void_p]
    return int(fn(cache.handle))


def _run(ins, n, method, nsteps, calls, **opts):
    from ins_amd import _lib

    sp = ins.Setup(x=tuple(np.linspace(0.0, 1.0, ni + 1) for ni in n), Re=500.0)
    ps = ins.psolver_spectral(sp)
    u = ins.random_field(sp, kp=3, seed=5, psolver=ps)
    m = getattr(ins.RKMethods, method)()
    with _lib.options(**opts):
        cache = ins.ode_method_cache(m, sp, ps)
        st = ins.create_stepper(m, setup=sp, psolver=ps, u=u, t=0.0)
        for _ in range(calls):
            st = ins.timesteps_(m, st, 2e-3, nsteps, cache=cache)
        torch.cuda.synchronize()
        return st.u.clone(), _replays(cache), float(ins.max_abs_divergence(st.u, sp))


@pytest.mark.parametrize("n,method", [((32, 32, 32), "RK44"), ((128, 16, 16), "Wray3"), ((16, 16, 16), "FE11"), ((64, 64), "RK44"), ((128, 32), "SSP33")])
def test_graph_replay_is_bitwise_the_plain_loop(ins, n, method):
    a, ra, diva = _run(ins, n, method, 6, 2, INS_STEP_GRAPH=1)
    b, rb, _ = _run(ins, n, method, 6, 2)
    assert rb == 0  # not the default
    # chained loops: the first and the last step of a call run directly (4 replays per call of 6); whole-step graphs (FE11): all but the first (5 per call)
    chained = method != "FE11"  # every fused periodic path chains its steps (64-wide, 62-wide and 2-D stage kernels); one-stage methods have nothing to fold
    assert ra == 2 * (4 if chained else 5)
    assert torch.equal(a, b)
    assert bool(torch.isfinite(a).all()) and diva < 1e-9


def test_graph_is_rebuilt_when_the_step_size_changes_and_off_switch_wins(ins):
    from ins_amd import _lib

    sp = ins.Setup(x=(np.linspace(0.0, 1.0, 33),) * 3, Re=500.0)
    ps = ins.psolver_spectral(sp)
    m = ins.RKMethods.RK44()
    u0 = ins.random_field(sp, kp=3, seed=5, psolver=ps)

    def run(**opts):
        with _lib.options(**opts):
            cache = ins.ode_method_cache(m, sp, ps)
            st = ins.create_stepper(m, setup=sp, psolver=ps, u=u0.clone(), t=0.0)
            st = ins.timesteps_(m, st, 2e-3, 5, cache=cache)
            st = ins.timesteps_(m, st, 1e-3, 5, cache=cache)  # another Δt: the kernel arguments of the captured step no longer apply
            st = ins.timesteps_(m, st, 1e-3, 2, cache=cache)  # too short for a replay
            torch.cuda.synchronize()
            return st.u.clone(), _replays(cache)

    a, ra = run(INS_STEP_GRAPH=1)
    b, rb = run(INS_STEP_GRAPH=1, INS_DISABLE_STEP_GRAPH=1)
    assert ra == 6 and rb == 0 and torch.equal(a, b)  # chained: 3 replays in each call of 5 (first and last step direct), none in the call of 2

"""Communication behind the C ABI (`ins_comm_*`, csrc/ins_comm.hip — RCCL loaded with dlopen, no torch.distributed involved), exercised
with ONE-rank communicators on the one GPU of the test box:

  * ins_comm_unique_id / ins_comm_create / ins_comm_create_local / ins_comm_rank / ins_comm_destroy;
  * ins_comm_sendrecv_f64: two messages to the same peer arrive in posting order (the P = 2 situation: lower and upper neighbour are one peer);
  * ins_halo_exchange_f64 on a slab grid = the periodic z wrap of apply_bc_u! (boundary_conditions.jl:276-288), bit for bit;
  * ins_halo_exchange_p_f64, ins_ztri_allgather_f64 (direct and ring), ins_comm_allreduce_f64, ins_comm_alltoall_f64;
  * the whole SlabStepper with every exchange through these entry points (`AbiSlabComm(loopback=True)`, asynchronous ones on its side
    stream) against the same stepper with local copies (bitwise) and against the single-GPU fused path (<= 1e-11).
More than one rank cannot run on this box (one GPU; RCCL refuses two ranks on one device)."""
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


def test_comm_handles_and_raw_exchanges(ins):
    from ins_amd import _lib

    lib = _lib.load()
    comm = ins.AbiSlabComm(1, 0, ins.AbiSlabComm.unique_id(), loopback=True)
    r, n = C.c_int(-1), C.c_int(-1)
    _lib.call("ins_comm_rank", comm._h, C.byref(r), C.byref(n))
    assert (r.value, n.value) == (0, 1)
    dev = comm.device
    a0 = torch.arange(0, 5000, dtype=torch.float64, device=dev)
    a1 = -torch.arange(0, 7000, dtype=torch.float64, device=dev)
    r0, r1 = torch.zeros_like(a0), torch.zeros_like(a1)
    comm.exchange([(a0, 0), (a1, 0)], [(r0, 0), (r1, 0)])
    torch.cuda.synchronize()
    assert torch.equal(r0, a0) and torch.equal(r1, a1)
    r0.zero_(); r1.zero_()
    for h in comm.exchange_async([(a0, 0), (a1, 0)], [(r0, 0), (r1, 0)]):
        h.wait()
    assert float(r0.sum()) == float(a0.sum()) and float(r1.sum()) == float(a1.sum())  # consumer on the current stream after wait()
    out = torch.zeros_like(a0)
    comm.all_gather(out, a0)
    assert torch.equal(out, a0)
    out.zero_()
    _lib.call("ins_ztri_allgather_f64", comm._h, C.c_void_p(a0.data_ptr()), C.c_void_p(out.data_ptr()), a0.numel(), 0, None)  # ring form
    torch.cuda.synchronize()
    assert torch.equal(out, a0)
    out.zero_()
    comm.all_to_all_async(out, a0, 1).wait()
    assert torch.equal(out, a0)
    t = torch.tensor([2.5, -1.0], dtype=torch.float64, device=dev)
    for op in ("sum", "max", "min"):
        assert torch.equal(comm.allreduce_(t.clone(), op), t)
    comm.barrier()
    # error paths come back as codes with a message
    assert lib.ins_comm_sendrecv_f64(None, 0, None, None, None, 0, None, None, None, None) == -1
    assert lib.ins_comm_allreduce_f64(comm._h, C.c_void_p(t.data_ptr()), 2, 7, None) == -1
    # single-process, n-device form (ncclCommInitAll) with n = 1
    hs = (C.c_void_p * 1)()
    _lib.call("ins_comm_create_local", 1, None, hs)
    _lib.call("ins_comm_rank", hs[0], C.byref(r), C.byref(n))
    assert (r.value, n.value) == (0, 1)
    _lib.call("ins_comm_group_begin")  # the bracket a single host thread puts around a round of per-communicator calls (nests with the calls' own groups)
    _lib.call("ins_comm_allreduce_f64", hs[0], C.c_void_p(t.data_ptr()), 2, 0, None)
    _lib.call("ins_comm_group_end")
    torch.cuda.synchronize()
    assert lib.ins_comm_destroy(hs[0]) == 0 and lib.ins_comm_destroy(None) == 0


def test_halo_exchange_equals_periodic_wrap(ins):
    """One rank owns the whole periodic z range: the exchanged ghost planes must be the periodic images apply_bc_u! writes."""
    n = (66, 12, 10)
    lay = ins.SlabLayout(n, 1, 0)
    K = ins.HipSlabKernels(lay, Re=100.0)
    comm = ins.AbiSlabComm(1, 0, ins.AbiSlabComm.unique_id(), loopback=True)
    u = K.vector()
    u.copy_(torch.randn(u.shape, dtype=torch.float64, device=u.device))
    want = u.clone()
    want[:, :, 0, :] = want[:, :, lay.nzl, :]
    want[:, :, lay.nzl + 1, :] = want[:, :, 1, :]
    comm.halo_u(K.setup, u)
    torch.cuda.synchronize()
    assert torch.equal(u, want)
    v = u.clone()
    v[:, :, 0, 2] = 7.0
    comm.halo_u(K.setup, v, comps=(2,), down_only=True)
    torch.cuda.synchronize()
    assert torch.equal(v, want)
    # extended pressure buffer
    plane = n[0] * n[1]
    pX = torch.randn(plane * (lay.nzl + 3), dtype=torch.float64, device=u.device)
    w = pX.clone()
    w[0:plane] = w[plane * lay.nzl : plane * (lay.nzl + 1)]
    w[plane * (lay.nzl + 1) : plane * (lay.nzl + 3)] = w[plane : plane * 3]
    from ins_amd import _lib

    _lib.call("ins_halo_exchange_p_f64", comm._h, C.c_void_p(pX.data_ptr()), plane, lay.nzl, None)
    torch.cuda.synchronize()
    assert torch.equal(pX, w)


@pytest.mark.parametrize("n", [(128, 16, 32), (66, 16, 24)])
def test_slab_stepper_through_the_comm_abi(ins, n):
    x = tuple(np.linspace(0.0, 1.0, ni + 1) for ni in n)
    sp = ins.Setup(x=x, Re=800.0)
    ps = ins.psolver_spectral(sp)
    rng = np.random.default_rng(5)
    u0 = ins.from_numpy(sp, np.asfortranarray(0.3 * rng.standard_normal(sp.grid.N + (3,))))
    ins.apply_bc_u_(u0, 0.0, sp)
    ins.project_(u0, sp, ps, ins.scalarfield(sp))
    ins.apply_bc_u_(u0, 0.0, sp)
    u0_h = ins.to_numpy(u0)
    (uref, _, _), _ = ins.solve_unsteady(setup=sp, tlims=(0.0, 0.03), ustart=u0, psolver=ps, Δt=0.01)
    uref = ins.to_numpy(uref)
    del ps, sp, u0
    import gc

    gc.collect()
    lay = ins.SlabLayout(n, 1, 0)
    K = ins.HipSlabKernels(lay, Re=800.0)
    for zsolve, chunks in (("tridiag", 1), ("fft", 3)):
        outs = []
        for loop in (True, False):
            comm = ins.AbiSlabComm(1, 0, ins.AbiSlabComm.unique_id(), loopback=loop)  # an id serves ONE communicator
            st = ins.SlabStepper(ins.RKMethods.RK44(), lay, K, comm, chunks=chunks, zsolve=zsolve)
            assert st.zsolve == zsolve
            u = K.from_global(u0_h)
            st.steps_(u, 0.01, 3)
            torch.cuda.synchronize()
            outs.append(ins.to_numpy(u))
            assert st.max_abs_divergence(u) < 1e-10
            del st, comm
        assert np.array_equal(outs[0], outs[1]), f"{zsolve}: RCCL through the C ABI differs from local copies"
        err = float(np.sqrt(np.sum((outs[0] - uref) ** 2)) / np.sqrt(np.sum(uref**2)))
        assert err < 1e-11, (zsolve, err)


def test_slab_cg_one_rank_equals_single_domain_cg(ins, oracle):
    """psolver_cg on a slab grid (z sides HaloBC) with a communicator: the dots / norm go through ins_comm all-reduces and the ghost planes
    of the search direction through ins_halo_exchange_scalar_f64 (RCCL, one rank = the whole periodic z range).  Same iteration count and
    the same solution as the single-domain CG on the periodic box; and against the oracle's CG."""
    from ins_amd import _lib
    from tests import fixtures as fx

    o = oracle
    n = (16, 12, 10)
    x = tuple(np.linspace(0.0, 1.0, ni + 1) for ni in n)
    so = o.make_setup(x, Re=100.0)
    sp = ins.Setup(x=x, Re=100.0)
    g = so.grid
    u_h = o.apply_bc_u(fx.randn_field(g.N + (3,), 8), 0.0, so)
    f = o.scalewithvolume(o.divergence(u_h, so), so)
    ip = tuple(slice(lo, hi) for lo, hi in g.Ip)
    single = ins.psolver_cg(sp)
    want = ins.to_numpy(ins.poisson(single, ins.from_numpy(sp, f)))
    it0, _ = single.last_info()
    lay = ins.SlabLayout(n, 1, 0)
    K = ins.HipSlabKernels(lay, Re=100.0)
    comm = ins.AbiSlabComm(1, 0, ins.AbiSlabComm.unique_id(), loopback=True)
    slab = ins.psolver_cg(K.setup)
    _lib.call("ins_poisson_cg_set_comm", slab.handle, comm._h)
    got = ins.to_numpy(ins.poisson(slab, ins.from_numpy(K.setup, f)))
    it1, _ = slab.last_info()
    assert it1 == it0
    err = float(np.sqrt(np.sum((got[ip] - want[ip]) ** 2)) / np.sqrt(np.sum(want[ip] ** 2)))
    assert err < 1e-12
    info = {}
    ref = o.poisson(o.psolver_cg(so, info=info), f)
    assert abs(it1 - info["iterations"]) <= 2
    assert float(np.sqrt(np.sum((got[ip] - ref[ip]) ** 2)) / np.sqrt(np.sum(ref[ip] ** 2))) < 1e-6

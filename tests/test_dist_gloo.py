"""World-size-2 rehearsal of the z-slab path on CPU (gloo): the product's `SlabStepper` bookkeeping and
torch.distributed exchanges, with oracle-backed local kernels injected, must reproduce the single-domain
oracle to round-off.  (The HIP slab kernels themselves are checked in tests/test_gpu_slab.py.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, nsteps, method_name, out_dir, chunks=1, own=False, zsolve="fft"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ins_amd as ins
        from oracle import ins_oracle as o
        from tests.slab_cpu_kernels import OracleSlabKernels

        x = tuple(np.linspace(0.0, 1.0, ni + 1) for ni in n)
        so = o.make_setup(x, Re=500.0)
        u0 = o.random_field(so, kp=2, seed=7)
        lay = ins.SlabLayout(n, world, rank)
        K = OracleSlabKernels(lay, Re=500.0, own=own)
        comm = ins.SlabComm(group2=dist.new_group(ranks=list(range(world))) if chunks > 1 else None)
        method = getattr(ins.RKMethods, method_name)()
        st = ins.SlabStepper(method, lay, K, comm, chunks=chunks, zsolve=zsolve)
        assert st.zsolve == zsolve
        assert len(st.chunks) == min(chunks, lay.kxn) and st.packed == own and st.inkernel == (own and len(method.b) > 1)
        u = K.from_global(u0)
        st.steps_(u, 0.01, nsteps)
        div = st.max_abs_divergence(u)
        np.save(os.path.join(out_dir, f"u_{rank}.npy"), u.numpy())
        np.save(os.path.join(out_dir, f"div_{rank}.npy"), np.array([div]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("method_name,chunks,own,zsolve,world",
                         [("RK44", 1, False, "fft", 2), ("Wray3", 1, False, "fft", 2), ("FE11", 1, False, "fft", 2), ("RK44", 3, False, "fft", 2),
                          ("RK44", 1, True, "fft", 2), ("RK44", 3, True, "fft", 2), ("Wray3", 2, True, "fft", 2), ("FE11", 1, True, "fft", 2),
                          ("RK44", 1, False, "tridiag", 2), ("RK44", 1, True, "tridiag", 2), ("Wray3", 1, True, "tridiag", 2),
                          ("FE11", 1, False, "tridiag", 2), ("RK44", 1, True, "tridiag", 3), ("RK44", 1, True, "tridiag-p2p", 3)])
def test_slab_stepper_two_ranks_matches_single_domain(tmp_path, oracle, method_name, chunks, own, zsolve, world):
    """chunks > 1: the transposes pipelined over kx-chunks on two process groups; own: the packed-pass branch;
    zsolve = tridiag: no transposes, the z direction as distributed tridiagonal systems (one all-gather per solve)."""
    o = oracle
    n, nsteps = (12, 6 * (world // 2 + world % 2) if world == 3 else 8, 12), 2
    if zsolve.endswith("-p2p"):  # the interface gather as direct sends to every peer (the RCCL default) instead of the collective
        zsolve = "tridiag"
        os.environ["INS_SLAB_GATHER"] = "p2p"
    else:
        os.environ.pop("INS_SLAB_GATHER", None)
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, nsteps, method_name, str(tmp_path), chunks, own, zsolve), nprocs=world, join=True)
    x = tuple(np.linspace(0.0, 1.0, ni + 1) for ni in n)
    so = o.make_setup(x, Re=500.0)
    ps = o.psolver_spectral(so)
    u0 = o.random_field(so, kp=2, seed=7)
    st = o.solve_unsteady(so, (0.0, 0.01 * nsteps), u0, method=getattr(o, method_name)(), psolver=ps, dt=0.01)
    nzl = n[2] // world
    for r in range(world):
        got = np.load(tmp_path / f"u_{r}.npy")
        ks = [(r * nzl + k - 1) % n[2] + 1 for k in range(nzl + 2)]
        want = st["u"][:, :, ks, :]
        err = np.sqrt(np.sum((got - want) ** 2)) / np.sqrt(np.sum(want**2))
        assert err < 1e-12, (r, err)
        assert float(np.load(tmp_path / f"div_{r}.npy")[0]) < 1e-10


def test_slab_layout_rejects_indivisible():
    import ins_amd as ins

    with pytest.raises(ValueError):
        ins.SlabLayout((8, 8, 10), 4, 0)
    lay = ins.SlabLayout((256, 512, 512), 4, 3)
    assert (lay.nzl, lay.nyl, lay.z0, lay.prev, lay.next, lay.kxn) == (128, 128, 384, 2, 0, 129)

"""First contact with RCCL (torch.distributed backend "nccl") on the ONE GPU of the test box: a world-size-1 process group whose
exchanges are real RCCL calls to self (SlabComm(loopback=True)).

What this covers that the gloo rehearsals (tests/test_dist_gloo.py, tests/test_gpu_slab.py) cannot:
  * `init_process_group("nccl", device_id=...)`, communicator start-up, a second communicator (`new_group`);
  * grouped `batch_isend_irecv` where BOTH neighbours are the same peer — with one rank prev = next = self, the same matching
    situation as P = 2 (sends and receives to one peer pair up in posting order): a wrong order would swap the upper and the lower
    ghost plane;
  * stream ordering of the asynchronous exchanges (`exchange_async` / `all_to_all_async` + `wait()`) against the HIP kernels that
    produce and consume the planes: the whole SlabStepper (tridiagonal route and transpose route) through RCCL must equal the same
    stepper with local copies bit for bit, and the single-GPU fused path to 1e-11.
Every failure path raises in the spawned rank => mp.spawn raises => the test fails (no re-exec, bounded by timeouts)."""
import datetime
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, port, n, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=180))
    try:
        import ins_amd as ins

        assert dist.get_backend() == "nccl"
        g2 = dist.new_group(ranks=[0])
        comm = ins.SlabComm(group2=g2, loopback=True)
        assert comm.backend == "nccl" and comm.world == 1
        # ---- raw exchanges: two messages to the same peer must arrive in posting order
        a0 = torch.arange(0, 4096, dtype=torch.float64, device=dev)
        a1 = -torch.arange(0, 8192, dtype=torch.float64, device=dev)
        r0, r1 = torch.zeros_like(a0), torch.zeros_like(a1)
        comm.exchange([(a0, 0), (a1, 0)], [(r0, 0), (r1, 0)])
        torch.cuda.synchronize()
        assert torch.equal(r0, a0) and torch.equal(r1, a1)
        r0.zero_(); r1.zero_()
        for req in comm.exchange_async([(a0, 0), (a1, 0)], [(r0, 0), (r1, 0)]):
            req.wait()
        s0 = float(r0.sum()) + float(r1.sum())  # consumer on the current stream after wait()
        assert s0 == float(a0.sum()) + float(a1.sum())
        # ---- collectives of the two Poisson routes
        out = torch.zeros(4096, dtype=torch.float64, device=dev)
        os.environ["INS_SLAB_GATHER"] = "collective"
        comm.all_gather(out, a0)
        assert torch.equal(out, a0)
        os.environ["INS_SLAB_GATHER"] = "p2p"
        out.zero_()
        comm.all_gather(out, a0)
        for req in comm.all_gather_async(out, a0):
            req.wait()
        assert torch.equal(out, a0)
        del os.environ["INS_SLAB_GATHER"]
        out.zero_()
        comm.all_to_all(out, a0)
        assert torch.equal(out, a0)
        out.zero_()
        comm.all_to_all_async(out, a0, 1).wait()
        assert torch.equal(out, a0)
        t = torch.tensor([3.5], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t) == 3.5
        # ---- the whole stepper through RCCL vs local copies vs the single-GPU path
        x = tuple(np.linspace(0.0, 1.0, ni + 1) for ni in n)
        sp = ins.Setup(x=x, Re=800.0, device=dev)
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
        K = ins.HipSlabKernels(lay, Re=800.0, device=dev)
        res = {}
        for zsolve, chunks in (("tridiag", 1), ("fft", 3)):
            outs = []
            for loop in (True, False):
                c = ins.SlabComm(group2=g2, loopback=loop)
                st = ins.SlabStepper(ins.RKMethods.RK44(), lay, K, c, chunks=chunks, zsolve=zsolve)
                assert st.zsolve == zsolve
                u = K.from_global(u0_h)
                st.steps_(u, 0.01, 3)
                torch.cuda.synchronize()
                outs.append(ins.to_numpy(u))
                res[(zsolve, loop)] = st.max_abs_divergence(u)
                del st
            assert np.array_equal(outs[0], outs[1]), f"{zsolve}: RCCL loopback differs from local copies"
            err = float(np.sqrt(np.sum((outs[0] - uref) ** 2)) / np.sqrt(np.sum(uref**2)))
            assert err < 1e-11, (zsolve, err)
            assert res[(zsolve, True)] < 1e-10
        np.save(os.path.join(out_dir, "ok.npy"), np.array([1.0]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("n", [(128, 16, 32), (66, 16, 24)])  # own-FFT box with the 64-wide correcting stage kernel; rocFFT box
def test_rccl_world1_loopback(tmp_path, n):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    mp.spawn(_worker, args=(_free_port(), n, str(tmp_path)), nprocs=1, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "ok.npy"))

"""Multi-GPU leg of bench.py: z-slab decomposition, one rank per GPU over RCCL (torch.distributed "nccl").

Headline line (`value`): weak scaling, every rank holds 256^3 cells — the global box is the TGV on [0,1]^3 with
(nx, ny, nz) = (256, 256, 512) at N=2, (256, 512, 512) at N=4 and 512^3 at N=8 (= BASELINE configs[3]).
`strong_512`: the strong-scaling leg the north star asks for — the SAME 512^3 box at every N (z-slabs of 512/N planes);
bench.py's N = 1 line carries the single-GPU point of that curve."""
import importlib
import json
import os
import time

import numpy as np
import torch
import torch.distributed as dist

GLOBAL_GRIDS = {1: (256, 256, 256), 2: (256, 256, 512), 4: (256, 512, 512), 8: (512, 512, 512)}
STRONG_GRID = (512, 512, 512)  # BASELINE configs[3]: the same box at every N


def tgv_local(lay, L=(1.0, 1.0, 1.0)):
    """TGV initial condition (examples/TaylorGreenVortex3D.jl:30-37) evaluated directly on this rank's padded slab
    (face positions xu[α]); periodic in all directions, so ghosts are just the same formula."""
    nx, ny, nz = lay.n
    h = [L[a] / lay.n[a] for a in range(3)]
    xi = (np.arange(-1, nx + 1) + 0.5) * h[0]
    yj = (np.arange(-1, ny + 1) + 0.5) * h[1]
    zk = (np.arange(-1, lay.nzl + 1) + lay.z0 + 0.5) * h[2]
    xf, yf = xi + 0.5 * h[0], yj + 0.5 * h[1]
    X, Y, Z = xf[:, None, None], yj[None, :, None], zk[None, None, :]
    u = np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y) * np.sin(2 * np.pi * Z) / 2
    X, Y = xi[:, None, None], yf[None, :, None]
    v = -np.cos(2 * np.pi * X) * np.sin(2 * np.pi * Y) * np.sin(2 * np.pi * Z) / 2
    out = np.zeros((nx + 2, ny + 2, lay.nzl + 2, 3), order="F")
    out[..., 0] = u
    out[..., 1] = v
    return out


def _rehearsal_kernels():
    """INS_BENCH_REHEARSAL_KERNELS="module:Class" (honoured with INS_BENCH_BACKEND=gloo only): rank-local kernels injected by a TEST so that the launch path,
    the rank bookkeeping and the JSON line can be exercised on a machine without a GPU (tests/test_bench_launch.py injects the oracle-backed CPU kernels of
    tests/slab_cpu_kernels.py).  The line then says `"rehearsal"` and its numbers mean nothing; no measured run ever takes this branch."""
    spec = os.environ.get("INS_BENCH_REHEARSAL_KERNELS")
    if not spec or os.environ.get("INS_BENCH_BACKEND") != "gloo":
        return None
    mod, cls = spec.split(":")
    return getattr(importlib.import_module(mod), cls)


def _sync(dev):
    if dev.type == "cuda":
        torch.cuda.synchronize()


def run_slab(ins, n, dt, steps, warmup, dev, backend, profile=True):
    """Warm-up + timed `steps` chained RK44 steps of the TGV on the global box `n`, this rank's z-slab.  Returns (max-over-ranks seconds,
    diagnostics).  Barrier + device synchronisation on both sides of the timed region."""
    world, rank = dist.get_world_size(), dist.get_rank()
    lay = ins.SlabLayout(n, world, rank)
    reh = _rehearsal_kernels()
    K = reh(lay, Re=1000.0, own=True) if reh else ins.HipSlabKernels(lay, Re=1000.0, device=dev)
    profile = profile and reh is None
    # zsolve (INS_SLAB_ZSOLVE): "tridiag" (default for > 1 rank) needs no transposes; "fft" pipelines them over kx-chunks on a second
    # communicator so that back-transposes of finished chunks run beside forward ones (full-duplex xGMI links)
    zs = os.environ.get("INS_SLAB_ZSOLVE") or ("tridiag" if world > 1 else "fft")
    nchunks = int(os.environ.get("INS_SLAB_CHUNKS", "4"))
    g2 = dist.new_group(ranks=list(range(world))) if (nchunks > 1 and zs == "fft") else None
    if os.environ.get("INS_BENCH_COMM") == "abi":
        # every exchange through the library's own RCCL entry points (ins_comm_*, csrc/ins_comm.hip) — the route a Julia host takes;
        # torch.distributed only ships the 128-byte communicator id and times the run
        ids = [ins.AbiSlabComm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        comm = ins.AbiSlabComm(world, rank, ids[0], device=dev)
    else:
        comm = ins.SlabComm(group2=g2)
    st = ins.SlabStepper(ins.RKMethods.RK44(), lay, K, comm, chunks=nchunks)
    u = K.vector()
    u.copy_(torch.from_numpy(np.ascontiguousarray(tgv_local(lay))).to(dev))
    st.project_(u)  # velocityfield(...; doproject = true)  (initializers.jl:38-42)
    st.halo_u(u)
    # the fixed-Δt loop of solve_unsteady as on one GPU (bench.py): K steps per call, u valid before and after
    if warmup:
        st.steps_(u, dt, warmup)
    dist.barrier()
    _sync(dev)
    t0 = time.perf_counter()
    st.steps_(u, dt, steps)
    _sync(dev)
    dist.barrier()
    t1 = time.perf_counter()
    el = torch.tensor([t1 - t0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    info = {"zsolve": st.zsolve, "kx_chunks": len(st.chunks), "nzl": lay.nzl, "stage_ms": 0.0, "nstage": 0, "stage_bytes": 0.0,
            "cells_rank": float(n[0]) * n[1] * lay.nzl}
    if profile:
        # roofline of the dominant kernel (the correcting stage kernel, K1 + K6 + the previous projection's gradient-subtract): two more
        # steps with HIP events around its launches on the stream it runs on; outside the timed region
        K.prof = []
        st.steps_(u, dt, 2)
        _sync(dev)
        prof, K.prof = K.prof, None
        info["stage_ms"] = sum(e0.elapsed_time(e1) for e0, e1, _ in prof)
        info["nstage"] = sum(1 for _, _, b in prof if b)
        info["cells_rank"] = float(n[0]) * n[1] * lay.nzl
        info["stage_bytes"] = sum(b for _, _, b in prof) * info["cells_rank"]
    info["div"] = st.max_abs_divergence(u)
    info["finite"] = bool(torch.isfinite(u).all())
    del st, K, u
    if dev.type == "cuda":
        torch.cuda.empty_cache()
    return float(el), info


def run_distributed(args, ins):
    world = int(os.environ["WORLD_SIZE"])
    rank = int(os.environ["RANK"])
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("INS_BENCH_BACKEND", "nccl")  # "gloo": rehearsal with several ranks on one GPU
    ndev = torch.cuda.device_count()
    rehearsal = _rehearsal_kernels() is not None
    if rehearsal:
        dev = torch.device("cpu")
    else:
        if backend == "nccl" and local_rank >= ndev:
            raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({ndev} visible); one rank per GPU")
        dev = torch.device("cuda", local_rank if backend == "nccl" else local_rank % ndev)
        torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
        # the communicator's size as RCCL itself reports it: a sum of ones over the ranks, reduced on the devices
        ones = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(ones)
        rccl_ranks = int(ones.item())
        assert rccl_ranks == world, f"RCCL communicator spans {rccl_ranks} ranks, expected {world}"
    else:
        dist.init_process_group(backend)
        rccl_ranks = None  # gloo rehearsal: no RCCL communicator exists
    n = GLOBAL_GRIDS.get(world)
    if n is None:
        n = (args.n, args.n, args.n * world)
    elif args.n != 256:  # --n: cells per direction and GPU (default 256); the weak-scaling boxes keep their aspect
        n = tuple(v * args.n // 256 for v in n)
    el, info = run_slab(ins, n, 1e-3, args.steps, args.warmup, dev, backend)
    stage_ms, nstage, cells_rank, stage_bytes, div, finite = (info[k] for k in ("stage_ms", "nstage", "cells_rank", "stage_bytes", "div", "finite"))
    # strong-scaling leg (north star / BASELINE configs[3]): the SAME 512^3 box on every N, z-slabs of 512/N planes
    strong = None
    n5 = STRONG_GRID if not os.environ.get("INS_BENCH_STRONG_GRID") else tuple(int(v) for v in os.environ["INS_BENCH_STRONG_GRID"].split("x"))
    if not os.environ.get("INS_BENCH_SKIP_STRONG_512") and n5[2] % world == 0 and n5[1] % world == 0 and n5[2] // world >= 2:
        ssteps, swarm = max(2, min(args.steps, 10)), min(args.warmup, 2)
        el5, info5 = run_slab(ins, n5, 2.5e-4, ssteps, swarm, dev, backend, profile=False)
        ms5 = el5 * 1e3 / ssteps
        # the N = 1 point of the same curve on the same node: rank 0 runs the single-GPU path on the same box while the other ranks wait at the barrier below
        n1 = None
        if rank == 0 and not rehearsal and not os.environ.get("INS_BENCH_SKIP_N1_REF") and n5[0] == n5[1] == n5[2]:
            from bench import strong_single_gpu

            n1 = strong_single_gpu(ins, dev, n=n5[0], steps=ssteps, warmup=swarm)["ms_per_step"]
        dist.barrier()
        strong = {"workload": f"TaylorGreenVortex3D {n5[0]}x{n5[1]}x{n5[2]} periodic fp64, RK44 + distributed spectral Poisson, dt=2.5e-4 (BASELINE configs[3]), "
                              "total work fixed over N", "scaling": "strong", "n_gpus": world, "grid": list(n5), "planes_per_rank": info5["nzl"],
                  "steps": ssteps, "warmup": swarm, "ms_per_step": ms5, "value": float(n5[0]) * n5[1] * n5[2] / (ms5 * 1e-3) / 1e6, "unit": "M cell-updates/s",
                  "max_abs_div_times_dx": info5["div"] / n5[0], "finite": info5["finite"], "zsolve": info5["zsolve"],
                  "speedup_vs_n1_hint": {"n1_ms_per_step": n1, "speedup": (n1 / ms5) if n1 else None,
                                         "source": "single-GPU path on the same box, timed by rank 0 on its GPU right after the slab run (same steps / warm-up); "
                                                   "the driver's own N = 1 run is the figure to divide by" if n1 else "not measured in this run: divide by the N = 1 line's strong_512.ms_per_step"}}
    if rank == 0:
        ms = el * 1e3 / args.steps
        cells = float(n[0]) * n[1] * n[2]
        out = {
            "metric": "M lattice-cell updates/sec (RK4 step incl. Poisson), 3D TGV fp64",
            "value": cells / (ms * 1e-3) / 1e6,
            "unit": "M cell-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"TaylorGreenVortex3D {n[0]}x{n[1]}x{n[2]} periodic fp64, RK44 + distributed spectral Poisson, dt=1e-3, Re=1e3",
                       "grid": list(n), "decomposition": f"z-slabs x{world} ({args.n}^3 cells per GPU), RCCL halo planes + "
                                        + ("one all-gather of interface values per solve (distributed tridiagonal z solve)" if info["zsolve"] == "tridiag"
                                           else "all-to-all transposes around the z-FFT")},
            "roofline": {"kernel": "k_flux64 CORR (slab): momentum-RHS stencil + RK stage combination + in-register pressure correction, per rank (rank 0)",
                         "bound": "hbm", "achieved": stage_bytes / (stage_ms * 1e-3) / 1e9 if stage_ms > 0 else None, "peak": 8000.0, "unit": "GB/s",
                         "frac": stage_bytes / (stage_ms * 1e-3) / 1e9 / 8000.0 if stage_ms > 0 else None, "traffic": None,
                         "bytes_per_cell": stage_bytes / cells_rank / max(nstage, 1), "avg_launch_ms": stage_ms / max(nstage, 1), "launches": nstage,
                         "note": "algorithmic bytes of the stages measured (2 chained RK44 steps after the timed region) / HIP-event time of their launches "
                                 "(interior + boundary plane ranges summed per stage); N = 1 has the PMC traffic figure"},
            "strong_512": strong,
            "rccl_ranks": rccl_ranks,
            "rehearsal": ("CPU stand-in kernels injected by a test (INS_BENCH_REHEARSAL_KERNELS): launch-path check only, the numbers mean nothing" if rehearsal else None),
            "check": {"max_abs_div_times_dx": div * (1.0 / n[0]), "finite": finite, "backend": backend, "zsolve": info["zsolve"], "kx_chunks": info["kx_chunks"]},
        }
        print(json.dumps(out))
    dist.destroy_process_group()

#!/usr/bin/env python3
"""Headline benchmark: M lattice-cell updates / s of one RK4 step incl. the 4 Poisson projections,
3-D Taylor-Green vortex, fp64 (BASELINE.json `metric`).

  python bench.py --gpus 1 --steps K --warmup W          (N = 1: TGV3D 256^3 = BASELINE configs[1])
  python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N rank processes — `self_launch` — before anything touches the GPU)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (the same ranks started by the caller; z-slab decomposition)

Prints ONE JSON line on rank 0.  `roofline` is the dominant kernel of the step — the momentum-RHS stencil with the
RK stage combination fused behind it (K1+K6) — timed live with HIP events recorded on the step's own
stream inside the timed region; its algorithmic bytes per launch are the compulsory reads/writes of each RK stage
(RK44: 72/104/104/152 B per cell — stages >= 2 read p as well and apply the previous projection in registers).  `roofline_k1` is the plain momentum-RHS stencil (48 B/cell) timed
after the run on the final state.  `cpu_baseline` is the CPU oracle (a numpy restatement, NOT Julia) timed on a
bounded sample on this host.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec HBM3E peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
K1_BYTES_PER_CELL = 48.0  # read u (3 x 8 B) + write F (3 x 8 B)                       SURVEY.md §8d


def tgv3d(al, x, y, z):
    """examples/TaylorGreenVortex3D.jl:30-37"""
    if al == 0:
        return np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y) * np.sin(2 * np.pi * z) / 2
    if al == 1:
        return -np.cos(2 * np.pi * x) * np.sin(2 * np.pi * y) * np.sin(2 * np.pi * z) / 2
    return 0 * (x + y + z)


def host_cores():
    """Threads this process may really use: cgroup CPU quota if set (a 1-GPU box gets a 16-CPU share of a big host),
    else the affinity mask, capped at 16."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(n=256, budget_s=20.0):
    """Time the CPU restatement (oracle/c: C + OpenMP stencil passes in the reference's unfused pass structure, scipy
    pocketfft for the FFTs; NOT Julia) on the bench's own grid (TGV3D n^3, same dt), a bounded number of steps, on this host's cores."""
    from oracle import ins_oracle as o
    from oracle.c_port import CPort

    cores = host_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)
    x = (np.linspace(0.0, 1.0, n + 1),) * 3
    so = o.make_setup(x, Re=1000.0)
    ps = o.psolver_spectral(so)
    u = np.asfortranarray(o.velocityfield(so, o.tgv3d_ufunc, 0.0, psolver=ps))
    m = o.RK44()
    cache = o.ode_method_cache(m, so)
    port = CPort(so, workers=cores)
    port.timestep_(m, u, 1e-3, cache)  # warm-up (page faults of the cache arrays, pocketfft plans)
    t0 = time.perf_counter()
    steps = 0
    while steps < 2 or (time.perf_counter() - t0 < budget_s and steps < 200):
        port.timestep_(m, u, 1e-3, cache)
        steps += 1
    dt = time.perf_counter() - t0
    return {
        "value": n**3 * steps / dt / 1e6,
        "unit": "M cell-updates/s",
        "cores": cores,
        "kind": "port",
        "sample": f"C+OpenMP CPU restatement (not Julia; reference's unfused pass structure, pocketfft), TGV3D {n}^3 fp64 (the bench grid), "
                  f"{steps} RK4 steps in {dt:.1f} s on {cores} threads after 1 warm-up step",
    }


def stage_bytes_per_cell(method, steps, chained):
    """Compulsory bytes per cell of the fused stage kernel, stage by stage           step_explicit_runge_kutta.jl:35-38
         k-basis:              R u_in + (R ustart, not for stage 1) + R k_j (non-zero a_ij) + W u* + (W k_i when a later stage needs it)
         stage-velocity basis: R u_in + (R ustart) + R V_m (non-zero β_im) + W u*       (csrc/ins_rk.hip; no stage force is stored)
    Returns (list per stage, K4 bytes per cell and step)."""
    A = np.asarray(method.A, dtype=float)
    ns = len(method.b)
    inkernel = ns > 1 and not os.environ.get("INS_DISABLE_INKERNEL_CORR")
    vbasis = inkernel and not os.environ.get("INS_RK_KEEP_K") and all(A[i, i] != 0.0 for i in range(ns))
    stage_bytes = []
    for i in range(ns):
        inkernel_p = 8 if (i > 0 and inkernel) else 0  # stages >= 2 also read p
        if vbasis:
            beta = np.linalg.solve(A[:i, :i].T, A[i, :i]) if i else np.zeros(0)
            nk, wk = int(np.count_nonzero(beta[: max(i - 1, 0)])), False  # β[i-1] multiplies the stencil input itself: taken from registers
        else:
            nk = sum(1 for j in range(i) if A[i, j] != 0.0)
            wk = any(A[i2, i] != 0.0 for i2 in range(i + 1, ns))
        stage_bytes.append(24 * (1 + (1 if i > 0 else 0) + nk + 1 + (1 if wk else 0)) + inkernel_p)
    k4_bytes = 64.0  # final gradient-subtract + ghost images + padded p: once per step, or once per call when the steps are chained
    if chained and vbasis and steps > 1:
        # chained steps: the first stage of steps 2..K also reads p and stores the corrected start field (+8 + 24 B), and K4 runs once
        stage_bytes[0] += 32.0 * (steps - 1) / steps
        k4_bytes /= steps
    return stage_bytes, k4_bytes


def committed_traffic(name, launches=None, chained=True):
    """HBM bytes per launch of the stage kernel from a committed `rocprofv3 --pmc` result under profiles/ (FETCH_SIZE and WRITE_SIZE in separate passes, FETCH
    doubled: gfx950 rule; tools/run_r03_profiles.sh) — PMC counters cannot be read inside the timed run.  The file holds the per-launch averages of the
    correcting form and of the first-stage form; they are weighted with the launch mix of THIS run's timed region (chained: one first-stage launch per call,
    every other launch corrects; single steps: one first-stage launch per step)."""
    tfile = os.path.join(ROOT, "profiles", name)
    try:
        pk = json.load(open(tfile))["per_kernel"]
        if launches and "stage kernel, corr" in pk and "stage kernel, first" in pk:
            nfirst = 1 if chained else max(launches // 4, 1)
            return (pk["stage kernel, corr"]["hbm_total_GB"] * (launches - nfirst) + pk["stage kernel, first"]["hbm_total_GB"] * nfirst) / launches * 1e9
        return pk["stage kernel, RK44 step average"]["hbm_total_GB"] * 1e9
    except Exception:
        return None


def strong_single_gpu(ins, dev, n=512, steps=10, warmup=2):
    """The strong-scaling workload of BASELINE configs[3] (TGV3D 512^3, RK44 + spectral Poisson, dt = 2.5e-4) on ONE GPU: the N = 1 point of the
    1/2/4/8-GPU curve (`bench.py --gpus N` runs the same box on z-slabs of 512/N planes and reports the same object)."""
    import torch

    setup = ins.Setup(x=(np.linspace(0.0, 1.0, n + 1),) * 3, Re=1000.0, device=dev)
    ps = ins.psolver_spectral(setup)
    u = ins.velocityfield(setup, tgv3d, 0.0, psolver=ps)
    method = ins.RKMethods.RK44()
    cache = ins.ode_method_cache(method, setup, ps)
    st = ins.create_stepper(method, setup=setup, psolver=ps, u=u, t=0.0)
    dt = 2.5e-4
    if warmup:
        st = ins.timesteps_(method, st, dt, warmup, cache=cache)
    import ctypes as C

    ins._lib.call("ins_rk_profile_enable", cache.handle, 1)  # HIP events around the stage-kernel launches, on their stream
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = ins.timesteps_(method, st, dt, steps, cache=cache)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    k_ms, k_n = C.c_double(), C.c_int64()
    ins._lib.call("ins_rk_profile_read", cache.handle, C.byref(k_ms), C.byref(k_n))
    ins._lib.call("ins_rk_profile_enable", cache.handle, 0)
    sb, _ = stage_bytes_per_cell(method, steps, True)
    avg_ms = k_ms.value / max(k_n.value, 1)
    gbs = float(np.mean(sb)) * float(n) ** 3 / (avg_ms * 1e-3) / 1e9
    roof = {"kernel": "k_flux64 FUSE[/CORR] (stage kernel: K1 + K6 + the previous projection's gradient-subtract in registers)", "bound": "hbm", "achieved": gbs,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": committed_traffic("r03_pmc_traffic_512.json", k_n.value) if n == 512 else None,
            "traffic_note": "B per launch from profiles/r03_pmc_traffic_512.json (rocprofv3 --pmc of `bench.py --n 512`, two passes)",
            "bytes_per_cell": float(np.mean(sb)), "bytes_per_cell_by_stage": sb, "avg_launch_ms": avg_ms, "launches": k_n.value}
    div = ins.max_abs_divergence(st.u, setup)
    assert div / n < 1e-10, f"{n}^3 state is not divergence-free: {div}"
    return {"roofline": roof, "workload": f"TaylorGreenVortex3D {n}^3 periodic fp64, RK44 + spectral Poisson, dt=2.5e-4 (BASELINE configs[3]), total work fixed over N",
            "scaling": "strong", "n_gpus": 1, "grid": [n, n, n], "steps": steps, "warmup": warmup, "ms_per_step": ms,
            "value": float(n) ** 3 / (ms * 1e-3) / 1e6, "unit": "M cell-updates/s", "max_abs_div_times_dx": div / n,
            "speedup_vs_n1_hint": {"n1_ms_per_step": ms, "speedup": 1.0, "source": "this run (N = 1)"}}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks as CHILD processes (torch.distributed.run, rendezvous on 127.0.0.1, a free
    port) — this parent never imports torch and never touches the GPU, so nothing is exec'ed from a GPU-initialised process —, relay rank 0's single JSON line and
    exit non-zero if any rank failed, no line came back or the run exceeded INS_BENCH_LAUNCH_TIMEOUT seconds (default 1500; the whole process group is killed)."""
    import signal
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__),
           # spelled out (not sys.argv): torch.distributed.run's parser claims abbreviations such as `--n` even behind the script name
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup), "--cells", str(args.n)]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL across processes)
    env["INS_BENCH_SELF_LAUNCHED"] = "1"
    limit = float(os.environ.get("INS_BENCH_LAUNCH_TIMEOUT", "1500"))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    try:
        stdout, _ = proc.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        proc.wait()
        print(f"bench.py: the {args.gpus} ranks did not finish within {limit:.0f} s; killed", file=sys.stderr)
        sys.exit(124)
    line = None
    for ln in stdout.splitlines():
        if ln.startswith("{"):
            try:
                if "metric" in json.loads(ln):
                    line = ln
            except ValueError:
                pass
    if proc.returncode != 0 or line is None:
        sys.stderr.write(stdout)
        print(f"bench.py: rank launcher exited with {proc.returncode}" + ("" if line else "; no result line from rank 0"), file=sys.stderr)
        sys.exit(proc.returncode or 1)
    print(line)
    sys.exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", "--cells", dest="n", type=int, default=256, help="cells per direction on one GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)  # before torch / the library are imported: the parent stays off the GPU

    import torch

    import ins_amd as ins

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU")
    if world > 1:
        from bench_dist import run_distributed  # z-slab path

        return run_distributed(args, ins)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    n = args.n
    x = (np.linspace(0.0, 1.0, n + 1),) * 3
    setup = ins.Setup(x=x, Re=1000.0, device=dev)
    ps = ins.psolver_spectral(setup)
    u = ins.velocityfield(setup, tgv3d, 0.0, psolver=ps)
    method = ins.RKMethods.RK44()
    cache = ins.ode_method_cache(method, setup, ps)
    stepper = ins.create_stepper(method, setup=setup, psolver=ps, u=u, t=0.0)
    dt = 1e-3 if n <= 256 else 2.5e-4  # (--n 512: the strong-scaling box of BASELINE configs[3] as the headline workload, for profiling)
    # The timed region is the fixed-Δt loop of solve_unsteady (solver.jl:74-83, no processors) = `timesteps_`: K steps in one native call,
    # u valid before and after; INS_BENCH_SINGLE_STEPS=1 times K calls of `timestep_` instead (u materialised after every step).
    chained = not os.environ.get("INS_BENCH_SINGLE_STEPS")

    def advance(st, k):
        if chained:
            return ins.timesteps_(method, st, dt, k, cache=cache)
        for _ in range(k):
            st = ins.timestep_(method, st, dt, cache=cache)
        return st

    if args.warmup:
        stepper = advance(stepper, args.warmup)
    ins._lib.call("ins_rk_profile_enable", cache.handle, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stepper = advance(stepper, args.steps)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    import ctypes as C

    k1_ms, k1_n = C.c_double(), C.c_int64()
    ins._lib.call("ins_rk_profile_read", cache.handle, C.byref(k1_ms), C.byref(k1_n))
    ins._lib.call("ins_rk_profile_enable", cache.handle, 0)
    # for transparency: the same K steps as K separate timestep_ calls (u materialised after every step)
    single_ms = None
    if chained:
        torch.cuda.synchronize()
        ts0 = time.perf_counter()
        for _ in range(args.steps):
            stepper = ins.timestep_(method, stepper, dt, cache=cache)
        torch.cuda.synchronize()
        single_ms = (time.perf_counter() - ts0) * 1e3 / args.steps

    ms_per_step = (t1 - t0) * 1e3 / args.steps
    cells = float(n) ** 3
    value = cells / (ms_per_step * 1e-3) / 1e6
    k1_avg_ms = k1_ms.value / max(k1_n.value, 1)
    stage_bytes, k4_bytes = stage_bytes_per_cell(method, args.steps, chained)
    fused_bytes_per_cell = float(np.mean(stage_bytes))
    k1_gbs = fused_bytes_per_cell * cells / (k1_avg_ms * 1e-3) / 1e9
    # plain K1 (momentum! only), 48 B/cell, on the final state
    F = ins.vectorfield(setup)
    for _ in range(3):
        ins.momentum_(F, stepper.u, None, 0.0, setup)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(20):
        ins.momentum_(F, stepper.u, None, 0.0, setup)
    ev1.record()
    torch.cuda.synchronize()
    k1_plain_ms = ev0.elapsed_time(ev1) / 20
    k1_plain_gbs = K1_BYTES_PER_CELL * cells / (k1_plain_ms * 1e-3) / 1e9
    del F
    # the north-star size for the stencil alone: K1 at 512^3 on a solenoidal-free random field (2 x 3.3 GB, 10 launches)
    k1_512 = None
    if n != 512 and not os.environ.get("INS_BENCH_SKIP_K1_512"):
        s5 = ins.Setup(x=(np.linspace(0.0, 1.0, 513),) * 3, Re=1000.0, device=dev)
        u5, F5 = ins.vectorfield(s5), ins.vectorfield(s5)
        u5.copy_(torch.randn(u5.shape, dtype=torch.float64, device=dev))
        ins.apply_bc_u_(u5, 0.0, s5)
        for _ in range(3):
            ins.momentum_(F5, u5, None, 0.0, s5)
        ev0.record()
        for _ in range(10):
            ins.momentum_(F5, u5, None, 0.0, s5)
        ev1.record()
        torch.cuda.synchronize()
        ms5 = ev0.elapsed_time(ev1) / 10
        k1_512 = {"kernel": "k_flux64 (K1 alone: ins_momentum_f64) at 512^3, random data", "bound": "hbm", "achieved": K1_BYTES_PER_CELL * 512.0**3 / (ms5 * 1e-3) / 1e9,
                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "bytes_per_cell": K1_BYTES_PER_CELL, "avg_launch_ms": ms5}
        k1_512["frac"] = k1_512["achieved"] / HBM_PEAK_GBS
        del u5, F5, s5
    traffic = (committed_traffic("r03_pmc_traffic.json", k1_n.value, chained) if n == 256
               else (committed_traffic("r03_pmc_traffic_512.json", k1_n.value, chained) if n == 512 else None))
    div = ins.max_abs_divergence(stepper.u, setup)
    energy = ins.total_kinetic_energy(stepper.u, setup)
    assert np.isfinite(energy) and div / n < 1e-10, f"bench state is not a valid flow: div={div}, E={energy}"

    out = {
        "metric": "M lattice-cell updates/sec (RK4 step incl. Poisson), 3D TGV fp64",
        "value": value,
        "unit": "M cell-updates/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "ms_per_step_single_calls": single_ms,
        "higher_is_better": True,
        "scaling": "single",  # one GPU: the N = 1 point of both curves (weak: 256^3 per GPU = this line's value; strong: `strong_512`)
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"TaylorGreenVortex3D {n}^3 periodic fp64, RK44 + spectral Poisson, dt={dt:g}, Re=1e3",
                   "grid": [n, n, n], "decomposition": "single GPU",
                   "loop": "timesteps_ (K steps, one native call)" if chained else "timestep_ x K"},
        "roofline": {
            "kernel": "k_flux64 FUSE[/CORR] (K1+K6: momentum-RHS stencil + RK stage combination; stages >= 2 also apply "
                      "the previous projection's gradient-subtract in registers)",
            "bound": "hbm",
            "achieved": k1_gbs,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": k1_gbs / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_note": "B per launch, HBM FETCH(x2 gfx950)+WRITE from rocprofv3 --pmc (two passes of this command with the same --steps/--warmup), profiles/r03_pmc_traffic[_512].json; PMC counters cannot be read inside the timed run",
            "bytes_per_cell": fused_bytes_per_cell,
            "bytes_per_cell_by_stage": stage_bytes,
            "avg_launch_ms": k1_avg_ms,
            "launches": k1_n.value,
        },
        "roofline_k1": {
            "kernel": "k_flux64 (K1: momentum-RHS stencil alone, ins_momentum_f64)",
            "bound": "hbm",
            "achieved": k1_plain_gbs,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": k1_plain_gbs / HBM_PEAK_GBS,
            "bytes_per_cell": K1_BYTES_PER_CELL,
            "avg_launch_ms": k1_plain_ms,
        },
        "roofline_k1_512": k1_512,
        "step_bandwidth": {"design_bytes_per_cell": 384 + k4_bytes + sum(stage_bytes), "achieved_GBs": (384 + k4_bytes + sum(stage_bytes)) * cells / (ms_per_step * 1e-3) / 1e9,
                           "note": "per cell and step: 4 Poisson solves x (32 + 16 + 16 + 16 + 16) B + the final K4 (64 B, once per call when chained) + the stage kernels"},
        "check": {"max_abs_div_times_dx": div / n, "kinetic_energy": energy},
    }
    del stepper, cache, u, ps, setup
    torch.cuda.empty_cache()
    if not os.environ.get("INS_BENCH_SKIP_STRONG_512"):
        out["strong_512"] = strong_single_gpu(ins, dev)
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(n)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

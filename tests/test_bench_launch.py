"""`python bench.py --gpus 2` through its own rank launcher (bench.py `self_launch`): the parent starts two rank processes as children before anything
touches a GPU, relays rank 0's single JSON line and returns a non-zero exit code when a rank fails.  No GPU here, so the ranks run the product's
SlabStepper / SlabComm / torch.distributed (gloo) bookkeeping on the oracle-backed stand-in kernels of tests/slab_cpu_kernels.py, injected through
INS_BENCH_REHEARSAL_KERNELS (tiny grids; the line is labelled a rehearsal and its numbers mean nothing)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(INS_BENCH_BACKEND="gloo", INS_BENCH_REHEARSAL_KERNELS="tests.slab_cpu_kernels:OracleSlabKernels", INS_BENCH_STRONG_GRID="16x8x8",
               PYTHONPATH=ROOT + os.pathsep + env.get("PYTHONPATH", ""), OMP_NUM_THREADS="1")
    env.update(kw)
    return env


def test_bench_gpus2_self_launch_prints_one_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--n", "8", "--no-cpu-baseline"],
                       cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["rehearsal"] and out["rccl_ranks"] is None and out["check"]["backend"] == "gloo" and out["check"]["finite"]
    assert out["config"]["grid"] == [8, 8, 16]
    s = out["strong_512"]
    assert s["scaling"] == "strong" and s["n_gpus"] == 2 and s["grid"] == [16, 8, 8] and s["planes_per_rank"] == 4
    assert s["ms_per_step"] > 0 and s["value"] > 0 and s["finite"] and "speedup_vs_n1_hint" in s
    assert s["max_abs_div_times_dx"] < 1e-10 and out["check"]["max_abs_div_times_dx"] < 1e-10


def test_bench_self_launch_reports_rank_failure():
    """A rank that dies (here: a strong grid the slab layout refuses is not the point — the kernels' module does not exist) makes the parent exit non-zero
    without printing a result line."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--n", "8", "--no-cpu-baseline"],
                       cwd=ROOT, env=_env(INS_BENCH_REHEARSAL_KERNELS="tests.no_such_module:Nothing"), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_refuses_mismatched_world():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], cwd=ROOT,
                       env=_env(WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stderr + r.stdout)


def test_bench_gpus8_self_launch_runs_eight_ranks():
    """The N = 8 shape of the driver's scaling run (BASELINE configs[3]: 512^3 over 8 z-slabs) on the same launch path with eight gloo ranks and tiny grids:
    the weak box of GLOBAL_GRIDS[8] scaled by --n, the strong leg with two planes per rank."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0", "--n", "8", "--no-cpu-baseline"],
                       cwd=ROOT, env=_env(INS_BENCH_STRONG_GRID="16x16x16"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["grid"] == [16, 16, 16] and out["check"]["finite"] and out["check"]["max_abs_div_times_dx"] < 1e-10
    s = out["strong_512"]
    assert s["n_gpus"] == 8 and s["grid"] == [16, 16, 16] and s["planes_per_rank"] == 2 and s["finite"] and s["max_abs_div_times_dx"] < 1e-10

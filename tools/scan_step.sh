#!/bin/bash
# whole-step A/B of the flux64 knobs: TGV 256^3 (bench.py) and decaying turbulence 512^3
cd "$(dirname "$0")/.."
B='import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])'
for zc in 8 16 32 64; do for lds in 0 70000; do for r in 4; do
  echo "== bench 256 rows=$r rows_corr=2 zc=$zc lds=$lds"
  INS_FLUX64_LDS=$lds INS_FLUX64_ROWS=$r INS_FLUX64_ROWS_CORR=2 INS_FLUX64_ZC=$zc INS_BENCH_SKIP_K1_512=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline | python -c "$B"
done; done; done
echo "== bench 256 old"; INS_DISABLE_FLUX64=1 INS_BENCH_SKIP_K1_512=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline | python -c "$B"
for zc in 8 32; do for lds in 0 70000; do
  echo "== turb 512 zc=$zc lds=$lds"; INS_FLUX64_LDS=$lds INS_FLUX64_ZC=$zc python -c "
import sys; sys.path.insert(0,'tools'); import configs_sanity as c; c.run_turb(512)" 2>&1 | grep config3
done; done
echo "== turb 512 old"; INS_DISABLE_FLUX64=1 python -c "
import sys; sys.path.insert(0,'tools'); import configs_sanity as c; c.run_turb(512)" 2>&1 | grep config3

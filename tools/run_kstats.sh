#!/bin/bash
# kernel stats of the bench's timed step at 256^3 and 512^3 (rocprofv3 --kernel-trace --stats; the program directly behind `--`): tools/run_kstats.sh tag
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${1:-k}
export INS_BENCH_SKIP_K1_512=1 INS_BENCH_SKIP_STRONG_512=1
for n in 256 512; do
  rm -rf $R/gpurun_out/ks_${T}_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_${T}_$n -o b -- python3 $R/bench.py --n $n --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/ks_${T}_$n.log 2>&1 || exit 1
  echo "n=$n"; python3 - $R/gpurun_out/ks_${T}_$n/b_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print("   %-72s calls %4s avg %8.1f us" % (r["Name"].replace("void (anonymous namespace)::","")[:72], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/ks_${T}_$n.log | head -1
done

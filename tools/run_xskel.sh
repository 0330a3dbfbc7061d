#!/bin/bash
# What bounds the x passes of the spectral solve?  Kernel times of one solve with and without their LDS transform stages (INS_X_SKEL=1: loads, LDS scatter, stores only).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in 256 512; do
  for sk in 0 1; do
    export INS_X_SKEL=$sk
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/xskel_${n}_$sk -o x -- python3 $R/tools/yz_lab.py $n base: > /dev/null 2>&1 || exit 1
    f=$(ls $R/gpurun_out/xskel_${n}_$sk/x_kernel_stats.csv $R/gpurun_out/xskel_${n}_$sk/*/x_kernel_stats.csv 2>/dev/null | head -1)
    echo "n=$n INS_X_SKEL=$sk"
    python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("k_xfwd", "k_xinv", "k_line3", "k_zsolve3", "k_yfft")):
        print("   %-70s calls %4s avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  done
done

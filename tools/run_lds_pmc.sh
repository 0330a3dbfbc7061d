#!/bin/bash
# LDS bank-conflict share of the solver passes (rocprofv3 --pmc, own run without trace domains): SQ_LDS_BANK_CONFLICT / SQ_LDS_ACTIVE cycles per kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/ldspmc
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/ldspmc -o l --output-format csv -- python3 $R/tools/yz_lab.py ${1:-256} base: > $R/gpurun_out/ldspmc.log 2>&1 || { tail -5 $R/gpurun_out/ldspmc.log; exit 1; }
python3 - $R/gpurun_out/ldspmc <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void (anonymous namespace)::", "")[:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, c in acc.items():
    if "k_" not in k: continue
    a = c.get("SQ_LDS_ACTIVE", 0); b = c.get("SQ_LDS_BANK_CONFLICT", 0); i = c.get("SQ_LDS_IDX_ACTIVE", 0)
    print(f"{k:62s} LDS_ACTIVE {a:.3e}  BANK_CONFLICT {b:.3e}  ({100*b/max(a,1):5.1f} % of active)  IDX_ACTIVE {i:.3e}")
PY

# HBM traffic of K1 variants at 512^3 (one launch each): two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), as MI355X_MICROARCH.md prescribes
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
V="old62:INS_DISABLE_FLUX64=1 nw4_zc32_nobar:INS_FLUX64_NW=4,INS_FLUX64_NOBAR=1,INS_FLUX64_ZC=32 nw4_zc32_bar:INS_FLUX64_NW=4,INS_FLUX64_ZC=32 nw8_zc64_nobar:INS_FLUX64_NW=8,INS_FLUX64_NOBAR=1,INS_FLUX64_ZC=64 default_nw8_zc64_bar: nw8_zc32_bar:INS_FLUX64_NW=8,INS_FLUX64_ZC=32 nw8_zc128_bar:INS_FLUX64_NW=8,INS_FLUX64_ZC=128 nw8_xw4_zc64_bar:INS_FLUX64_NW=8,INS_FLUX64_XW=4,INS_FLUX64_ZC=64"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_f -o f --output-format csv -- python3 tools/k1_lab.py 512 --once $V > gpurun_out/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_w -o w --output-format csv -- python3 tools/k1_lab.py 512 --once $V > gpurun_out/pmc_w.log 2>&1
ls gpurun_out/pmc_f gpurun_out/pmc_w
python3 tools/k1_lab.py parse 512 gpurun_out/pmc_f/f_counter_collection.csv gpurun_out/pmc_w/w_counter_collection.csv $V > gpurun_out/k1_traffic_512.txt 2>&1
cat gpurun_out/k1_traffic_512.txt
python bench.py > gpurun_out/bench2.json 2> gpurun_out/bench2.err; cat gpurun_out/bench2.json

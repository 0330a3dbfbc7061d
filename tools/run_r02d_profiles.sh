# round-2 (fourth batch) measurements kept under profiles/r02d_*: cavity and extended-loop kernel stats, field operators, 3*2^m FFT sizes
set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
rocprofv3 --kernel-trace --stats -d gpurun_out/r02d/cav -o cav --output-format csv -- python3 tools/cavity_prof.py 256 5 > gpurun_out/r02d/cav.log 2>&1 &&
rocprofv3 --kernel-trace --stats -d gpurun_out/r02d/temp -o ext --output-format csv -- python3 tools/ext_prof.py 256 temp 5 > gpurun_out/r02d/temp.log 2>&1 &&
rocprofv3 --kernel-trace --stats -d gpurun_out/r02d/smag -o ext --output-format csv -- python3 tools/ext_prof.py 256 smag 5 > gpurun_out/r02d/smag.log 2>&1 &&
python3 tools/fields_bench.py 256 > gpurun_out/r02d/fields_bench.txt 2>&1 &&
python3 tools/temp_time.py 256 > gpurun_out/r02d/ext_loops.txt 2>&1 &&
INS_HOST_STAGE_LOOP=1 python3 tools/temp_time.py 256 > gpurun_out/r02d/ext_loops_host.txt 2>&1 &&
python3 tools/fft_r3_lab.py > gpurun_out/r02d/fft_r3_lab.txt 2>&1 &&
python3 tools/cavity_lab.py 256 base: wide62:INS_DISABLE_FLUX64M=1 keepk:INS_RK_KEEP_K=1 gather:INS_DISABLE_FDM_UNFOLD4=1 > gpurun_out/r02d/cavity_lab.txt 2>&1
for f in gpurun_out/r02d/*.txt; do tail -n 3 $f; done

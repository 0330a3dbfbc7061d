# round-2 final measurement batch (after the one-kernel closure force, the w-output of the masked stage kernel and the Newton form of λ2) kept under profiles/r02g_*: kernel stats of the cavity, all-walls and extended-loop steps, and the labs
# (field operators, extended loops native vs host-driven, wall-bounded temperature loop, 3*2^m FFT sizes, cavity A/B, fp32 family)
set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02g
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/cav -o cav --output-format csv -- python3 tools/cavity_prof.py 256 5 > $O/cav.log 2>&1 &&
rocprofv3 --kernel-trace --stats -d $O/walls -o w --output-format csv -- python3 tools/walls_prof.py 256 5 > $O/walls.log 2>&1 &&
rocprofv3 --kernel-trace --stats -d $O/temp -o ext --output-format csv -- python3 tools/ext_prof.py 256 temp 5 > $O/temp.log 2>&1 &&
rocprofv3 --kernel-trace --stats -d $O/smag -o ext --output-format csv -- python3 tools/ext_prof.py 256 smag 5 > $O/smag.log 2>&1 &&
rocprofv3 --kernel-trace --stats -d $O/wt -o w --output-format csv -- python3 tools/walls_temp_prof.py 256 5 > $O/wt.log 2>&1 &&
rocprofv3 --kernel-trace --stats -d $O/f32 -o f --output-format csv -- python3 tools/f32_prof.py 512 5 > $O/f32.log 2>&1 &&
python3 tools/fields_bench.py 256 > $O/fields_bench.txt 2>&1 &&
python3 tools/temp_time.py 256 > $O/ext_loops.txt 2>&1 &&
INS_HOST_STAGE_LOOP=1 python3 tools/temp_time.py 256 > $O/ext_loops_host.txt 2>&1 &&
python3 tools/temp_walls_time.py 256 > $O/walls_temp.txt 2>&1 &&
INS_HOST_STAGE_LOOP=1 python3 tools/temp_walls_time.py 256 > $O/walls_temp_host.txt 2>&1 &&
python3 tools/fft_r3_lab.py > $O/fft_r3_lab.txt 2>&1 &&
python3 tools/smagforce_lab.py 256 > $O/smagforce_lab.txt 2>&1 &&
python3 tools/small_grid_lab.py 64 128 192 > $O/small_grid_lab.txt 2>&1 &&
python3 tools/cavity_lab.py 256 base: wide62:INS_DISABLE_FLUX64M=1 keepk:INS_RK_KEEP_K=1 gather:INS_DISABLE_FDM_UNFOLD4=1 > $O/cavity_lab.txt 2>&1 &&
python3 tools/f32_bench.py 512 > $O/f32_bench.txt 2>&1 &&
INS_F32_FP64_SPECTRA=1 python3 tools/f32_bench.py 512 > $O/f32_bench_fp64_spectra.txt 2>&1 &&
python3 tools/f32_bench.py 256 > $O/f32_bench_256.txt 2>&1
L=$O/labs.txt
: > $L
for f in $O/*.txt; do [ "$f" = "$L" ] && continue; echo "== tools: $(basename $f .txt) (tools/run_r02g_profiles.sh)" >> $L; grep -v '^{' $f | grep -v amdgpu.ids >> $L; done
tail -n 5 $L

set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/proff32 -o f --output-format csv -- python3 tools/f32_prof.py 512 5 > gpurun_out/proff32.log 2>&1
tail -n 1 gpurun_out/proff32.log

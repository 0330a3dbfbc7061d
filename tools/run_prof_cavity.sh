set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/profcav -o cav --output-format csv -- python3 tools/cavity_prof.py 256 5 > gpurun_out/profcav.log 2>&1
tail -2 gpurun_out/profcav.log

set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/profch -o c --output-format csv -- python3 tools/channel_prof.py 256 5 > gpurun_out/profch.log 2>&1
tail -n 1 gpurun_out/profch.log

#!/bin/bash
# kernel trace of a short bench run; summary lands in gpurun_out/prof/<tag>_*
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o $tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log

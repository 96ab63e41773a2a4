#!/bin/bash
# kernel stats of one shape of tools/shapes.py: tools/prof_shape.sh <tag> "<name filter>"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o $1 -- python3 $GRAFT_REPO_ROOT/tools/shapes.py "$2" > $GRAFT_REPO_ROOT/gpurun_out/prof_$1.log 2>&1
grep shape $GRAFT_REPO_ROOT/gpurun_out/prof_$1.log

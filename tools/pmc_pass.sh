#!/bin/bash
# one rocprofv3 --pmc pass (own run, kernel-trace only): tools/pmc_pass.sh <tag> <counter> [<counter> ...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc -o $tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
grep -c "" $GRAFT_REPO_ROOT/gpurun_out/pmc/${tag}_counter_collection.csv

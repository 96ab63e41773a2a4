#!/bin/bash
# kernel trace (with timestamps) of a short bench run; keeps the trace CSV for tools/timeline.py.  tools/trace_step.sh <tag> [bench args]
tag=${1:-x}; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o $tag -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-events --no-extras "$@" > $R/gpurun_out/prof_$tag.log 2>&1
tail -1 $R/gpurun_out/prof_$tag.log | cut -c1-300
cd $R && python3 tools/timeline.py gpurun_out/prof/${tag}_kernel_trace.csv > gpurun_out/timeline_$tag.txt 2>&1; tail -1 gpurun_out/timeline_$tag.txt

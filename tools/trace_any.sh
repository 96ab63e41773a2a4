#!/bin/bash
# kernel trace of any python tool; the last N launches as a timeline: tools/trace_any.sh <tag> <N> <script> [args...]
tag=$1; n=$2; shift 2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o $tag -- python3 $R/"$@" > $R/gpurun_out/prof_$tag.log 2>&1
tail -2 $R/gpurun_out/prof_$tag.log | cut -c1-300
cd $R && python3 tools/timeline_tail.py gpurun_out/prof/${tag}_kernel_trace.csv $n > gpurun_out/timeline_$tag.txt 2>&1
rm -f gpurun_out/prof/${tag}_kernel_trace.csv

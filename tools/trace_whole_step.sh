#!/bin/bash
# kernel trace of the whole training step (tools/step_bench.py diora|cliora) + the folded timeline of its last step.  tools/trace_whole_step.sh <diora|cliora> <tag>
which=${1:-diora}; tag=${2:-ws_$which}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o $tag -- python3 $R/tools/step_bench.py $which > $R/gpurun_out/prof_$tag.log 2>&1
cat $R/gpurun_out/prof_$tag.log | grep '^{'
cd $R && python3 tools/step_timeline.py gpurun_out/prof/${tag}_kernel_trace.csv > gpurun_out/step_timeline_$tag.txt 2>&1; tail -3 gpurun_out/step_timeline_$tag.txt
rm -f gpurun_out/prof/${tag}_kernel_trace.csv

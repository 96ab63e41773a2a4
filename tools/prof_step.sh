#!/bin/bash
# kernel stats of the whole training step (tools/step_bench.py [diora|cliora|both]); summary of the top kernels
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o stepc -- python3 $GRAFT_REPO_ROOT/tools/step_bench.py ${1:-both} > $GRAFT_REPO_ROOT/gpurun_out/prof_stepc.log 2>&1
python3 - $GRAFT_REPO_ROOT/gpurun_out/prof/stepc_kernel_stats.csv <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:45]:
    n = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("cliora::", "")[:90]
    print("%-90s calls %6s avg %8.1f total %9.1f us  %s%%" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 1e3, r["Percentage"]))
PY

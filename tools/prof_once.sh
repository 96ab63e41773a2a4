#!/bin/bash
# kernel trace of a short bench run (headline workload only); summary lands in gpurun_out/prof/<tag>_*
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o $tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --no-extras > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log | cut -c1-200
python3 - $GRAFT_REPO_ROOT/gpurun_out/prof/${tag}_kernel_stats.csv <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows:
    n = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("cliora::", "")[:64]
    us = int(r["TotalDurationNs"]) / 1e3 / 13
    tot += us
    if us > 15: print("%-64s calls/step %5.1f avg %7.1f us/step %7.1f" % (n, int(r["Calls"]) / 13, float(r["AverageNs"]) / 1e3, us))
print("sum of kernel time per step: %.0f us" % tot)
PY

#!/bin/bash
# kernel stats of the training-mode forward alone under the given environment: tools/fwd_prof.sh <tag> [VAR=val ...]
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o fp_$tag -- python3 $R/tools/fwd_bench.py --only fwd --steps 20 > $R/gpurun_out/fp_$tag.log 2>&1
grep "forward" $R/gpurun_out/fp_$tag.log
python3 - $R/gpurun_out/prof/fp_${tag}_kernel_stats.csv <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    n = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("cliora::", "")[:60]
    print("   %-60s calls %6d avg %8.2f us" % (n, int(r["Calls"]), float(r["AverageNs"]) / 1e3))
PY
rm -f $R/gpurun_out/prof/fp_${tag}_kernel_trace.csv

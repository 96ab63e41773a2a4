#!/bin/bash
# kernel stats of one tools/shapes.py case: tools/prof_shape.sh <tag> "<case substring>"   (on the MI355X box)
tag=${1:-x}; what=${2:-c5 DioraTreeLSTM len 40}
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o $tag -- python3 $GRAFT_REPO_ROOT/tools/shapes.py "$what" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log | cut -c1-200
python3 - $GRAFT_REPO_ROOT/gpurun_out/prof/${tag}_kernel_stats.csv <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    n = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("cliora::", "")[:70]
    print("%-70s calls %6d avg %8.1f us  %5.1f %%" % (n, int(r["Calls"]), float(r["AverageNs"]) / 1e3, 100.0 * int(r["TotalDurationNs"]) / tot))
PY

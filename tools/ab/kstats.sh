#!/bin/bash
# per-kernel average durations of a short bench run under an environment setting: tools/ab/kstats.sh "VAR=x" <grep pattern> [bench args]
E=$1; P=$2; shift 2
export $E
bash tools/trace_step.sh ks "$@" > /dev/null
echo "== $E"
python3 - "$P" <<'PY'
import csv, sys, re
pat = re.compile(sys.argv[1])
for r in csv.reader(open('gpurun_out/prof/ks_kernel_stats.csv')):
    if pat.search(r[0]):
        print(f"{r[0][:72]:72s} calls {r[1]:>5s} avg_us {float(r[3])/1e3:9.1f} total_ms {float(r[2])/1e6:8.2f}")
PY

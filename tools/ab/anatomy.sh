#!/bin/bash
# Latency anatomy of the compose launches with WRONG-RESULT build variants (tools/ab/build_variant.sh k6 -DCLIORA_DIAG_KSTEPS=6, k1, nostage,
# k1nostage): step time and per-kernel averages under each, headline workload.  tools/ab/anatomy.sh [bench args] -> gpurun_out/anatomy.txt
R=$GRAFT_REPO_ROOT; cd $R
out=gpurun_out/anatomy.txt; : > $out
for v in "" ${VARIANTS:-k6 k1 nostage k1nostage}; do
  lib=$R/cliora_amd/libcliora_chart.so; [ -n "$v" ] && lib=$R/cliora_amd/libvar_$v.so
  [ -f $lib ] || continue
  export CLIORA_CHART_LIB=$lib
  echo "== ${v:-default}" >> $out
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-events --no-extras --steps 50 --warmup 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   step ms', d['ms_per_step'], 'median', d['step_ms']['median'])" >> $out
  bash tools/trace_step.sh an_${v:-default} "$@" > /dev/null 2>&1
  python3 - gpurun_out/prof/an_${v:-default}_kernel_stats.csv >> $out <<'PY'
import csv, re, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    n = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("cliora::", "")[:50]
    print("   %-50s calls %5d avg %8.2f us" % (n, int(r["Calls"]), float(r["AverageNs"]) / 1e3))
PY
  rm -f gpurun_out/prof/an_${v:-default}_kernel_trace.csv
done
cat $out

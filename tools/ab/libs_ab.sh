#!/bin/bash
# alternate several builds of the library on a bench workload: tools/ab/libs_ab.sh "<bench args>" <rounds> <lib> <lib> ...
ARGS=$1; N=$2; shift 2
R="python bench.py --no-cpu-baseline --no-kernel-events --no-extras --steps 50 --warmup 10 $ARGS"
for i in $(seq $N); do
  for L in "$@"; do
    CLIORA_CHART_LIB=$PWD/$L timeout 300 $R 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['ms_per_step'], d['step_ms']['median'])"
  done
done

#!/bin/bash
# alternate two builds of the library on the headline workload: tools/ab/ab.sh <libA> <libB> [rounds]
A=$1; B=$2; N=${3:-3}
R="python bench.py --no-cpu-baseline --no-kernel-events --no-extras --steps 50 --warmup 10"
for i in $(seq $N); do
  for L in $A $B; do
    CLIORA_CHART_LIB=$PWD/$L $R 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['ms_per_step'], d['step_ms']['median'])"
  done
done

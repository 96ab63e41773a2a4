#!/bin/bash
# alternate two environment settings of the library on the headline workload: tools/ab/env_ab.sh "VAR=a" "VAR=b" [rounds] [bench args]
A=$1; B=$2; N=${3:-3}; shift 3
R="python bench.py --no-cpu-baseline --no-kernel-events --no-extras --steps 50 --warmup 10 $@"
for i in $(seq $N); do
  for E in "$A" "$B"; do
    env $E $R 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$E', d['ms_per_step'], d['step_ms']['median'])"
  done
done

#!/bin/bash
# one bench line per environment setting, with extra bench arguments first: tools/ab/sweep_any.sh "<bench args>" "A=1" "A=2" ...
ARGS=$1; shift
R="python bench.py --no-cpu-baseline --no-kernel-events --no-extras $ARGS"
for E in "$@"; do
  env $E $R 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$E', d['ms_per_step'], d['step_ms']['median'])"
done

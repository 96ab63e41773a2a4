#!/bin/bash
# one bench line per environment setting: tools/ab/sweep_env.sh "A=1 B=2" "A=3" ...   (prints ms_per_step and the per-step median)
R="python bench.py --no-cpu-baseline --no-kernel-events --no-extras --steps 50 --warmup 10"
for E in "$@"; do
  env $E $R 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$E', d['ms_per_step'], d['step_ms']['median'])"
done

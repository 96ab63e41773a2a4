#!/bin/bash
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for k in 1 2; do python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['classes'])"; done

#!/bin/bash
for mb in 400 600 1000 2000 400; do echo "min blocks $mb"; CLIORA_KSPLIT_MIN_BLOCKS=$mb python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done

#!/bin/bash
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
export CLIORA_COMPOSE_KSPLIT_ROWS=1500
for pm in f32 bf16x3; do echo "proj $pm"; CLIORA_PROJ_MFMA=$pm python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['classes'])"; done
D=50 B=8 L=10 python tools/accuracy.py 2>/dev/null | tail -1 | cut -c1-400

for t in 1500 2500 3500 5000; do echo "thr $t"; CLIORA_COMPOSE_KSPLIT_ROWS=$t python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
bash tools/prof_once.sh ws3

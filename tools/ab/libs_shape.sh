#!/bin/bash
# alternate several builds of the library on one tools/shapes.py case: tools/ab/libs_shape.sh "<case substring>" <rounds> <lib> <lib> ...
W=$1; N=$2; shift 2
for i in $(seq $N); do
  for L in "$@"; do
    CLIORA_CHART_LIB=$PWD/$L SHAPES_STEPS=${SHAPES_STEPS:-20} SHAPES_WARMUP=5 timeout 300 python tools/shapes.py "$W" 2>/dev/null | grep ms_per_step | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['shape'], d['ms_per_step'])"
  done
done

#!/bin/bash
# the pair rows' weight gradient as ONE launch with nothing beside it (one stream, no early part): kernel durations of both operand forms
# tools/ab/wgrad_alone.sh [bench args]
export CLIORA_WAVEFRONT=0 CLIORA_WGRAD_EARLY_STEP=-1
for T in 0 1; do
  CLIORA_PAIR_TILES=$T bash tools/trace_step.sh alone$T "$@" > /dev/null
  echo "CLIORA_PAIR_TILES=$T"
  grep -E "tn_gemm_tiles|tn_gemm_dma3x|level_compose_bwd|slab_reduce" gpurun_out/prof/alone${T}_kernel_stats.csv | cut -c1-60,60-400 | awk -F, '{print substr($1,1,70), $2, $3, $4}'
done

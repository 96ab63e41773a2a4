#!/bin/bash
# duration of tn_gemm_tiles launched alone (one stream, no early part) under an environment setting: tools/ab/tiles_alone.sh "VAR=x" [bench args]
E=$1; shift
export CLIORA_WAVEFRONT=0 CLIORA_WGRAD_EARLY_STEP=-1 $E
bash tools/trace_step.sh ta "$@" > /dev/null
echo "$E: $(grep -E 'tn_gemm_tiles' gpurun_out/prof/ta_kernel_stats.csv | awk -F, '{print $(NF-5), $(NF-4)}')"

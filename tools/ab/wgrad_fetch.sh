#!/bin/bash
# HBM-side bytes fetched by the pair rows' weight-gradient kernel, launched alone (one stream, no early part): tools/ab/wgrad_fetch.sh <tag> [bench args]
tag=$1; shift
export CLIORA_WAVEFRONT=0 CLIORA_WGRAD_EARLY_STEP=-1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O -o fetch -- python3 $R/bench.py --no-cpu-baseline --no-kernel-events --no-extras --steps 2 --warmup 1 "$@" > $O/log.txt 2>&1
cd $R
python tools/summarize_pmc.py $O/fetch_counter_collection.csv $O/fetch_by_kernel.csv
grep -E "tn_gemm|level_compose_bwd" $O/fetch_by_kernel.csv | cut -c1-40,100-300
rm -f $O/fetch_counter_collection.csv $O/fetch_kernel_trace.csv

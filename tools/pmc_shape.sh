#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / MFMA-busy passes (each its own rocprofv3 --pmc run, --kernel-trace only) of one tools/shapes.py case, 3 steps:
#   tools/pmc_shape.sh <tag> "<case substring>"        -> gpurun_out/pmc_<tag>_{fetch,write,mfma}_by_kernel.csv, gpurun_out/traffic_shapes.json[tag]
tag=${1:-l40}; what=${2:-DioraMLP len 40}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export SHAPES_STEPS=2 SHAPES_WARMUP=1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_$tag -o fetch -- python3 $R/tools/shapes.py "$what" > $O/pmc_${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_$tag -o write -- python3 $R/tools/shapes.py "$what" > $O/pmc_${tag}_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_$tag -o mfma -- python3 $R/tools/shapes.py "$what" > $O/pmc_${tag}_mfma.log 2>&1
cd $R
python3 tools/summarize_pmc.py $O/pmc_$tag/fetch_counter_collection.csv $O/pmc_${tag}_fetch_by_kernel.csv
python3 tools/summarize_pmc.py $O/pmc_$tag/write_counter_collection.csv $O/pmc_${tag}_write_by_kernel.csv
python3 tools/summarize_pmc.py $O/pmc_$tag/mfma_counter_collection.csv $O/pmc_${tag}_mfma_busy.csv --mfma-busy
python3 tools/pmc_step_total.py $O/pmc_$tag/fetch_counter_collection.csv $O/pmc_$tag/write_counter_collection.csv 3 $O/traffic_shapes.json $tag
rm -rf $O/pmc_$tag

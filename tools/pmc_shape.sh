#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / MFMA-busy passes (each its own rocprofv3 --pmc run, --kernel-trace only; every run under its own timeout) of a
# command that makes <nsteps> steps of one workload:
#   tools/pmc_shape.sh <tag> <nsteps> python3 <script> [args]     e.g.  SHAPES_STEPS=2 SHAPES_WARMUP=1 tools/pmc_shape.sh l40 3 python3 tools/shapes.py "DioraMLP len 40"
# -> gpurun_out/pmc_<tag>_{fetch,write}_by_kernel.csv, pmc_<tag>_mfma_busy.csv, and gpurun_out/traffic_shapes.json[tag] (bytes per step)
tag=$1; nsteps=$2; shift 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
prog=$1; shift
args=()
for a in "$@"; do case "$a" in tools/*|bench.py) args+=("$R/$a");; *) args+=("$a");; esac; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_$tag -o fetch -- $prog "${args[@]}" > $O/pmc_${tag}_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_$tag -o write -- $prog "${args[@]}" > $O/pmc_${tag}_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_$tag -o mfma -- $prog "${args[@]}" > $O/pmc_${tag}_mfma.log 2>&1
cd $R
python3 tools/summarize_pmc.py $O/pmc_$tag/fetch_counter_collection.csv $O/pmc_${tag}_fetch_by_kernel.csv
python3 tools/summarize_pmc.py $O/pmc_$tag/write_counter_collection.csv $O/pmc_${tag}_write_by_kernel.csv
python3 tools/summarize_pmc.py $O/pmc_$tag/mfma_counter_collection.csv $O/pmc_${tag}_mfma_busy.csv --mfma-busy
python3 tools/pmc_step_total.py $O/pmc_$tag/fetch_counter_collection.csv $O/pmc_$tag/write_counter_collection.csv $nsteps $O/traffic_shapes.json $tag
rm -rf $O/pmc_$tag

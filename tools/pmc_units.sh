#!/bin/bash
# Which unit does a kernel keep busy?  Four rocprofv3 --pmc passes (each its own run, --kernel-trace only, under a timeout) of one
# tools/shapes.py case: LDS pipe, vector L1 (TCP), L2 hit / miss, instruction mix.   tools/pmc_units.sh <tag> "<case substring>"
# -> gpurun_out/units_<tag>.txt (tools/pmc_units_table.py)
tag=${1:-l40}; what=${2:-DioraMLP len 40}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export SHAPES_STEPS=2 SHAPES_WARMUP=1
# (a TA pass -- TA_TA_BUSY_sum with the two TA_*_STALLED_BY_TC counters -- is refused by the profiler: "exceeds the capabilities of the hardware to
#  collect", and its abort then hangs until the timeout; the vector L1's own counters in pass 2 carry the same information)
i=0
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/units_$tag -o p$i -- python3 $R/tools/shapes.py "$what" > $O/units_${tag}_p$i.log 2>&1
  python3 $R/tools/summarize_pmc.py $O/units_$tag/p${i}_counter_collection.csv $O/units_${tag}_p$i.csv
done
rm -rf $O/units_$tag
python3 $R/tools/pmc_units_table.py $O/units_${tag} > $O/units_$tag.txt
cat $O/units_$tag.txt

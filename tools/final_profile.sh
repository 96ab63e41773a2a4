#!/bin/bash
# end-of-round measurement set (run on the MI355X box from the repo root): bench line, kernel stats, PMC passes.
# Every rocprofv3 run is its own pass (kernel-trace only next to --pmc); summaries are copied to profiles/ by hand.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
CLIORA_PERSISTENT=1 python bench.py --no-cpu-baseline --no-extras > $O/bench_persistent_on.json 2> $O/bench_persistent_on.err
python tools/persist_ab.py > $O/persist_ab.txt 2>&1
CLIORA_PERSIST_TRACE=1 python tools/persist_trace.py > $O/persist_trace.txt 2>&1
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-kernel-events --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ks -- $B --steps 10 --warmup 3 > $O/prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc -o fetch -- $B --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc -o write -- $B --steps 2 --warmup 1 > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc -o mfma -- $B --steps 2 --warmup 1 > $O/pmc_mfma.log 2>&1
cd $R
python tools/summarize_pmc.py $O/pmc/fetch_counter_collection.csv $O/pmc_fetch_by_kernel.csv
python tools/summarize_pmc.py $O/pmc/write_counter_collection.csv $O/pmc_write_by_kernel.csv
python tools/summarize_pmc.py $O/pmc/mfma_counter_collection.csv $O/pmc_mfma_busy.csv --mfma-busy
python tools/pmc_traffic.py $O/pmc/fetch_counter_collection.csv $O/pmc/write_counter_collection.csv $O/traffic.json
rm -rf $O/pmc/*_counter_collection.csv $O/prof/ks_kernel_trace.csv    # big raw files stay on the box
ls -la $O $O/prof

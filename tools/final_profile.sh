#!/bin/bash
# end-of-round measurement set (run on the MI355X box from the repo root): bench line, kernel stats, PMC passes, other shapes.
# Every rocprofv3 run is its own pass (kernel-trace only next to --pmc); summaries are copied to profiles/ by hand.
#   gpurun -- "GRAFT_COMMIT=$(git rev-parse --short HEAD) bash tools/final_profile.sh"      (the box has no .git)
set -x
# every long command under its own `timeout` (a hung profiler run once cost a whole gpurun call)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
# A step that fails must not leave its error text behind as "evidence" (round 5 committed a traceback as r05_timeline.txt): `ck <file> <cmd...>`
# runs the command into <file>; on a non-zero exit the output is moved to <file>.FAILED, the step is listed in $O/FAILED and the script
# exits non-zero at the end.
FAILS=0
ck() { local out=$1; shift; if ! "$@" > "$out" 2>&1; then mv "$out" "$out.FAILED"; echo "$out: $*" >> $O/FAILED; FAILS=$((FAILS + 1)); fi; }
rm -f $O/FAILED
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err || { echo "$O/bench.json: bench.py" >> $O/FAILED; FAILS=$((FAILS + 1)); }
timeout 600 python bench.py --workload c3 --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err || { echo "$O/bench_c3.json: bench.py --workload c3" >> $O/FAILED; FAILS=$((FAILS + 1)); }
timeout 600 python tools/shapes.py > $O/shapes.jsonl 2> $O/shapes.err || { echo "$O/shapes.jsonl: tools/shapes.py" >> $O/FAILED; FAILS=$((FAILS + 1)); }
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-kernel-events --no-extras"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ks -- $B --steps 10 --warmup 3 > $O/prof.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc -o fetch -- $B --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc -o write -- $B --steps 2 --warmup 1 > $O/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc -o mfma -- $B --steps 2 --warmup 1 > $O/pmc_mfma.log 2>&1
cd $R
ck $O/timeline.txt python tools/timeline.py $O/prof/ks_kernel_trace.csv
python tools/summarize_pmc.py $O/pmc/fetch_counter_collection.csv $O/pmc_fetch_by_kernel.csv
python tools/summarize_pmc.py $O/pmc/write_counter_collection.csv $O/pmc_write_by_kernel.csv
python tools/summarize_pmc.py $O/pmc/mfma_counter_collection.csv $O/pmc_mfma_busy.csv --mfma-busy
python tools/pmc_traffic.py $O/pmc/fetch_counter_collection.csv $O/pmc/write_counter_collection.csv $O/traffic.json
rm -rf $O/pmc/*_counter_collection.csv $O/prof/ks_kernel_trace.csv $O/pmc/*_kernel_trace.csv    # big raw files stay on the box
for c in "c1 DioraMLP" "c3 CLIORA" "DioraMLP len 40" "c5 DioraTreeLSTM len 40"; do
  t=$(echo $c | tr -d ' ' | tr 'A-Z' 'a-z'); bash tools/prof_shape.sh fin_$t "$c" > $O/stats_$t.txt 2>&1; cp gpurun_out/prof/fin_${t}_kernel_stats.csv $O/ 2>/dev/null
done
rm -f gpurun_out/prof/fin_*_kernel_trace.csv
# PMC passes of the other workloads (round 5: traffic per step of the throughput-bound shapes and of the CLIORA training step)
export SHAPES_STEPS=2 SHAPES_WARMUP=1
bash tools/pmc_shape.sh l40 3 python3 tools/shapes.py "DioraMLP len 40" > $O/pmc_l40.txt 2>&1
bash tools/pmc_shape.sh c5 3 python3 tools/shapes.py "c5 DioraTreeLSTM len 40" > $O/pmc_c5.txt 2>&1
bash tools/pmc_shape.sh c3 3 python3 tools/shapes.py "c3 CLIORA" > $O/pmc_c3.txt 2>&1
bash tools/pmc_shape.sh c1 3 python3 tools/shapes.py "c1 DioraMLP" > $O/pmc_c1.txt 2>&1
bash tools/pmc_shape.sh c3_step 3 python3 bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_c3_step.txt 2>&1
unset SHAPES_STEPS SHAPES_WARMUP
cp gpurun_out/traffic_shapes.json gpurun_out/pmc_*_by_kernel.csv gpurun_out/pmc_*_mfma_busy.csv $O/ 2>/dev/null
ck $O/wavefront_sweep.txt timeout 900 python tools/wavefront_sweep.py
ls -la $O $O/prof
if [ $FAILS -ne 0 ]; then echo "final_profile.sh: $FAILS step(s) failed:"; cat $O/FAILED; exit 1; fi

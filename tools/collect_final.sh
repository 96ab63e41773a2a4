#!/bin/bash
# copy the judged summaries of a tools/final_profile.sh run (gpurun_out/final, gpurun_out/pmc_*) into profiles/ under the round's prefix:
#   tools/collect_final.sh r05
P=${1:?round prefix}; F=gpurun_out/final; D=profiles
cp $F/prof/ks_kernel_stats.csv $D/${P}_kernel_stats.csv
for s in c1dioramlp:c1 c3cliora:c3 dioramlplen40:l40 c5dioratreelstmlen40:c5; do cp $F/fin_${s%%:*}_kernel_stats.csv $D/${P}_kernel_stats_${s##*:}.csv; done
cp $F/pmc_fetch_by_kernel.csv $D/${P}_pmc_fetch_by_kernel.csv
cp $F/pmc_write_by_kernel.csv $D/${P}_pmc_write_by_kernel.csv
cp $F/pmc_mfma_busy.csv $D/${P}_pmc_mfma_busy.csv
cp $F/traffic.json $D/${P}_traffic.json; cp $F/traffic.json $D/traffic.json
for t in l40 c5 c3 c3_step c1; do for k in fetch_by_kernel write_by_kernel mfma_busy; do cp $F/pmc_${t}_$k.csv $D/${P}_pmc_${k}_$t.csv; done; done
cp $F/traffic_shapes.json $D/${P}_traffic_shapes.json
cp $F/timeline.txt $D/${P}_timeline.txt
cp $F/wavefront_sweep.txt $D/${P}_wavefront_sweep.txt
cp $F/bench_c3.json $D/${P}_bench_c3.json
cp $F/shapes.jsonl $D/${P}_shapes.jsonl
git status --short $D | head -40

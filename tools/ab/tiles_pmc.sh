#!/bin/bash
# PMC counters of tn_gemm_tiles launched alone: tools/ab/tiles_pmc.sh "<counters>" [bench args]
C=$1; shift
export CLIORA_WAVEFRONT=0 CLIORA_WGRAD_EARLY_STEP=-1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_tiles; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O -o c -- python3 $R/bench.py --no-cpu-baseline --no-kernel-events --no-extras --steps 2 --warmup 1 "$@" > $O/log.txt 2>&1
cd $R
python3 - <<PY
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open('$O/c_counter_collection.csv')):
    k = r['Kernel_Name']
    if 'tn_gemm_tiles' in k or 'tn_gemm_dma3x' in k:
        acc[k[:40]][r['Counter_Name']] += float(r['Counter_Value']); n[(k[:40], r['Counter_Name'])] += 1
for k, d in acc.items():
    print(k, {c: round(v / n[(k, c)]) for c, v in d.items()})
PY
rm -f $O/c_counter_collection.csv $O/c_kernel_trace.csv

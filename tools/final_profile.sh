#!/bin/bash
# end-of-round measurement set (run on the MI355X box from the repo root): bench lines, kernel stats, PMC passes, other shapes
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --mfma f32 --no-cpu-baseline > $O/bench_f32.json 2>> $O/bench.err
python tools/shapes.py > $O/shapes.jsonl 2>> $O/bench.err
CLIORA_MFMA=f32 python tools/shapes.py > $O/shapes_f32.jsonl 2>> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ks -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events > $O/prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc -o fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc -o write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc -o mfma -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > $O/pmc_mfma.log 2>&1
ls -la $O $O/prof $O/pmc

#!/usr/bin/env python
"""HBM traffic per launch of each MFMA kernel class from two rocprofv3 --pmc passes.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py ...
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py ...
  python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv profiles/traffic.json

gfx950 corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; FETCH_SIZE reports
half the bytes of wide (16 B/lane) coalesced reads, so it is doubled; WRITE_SIZE is exact for 16-B stores.
"""
import csv
import json
import sys
from collections import defaultdict

CLASSES = {
    'compose_fwd': ('level_compose_fwd',),
    'compose_bwd': ('level_compose_bwd',),
    'wgrad': ('tn_gemm_tiles',),          # the pair rows' dW2 (round 4: the tiled split-bf16 operands; the projections' gradients stay on tn_gemm_dma3x)
}


def per_class(path, counter):
    tot, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        name = r['Kernel_Name']
        for cls, pats in CLASSES.items():
            if all(p in name for p in pats):
                tot[cls] += float(r['Counter_Value'])
                n[cls] += 1
    return {c: (tot[c] / n[c] if n[c] else None) for c in CLASSES}, dict(n)


if __name__ == '__main__':
    fetch, nf = per_class(sys.argv[1], 'FETCH_SIZE')
    write, nw = per_class(sys.argv[2], 'WRITE_SIZE')
    out = {}
    for c in CLASSES:
        if fetch[c] is None or write[c] is None:
            continue
        out[c] = round((2.0 * fetch[c] + write[c]) * 1024.0)
        out[c + '_detail'] = dict(fetch_kib_raw=fetch[c], write_kib=write[c], launches_fetch=nf.get(c, 0), launches_write=nw.get(c, 0),
                                  note='bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch (gfx950: FETCH_SIZE counts wide reads at half)')
    import datetime
    out['measured'] = 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --no-cpu-baseline --no-kernel-events --no-extras --steps 2 --warmup 1` on %s (tools/final_profile.sh)' % datetime.date.today().isoformat()
    import os
    out['commit'] = os.environ.get('GRAFT_COMMIT', 'not recorded')       # the GPU box has no .git: pass GRAFT_COMMIT=$(git rev-parse --short HEAD) in the gpurun command
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
    print(json.dumps(out))

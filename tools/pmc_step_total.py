#!/usr/bin/env python
"""Traffic per STEP of a whole workload from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes, --kernel-trace only):
every kernel's 2 x FETCH_SIZE + WRITE_SIZE (KiB; gfx950: FETCH_SIZE counts wide reads at half, MI355X_MICROARCH.md HBM section), summed
over the run and divided by the number of steps the run made.  lane_probe (the one-time stream probe) is left out.

  python tools/pmc_step_total.py <fetch counter_collection.csv> <write counter_collection.csv> <steps in the run> <out.json> <label>

Appends / replaces `label` in out.json: {total_bytes_per_step, by_kernel: {name: bytes per step}, mfma (optional)}."""
import csv
import json
import os
import re
import sys
from collections import defaultdict


def sums(path, counter):
    tot = defaultdict(float)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter or 'lane_probe' in r['Kernel_Name']:
            continue
        name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '').replace('cliora::', '')[:60]
        tot[name] += float(r['Counter_Value'])
    return tot


if __name__ == '__main__':
    fetch, write = sums(sys.argv[1], 'FETCH_SIZE'), sums(sys.argv[2], 'WRITE_SIZE')
    steps, out, label = int(sys.argv[3]), sys.argv[4], sys.argv[5]
    by = {k: (2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024.0 / steps for k in set(fetch) | set(write)}
    rec = dict(total_bytes_per_step=round(sum(by.values())), steps_in_run=steps,
               by_kernel={k: round(v) for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:16]},
               formula='(2 x FETCH_SIZE + WRITE_SIZE) x 1024 over every kernel of the run / steps', commit=os.environ.get('GRAFT_COMMIT', 'not recorded'))
    j = json.load(open(out)) if os.path.exists(out) else {}
    j[label] = rec
    json.dump(j, open(out, 'w'), indent=1)
    print(label, json.dumps(rec)[:600])

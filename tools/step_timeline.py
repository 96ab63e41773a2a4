#!/usr/bin/env python
"""Timeline of the LAST whole training step in a rocprofv3 kernel trace of tools/step_bench.py: the kernels around the chart one by one
(queue, start in us since the step's first kernel, duration, gap on its queue), the chart's own launches folded into one line per run.
A step is cut at the last two launches of the kernel that ends a step (the fused clip + Adam apply).
  python tools/step_timeline.py <kernel_trace.csv> [end-kernel substring, default: adam]"""
import csv
import re
import sys

CHART = ('level_', 'cell_', 'tn_gemm', 'rows_gemm_ksplit', 'slab_reduce', 'copy2d_multi', 'weight_images_all', 'unit_norm_rows', 'root_bwd', 'leaf_bwd',
         'resident_', 'lane_probe', 'lstm_', 'pair_scores')
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
endk = (sys.argv[2] if len(sys.argv) > 2 else 'adam').lower()
ends = [i for i, r in enumerate(rows) if endk in r['Kernel_Name'].lower() and (i + 1 == len(rows) or endk not in rows[i + 1]['Kernel_Name'].lower())]
if len(ends) < 2:
    sys.exit('step_timeline.py: fewer than two step ends (%r) in %s' % (endk, sys.argv[1]))
step = rows[ends[-2] + 1: ends[-1] + 1]
t0 = int(step[0]['Start_Timestamp'])
short = lambda s: re.sub(r'\(.*', '', s).replace('void ', '').replace('cliora::', '').replace('at::native::', '')[:70]
last, run = {}, None


def flush():
    global run
    if run:
        print('     %9.2f  span %7.2f            [chart: %d launches on %d queue(s), %.1f us of kernels]' % ((run['s'] - t0) / 1e3, (run['e'] - run['s']) / 1e3, run['n'], len(run['q']), run['busy'] / 1e3))
    run = None


for r in step:
    q, s, e = r.get('Queue_Id', '?'), int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = short(r['Kernel_Name'])
    if any(k in name for k in CHART):
        if run is None:
            run = dict(s=s, e=e, n=0, q=set(), busy=0)
        run['e'] = max(run['e'], e); run['n'] += 1; run['q'].add(q); run['busy'] += e - s
        last[q] = e
        continue
    flush()
    gap = (s - last[q]) / 1e3 if q in last else 0.0
    last[q] = e
    print('q%-3s %9.2f  dur %7.2f  gap %6.2f  %s' % (q, (s - t0) / 1e3, (e - s) / 1e3, gap, name))
flush()
print('step span %.1f us, %d launches' % ((max(int(r['End_Timestamp']) for r in step) - t0) / 1e3, len(step)))

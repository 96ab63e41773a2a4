#!/usr/bin/env python
"""The last N launches of a rocprofv3 kernel-trace CSV (queue, start in us since the first of them, duration, gap on the queue, name) --
for workloads tools/timeline.py cannot cut into steps.   python tools/timeline_tail.py <kernel_trace.csv> [N] [skip-from-end]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
sel = rows[len(rows) - n - skip: len(rows) - skip]
t0 = int(sel[0]['Start_Timestamp'])
last = {}
short = lambda s: re.sub(r'\(.*', '', s).replace('void ', '').replace('cliora::', '').replace('at::native::', '')[:60]
for r in sel:
    q = r.get('Queue_Id', '?')
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - last[q]) / 1e3 if q in last else 0.0
    last[q] = e
    print('q%-3s %9.2f  dur %7.2f  gap %6.2f  %s' % (q, (s - t0) / 1e3, (e - s) / 1e3, gap, short(r['Kernel_Name'])))

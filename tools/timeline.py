#!/usr/bin/env python
"""Timeline of ONE step from a rocprofv3 kernel-trace CSV: every launch with its queue, start (us since the step's first
kernel), duration and the gap to the previous launch on the same queue.  The last step of the trace is taken (between two
copy2d_multi pack launches).   python tools/timeline.py <kernel_trace.csv> [max rows]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# a step starts with the forward's parameter pack (copy2d_multi followed by split_weight_image)
starts = [i for i in range(len(rows) - 1) if 'copy2d_multi' in names[i] and 'split_weight_image' in names[i + 1]]
a, b = starts[-2], starts[-1]
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp'])
last_end = {}
short = lambda n: re.sub(r'\(.*', '', n).replace('void ', '').replace('cliora::', '')[:44]
busy = {}
for r in step[: int(sys.argv[2]) if len(sys.argv) > 2 else 100000]:
    q = r.get('Queue_Id', '?')
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    busy[q] = busy.get(q, 0) + (e - s)
    print('q%-3s %9.2f  dur %7.2f  gap %6.2f  %s' % (q, (s - t0) / 1e3, (e - s) / 1e3, gap, short(r['Kernel_Name'])))
print('step span %.1f us; busy per queue: %s' % ((max(int(r['End_Timestamp']) for r in step) - t0) / 1e3, {q: round(v / 1e3, 1) for q, v in busy.items()}))

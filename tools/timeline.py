#!/usr/bin/env python
"""Timeline of ONE step from a rocprofv3 kernel-trace CSV: every launch with its queue, start (us since the step's first
kernel), duration and the gap to the previous launch on the same queue.  The last step of the trace is taken (between two
copy2d_multi pack launches that open a forward).   python tools/timeline.py <kernel_trace.csv> [max rows]
Exits non-zero (with a message, no traceback) when the trace holds no complete step."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# a step starts with the forward's parameter pack: copy2d_multi followed by the launch that builds the weight images
# (weight_images_all since round 5's ffae15a; split_weight_image / frag_weight_image before)
IMAGE_KERNELS = ('weight_images_all', 'split_weight_image', 'frag_weight_image')
starts = [i for i in range(len(rows) - 1) if 'copy2d_multi' in names[i] and any(k in names[i + 1] for k in IMAGE_KERNELS)]
if len(starts) < 2:
    sys.exit('timeline.py: %d step starts (copy2d_multi + %s) in %s: need at least 2 for one complete step; kernels seen: %s'
             % (len(starts), ' / '.join(IMAGE_KERNELS), sys.argv[1], sorted({re.sub(r'[<(].*', '', n) for n in names})[:12]))
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

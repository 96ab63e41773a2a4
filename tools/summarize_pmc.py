#!/usr/bin/env python
"""Per-kernel summary of a rocprofv3 --pmc counter_collection.csv: dispatches and the sum / mean of every counter.
  python tools/summarize_pmc.py <counter_collection.csv> <out.csv> [--mfma-busy]
--mfma-busy adds mfma_busy_pct = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCDs) * 1024 SIMDs)."""
import csv
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
busy = '--mfma-busy' in sys.argv
tot = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for r in csv.DictReader(open(src)):
    k = r['Kernel_Name'][:110]
    tot[k][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[k][r['Counter_Name']] += 1
counters = sorted({c for v in tot.values() for c in v})
with open(dst, 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['kernel', 'dispatches'] + ['%s_sum' % c for c in counters] + ['%s_mean' % c for c in counters] + (['mfma_busy_pct'] if busy else []))
    for k in sorted(tot, key=lambda k: -sum(tot[k].values())):
        n = max(cnt[k].values())
        row = [k, n] + ['%.6g' % tot[k][c] for c in counters] + ['%.6g' % (tot[k][c] / max(1, cnt[k][c])) for c in counters]
        if busy:
            ga = tot[k].get('GRBM_GUI_ACTIVE', 0.0)
            row.append('%.1f' % (100.0 * tot[k].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (ga / 8 * 1024)) if ga else '')
        w.writerow(row)

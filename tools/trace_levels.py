#!/usr/bin/env python
"""Per-launch durations of one kernel family from a rocprofv3 kernel trace CSV."""
import csv
import sys
path, pat = sys.argv[1], sys.argv[2]
rows = [r for r in csv.DictReader(open(path)) if pat in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 45
for r in rows[-n:]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    print('%8.1f us grid %s wg %s  %s' % (d, r.get('Grid_Size_X', '?') + 'x' + r.get('Grid_Size_Y', '?'), r.get('Workgroup_Size_X', '?'), r['Kernel_Name'][:70]))

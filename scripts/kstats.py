#!/usr/bin/env python
"""Per-kernel summary of a rocprofv3 --kernel-trace CSV, grouped by (kernel, grid size) so that
launches of one kernel at different shapes are not averaged together.
    python scripts/kstats.py <..._kernel_trace.csv> [name filter]"""
import csv
import re
import sys
from collections import defaultdict

rows = defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        name = re.sub(r'^void ', '', name)
        name = re.sub(r'\(.*$', '', name)
        if len(sys.argv) > 2 and sys.argv[2] not in name:
            continue
        key = (name, int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1),
               int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
        rows[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print('%-44s %14s %6s %9s %9s %9s' % ('kernel', 'grid (wgs)', 'calls', 'median us', 'min us', 'max us'))
for key, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print('%-44s %14s %6d %9.2f %9.2f %9.2f' % (key[0][:44], 'x'.join(str(k) for k in key[1:]),
                                              len(v), v[len(v) // 2], v[0], v[-1]))

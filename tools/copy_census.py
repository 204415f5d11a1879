"""Where do the __amd_rocclr_copyBuffer / fillBuffer launches of a step come from?  From a rocprofv3 --kernel-trace directory: for every
such dispatch, the kernels right before and after it on the same queue; prints the most frequent (previous, next) contexts.
    python tools/copy_census.py <dir> [steps]"""
import collections
import csv
import glob
import os
import sys

f = glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True)[0]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
short = lambda n: n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60]
by_q = collections.defaultdict(list)
for r in rows:
    by_q[r.get('Queue_Id', '0')].append(r)
ctx = collections.Counter()
sizes = collections.Counter()
for q, rs in by_q.items():
    for i, r in enumerate(rs):
        n = r['Kernel_Name']
        if 'copyBuffer' not in n and 'fillBuffer' not in n:
            continue
        prev = short(rs[i - 1]['Kernel_Name']) if i > 0 else '-'
        nxt = short(rs[i + 1]['Kernel_Name']) if i + 1 < len(rs) else '-'
        ctx[(short(n), prev, nxt)] += 1
        sizes[(short(n), r.get('Grid_Size', '?'))] += 1
print(f'# copy / fill launches per step by context (kernel, previous on the queue, next on the queue); {steps:g} profiled steps')
for (n, p, x), c in ctx.most_common(40):
    print(f'{c / steps:7.1f}  {n:28s} after {p:60s} before {x}')
print('# by grid size')
for (n, g), c in sizes.most_common(12):
    print(f'{c / steps:7.1f}  {n:28s} grid {g}')

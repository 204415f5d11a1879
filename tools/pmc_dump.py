"""Per-kernel averages of every counter in a rocprofv3 --pmc output directory.   python tools/pmc_dump.py <dir> [name filter]"""
import csv, glob, collections, os, sys
f = glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True)
if not f:
    print('no counter_collection.csv under', sys.argv[1]); sys.exit(1)
flt = sys.argv[2] if len(sys.argv) > 2 else ''
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    if flt in k:
        per[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in sorted(per.items()):
    print(k[:90])
    print('   ', '  '.join(f'{n}={sum(v) / len(v):.4g} (x{len(v)})' for n, v in sorted(c.items())))

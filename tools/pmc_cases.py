"""HBM bytes per launch of each CASE of tools/decoder_gemms.py / tools/encoder_gemms.py from two rocprofv3 PMC passes
(FETCH_SIZE x 2 for the gfx950 wide-read undercount -- MI355X_MICROARCH.md, HBM -- and WRITE_SIZE; KiB in the CSVs), the cases
separated by the marker fills the tools enqueue.    python tools/pmc_cases.py <fetch dir> <write dir> <n cases> [names...]"""
import sys, glob, os, csv


def load(d, name):
    f = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == name]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    return rows


def cases(rows, n):
    marks = [i for i, r in enumerate(rows) if 'FillFunctor<int>' in r['Kernel_Name']][-n:]
    marks.append(len(rows))
    out = []
    for c in range(n):
        seg = [r for r in rows[marks[c] + 1:marks[c + 1]] if 'gemm' in r['Kernel_Name'] or 'reduce' in r['Kernel_Name']]
        per = {}
        for r in seg:
            per.setdefault(r['Kernel_Name'].split('(')[0][-36:], []).append(float(r['Counter_Value']))
        out.append({k: sum(v) / len(v) * 1024.0 for k, v in per.items()})
    return out


n = int(sys.argv[3])
names = sys.argv[4:] or [str(i) for i in range(n)]
fe, wr = cases(load(sys.argv[1], 'FETCH_SIZE'), n), cases(load(sys.argv[2], 'WRITE_SIZE'), n)
for i in range(n):
    f = sum(fe[i].values()) * 2.0
    w = sum(wr[i].values())
    print(f'{names[i] if i < len(names) else i:34s} fetch {f / 1e6:8.1f} MB   write {w / 1e6:8.1f} MB   total {(f + w) / 1e6:8.1f} MB')

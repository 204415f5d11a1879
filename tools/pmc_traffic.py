"""HBM traffic of the bf16 MFMA GEMM launches of bench.py from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE need separate
passes: 3 + 2 of the 4 TCC slots).  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half the bytes of wide
coalesced reads -> doubled here.  rocprofv3 reports both in KiB.
    rocprofv3 --pmc FETCH_SIZE -d out/f -o p --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing
    rocprofv3 --pmc WRITE_SIZE -d out/w -o p --output-format csv -- python bench.py ... (same)
    python tools/pmc_traffic.py out/f out/w profiles/r01_pmc_traffic.json"""
import sys, os, glob, csv, json, collections
def load(d, name):
    f = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)[0]
    per = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != name:
            continue
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        per[k][0] += float(r['Counter_Value']); per[k][1] += 1
    return per
fe, wr = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
out = {'unit': 'bytes per launch', 'fetch_correction': 'x2 (gfx950 wide-read undercount)', 'kernels': {}}
tot_b, tot_n = 0.0, 0
for k in sorted(fe, key=lambda k: -fe[k][0]):
    n = fe[k][1]
    fb = 2.0 * 1024.0 * fe[k][0] / n
    wb = 1024.0 * wr[k][0] / max(wr[k][1], 1) if k in wr else 0.0
    out['kernels'][k] = {'launches': n, 'fetch': round(fb), 'write': round(wb), 'traffic': round(fb + wb)}
    if k.startswith('gemm_bf16'):
        tot_b += (fb + wb) * n; tot_n += n
out['gemm_bf16_traffic_per_launch'] = round(tot_b / max(tot_n, 1))
out['gemm_bf16_launches'] = tot_n
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != 'kernels'}))
for k, v in list(out['kernels'].items())[:12]:
    print(k[:60], v)

"""GPU busy time (union of kernel intervals) vs wall for the steady-state steps of a rocprofv3 --kernel-trace of bench.py.
usage: python tools/trace_union.py <trace dir> [kernel-name marker of a once-per-step kernel, default adamw]"""
import sys, glob, csv, os
f = glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True)[0]
mark = sys.argv[2] if len(sys.argv) > 2 else 'adamw'
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))))
steps = [s for s, e, n in rows if mark in n]
lo, hi = steps[-4], steps[-1]            # three full steps
iv = [(s, e) for s, e, n in rows if s >= lo and s < hi]
busy, cur_s, cur_e = 0, None, None
for s, e in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = hi - lo
gaps = []
pe = None
for s, e in sorted(iv):
    if pe is not None and s > pe: gaps.append(s - pe)
    pe = e if pe is None else max(pe, e)
print(f'steps: 3  wall/step {wall / 3e6:.2f} ms  busy/step {busy / 3e6:.2f} ms  idle/step {(wall - busy) / 3e6:.2f} ms  kernels/step {len(iv) / 3:.0f}')
print(f'gaps/step: {len(gaps) / 3:.0f}, >20us: {sum(1 for g in gaps if g > 20000) / 3:.0f} totalling {sum(g for g in gaps if g > 20000) / 3e6:.2f} ms; sum kernel time/step {sum(e - s for s, e in iv) / 3e6:.2f} ms')
# the largest gaps and the kernels around them
ev = sorted([(s, e, n) for s, e, n in rows if s >= lo and s < hi])
pe, pn, big = None, None, []
for s, e, n in ev:
    if pe is not None and s > pe: big.append((s - pe, pn, n))
    if pe is None or e > pe: pe, pn = e, n
for g, a, b in sorted(big, reverse=True)[:24]:
    print(f'{g / 1e3:8.1f} us  after {a[:60]:60s} before {b[:60]}')

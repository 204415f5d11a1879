"""MFMA-pipe occupancy per GEMM kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE pass of
tools/encoder_gemms.py.  SQ_VALU_MFMA_BUSY_CYCLES is summed over all 1 024 SIMDs (check: fc1 = 3.6 M MFMAs x 32 cycles = 117 M);
GRBM_GUI_ACTIVE is summed over the 8 XCDs (fc1: 2.79 M = 8 x 172 us x 2.02 GHz).  So the MFMA pipe's share of the kernel's wall
time is (busy / 1024) / (GUI / 8), and GUI / 8 / duration is the effective clock.
    python tools/pmc_mfma.py <dir>"""
import collections
import csv
import glob
import os
import sys

f = glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    if 'gemm' not in k:
        continue
    per[k][r['Counter_Name']].append(float(r['Counter_Value']))
print(f'{"kernel":48s} {"launches":>8s} {"MFMA busy cyc/SIMD":>19s} {"GUI cyc/XCD":>12s} {"MFMA pipe busy":>15s}')
for k, c in sorted(per.items()):
    n = len(c.get('GRBM_GUI_ACTIVE', []))
    if not n:
        continue
    mf = sum(c.get('SQ_VALU_MFMA_BUSY_CYCLES', [0.0])) / max(len(c.get('SQ_VALU_MFMA_BUSY_CYCLES', [1])), 1)
    gui = sum(c['GRBM_GUI_ACTIVE']) / n
    print(f'{k[:48]:48s} {n:8d} {mf / 1024.0:19.0f} {gui / 8.0:12.0f} {(mf / 1024.0) / (gui / 8.0) if gui else 0:15.3f}')

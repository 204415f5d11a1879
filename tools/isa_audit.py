#!/usr/bin/env python
"""ISA audit of the built ping-pong GEMM objects: no scratch in the flavoured kernels, and no instruction touches the
destination registers of an inline-asm `ds_read_b64_tr_b16` before the `s_waitcnt lgkmcnt(0)` that makes them valid (the compiler
neither counts nor waits for an asm load, gemm_pp_body.h: frag_ks / frag_fence).

    python tools/isa_audit.py [multimae_amd/csrc/build/gemm_bf16_pp_fl.o ...]      # default: the three pp objects

Works on the host object files: the gfx950 code object is pulled out with `llvm-objdump --offloading`, kernel resource usage read
from its notes, the hazard check run over its disassembly.  Exit code 1 on any finding."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
DEFAULT = [os.path.join(ROOT, 'multimae_amd', 'csrc', 'build', f) for f in ('gemm_bf16_pp_fl.o', 'gemm_bf16_pp.o', 'mxfp8.o')]


def code_object(obj: str, tmp: str) -> str:
    dst = os.path.join(tmp, os.path.basename(obj))
    shutil.copy(obj, dst)
    subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '--offloading', dst], check=True, capture_output=True)
    for f in os.listdir(tmp):
        if f.startswith(os.path.basename(obj)) and 'gfx950' in f:
            return os.path.join(tmp, f)
    raise RuntimeError(f'no gfx950 bundle in {obj}')


def resources(co: str):
    """{kernel: dict(vgpr, agpr, scratch, spill)} from the code object's metadata note."""
    txt = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', co], check=True, capture_output=True, text=True).stdout
    out, cur = {}, {}
    for line in txt.splitlines():
        m = re.match(r'\s+(?:- )?\.(\w+):\s+(\S+)', line)
        if not m:
            continue
        k, v = m.groups()
        if k in ('agpr_count', 'vgpr_count', 'private_segment_fixed_size', 'vgpr_spill_count', 'sgpr_spill_count'):
            cur[k] = int(v)
        elif k == 'name' and not v.startswith('.'):
            cur['name'] = v
        if k == 'wavefront_size' or (k == 'vgpr_spill_count' and 'name' in cur):
            pass
        if 'name' in cur and all(x in cur for x in ('vgpr_count', 'private_segment_fixed_size', 'vgpr_spill_count')):
            out[cur['name']] = dict(vgpr=cur['vgpr_count'], agpr=cur.get('agpr_count', 0), scratch=cur['private_segment_fixed_size'],
                                    spill=cur['vgpr_spill_count'])
            cur = {}
    return out


def regs_of(tok: str):
    """VGPR numbers named by one operand token: v12, v[12:15]."""
    m = re.fullmatch(r'v(\d+)', tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def hazards(co: str):
    """[(kernel, instruction line, registers)] -- uses of a transposing read's destination before the wait that covers it."""
    txt = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--no-show-raw-insn', co], check=True, capture_output=True, text=True).stdout
    found, counts = [], {}
    kernel, pending = None, set()
    for line in txt.splitlines():
        m = re.match(r'^[0-9a-f]+ <(\S+)>:', line)
        if m:
            kernel, pending = m.group(1), set()
            continue
        ins = line.split('//')[0].strip()
        if not ins or kernel is None:
            continue
        parts = ins.replace(',', ' ').split()
        op, toks = parts[0], parts[1:]
        if op == 'ds_read_b64_tr_b16':
            counts[kernel] = counts.get(kernel, 0) + 1
            used = set().union(*[regs_of(t) for t in toks[1:]]) if len(toks) > 1 else set()
            if used & pending:
                found.append((kernel, ins, sorted(used & pending)))
            pending |= regs_of(toks[0])
            continue
        if op == 's_waitcnt' and re.search(r'lgkmcnt\(0\)', ins):
            pending = set()
            continue
        if op == 's_waitcnt' and 'lgkmcnt' not in ins and re.fullmatch(r'0x[0-9a-f]+|\d+', toks[0] if toks else ''):
            v = int(toks[0], 0)
            if ((v >> 8) & 0xf) == 0:            # raw immediate: lgkmcnt field (bits 11:8) zero
                pending = set()
            continue
        if pending:
            used = set().union(*[regs_of(t) for t in toks]) if toks else set()
            if used & pending:
                found.append((kernel, ins, sorted(used & pending)))
    return found, counts


def audit(objs):
    problems, report = [], []
    with tempfile.TemporaryDirectory() as tmp:
        for obj in objs:
            co = code_object(obj, tmp)
            res = resources(co)
            hz, counts = hazards(co)
            for k, r in sorted(res.items()):
                if 'gemm_bf16_pp' not in k and 'gemm_mxfp8' not in k:
                    continue
                report.append((os.path.basename(obj), k, r, counts.get(k, 0)))
            for k, ins, regs in hz:
                problems.append(f'{os.path.basename(obj)}: {k}: `{ins}` touches v{regs} before s_waitcnt lgkmcnt(0)')
    return report, problems


def flavoured(name: str) -> bool:
    """gemm_bf16_pp_kernel<TM, AKS, BKS, FL != 0, KF>: the instantiations the training step launches (epilogue fixed at compile time)."""
    m = re.search(r'gemm_bf16_pp_kernelILi(\d)ELb(\d)ELb(\d)ELi(\d+)ELb(\d)E', name)
    return bool(m) and int(m.group(4)) != 0


def main():
    objs = sys.argv[1:] or DEFAULT
    report, problems = audit(objs)
    for obj, k, r, n_tr in report:
        print(f'{obj:22s} vgpr {r["vgpr"]:3d} agpr {r["agpr"]:3d} scratch {r["scratch"]:5d} B spills {r["spill"]:3d}  tr-reads {n_tr:4d}  {k}')
        if flavoured(k) and (r['scratch'] or r['spill']):
            problems.append(f'{obj}: {k}: flavoured kernel with scratch ({r["scratch"]} B, {r["spill"]} spilled VGPRs)')
    for p in problems:
        print('PROBLEM:', p)
    print(f'{len(report)} kernels audited, {len(problems)} problem(s)')
    return 1 if problems else 0


if __name__ == '__main__':
    sys.exit(main())

"""How far is bench.py's CPU baseline "port" (oracle/multimae_oracle.py + torch autograd: what the GPU box can run) from the REFERENCE's
own classes (/root/reference, present only in the build container)?  Times the same cfg3 step (fwd -> 4 losses -> backward -> AdamW, fp32,
B = 16) with both at EQUAL thread counts in THIS container and writes the ratio bench.py carries in its line
(cpu_baseline.port_over_reference_time).  VERDICT r5 item 7c.
    python tools/cpu_port_vs_reference.py [threads] > profiles/r06_cpu_port_vs_reference.json"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B, steps = 16, 5
out = {'config': 'cfg3', 'B': B, 'threads': threads, 'steps': steps, 'cpu': bench.cpu_model_name(), 'torch': torch.__version__}
torch.set_num_threads(threads)
for kind, mk in (('reference', bench._reference_step_fn), ('port', bench._oracle_step_fn), ('reference_again', bench._reference_step_fn)):
    step = mk('cfg3', B)
    if step is None:
        out[kind] = None
        continue
    ts = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    ts = sorted(ts[1:])
    out[kind] = {'median_s_per_step': round(ts[len(ts) // 2], 4), 'img_s': round(B / ts[len(ts) // 2], 3), 'all_s': [round(t, 4) for t in ts]}
if out.get('reference') and out.get('port'):
    ref = min(out['reference']['median_s_per_step'], (out.get('reference_again') or out['reference'])['median_s_per_step'])
    out['port_over_reference_time'] = round(out['port']['median_s_per_step'] / ref, 4)
print(json.dumps(out, indent=1))

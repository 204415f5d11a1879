"""Which host-side calls of one production step become device copies (hipMemcpyAsync -> __amd_rocclr_copyBuffer)?  torch.profiler over one bench-shaped step with stacks:
prints the aten::copy_ / aten::_to_copy / aten::to calls grouped by input shapes and their innermost repository frame.   python tools/copy_sources.py"""
import collections
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import multimae_amd as M
from multimae_amd.optim import FusedAdamW

model, doms = bench.build_model('cfg3')
model.cuda(); model.build_arena()
M.engine.set_direct_grads(True); M.engine.set_adapter_streams(True); M.engine.set_wgrad_stream(True)
opt = FusedAdamW(model, lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05)
x = bench.synthetic_batch(doms, 256, torch.device('cuda'), seed=0)
tgt = dict(x, norm_rgb=x['rgb'])
fns = bench.loss_fns()


def step():
    opt.zero_grad()
    preds, masks = model(x, num_encoded_tokens=98, alphas=1.0, sample_tasks_uniformly=False, fp32_output_adapters=['semseg'])
    mk = dict(masks, norm_rgb=masks['rgb'])
    loss = sum(fns[k](preds[k].float(), tgt[k], mask=mk[k]) for k in preds)
    loss.backward()
    opt.step(loss)


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cnt = collections.Counter()
for e in prof.events():
    if e.name in ('aten::copy_', 'aten::_to_copy', 'aten::to', 'aten::pin_memory', 'aten::item', 'aten::_local_scalar_dense', 'aten::fill_', 'aten::zero_'):
        frame = next((f for f in (e.stack or []) if root in f and 'tools/' not in f), (e.stack or ['?'])[0] if e.stack else '?')
        cnt[(e.name, str(e.input_shapes)[:60], frame.replace(root + '/', '')[:110])] += 1
for (n, sh, fr), c in cnt.most_common(45):
    print(f'{c:4d}  {n:24s} {sh:62s} {fr}')
mem = collections.Counter()
for e in prof.events():
    if 'emcpy' in e.name or 'emset' in e.name:
        mem[e.name] += 1
print(dict(mem))

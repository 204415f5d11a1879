"""Which ATen operators (i.e. PyTorch kernels, not library calls) does one production step still launch?  torch.profiler over one
bench-shaped step; prints the operators with device time, their call counts and input shapes.   python tools/aten_census.py"""
import sys, os
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    step()
    torch.cuda.synchronize()
ev = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith('aten::') and e.device_time_total > 0]
ev.sort(key=lambda e: -e.device_time_total)
for e in ev[:40]:
    print(f'{e.device_time_total:9.1f} us  x{e.count:3d}  {e.key:28s} {str(e.input_shapes)[:110]}')

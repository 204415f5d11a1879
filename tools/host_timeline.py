"""Host-side timeline of the cfg3 step: where does the launching thread spend its time / block?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import multimae_amd as M
from multimae_amd.optim import FusedAdamW
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model, doms = bench.build_model('cfg3')
model.to(dev); model.build_arena()
M.engine.set_precision('bf16'); M.engine.set_direct_grads(True); M.engine.set_adapter_streams(True); M.engine.set_wgrad_stream(True)
opt = FusedAdamW(model, lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05)
x = bench.synthetic_batch(doms, 256, dev, seed=0)
tgt = dict(x, norm_rgb=x['rgb'])
fns = bench.loss_fns()
T = []
def step():
    t = [time.perf_counter()]
    opt.zero_grad(); t.append(time.perf_counter())
    preds, masks = model(x, num_encoded_tokens=98, alphas=1.0, sample_tasks_uniformly=False, fp32_output_adapters=['semseg']); t.append(time.perf_counter())
    mk = dict(masks, norm_rgb=masks['rgb'])
    loss = sum(fns[k](preds[k].float(), tgt[k], mask=mk[k]) for k in preds); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    T.append(t)
for _ in range(4): step()
torch.cuda.synchronize(); T.clear()
t0 = time.perf_counter()
for _ in range(8): step()
th = time.perf_counter() - t0
torch.cuda.synchronize()
tw = time.perf_counter() - t0
print(f'host {th / 8 * 1e3:.2f} ms/step, wall {tw / 8 * 1e3:.2f} ms/step')
names = ['zero_grad', 'forward', 'losses', 'backward', 'opt.step']
for i, t in enumerate(T):
    print(i, ' '.join(f'{n}={1e3 * (t[j + 1] - t[j]):6.2f}' for j, n in enumerate(names)), f' start@{1e3 * (t[0] - t0):7.2f}')

# where the launching thread's own time goes (the run-ahead bound's waits show up as Event.synchronize)
import cProfile, pstats
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(45)

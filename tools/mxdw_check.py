"""MX-fp8 mode, cfg5: how far the gradients move when the weight-gradient products run on the scaled MFMA (ops.mx_wgrad)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import multimae_amd as M
from multimae_amd import ops, engine
torch.manual_seed(0)
model, doms = bench.build_model('cfg5'); model.cuda(); arena = model.build_arena()
M.engine.set_direct_grads(True)
x = bench.synthetic_batch(doms, 16, torch.device('cuda'), seed=0)
fns = bench.loss_fns()
def grads(flag):
    ops.mx_wgrad(flag)
    arena.zero_grad(); arena.rebind_grads()
    torch.manual_seed(1)
    with engine.precision('mxfp8'):
        preds, masks = model(x, num_encoded_tokens=196, alphas=1.0, sample_tasks_uniformly=False, fp32_output_adapters=['semseg'])
        tgt = dict(x, norm_rgb=x['rgb']) if 'norm_rgb' in preds else x
        mk = dict(masks, norm_rgb=masks['rgb']) if 'norm_rgb' in preds else masks
        loss = sum(fns[k](preds[k].float(), tgt[k], mask=mk[k]) for k in preds)
        loss.backward()
    torch.cuda.synchronize()
    return {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}, float(loss)
g1, l1 = grads(True); g0, l0 = grads(False)
print('loss', l1, l0, 'prev policy', ops.mx_wgrad())
for n in ['encoder.0.attn.qkv.weight','encoder.5.mlp.fc1.weight','encoder.23.mlp.fc2.weight','encoder.3.attn.proj.weight','encoder.3.attn.qkv.bias']:
    a,b=g1[n],g0[n]; print(n, float((a-b).norm()/b.norm()))

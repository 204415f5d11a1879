"""Census of the GEMM / colsum / axpy launches of one cfg3 pre-training step (shapes x counts, by flops)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import multimae_amd as M
from multimae_amd import ops
from multimae_amd.optim import FusedAdamW
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model, doms = bench.build_model('cfg3')
model.to(dev)
model.build_arena()
M.engine.set_precision('bf16'); M.engine.set_direct_grads(True)
opt = FusedAdamW(model, lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05)
B = 256
x = bench.synthetic_batch(doms, B, dev, seed=0)
tgt = dict(x, norm_rgb=x['rgb'])
fns = bench.loss_fns()
def step():
    opt.zero_grad()
    preds, masks = model(x, num_encoded_tokens=98, alphas=1.0, sample_tasks_uniformly=False, fp32_output_adapters=['semseg'])
    mk = dict(masks, norm_rgb=masks['rgb'])
    loss = sum(fns[k](preds[k].float(), tgt[k], mask=mk[k]) for k in preds)
    loss.backward(); opt.step()
step(); torch.cuda.synchronize()
gem, cs = collections.Counter(), collections.Counter()
og, oc = ops.gemm, ops.colsum
def lg(A, Bm, C, Mm, N, K, **kw):
    gem[(str(A.dtype)[6:], str(C.dtype)[6:], Mm, N, K, int(kw.get('a_trans', False)), int(kw.get('b_trans', False)), kw.get('batch', 1),
         kw.get('epi', 0), kw.get('bias') is not None, kw.get('resid') is not None, kw.get('accumulate', False))] += 1
    return og(A, Bm, C, Mm, N, K, **kw)
def lc(xx, out, acc=False, *a, **k):
    cs[(str(xx.dtype)[6:], tuple(xx.shape))] += 1
    return oc(xx, out, acc, *a, **k)
ops.gemm, ops.colsum = lg, lc
import multimae_amd.functions as F_
step(); torch.cuda.synchronize()
tot = 0.0
print('dtype cdt M N K at bt batch epi bias resid acc : count  GF each  GF total')
for k, c in sorted(gem.items(), key=lambda kv: -kv[1] * kv[0][2] * kv[0][3] * kv[0][4] * kv[0][7]):
    gf = 2.0 * k[2] * k[3] * k[4] * k[7] / 1e9
    tot += gf * c
    print(k, ':', c, f'{gf:9.1f} {gf * c:10.1f}')
print('total GF', tot)
print('colsum calls:')
for k, c in sorted(cs.items(), key=lambda kv: -kv[1] * kv[0][1][0] * kv[0][1][-1]):
    print(k, c)

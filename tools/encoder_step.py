"""The north star's "ViT-B encoder step at bs=256/GPU": forward + backward of the 12-block encoder stack on
(256, 99, 768) tokens (13.19 TFLOP of GEMM work per step), timed with HIP events; prints TFLOP/s
and the fraction of the 2.5 PFLOP/s bf16 MFMA peak."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multimae_amd as M
from multimae_amd.multimae_utils import Block, run_blocks
from functools import partial
from torch import nn

B, N, D, L = 256, 99, 768, 12
torch.manual_seed(0)
enc = nn.Sequential(*[Block(D, 12, mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6)) for _ in range(L)]).cuda()
arena = M.engine.ParamArena(enc)
M.engine.set_direct_grads(True)
M.engine.set_wgrad_stream(True)
x = torch.randn(B, N, D, device='cuda', requires_grad=True)
g = torch.randn(B, N, D, device='cuda')


def step():
    arena.zero_grad()
    y = run_blocks(enc, x, root=enc)
    y.backward(g)
    M.engine.join_wgrad_streams()


for _ in range(3):
    step()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
iters = 10
torch.cuda.synchronize()
e0.record()
for _ in range(iters):
    step()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
tflop = 51.53e9 * B / 1e12
print(json.dumps({'encoder_step_ms': round(ms, 3), 'tflop_per_step': round(tflop, 2), 'tflops': round(tflop / ms * 1e3, 1),
                  'frac_of_2.5PF': round(tflop / ms * 1e3 / 2500, 4), 'batch': B, 'tokens': N}))

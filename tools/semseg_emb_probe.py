"""Time of the class-embedding gradient (SemSegInputAdapter's nn.Embedding backward) at the cfg3 geometry: the deterministic two-launch form
against the float-atomic kernel it replaced.  HIP events, 20 launches each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimae_amd import ops
B, H, W, E, ph, pw, n_cls, n_sel = 256, 56, 56, 64, 4, 4, 133, 98
nh, nw = H // ph, W // pw
torch.manual_seed(0)
cls = torch.randint(0, n_cls, (B, H, W), device='cuda')
sel = torch.stack([torch.randperm(3 * 196)[:n_sel] for _ in range(B)]).cuda()
d_rows = torch.randn(B * n_sel, E * ph * pw, device='cuda').to(torch.bfloat16)
g = torch.zeros(n_cls, E, device='cuda')
kw = dict(B=B, H=H, W=W, E=E, ph=ph, pw=pw, n_sel=n_sel, k_off=0, tok_off=392, n_patches=196, n_cls=n_cls)
for det in (True, False, True, False):
    f = lambda: ops.semseg_emb_bwd(d_rows, cls, sel, g, accumulate=True, deterministic=det, **kw)
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20):
        f()
    e1.record(); torch.cuda.synchronize()
    print(f"{'deterministic (two launches)' if det else 'float atomics':30s} {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us")

"""Round 6 probe: does the encoder step get faster as TWO half-batch pipelines on two streams (each pipeline's persistent GEMM
grids at most n_cu - reserve workgroups wide, so that the two are resident side by side and drift out of phase) than as one
full-batch pipeline that owns the chip?  The HBM-bound phases of one pipeline (LayerNorm, attention, the store bursts of the GELU
epilogues) would then sit under the MFMA loops of the other.  Two model copies stand in for one (timing only: the weight-gradient
accumulation order of a shared gradient arena is what a real implementation would add).
    python tools/two_pipe_probe.py [--fwd-only]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multimae_amd as M
from multimae_amd import ops
from multimae_amd.multimae_utils import Block, run_blocks
from functools import partial
from torch import nn

B, N, D, L = 256, 99, 768, 12
FWD_ONLY = '--fwd-only' in sys.argv
torch.manual_seed(0)
mk = lambda: nn.Sequential(*[Block(D, 12, mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6)) for _ in range(L)]).cuda()
encs = [mk(), mk()]
arenas = [M.engine.ParamArena(e) for e in encs]
M.engine.set_direct_grads(True)
M.engine.set_wgrad_stream(True)
x_full = torch.randn(B, N, D, device='cuda', requires_grad=True)
g_full = torch.randn(B, N, D, device='cuda')
xh = [torch.randn(B // 2, N, D, device='cuda', requires_grad=True) for _ in range(2)]
gh = [torch.randn(B // 2, N, D, device='cuda') for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def step_one():
    arenas[0].zero_grad()
    y = run_blocks(encs[0], x_full, root=encs[0])
    if not FWD_ONLY:
        y.backward(g_full)
    M.engine.join_wgrad_streams()


def step_two():
    cur = torch.cuda.current_stream()
    ys = []
    for a in arenas:
        a.zero_grad()
    for i in range(2):
        streams[i].wait_stream(cur)
        with torch.cuda.stream(streams[i]):
            ys.append(run_blocks(encs[i], xh[i], root=encs[i]))
    if not FWD_ONLY:
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                ys[i].backward(gh[i])
                M.engine.join_wgrad_streams()
    for s in streams:
        cur.wait_stream(s)


def timeit(fn, iters=8):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


n_cu = torch.cuda.get_device_properties(0).multi_processor_count
out = {'fwd_only': FWD_ONLY, 'n_cu': n_cu}
out['one pipeline, B=256, whole chip'] = round(timeit(step_one), 3)
for res in (0, n_cu // 2, n_cu // 2 - 16, n_cu // 4):
    ops.gemm_cu_reserve(res)
    out[f'two pipelines, B=128 each, GEMM grids <= {n_cu - res} workgroups'] = round(timeit(step_two), 3)
ops.gemm_cu_reserve(0)
out['one pipeline again'] = round(timeit(step_one), 3)
ops.gemm_cu_reserve(n_cu // 2)
out['one pipeline, GEMM grids <= half the chip (what a half costs alone)'] = round(timeit(step_one), 3)
ops.gemm_cu_reserve(0)
print(json.dumps(out, indent=1))

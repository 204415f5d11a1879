#!/usr/bin/env python
"""bench.py -- pre-training step throughput of the MI355X-native MultiMAE engine.

Metric (BASELINE.json): pre-train images/sec (whole node), ViT-B RGB+D+S 224^2, 98 visible tokens,
bs 256 / GPU, 1/2/4/8 MI355X.  One "step" = what run_pretraining_multimae.py:500-537 does per
iteration: masked forward (Dirichlet sampling, alphas 1.0) -> 4 masked losses (rgb MSE, depth L1,
semseg CE in the fp32 adapter, norm-pix MSE) -> backward -> gradient all-reduce (N > 1) ->
AdamW(betas .9/.95, wd .05 on every tensor).  Synthetic inputs resident in HBM, random-init weights.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 20 --warmup 5
    python bench.py --gpus 8 --steps 20 --warmup 5        # no launcher in the environment: re-executes itself under the line above

Prints ONE JSON line (rank 0) with the driver's contract plus "roofline" and "cpu_baseline".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

ALG_GFLOP_PER_IMG = {'cfg3': 65.4, 'cfg2': 57.4, 'cfg5': 382.0}   # GEMM flops fwd+bwd, visible-patch embedding (SURVEY 8d / BASELINE.md 2; cfg5: 385.39 with all patches embedded)
PEAK_BF16_TFLOPS = 2500.0                            # dense MFMA bf16, MI355X_MICROARCH.md


def build_model(cfg: str):
    import multimae_amd as M
    doms = ['rgb'] if cfg == 'cfg2' else ['rgb', 'depth', 'semseg']
    ins = {}
    for d in doms:
        if d == 'semseg':
            ins[d] = M.SemSegInputAdapter(num_classes=133, dim_class_emb=64, interpolate_class_emb=False, stride_level=4,
                                          patch_size_full=16)
        else:
            ins[d] = M.PatchedInputAdapter(num_channels=3 if d == 'rgb' else 1, stride_level=1, patch_size_full=16)
    outs = {}
    for key, task in [(d, d) for d in doms] + [('norm_rgb', 'rgb')]:
        ch = {'rgb': 3, 'depth': 1, 'semseg': 133}[task]
        outs[key] = M.SpatialOutputAdapter(num_channels=ch, stride_level=4 if task == 'semseg' else 1, patch_size_full=16,
                                           dim_tokens=256, depth=2, num_heads=8, use_task_queries=True, task=task,
                                           context_tasks=list(doms), use_xattn=True)
    model = M.create_model('pretrain_multimae_large' if cfg == 'cfg5' else 'pretrain_multimae_base', input_adapters=ins, output_adapters=outs, num_global_tokens=1,
                           drop_path_rate=0.0)
    return model.train(), doms


def synthetic_batch(doms, B, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    x = {}
    if 'rgb' in doms:
        x['rgb'] = torch.randn(B, 3, 224, 224, device=device, generator=g)
    if 'depth' in doms:
        x['depth'] = torch.randn(B, 1, 224, 224, device=device, generator=g)
    if 'semseg' in doms:
        x['semseg'] = torch.randint(0, 133, (B, 56, 56), device=device, generator=g)
    return x


def loss_fns():
    import multimae_amd as M
    return {'rgb': M.MaskedMSELoss(16, 1), 'depth': M.MaskedL1Loss(16, 1), 'semseg': M.MaskedCrossEntropyLoss(16, 4),
            'norm_rgb': M.MaskedMSELoss(16, 1, norm_pix=True)}


def pmc_traffic(dom):
    """HBM bytes per bf16-GEMM launch (average over the launches of one step) from the committed rocprofv3 PMC passes of this
    same command (tools/pmc_traffic.py -> profiles/*_pmc_traffic.json; FETCH_SIZE x2 gfx950 correction + WRITE_SIZE).
    Counters cannot be read from inside the process, so this is a RECORDED figure (the JSON line says which file it came from:
    roofline.traffic_source), or None if no PMC pass is committed."""
    if dom != torch.bfloat16:
        return None, None
    for name in ('r06_pmc_traffic.json', 'r05_pmc_traffic.json', 'r04_pmc_traffic.json', 'r03_pmc_traffic.json', 'r02_pmc_traffic.json', 'r01_pmc_traffic.json'):
        f = os.path.join(ROOT, 'profiles', name)
        try:
            return json.load(open(f)).get('gemm_bf16_traffic_per_launch'), f'recorded, not of this run: profiles/{name} (rocprofv3 --pmc FETCH_SIZE x 2 and WRITE_SIZE passes of this command on the build that file names, tools/gpu_r6.sh / gpu_r5.sh final; counters cannot be read in-process, and a --pmc pass may not be combined with the timed run)'
        except (OSError, ValueError):
            continue
    return None, None


def log(msg):
    print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)


def cpu_model_name():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown CPU'


REF = '/root/reference'


def _reference_step_fn(cfg: str, B: int):
    """The reference's own classes (SURVEY.md Appendix B import recipe) when its checkout is present (the build container;
    never on the GPU box): returns a callable running one fwd -> 4 losses -> backward -> AdamW step, or None."""
    if not os.path.isdir(os.path.join(REF, 'multimae')):
        return None
    import types
    saved_mods = {k: sys.modules.get(k) for k in ('utils', 'multimae')}
    try:
        pkg = types.ModuleType('utils')
        pkg.__path__ = [os.path.join(REF, 'utils')]
        sys.modules['utils'] = pkg
        for k in [k for k in sys.modules if k == 'multimae' or k.startswith('multimae.')]:
            del sys.modules[k]
        sys.path.insert(0, REF)
        sys.dont_write_bytecode = True
        import multimae.multimae as rm
        import multimae.input_adapters as ria
        import multimae.output_adapters as roa
        import multimae.criterion as rc
        sys.path.remove(REF)
    except Exception as e:                                  # noqa: BLE001 -- any import trouble: fall back to the oracle
        log(f'cpu_baseline: reference import failed ({e!r}); using the oracle port')
        return None
    doms = ['rgb'] if cfg == 'cfg2' else ['rgb', 'depth', 'semseg']
    ins = {}
    for d in doms:
        if d == 'semseg':
            ins[d] = ria.SemSegInputAdapter(num_classes=133, dim_class_emb=64, interpolate_class_emb=False, stride_level=4, patch_size_full=16)
        else:
            ins[d] = ria.PatchedInputAdapter(num_channels=3 if d == 'rgb' else 1, stride_level=1, patch_size_full=16)
    outs = {}
    for key, task in [(d, d) for d in doms] + [('norm_rgb', 'rgb')]:
        ch = {'rgb': 3, 'depth': 1, 'semseg': 133}[task]
        outs[key] = roa.SpatialOutputAdapter(num_channels=ch, stride_level=4 if task == 'semseg' else 1, patch_size_full=16, dim_tokens=256,
                                             depth=2, num_heads=8, use_task_queries=True, task=task, context_tasks=list(doms), use_xattn=True)
    torch.manual_seed(0)
    model = rm.pretrain_multimae_base(ins, outs, num_global_tokens=1, drop_path_rate=0.0).train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05)
    fns = {'rgb': rc.MaskedMSELoss(16, 1), 'depth': rc.MaskedL1Loss(16, 1), 'semseg': rc.MaskedCrossEntropyLoss(16, 4),
           'norm_rgb': rc.MaskedMSELoss(16, 1, norm_pix=True)}
    x = {}
    if 'rgb' in doms:
        x['rgb'] = torch.randn(B, 3, 224, 224)
    if 'depth' in doms:
        x['depth'] = torch.randn(B, 1, 224, 224)
    if 'semseg' in doms:
        x['semseg'] = torch.randint(0, 133, (B, 56, 56))
    tgt = dict(x, norm_rgb=x['rgb'])

    def step():
        preds, masks = model(x, num_encoded_tokens=98, alphas=1.0, fp32_output_adapters=['semseg'] if 'semseg' in doms else [])
        masks = dict(masks, norm_rgb=masks['rgb'])
        loss = sum(fns[k](preds[k].float(), tgt[k], mask=masks[k]) for k in preds)
        opt.zero_grad()
        loss.backward()
        opt.step()
    return step


def _oracle_step_fn(cfg: str, B: int):
    """The oracle (CPU restatement pinned to the reference, oracle/multimae_oracle.py): the same step in fp32."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import multimae_oracle as orc
    model, doms = build_model(cfg)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    train = {n for n, p in model.named_parameters() if p.requires_grad}
    del model
    ocfg = orc.standard_config(doms)
    torch.manual_seed(0)
    x = {}
    if 'rgb' in doms:
        x['rgb'] = torch.randn(B, 3, 224, 224)
    if 'depth' in doms:
        x['depth'] = torch.randn(B, 1, 224, 224)
    if 'semseg' in doms:
        x['semseg'] = torch.randint(0, 133, (B, 56, 56))
    m = {n: torch.zeros_like(sd[n]) for n in train}
    v = {n: torch.zeros_like(sd[n]) for n in train}
    count = [0]

    def step():
        count[0] += 1
        dist, tn, an = orc.draw_mask_randoms(B, [196] * len(doms), 1.0)
        spt = orc.samples_per_task_from_dirichlet(dist, 98)
        mask_all, ik, ir = orc.masks_from_noise(spt, tn, an, 98)
        sdo = {k: (t.requires_grad_(True) if k in train else t) for k, t in sd.items()}
        preds = orc.multimae_forward(x, sdo, ocfg, ik, ir)
        loss = sum(orc.pretrain_losses(preds, x, mask_all, ocfg, {d: 196 for d in doms}).values())
        loss.backward()
        with torch.no_grad():
            grads = {n: sdo[n].grad for n in train}
            for n in train:
                sdo[n].requires_grad_(False)
            orc.adamw_step({n: sd[n] for n in train}, grads, m, v, count[0], 1e-4, 0.05)
            for n in train:
                sd[n].grad = None
    return step


def port_vs_reference(kind: str) -> dict:
    """How far the oracle port is from the reference's own classes, measured where both can run (the build container, equal threads:
    tools/cpu_port_vs_reference.py -> profiles/r06_cpu_port_vs_reference.json) -- so a "port" baseline on the GPU box can be read as
    a "reference" one."""
    try:
        r = json.load(open(os.path.join(ROOT, 'profiles', 'r06_cpu_port_vs_reference.json')))
        ratio = r['port_over_reference_time']
    except (OSError, ValueError, KeyError):
        return {}
    out = {'port_over_reference_time': ratio,
           'port_over_reference_source': f"recorded in the build container (profiles/r06_cpu_port_vs_reference.json: cfg3 step, B = {r['B']}, {r['threads']} threads of a {r['cpu']}, "
                                         f"reference {r['reference']['median_s_per_step']} s/step, port {r['port']['median_s_per_step']} s/step)"}
    return out


def cpu_baseline(cfg: str, sample_B: int, steps: int, threads: int = 0):
    """BASELINE.md section 3 protocol: the same step (fwd -> 4 losses -> backward -> AdamW) in fp32 on the host cores, B = 16,
    1 warm-up + `steps` timed steps, median; the thread count is the best of a quick 8/16/32/64 sweep (1 + 1 steps each).
    Runs the reference's own classes where /root/reference exists (kind "reference"); on the GPU box, where it does not, the
    oracle port of the same arithmetic (kind "port")."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    step = _reference_step_fn(cfg, sample_B)
    kind = 'reference' if step is not None else 'port'
    if step is None:
        step = _oracle_step_fn(cfg, sample_B)

    def timed(n_threads, n_steps):
        torch.set_num_threads(n_threads)
        ts = []
        for i in range(n_steps + 1):
            t0 = time.perf_counter()
            step()
            ts.append(time.perf_counter() - t0)
        ts = sorted(ts[1:])
        return ts[(len(ts) - 1) // 2]
    sweep = {}
    if threads:
        cores = threads
    else:
        for c in sorted({min(c, avail) for c in (8, 16, 32, 64)}):
            sweep[c] = timed(c, 2)
            log(f'cpu_baseline sweep: {c} threads -> {sweep[c]:.2f} s/step ({sample_B / sweep[c]:.2f} img/s)')
        cores = min(sweep, key=sweep.get)
    med = timed(cores, steps)
    log(f'cpu_baseline: {kind}, {cores} of {avail} cores, B={sample_B}: median {med:.2f} s/step = {sample_B / med:.2f} img/s')
    # batch sweep: a larger CPU batch could amortise the per-step overheads -- B = 4 x sample_B at the best thread
    # count, one timed step after one warm-up (~15 s; on the EPYC 9575F box it is SLOWER per image than B = sample_B: 8.5 against 14.4
    # img/s); the best img/s over the batches is the reported value, so the baseline is not under-stated
    best = dict(img_s=sample_B / med, B=sample_B, cores=cores)
    batch_sweep = {f'B={sample_B},threads={cores}': round(sample_B / med, 2)}
    big_B = 4 * sample_B
    try:
        step_big = (_reference_step_fn(cfg, big_B) if kind == 'reference' else None) or _oracle_step_fn(cfg, big_B)
        step = step_big
        for c in (cores,):
            t = timed(c, 1)
            batch_sweep[f'B={big_B},threads={c}'] = round(big_B / t, 2)
            log(f'cpu_baseline batch sweep: B={big_B}, {c} threads -> {t:.2f} s/step ({big_B / t:.2f} img/s)')
            if big_B / t > best['img_s']:
                best = dict(img_s=big_B / t, B=big_B, cores=c)
    except Exception as e:      # noqa: BLE001 -- memory / time on a small host: keep the B = sample_B figure
        batch_sweep['error'] = repr(e)[:200]
    return {'value': round(best['img_s'], 3), 'unit': 'images/s', 'cores': best['cores'], 'kind': kind, 'cpu': cpu_model_name(),
            'cores_available': avail, 'thread_sweep_img_s': {str(c): round(sample_B / t, 2) for c, t in sweep.items()},
            'batch_sweep_img_s': batch_sweep,
            **port_vs_reference(kind),
            'sample': f'BASELINE.md section 3 protocol: the same {cfg} step (fwd, 4 losses, backward, AdamW) in fp32, '
                      + ('the reference classes (/root/reference, Appendix B import)' if kind == 'reference'
                         else 'oracle/multimae_oracle.py + torch autograd (the reference checkout does not exist on this box; ~15 % slower than the reference classes on equal cores, DESIGN section 8)')
                      + f'; threads = best of the 8/16/32/64 sweep at B={sample_B} (median of {steps} timed steps after 1 warm-up), then B={big_B} at that count '
                        f'(1 timed step after 1 warm-up); value = the best img/s seen: B={best["B"]}, {best["cores"]} threads'}


def encoder_step(B: int, N: int, D: int = 768, L: int = 12, heads: int = 12, iters: int = 6) -> dict:
    """Forward + backward of the L-block encoder stack alone on (B, N, D) tokens, weight gradients on the side stream, HIP-event timed
    (tools/encoder_step.py inside the bench): 51.53 GFLOP of GEMM work per image."""
    import multimae_amd as M
    from functools import partial
    from torch import nn
    from multimae_amd.multimae_utils import Block, run_blocks
    torch.manual_seed(0)
    enc = nn.Sequential(*[Block(D, heads, mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6)) for _ in range(L)]).cuda()
    arena = M.engine.ParamArena(enc)
    dg, ws = M.engine.direct_grads(), M.engine.wgrad_stream()
    M.engine.set_direct_grads(True)
    M.engine.set_wgrad_stream(True)
    try:
        x = torch.randn(B, N, D, device='cuda', requires_grad=True)
        g = torch.randn(B, N, D, device='cuda')

        def step():
            arena.zero_grad()
            run_blocks(enc, x, root=enc).backward(g)
            M.engine.join_wgrad_streams()
        for _ in range(3):
            step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            step()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
    finally:
        M.engine.set_direct_grads(dg)
        M.engine.set_wgrad_stream(ws)
    tflop = 51.53e9 * B / 1e12
    log(f'encoder step: {ms:.2f} ms = {tflop / ms * 1e3:.0f} TF/s')
    return {'encoder_step_ms': round(ms, 3), 'encoder_step_tflops': round(tflop / ms * 1e3, 1), 'encoder_step_frac': round(tflop / ms * 1e3 / PEAK_BF16_TFLOPS, 4),
            'encoder_step_what': f'forward + backward of the {L}-block ViT-B encoder stack alone on ({B}, {N}, {D}) tokens: the step the north star quotes its >= 50 % on'}


def result_line(args, B, world, ms_per_step, img_s, final_loss, counters, roof, cpu, dp_diag, host_ms, host_wait_ms, use_graph, n_vis, doms):
    """The ONE JSON line of the measurement contract (metric / value / unit / n_gpus / steps / warmup / ms_per_step /
    higher_is_better / scaling / vs_baseline / dtype / data / config.workload + roofline + cpu_baseline [+ data_parallel])."""
    out = {
        'metric': {'cfg3': 'pre-train images/sec (whole node), ViT-B RGB+D+S 224^2 98-vis-tok',
                   'cfg2': 'pre-train images/sec (whole node), ViT-B RGB-only 224^2 98-vis-tok',
                   'cfg5': 'pre-train images/sec (whole node), ViT-L RGB+D+S 224^2 196-vis-tok'}[args.config],
        'value': round(img_s, 1), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': ({'bf16': 'bf16', 'fp32': 'f32', 'mxfp8': 'mxfp8 (e4m3 + E8M0/32) encoder forward, dX and weight-gradient products, bf16 everywhere else'}[args.precision]
                  # the storage form of the reference's fp32_output_adapters is part of the precision statement
                  + ((' (fp32_output_adapters: ' + {'h16': 'fp16 storage', 'f16': 'f32 tensors, fp16 operands', 'x3': 'f32 tensors, split-bf16 x3 operands',
                                                     'exact': 'f32'}[getattr(args, 'fp32_adapter_gemm', 'h16')] + ')')
                     if ('semseg' in doms and args.precision != 'fp32') else '')), 'data': 'synthetic',
        'config': {'workload': f'BASELINE.json configs[{ {"cfg3": 2, "cfg2": 1, "cfg5": 4}[args.config] }]: '
                               + (('ViT-L (MX-fp8 encoder products), ' if args.precision == 'mxfp8' else 'ViT-L (bf16 run of the fp8 config), ') if args.config == 'cfg5' else 'ViT-B, ')
                               + ('RGB-only' if args.config == 'cfg2' else 'RGB+depth+semseg')
                               + f', 224^2, Dirichlet alpha=1.0, {n_vis} visible tokens, {len(doms) + 1} cross-attention decoders (dim 256, depth 2), '
                               + (('semseg adapter (fp32_output_adapters) with ' + {'h16': 'fp16 STORAGE (activations / saved tensors / gradients of the adapter as IEEE half in HBM: every matmul input carries the 11-bit significand TF32 gave the reference on A100; gradients stored times a power of two fixed by the loss kernel; residual stream, LayerNorm statistics, softmax, loss f32)', 'f16': 'fp32 activations, Linear products on fp16 operands (11-bit significand = TF32, the precision the reference got on A100 under torch 1.10; one MFMA, gradient operands pre-scaled by a power of two), attention cores split-bf16', 'x3': 'fp32 activations, x3 split-bf16 operands (a_hi.b_hi + a_hi.b_lo + a_lo.b_hi)', 'exact': 'fp32 activations, f32 MFMA'}[getattr(args, 'fp32_adapter_gemm', 'h16')] + ', fp32 accumulate, ') if 'semseg' in doms else '') + 'AdamW; fwd+losses+bwd+optimizer',
                   'per_gpu_batch': B, 'global_batch': B * world, 'parallelism': f'dp{world}'},
        'final_loss': round(final_loss, 5), 'launch': ("the reference's loop body against the drop-in boundary: DistributedDataParallel(world 1, nccl, find_unused_parameters) + autocast + GradScaler + FusedAdamW, gradients through autograd"
                                                      if getattr(args, 'dropin_ddp', 0) else 'hipGraph replay of the captured step' if use_graph else 'eager, one library call per encoder stack / output adapter / loss / optimiser step'),
        'host_enqueue_ms_per_step': round(host_ms, 3), 'host_throttle_wait_ms_per_step': round(host_wait_ms, 3), 'optimizer_counters': counters,
        'roofline': roof, 'cpu_baseline': cpu,
    }
    if dp_diag is not None:
        out['data_parallel'] = dp_diag
        out['rccl_ranks_seen'] = dp_diag.get('rccl_ranks_seen')      # top level: how many ranks the collective backend really connected
    return out


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher in the environment (no WORLD_SIZE): re-execute this command as N ranks under
    torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1 at a free port).  Rank 0 of the children prints the one
    JSON line on the inherited stdout; returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC (RCCL across processes on this driver)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f'--gpus {n} without a launcher: ' + ' '.join(cmd))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=0, help='per-GPU batch (default 256; 128 for cfg5)')
    ap.add_argument('--config', default='cfg3', choices=['cfg3', 'cfg2', 'cfg5'],
                    help='cfg3 = BASELINE.json configs[2] (the metric), cfg2 = configs[1] (RGB-only), cfg5 = configs[4] geometry (ViT-L, 196 visible tokens, B = 128; MX-fp8 encoder products by default, --precision bf16 for the bf16 run of the same geometry)')
    ap.add_argument('--precision', default=None, choices=['bf16', 'fp32', 'mxfp8'],
                    help="'mxfp8': encoder forward / dX / weight-gradient products on OCP MX-fp8 operands (block-scaled MFMA), everything else as bf16.  "
                         "Default: bf16; mxfp8 for --config cfg5 (BASELINE.json configs[4] names the fp8 MFMA path)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample-batch', type=int, default=16)
    ap.add_argument('--cpu-threads', type=int, default=0, help='0: best of an 8/16/32/64 sweep')
    ap.add_argument('--cpu-steps', type=int, default=5)
    ap.add_argument('--total-steps', type=int, default=1000, help='length of the cosine lr / wd tables the loop walks (run_pretraining_multimae.py:474-480)')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--adapter-streams', type=int, default=1, help='1: output adapters on separate HIP streams')
    ap.add_argument('--wgrad-stream', type=int, default=1, help='1: weight-gradient GEMMs on a side stream')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend for --gpus > 1 ('nccl' = RCCL; 'gloo' for functional tests)")
    ap.add_argument('--force-dist', type=int, default=0, help='1: run the data-parallel path (process group, bucketed all-reduce overlapped with backward, chunked encoder backward) even at world size 1 -- the one-rank all-reduces ARE issued (RCCL copy path), so the launch-stream / event ordering of the reducer runs on a single GPU')
    ap.add_argument('--bucket-mb', type=float, default=64.0, help='gradient all-reduce bucket size (MiB of fp32 gradients)')
    ap.add_argument('--dp-exchange', default='all_reduce', choices=['all_reduce', 'rs_ag'],
                    help="N > 1: how a gradient bucket is summed -- one RCCL all_reduce (ring on xGMI) or reduce_scatter + all_gather (the direct form of SURVEY 8e); multimae_amd/dist.py")
    ap.add_argument('--bf16-buckets', type=int, default=0, help='1: gradient buckets travel as bf16 (half the xGMI bytes, bf16-rounded sum; multimae_amd/dist.py)')
    ap.add_argument('--share-device', type=int, default=0, help='1: every rank uses cuda:0 (functional test of the multi-process path on one GPU; use with --backend gloo)')
    ap.add_argument('--graph', type=int, default=-1, help='1: capture the step once as a hipGraph and replay it (multimae_amd.graph.StepGraph; 1 GPU only). Default 0: on ROCm 7.2 the replay of this ~1 100-node, 10-stream graph measured 46.2 ms/step against 44.1 ms eager')
    ap.add_argument('--no-secondary', action='store_true', help='skip the secondary line (cfg5 = BASELINE configs[4] geometry, ViT-L, MX-fp8 encoder products, B = 128, a few steps) that the default single-GPU cfg3 run appends as `secondary`')
    ap.add_argument('--gemm-cu-reserve', type=int, default=-1, help='compute units the persistent GEMM grids leave free (for RCCL\'s channel kernels while gradient buckets are in flight); -1: 16 when gradient buckets are exchanged (N > 1 or --force-dist), else 0')
    ap.add_argument('--fp32-adapter-gemm', default='h16', choices=['h16', 'f16', 'x3', 'exact'], help="the fp32 output adapter (semseg) in the bf16 speed mode: 'h16' fp16 storage of its activations / gradients (bf16-pipeline kernels, TF32's significand; default), 'f16' f32 tensors with fp16 operands (one MFMA), 'x3' split bf16 (three), 'exact' f32 MFMA (multimae_amd.engine.set_fp32_adapter_gemm)")
    ap.add_argument('--adapter-cu-share', type=int, default=0, help='experiment: compute units the output adapters\' persistent GEMM grids leave free while they run on separate streams (engine.set_adapter_cu_share)')
    ap.add_argument('--enc-bwd-side-cus', type=int, default=0, help='experiment: compute units the encoder backward leaves to its weight-gradient stream (engine.set_enc_bwd_side_cus)')
    ap.add_argument('--dropin-ddp', type=int, default=0, help="1: time the REFERENCE'S loop body against the drop-in boundary instead of the native loop (VERDICT r3 item 7): the model wrapped in torch DistributedDataParallel (world 1, nccl, find_unused_parameters=True, run_pretraining_multimae.py:380-387), forward + losses inside torch.cuda.amp.autocast(), the NativeScaler sequence (GradScaler.scale(loss).backward(), unscale_, gradient norm, GradScaler.step(FusedAdamW), update()); gradients travel through autograd / DDP's reducer (engine.set_direct_grads(False))")
    ap.add_argument('--dry-run', type=int, default=0, help='1: CPU tensors and a type-checking stub of the C ABI (tests/dryrun_harness.py): exercises the LAUNCH path of this script (self-launch, process group, reducer, the JSON line) without a GPU -- the numbers are meaningless and the line says so')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args.gpus))
    if args.precision is None:
        args.precision = 'mxfp8' if args.config == 'cfg5' else 'bf16'
    out = run_once(args)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if out is not None and world == 1 and not args.force_dist and args.config == 'cfg3' and not args.no_secondary and not args.batch:
        # the fp8 configuration of BASELINE.json (configs[4]) measured by the same command, so that the driver's run records it:
        # a short run (the primary line above is the metric; this one is reported beside it, never as `value`)
        import copy
        import gc
        a2 = copy.copy(args)
        a2.config, a2.precision, a2.steps, a2.warmup, a2.no_cpu_baseline = 'cfg5', 'mxfp8', 5, 2, True
        gc.collect()
        torch.cuda.empty_cache()
        try:
            o2 = run_once(a2)
            out['secondary'] = {k: o2[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'warmup', 'dtype', 'config', 'final_loss', 'roofline')}
        except Exception as e:      # noqa: BLE001 -- the secondary line must never cost the primary one
            out['secondary'] = {'error': repr(e)[:300]}
        # ... and two more short runs beside it: the headline step with the fp32 (semseg) adapter as f32 tensors and
        # split-bf16 ('x3', ~16-bit operand significand, 8-bit exponent) products instead of fp16 storage -- the strictest of the speed-mode
        # adapter forms, so that the driver's line shows what the default's storage format buys -- and BASELINE configs[1] (RGB only)
        for key, mods in (('secondary_x3_adapter', dict(fp32_adapter_gemm='x3')), ('secondary_cfg2', dict(config='cfg2'))):
            a3 = copy.copy(args)
            a3.steps, a3.warmup, a3.no_cpu_baseline, a3.no_kernel_timing = 10, 3, True, True
            for k, v in mods.items():
                setattr(a3, k, v)
            gc.collect()
            torch.cuda.empty_cache()
            try:
                o3 = run_once(a3)
                out[key] = {k: o3[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'warmup', 'dtype', 'config', 'final_loss')}
            except Exception as e:      # noqa: BLE001
                out[key] = {'error': repr(e)[:300]}
    if out is not None:
        print(json.dumps(out))
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


def run_once(args):
    """One measured configuration; returns the result line (rank 0) or None."""

    import torch.distributed as dist
    import multimae_amd as M
    from multimae_amd import ops
    from multimae_amd.dist import GradAllReducer, attach, broadcast_parameters
    from multimae_amd.optim import FusedAdamW

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if args.share_device:
        local = 0
    dry = bool(args.dry_run)
    if dry:
        # launch-path test on CPU (tests/test_bench_launch_cpu.py): the C ABI is the type-checking stub of tests/dryrun_harness.py
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        import dryrun_harness
        dryrun_harness.install()
        args.adapter_streams = args.wgrad_stream = 0
        args.no_kernel_timing = args.no_cpu_baseline = args.no_secondary = True
        args.backend = 'gloo'
    use_dist = world > 1 or bool(args.force_dist)
    if args.dropin_ddp:
        assert world == 1 and not args.force_dist and not dry, '--dropin-ddp times the one-GPU reference loop'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29534')
        os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    if args.force_dist and 'MASTER_ADDR' not in os.environ:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    if args.gpus > 1 or use_dist:
        if world != args.gpus and not args.force_dist:
            raise SystemExit(f'bench.py --gpus {args.gpus} inside a launcher environment with WORLD_SIZE={world}: the two must agree')
        if not dry:
            torch.cuda.set_device(local)
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(args.backend)
    device = torch.device('cpu') if dry else torch.device('cuda', local)
    if not dry:
        torch.cuda.set_device(device)

    torch.manual_seed(0 + rank)                                   # run_pretraining_multimae.py:300
    model, doms = build_model(args.config)
    model.to(device)
    arena = model.build_arena()
    reducer = None
    if use_dist:
        broadcast_parameters(arena)
        reducer = GradAllReducer.for_arena(arena, bucket_mb=args.bucket_mb, bf16_buckets=bool(args.bf16_buckets), force_collective=bool(args.force_dist), exchange=args.dp_exchange)
    M.engine.set_precision(args.precision)
    M.engine.set_fp32_adapter_gemm(args.fp32_adapter_gemm)
    M.engine.set_adapter_cu_share(args.adapter_cu_share)
    M.engine.set_enc_bwd_side_cus(args.enc_bwd_side_cus)
    M.engine.set_direct_grads(not args.dropin_ddp)
    M.engine.set_adapter_streams(bool(args.adapter_streams))
    M.engine.set_wgrad_stream(bool(args.wgrad_stream))
    B = args.batch or (2 if dry else (128 if args.config == 'cfg5' else 256))
    # compute units the persistent GEMM grids leave to RCCL's channel kernels while gradient buckets are in flight
    cu_reserve = args.gemm_cu_reserve if args.gemm_cu_reserve >= 0 else (16 if use_dist else 0)
    if reducer is not None:
        reducer.gemm_cu_reserve = cu_reserve
    elif cu_reserve > 0 and not dry:
        from multimae_amd import ops as _ops
        _ops.gemm_cu_reserve(cu_reserve)                          # single process, no gradient exchange: narrower grids for the whole run (an experiment knob)
    n_vis = 196 if args.config == 'cfg5' else 98
    lr = 1e-4 * B * world / 256                                   # blr * global_bs / 256 (:372-373)
    opt = FusedAdamW(model, lr=lr, betas=(0.9, 0.95), weight_decay=0.05)
    if reducer is not None:
        attach(model, reducer, opt)                               # buckets are summed; 1 / world is folded into the optimiser's scale
    # per-iteration cosine tables, assigned to the param group before every step as the reference loop does (:474-480)
    import math
    n_tab = max(args.total_steps, args.steps + args.warmup + 8)
    warm = max(1, n_tab // 20)
    lr_tab = [lr * (i + 1) / warm if i < warm else 1e-6 + 0.5 * (lr - 1e-6) * (1 + math.cos(math.pi * (i - warm) / max(1, n_tab - warm)))
              for i in range(n_tab)]
    wd_tab = [0.05] * n_tab
    it = [0]
    x = synthetic_batch(doms, B, device, seed=rank)
    tgt = dict(x, norm_rgb=x['rgb'])
    fns = loss_fns()
    fp32_adapters = ['semseg'] if 'semseg' in doms else []
    last = {}

    net, scaler = model, None
    if args.dropin_ddp:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local], find_unused_parameters=True)
        scaler = torch.cuda.amp.GradScaler()

    def dropin_step():
        """run_pretraining_multimae.py:474-537 with utils/native_scaler.py:20-40 inlined"""
        for g in opt.param_groups:
            g['lr'] = lr_tab[min(it[0], n_tab - 1)] * g['lr_scale']
            if g['weight_decay'] > 0:
                g['weight_decay'] = wd_tab[min(it[0], n_tab - 1)]
        it[0] += 1
        with torch.cuda.amp.autocast():
            preds, masks = net(x, num_encoded_tokens=n_vis, alphas=1.0, sample_tasks_uniformly=False, fp32_output_adapters=fp32_adapters)
            mk = dict(masks, norm_rgb=masks['rgb'])
            losses = {k: fns[k](preds[k].float(), tgt[k], mask=mk[k]) for k in preds}
            loss = sum(losses.values())
        opt.zero_grad()
        scaler.scale(loss).backward()
        scaler.unscale_(opt)
        torch.norm(torch.stack([torch.norm(p.grad.detach(), 2.0) for p in model.parameters() if p.grad is not None]), 2.0)
        scaler.step(opt)
        scaler.update()
        last['loss'] = loss

    def step():
        if args.dropin_ddp:
            return dropin_step()
        g = opt.param_groups[0]
        g['lr'], g['weight_decay'] = lr_tab[min(it[0], n_tab - 1)] * g['lr_scale'], wd_tab[min(it[0], n_tab - 1)]     # as the reference loop (:474-480)
        it[0] += 1
        opt.zero_grad()
        preds, masks = model(x, num_encoded_tokens=n_vis, alphas=1.0, sample_tasks_uniformly=False, fp32_output_adapters=fp32_adapters)
        mk = dict(masks, norm_rgb=masks['rgb'])
        losses = {k: fns[k](preds[k].float(), tgt[k], mask=mk[k]) for k in preds}
        loss = sum(losses.values())
        loss.backward()
        if reducer is not None:
            reducer.finish()
        opt.step(loss)                          # isfinite(loss) guard + grad-norm + AdamW: one call, every decision on the device
        last['loss'] = loss

    def sync():
        if use_dist:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    use_graph = False if args.graph < 0 else bool(args.graph)
    if use_graph and world > 1:
        raise SystemExit('--graph 1 with more than one rank is not supported: the gradient all-reduce is launched from the host')
    log(f'model built on {device}, B={B}, world={world}; warmup {args.warmup}; launch mode: {"hipGraph replay" if use_graph else "eager"}')
    for _ in range(args.warmup):
        step()
    run = step
    if use_graph:
        # the whole step (all streams) captured once; the capture itself and the first replay are extra untimed warm-up steps
        from multimae_amd.graph import StepGraph
        run = StepGraph(step)
        sync()
        run()
        run()
        log(f'step captured as one hipGraph ({run.n_host_inputs} host inputs refreshed per replay)')
    sync()
    log('warmup done; timing')
    opt.host_wait_s = 0.0
    if reducer is not None:
        reducer.timing = True
        reducer.exposed_wait_ms()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    host_total = time.perf_counter() - t0
    # launch-side WORK per step: loop time before the final sync minus the time the host sat in FusedAdamW's run-ahead bound
    # (it may lead the GPU by at most 2 steps, so its loop time alone would just mirror the GPU's)
    host_ms = (host_total - opt.host_wait_s) / args.steps * 1e3
    host_wait_ms = opt.host_wait_s / args.steps * 1e3
    sync()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    ms_per_step = dt / args.steps * 1e3
    img_s = B * world * args.steps / dt
    dp_diag = None
    if reducer is not None:
        # self-diagnosis of the data-parallel run: how much of the gradient exchange was NOT hidden by backward
        reducer.timing = False
        bt = reducer.bucket_timings()          # per bucket: last gradient kernel done -> all-reduce done, while backward kept running
        seen = torch.ones(1, device=device)
        dist.all_reduce(seen)
        dp_diag = {'exposed_allreduce_ms_per_step': round(reducer.exposed_wait_ms() / args.steps, 3), 'bucket_mb': args.bucket_mb,
                   'buckets': len(reducer.buckets), 'bucket_dtype': 'bf16' if args.bf16_buckets else 'f32',
                   'allreduce_mb_per_step': round(sum(e - s for s, e, _ in reducer.buckets) * (2 if args.bf16_buckets else 4) / 2 ** 20, 1),
                   'rccl_ranks_seen': int(seen.item()), 'backend': args.backend, 'exchange': args.dp_exchange, 'encoder_bwd_chunk_layers': int(getattr(model, '_bwd_chunk_layers', 0)),
                   'gemm_cu_reserved': cu_reserve, 'per_bucket': bt['buckets'],
                   'backward_ms_on_reduced_width_grids': bt['backward_ms_on_reduced_width_grids'],
                   'reading': 'a bucket whose launch_to_done_ms approaches the reduced-width stretch is what finish() waits for; exposed_allreduce_ms_per_step '
                              'is that wait; algbw_gb_s is per bucket under the load of the concurrent backward (8 GPUs, 7 xGMI links x ~153 GB/s: ring bound ~2 x 64 MiB x 7/8 / 153 GB/s = 0.77 ms per bucket unloaded)'}
    if args.dropin_ddp:
        args.no_kernel_timing = True
    final_loss = float(last['loss'].detach())
    if dry and final_loss != final_loss:
        final_loss = 0.0
    counters = opt.counters()
    log(f'timed region done: {ms_per_step:.2f} ms/step, {img_s:.1f} img/s, loss {final_loss:.4f} (host enqueue work {host_ms:.2f} ms/step + {host_wait_ms:.2f} ms waiting in the 2-step run-ahead bound)')

    # per-kernel timing pass (outside the timed region): HIP events around every MFMA GEMM launch of the PRODUCTION path -- the
    # library brackets each mmae_gemm / mmae_gemm_dw_group call on its launch stream (mmae_gemm_timing_*), also inside the
    # composite per-stack / per-adapter calls.  The pass runs with the multi-stream options OFF, so every launch is alone on the
    # GPU: a bracket is then that launch's own duration (with the adapters / weight-gradient GEMMs on other streams it would
    # also contain the time the launch shares the CUs with them).  rocprofv3 of `bench.py --adapter-streams 0 --wgrad-stream 0`
    # gives the same per-kernel averages (profiles/*_serialized*).
    roof = None
    if not args.no_kernel_timing:
        import ctypes
        from multimae_amd import _lib
        lib = _lib.load()
        M.engine.set_adapter_streams(False)
        M.engine.set_wgrad_stream(False)
        step()
        torch.cuda.synchronize()
        # park the GPU for ~80 ms first, so the whole step is already queued when it starts executing: the events then
        # bracket back-to-back kernels (with the GPU waiting for the host, a bracket would include host launch time)
        try:
            torch.cuda._sleep(int(0.08 * 2.0e9))
        except Exception:                      # noqa: BLE001 -- no spin kernel in this torch build: fill memory instead
            junk = torch.empty(1 << 30, device=device, dtype=torch.uint8)
            for _ in range(40):
                junk.zero_()
        lib.mmae_gemm_timing_enable(1)
        step()
        torch.cuda.synchronize()
        ms2, fl2, n2 = (ctypes.c_double * 3)(), (ctypes.c_double * 3)(), (ctypes.c_int64 * 3)()
        lib.mmae_gemm_timing_read(ms2, fl2, n2)
        by2 = (ctypes.c_double * 3)()
        lib.mmae_gemm_timing_read_bytes(by2)
        lib.mmae_gemm_timing_enable(0)
        M.engine.set_adapter_streams(bool(args.adapter_streams))
        M.engine.set_wgrad_stream(bool(args.wgrad_stream))
        log('kernel timing pass done')
        tot_ms = {torch.bfloat16: ms2[0], torch.float32: ms2[1]}
        tot_fl = {torch.bfloat16: fl2[0], torch.float32: fl2[1]}
        cnt = {torch.bfloat16: int(n2[0]), torch.float32: int(n2[1])}
        dom = torch.float32 if args.precision == 'fp32' else torch.bfloat16
        peak = PEAK_BF16_TFLOPS if dom == torch.bfloat16 else 157.3
        ach = tot_fl[dom] / (tot_ms[dom] * 1e-3) / 1e12 if tot_ms[dom] > 0 else 0.0
        traffic, traffic_src = pmc_traffic(dom) if args.config == 'cfg3' else (None, 'no PMC pass of this configuration is committed (the recorded one is cfg3)')
        roof = {'bound': 'mfma', 'kernel': 'gemm_bf16_pp_kernel (+ its grouped weight-gradient form gemm_bf16_pp_dwgroup_kernel) / gemm_bf16_kernel: all bf16 MFMA GEMM calls of one production step, each bracketed by HIP events on its launch stream inside the library, single-stream' if dom == torch.bfloat16 else 'gemm_f32_kernel',
                'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4), 'traffic': traffic, 'traffic_source': traffic_src,
                # what `traffic` is read against: operands once, C and every epilogue stream once, per launch, summed by the
                # library over the SAME launches `achieved` is taken over (mmae_gemm_timing_read_bytes) -- of THIS run, unlike the recorded counters
                'algorithmic_bytes_per_launch': int(by2[0 if dom == torch.bfloat16 else 1] / max(cnt[dom], 1)),
                'traffic_over_algorithmic': (round(traffic / (by2[0] / max(cnt[dom], 1)), 3) if (traffic and dom == torch.bfloat16 and by2[0] > 0) else None),
                'launches_per_step': cnt[dom], 'gemm_ms_per_step': round(tot_ms[dom], 3),
                'gemm_gflop_per_step': round(tot_fl[dom] / 1e9, 1),
                'f32_adapter_gemm_ms_per_step': round(tot_ms[torch.float32], 3) if dom == torch.bfloat16 else None,
                'f32_adapter_gemm_tflops': round(tot_fl[torch.float32] / max(tot_ms[torch.float32], 1e-9) / 1e9, 2) if dom == torch.bfloat16 else None,
                'whole_step_frac_of_peak': round(ALG_GFLOP_PER_IMG[args.config] * 1e9 * B / (ms_per_step * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)}
        if args.precision == 'mxfp8':
            # the dominant kernel of this mode is the MX-fp8 GEMM: ITS launches against the 5 PFLOP/s dense MX-fp8 peak are the
            # roofline line; the bf16 launches of the same step (weight gradients, adapters) are kept beside it
            mx_ach = fl2[2] / max(ms2[2], 1e-9) / 1e9
            roof = {'bound': 'mfma', 'kernel': 'gemm_mxfp8_kernel (v_mfma_scale_f32_32x32x64_f8f6f4, e4m3 x E8M0 / 32): the encoder forward + dX products of one production step, each launch bracketed by HIP events inside the library, single-stream',
                    'achieved': round(mx_ach, 2), 'peak': 5000.0, 'unit': 'TFLOP/s', 'frac': round(mx_ach / 5000.0, 4),
                    'traffic': None, 'traffic_source': 'PMC passes of this mode: profiles/r02_mxfp8_pmc_traffic.json (per kernel; not averaged into one figure)',
                    'launches_per_step': int(n2[2]), 'gemm_ms_per_step': round(ms2[2], 3), 'gemm_gflop_per_step': round(fl2[2] / 1e9, 1),
                    'bf16_products': roof}
    if roof is not None and args.config in ('cfg3', 'cfg2') and args.precision == 'bf16' and world == 1 and not dry:
        # the number the north star is quoted on: the ViT-B ENCODER step (12 blocks, forward + backward, B x 99 tokens) against the 2.5 PF peak
        try:
            roof.update(encoder_step(B, 98 + 1))
        except Exception as e:                   # noqa: BLE001 -- never lose the line over the beside-measurement
            roof['encoder_step_error'] = repr(e)[:200]
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config != 'cfg5':      # the CPU leg is sized for the ViT-B configs
        M.engine.set_direct_grads(False)
        cpu = cpu_baseline(args.config, args.cpu_sample_batch, args.cpu_steps, args.cpu_threads)

    M.engine.set_direct_grads(False)
    M.engine.set_precision('bf16')
    if rank == 0:
        out = result_line(args, B, world, ms_per_step, img_s, final_loss, counters, roof, cpu, dp_diag, host_ms, host_wait_ms, use_graph, n_vis, doms)
        if dry:
            out['data'] = 'DRY RUN on CPU against a stub of the C ABI (launch-path test): every number in this line is meaningless'
        return out
    return None


if __name__ == '__main__':
    main()

"""Loss curves at the HEADLINE geometry (VERDICT r4 "next round" item 2).

  (a) cfg3 (ViT-B, RGB + depth + semseg, 224^2, 98 of 588 tokens, four output adapters; B = 4), 12 AdamW steps through
      mmae_opt_step against the REFERENCE's own curves (tests/golden/curve_cfg3.json, written by tests/golden/make_golden_curve_cfg3.py
      from the reference classes with the same seeded init / inputs / per-step masks):
        fp32 parity mode  within 1e-4 per step of the reference's fp32 curve (north_star's number);
        bf16 speed mode   (every storage form of the fp32 semseg adapter: 'h16', 'f16', 'x3') within max(5e-3, 2x the reference's OWN
                          bf16-autocast deviation at that step or a neighbour) per step and 2x its mean deviation on average.
  (b) 200 steps of cfg3 at B = 32 from one initialisation, identical batch and masks, with the fp32 semseg adapter stored as
      fp16 ('h16', the default), as f32 with fp16 operands ('f16'), as f32 with split-bf16 operands ('x3') and computed in exact f32
      ('exact'): mean loss of the last 20 steps within 2e-3 of the exact-f32 adapter's, the semseg task's loss alone within 2e-3,
      no skipped / non-finite step (an fp16 overflow would surface as one: common.h f16_cvt).
  (c) the same protocol at the cfg5 geometry (ViT-L, 196 visible tokens, B = 8, 120 steps) for the MX-fp8 encoder mode against bf16:
      last-20-step mean within 1.5e-2.
The curves of (b) and (c) are also written to gpurun_out/ so that a collection visit can keep them under profiles/."""
import json
import os

import pytest
import torch

import multimae_oracle as orc
from helpers import GOLD, build_engine_model, make_inputs

pytestmark = pytest.mark.gpu
DEV = 'cuda'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
DOMS, P, S, NVIS = ['rgb', 'depth', 'semseg'], 16, 224, 98


def _loss_fns():
    import multimae_amd as M
    return {'rgb': M.MaskedMSELoss(P, 1), 'depth': M.MaskedL1Loss(P, 1), 'semseg': M.MaskedCrossEntropyLoss(P, 4),
            'norm_rgb': M.MaskedMSELoss(P, 1, norm_pix=True)}


def _golden_curve(mode, adapter_mode=None):
    """The generator's recipe (make_golden_curve_cfg3.py): seed 0 -> model, inputs; seed 2000 + step -> that step's mask draws."""
    import multimae_amd as M
    from multimae_amd.optim import FusedAdamW
    B, steps = 4, 12
    torch.manual_seed(0)
    model = build_engine_model(DOMS, P, S)
    x = make_inputs(DOMS, B, S)
    model.to(DEV)
    model.build_arena()
    xd = {k: v.to(DEV) for k, v in x.items()}
    tgt = dict(xd, norm_rgb=xd['rgb'])
    fns = _loss_fns()
    prev_adapter = M.engine.fp32_adapter_gemm()
    M.engine.set_direct_grads(True)
    if adapter_mode:
        M.engine.set_fp32_adapter_gemm(adapter_mode)
    opt = FusedAdamW(model, lr=1.5e-4, betas=(0.9, 0.95), weight_decay=0.05)
    total, per_task = [], {k: [] for k in fns}
    try:
        with M.engine.precision(mode):
            for step in range(1, steps + 1):
                torch.manual_seed(2000 + step)
                dist, tn, an = orc.draw_mask_randoms(B, [196, 196, 196], 1.0)
                spt = orc.samples_per_task_from_dirichlet(dist, NVIS)
                mask_all, ik, ir = orc.masks_from_noise(spt, tn, an, NVIS)
                tm = {d: mask_all[:, i * 196:(i + 1) * 196].contiguous().to(DEV) for i, d in enumerate(DOMS)}
                ikd, ird = ik.to(DEV), ir.to(DEV)
                model.generate_random_masks = lambda *a, **k: (tm, ikd, ird)
                opt.zero_grad()
                preds, masks = model(xd, num_encoded_tokens=NVIS, alphas=1.0, fp32_output_adapters=['semseg'] if mode != 'fp32' else [])
                mk = dict(masks, norm_rgb=masks['rgb'])
                losses = {k: fns[k](preds[k].float(), tgt[k], mask=mk[k]) for k in preds}
                loss = sum(losses.values())
                loss.backward()
                opt.step(loss)
                total.append(float(loss))
                for k, v in losses.items():
                    per_task[k].append(float(v))
    finally:
        M.engine.set_direct_grads(False)
        M.engine.set_fp32_adapter_gemm(prev_adapter)
    assert opt.counters() == dict(steps=steps, nonfinite_loss=0, skipped=0)
    return total, per_task


def test_loss_curve_fp32_mode_cfg3_headline_geometry_within_1e_4_of_the_reference():
    """north_star: "loss curve matching CPU reference to 1e-4" -- at the headline geometry, against the reference's own run."""
    gold = json.load(open(os.path.join(GOLD, 'curve_cfg3.json')))
    ce, per = _golden_curve('fp32')
    d = [abs(a - b) for a, b in zip(ce, gold['reference_fp32'])]
    for k, ref in gold['reference_fp32_per_task'].items():
        dk = max(abs(a - b) for a, b in zip(per[k], ref))
        assert dk < 1e-4, (k, dk)
    assert max(d) < 1e-4, (max(d), ce, gold['reference_fp32'])
    assert ce[-1] < ce[0]


@pytest.mark.parametrize('adapter_mode', ['h16', 'f16', 'x3'])
def test_loss_curve_bf16_mode_cfg3_headline_geometry(adapter_mode):
    """bf16 speed mode with each storage form of the fp32 (semseg) adapter.  Calibration: the reference under bf16 autocast with
    fp32_output_adapters=['semseg'] deviates from its own fp32 run by up to 7.7e-3 (mean 2.0e-3) on this recipe (curve_cfg3.json)."""
    gold = json.load(open(os.path.join(GOLD, 'curve_cfg3.json')))
    ref32, ref16 = gold['reference_fp32'], gold['reference_bf16_autocast']
    ref_dev = [abs(a - b) for a, b in zip(ref16, ref32)]
    ce, per = _golden_curve('bf16', adapter_mode)
    dev = [abs(a - b) for a, b in zip(ce, ref32)]
    try:
        os.makedirs(OUT, exist_ok=True)
        with open(os.path.join(OUT, f'curve_cfg3_bf16_{adapter_mode}.json'), 'w') as f:
            json.dump({'engine': ce, 'engine_per_task': per, 'dev_vs_reference_fp32': dev, 'reference_own_bf16_dev': ref_dev}, f, indent=1)
    except OSError:
        pass
    for i, d in enumerate(dev):
        r = max(ref_dev[max(i - 1, 0):i + 2])
        assert d < max(5e-3, 2.0 * r), (adapter_mode, i, d, r, ce, ref32)
    assert sum(dev) / len(dev) < 2.0 * sum(ref_dev) / len(ref_dev), (sum(dev) / len(dev), sum(ref_dev) / len(ref_dev))
    # the semseg task alone (the adapter whose storage changes): within 2x the reference's own deviation on that task, floor 3e-3
    rs32, rs16 = gold['reference_fp32_per_task']['semseg'], gold['reference_bf16_autocast_per_task']['semseg']
    for i, (a, b) in enumerate(zip(per['semseg'], rs32)):
        r = max(abs(p - q) for p, q in zip(rs16[max(i - 1, 0):i + 2], rs32[max(i - 1, 0):i + 2]))
        assert abs(a - b) < max(3e-3, 2.0 * r), (adapter_mode, 'semseg', i, a, b, r)
    assert ce[-1] < ce[0]


def _train(model_key, mode, adapter_mode, steps, batch, nvis, lr=1e-4, warm=20):
    """tools/mx_trains.py's protocol: seeded init, one fixed synthetic batch, the sampler's draws replayed from one seed per step."""
    import multimae_amd as M
    from multimae_amd.optim import FusedAdamW
    import bench
    torch.manual_seed(0)
    model, doms = bench.build_model(model_key)
    model.to(DEV)
    model.build_arena()
    prev_adapter = M.engine.fp32_adapter_gemm()
    M.engine.set_direct_grads(True)
    M.engine.set_fp32_adapter_gemm(adapter_mode)
    opt = FusedAdamW(model, lr=lr, betas=(0.9, 0.95), weight_decay=0.05)
    x = bench.synthetic_batch(doms, batch, torch.device(DEV), seed=0)
    tgt = dict(x, norm_rgb=x['rgb'])
    fns = bench.loss_fns()
    tot, sem = [], []
    try:
        with M.engine.precision(mode):
            for it in range(steps):
                opt.param_groups[0]['lr'] = lr * min(1.0, (it + 1) / warm)
                torch.manual_seed(1000 + it)
                torch.cuda.manual_seed(1000 + it)
                opt.zero_grad()
                preds, masks = model(x, num_encoded_tokens=nvis, alphas=1.0, sample_tasks_uniformly=False, fp32_output_adapters=['semseg'])
                mk = dict(masks, norm_rgb=masks['rgb'])
                losses = {k: fns[k](preds[k].float(), tgt[k], mask=mk[k]) for k in preds}
                loss = sum(losses.values())
                loss.backward()
                opt.step(loss)
                tot.append(loss.detach())
                sem.append(losses['semseg'].detach())
        torch.cuda.synchronize()
    finally:
        M.engine.set_direct_grads(False)
        M.engine.set_fp32_adapter_gemm(prev_adapter)
    c = opt.counters(detail=True)
    del opt, model
    torch.cuda.empty_cache()
    return [float(v) for v in tot], [float(v) for v in sem], c


def test_fp32_adapter_storage_forms_train_alike_at_the_headline_geometry():
    """(b): 'h16' (default) / 'f16' / 'x3' against the exact-f32 adapter over 200 steps of cfg3 at B = 32."""
    steps, tail = 200, 20
    res = {m: _train('cfg3', 'bf16', m, steps, 32, NVIS) for m in ('exact', 'x3', 'f16', 'h16')}
    mean = lambda v: sum(v[-tail:]) / tail
    rows = {m: dict(total=mean(t), semseg=mean(s), counters=c, first=t[0], last=t[-1]) for m, (t, s, c) in res.items()}
    gaps = {m: dict(total=abs(rows[m]['total'] - rows['exact']['total']), semseg=abs(rows[m]['semseg'] - rows['exact']['semseg']),
                    max_step=max(abs(a - b) for a, b in zip(res[m][0], res['exact'][0]))) for m in ('x3', 'f16', 'h16')}
    try:
        os.makedirs(OUT, exist_ok=True)
        with open(os.path.join(OUT, 'adapter_modes_train_cfg3.json'), 'w') as f:
            json.dump({'protocol': f'cfg3, B = 32, {steps} AdamW steps (lr 1e-4 after 20 warm-up steps), identical init / batch / masks; means over the last {tail} steps',
                       'rows': rows, 'gap_vs_exact_f32_adapter': gaps,
                       'curves_every_10': {m: res[m][0][::10] for m in res}, 'semseg_every_10': {m: res[m][1][::10] for m in res}}, f, indent=1)
    except OSError:
        pass
    for m, (t, s, c) in res.items():
        assert c['skipped'] == 0 and c['nonfinite_loss'] == 0 and c['nonfinite_grad'] == 0, (m, c)
        assert t[-1] < 0.75 * t[0], (m, t[0], t[-1])                 # it trains
    for m, g in gaps.items():
        assert g['total'] < 2e-3, (m, g, rows)
        assert g['semseg'] < 2e-3, (m, g, rows)


def test_mxfp8_encoder_trains_like_bf16_at_the_cfg5_geometry():
    """(c): BASELINE configs[4]'s geometry (ViT-L, 196 visible tokens), B = 8, 120 steps: MX-fp8 encoder products against bf16."""
    steps, tail = 120, 20
    a, sa, ca = _train('cfg5', 'bf16', 'h16', steps, 8, 196)
    b, sb, cb = _train('cfg5', 'mxfp8', 'h16', steps, 8, 196)
    mean = lambda v: sum(v[-tail:]) / tail
    gap = abs(mean(a) - mean(b))
    run = lambda v, i, w=20: sum(v[max(0, i - w + 1):i + 1]) / (i - max(0, i - w + 1) + 1)
    run_gap = max(abs(run(a, i) - run(b, i)) for i in range(steps))
    try:
        os.makedirs(OUT, exist_ok=True)
        with open(os.path.join(OUT, 'mxfp8_trains_cfg5.json'), 'w') as f:
            json.dump({'protocol': f'cfg5 geometry (ViT-L, 196 visible tokens), B = 8, {steps} steps, identical init / batch / masks',
                       'bf16_last20': mean(a), 'mxfp8_last20': mean(b), 'gap_last20': gap, 'max_gap_of_20_step_running_means': run_gap,
                       'max_single_step_gap': max(abs(p - q) for p, q in zip(a, b)), 'counters': {'bf16': ca, 'mxfp8': cb},
                       'bf16_every_10': a[::10], 'mxfp8_every_10': b[::10]}, f, indent=1)
    except OSError:
        pass
    assert ca['skipped'] == 0 and cb['skipped'] == 0, (ca, cb)
    assert a[-1] < 0.8 * a[0] and b[-1] < 0.8 * b[0]
    # round 6: every reduction of a step has a fixed order (the class-embedding gradient was the last float-atomic one), so the curves -- and these
    # gaps -- are the same from run to run (measured 1.6e-3 / 1.6e-2); the bounds are the round-5 verdict's
    assert gap < 8e-3, (gap, mean(a), mean(b))
    assert run_gap < 3e-2, run_gap


@pytest.mark.parametrize('cfg, precision, B, nvis', [('cfg3', 'bf16', 8, NVIS), ('cfg5', 'mxfp8', 4, 196)])
def test_a_training_run_is_bit_reproducible(cfg, precision, B, nvis):
    """Two runs of the same 8 steps (identical init / batch / mask seeds) give bit-identical loss curves: no float atomics are left in the step
    (mmae_semseg_emb_bwd_det, round 6; split-K slabs, column sums and the gradient-norm reduction always had fixed orders)."""
    r1 = _train(cfg, precision, 'h16', 8, B, nvis)
    r2 = _train(cfg, precision, 'h16', 8, B, nvis)
    assert r1[0] == r2[0], [abs(p - q) for p, q in zip(r1[0], r2[0])]
    assert r1[1] == r2[1]

"""CPU: full host-side control flow of the engine (all Functions, both precisions, both gradient
routing modes, optimiser, reducer hooks) against a type-checking stub of the C ABI."""
import pytest
import torch

from helpers import MINI, build_mini_engine, load_mini


@pytest.fixture()
def stubbed(monkeypatch):
    from multimae_amd import _lib, ops
    import dryrun_harness
    old = (_lib._lib, ops._require_gpu, ops._stream, ops._device_ok, ops._WS_ELEMS[0])
    dryrun_harness.install()
    yield
    _lib._lib, ops._require_gpu, ops._stream, ops._device_ok, ops._WS_ELEMS[0] = old
    ops._WS.clear()
    ops.set_composite_blocks(True)
    ops.set_stack_composites(True)


def _step(model, mode, direct, x, fp32_adapters=()):
    import multimae_amd as M
    P = MINI['P']
    fns = {'rgb': M.MaskedMSELoss(P, 1), 'depth': M.MaskedL1Loss(P, 1), 'semseg': M.MaskedCrossEntropyLoss(P, 4),
           'norm_rgb': M.MaskedMSELoss(P, 1, norm_pix=True)}
    M.engine.set_direct_grads(direct)
    try:
        with M.engine.precision(mode):
            preds, masks = model(x, num_encoded_tokens=MINI['nvis'], alphas=1.0, fp32_output_adapters=list(fp32_adapters))
            tgt = dict(x, norm_rgb=x['rgb'])
            mk = dict(masks, norm_rgb=masks['rgb'])
            losses = {k: fns[k](preds[k].float(), tgt[k], mask=mk[k]) for k in preds}
            sum(losses.values()).backward()
    finally:
        M.engine.set_direct_grads(False)
    return preds, masks


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
@pytest.mark.parametrize('direct', [False, True])
@pytest.mark.parametrize('composite', ['stack', 'block', 'none'])
def test_full_step_control_flow(stubbed, mode, direct, composite):
    """composite: one library call per stack / adapter, one per transformer block, or one per kernel."""
    import multimae_amd as M
    from multimae_amd import ops
    ops.set_composite_blocks(composite != 'none')
    ops.set_stack_composites(composite == 'stack')
    g = load_mini()
    model = build_mini_engine()
    model.load_state_dict(g['sd'])
    if direct:
        arena = model.build_arena()
        assert arena.intact()
    ready = []
    model._grad_ready_cb = ready.append
    preds, masks = _step(model, mode, direct, g['x'], fp32_adapters=('semseg',))
    assert preds['rgb'].shape == (3, 3, 32, 32) and preds['semseg'].shape == (3, 133, 8, 8)
    assert masks['depth'].shape == (3, 16) and masks['depth'].dtype == torch.int64
    for n, p in model.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and p.grad.shape == p.shape, n
        else:
            assert p.grad is None, n
    if not direct:
        # gradients handed to autograd are written by AccumulateGrad AFTER the node returns: no unit may be reported ready
        # from inside backward; the reducer's finish() takes everything
        assert ready == []
        return
    assert ready[:2] == ['output_adapters.norm_rgb', 'output_adapters.semseg'] or set(ready) >= {'encoder.0', 'encoder.1'}
    assert ready[-3:] == ['encoder.0', 'global_tokens', 'input_adapters']           # readiness order = arena order


def test_optimizer_and_arena_flow(stubbed):
    import multimae_amd as M
    from multimae_amd.optim import FusedAdamW
    g = load_mini()
    model = build_mini_engine()
    model.load_state_dict(g['sd'])
    arena = model.build_arena()
    sd_after = model.state_dict()
    for k, v in g['sd'].items():
        assert torch.equal(sd_after[k], v), k           # the arena move preserves every value and key
    assert arena.n_trainable <= arena.numel and arena.grad.numel() == arena.n_trainable
    opt = FusedAdamW(model, lr=1e-3, clip_grad=1.0)
    opt.zero_grad()
    _step(model, 'bf16', True, g['x'])
    gn = opt.step()
    assert gn.shape == (1,) and opt.step_count == 0      # Adam's step counter lives on the device (the stub advances nothing)
    assert set(opt.counters()) == {'steps', 'nonfinite_loss', 'skipped'}
    assert model.encoder[0].attn.qkv.weight.grad.data_ptr() == arena.grad.data_ptr() + 4 * arena.offsets['encoder.0.attn.qkv.weight']


def test_standalone_modules_flow(stubbed):
    import multimae_amd as M
    from multimae_amd.multimae_utils import Attention, Block, CrossAttention, LayerNorm, Linear, Mlp
    x = torch.randn(2, 5, 64, requires_grad=True)
    for mod in (Linear(64, 32), LayerNorm(64), Mlp(64, 128), Attention(64, 2, qkv_bias=True), Block(64, 2, qkv_bias=True)):
        y = mod(x)
        y.sum().backward()
        assert all(p.grad is not None for p in mod.parameters())
    ca = CrossAttention(64, 2, qkv_bias=True)
    ca(x, torch.randn(2, 7, 64)).sum().backward()
    vit = M.multivit_base({'rgb': M.PatchedInputAdapter(3, 1, 16)}, None)
    with torch.no_grad():
        outs = vit(torch.randn(1, 3, 32, 32), return_all_layers=True)
    assert len(outs) == 12 and outs[0].shape == (1, 5, 768)


def test_mxfp8_mode_and_option_plumbing(stubbed):
    """Host logic of the 'mxfp8' precision mode (weight table + refresh, the stack descriptor's mx_w, slab sizing calls) and of
    the qkv_bias=False / learnable_pos_emb constructor options, against the type-checking stub of the C ABI."""
    import multimae_amd as M
    from multimae_amd import ops
    from helpers import build_engine_model, make_inputs
    from functools import partial
    from torch import nn
    doms, P, S, B, nvis = ['rgb', 'semseg'], 8, 32, 2, 9
    model = build_engine_model(doms, P, S, enc=(256, 2, 4), posemb_size=32)
    x = make_inputs(doms, B, S)
    calls = []
    lib = M._lib.load()
    orig = lib.__getattr__

    def spy(name):
        fn = orig(name)

        def wrapped(*a):
            calls.append(name)
            return fn(*a)
        return wrapped
    type(lib).__getattr__ = lambda self, name: spy(name)
    try:
        M.engine.set_direct_grads(True)
        model.build_arena()
        old = ops._X3_PRESPLIT
        ops._X3_PRESPLIT = True
        with M.engine.precision('mxfp8'):
            assert M.engine.mx_encoder() and M.engine.precision_mode() == 'mxfp8' and M.engine.act_dtype() == torch.bfloat16
            preds, masks = model(x, num_encoded_tokens=nvis, alphas=1.0, fp32_output_adapters=['semseg'])
            sum(p.float().sum() for p in preds.values()).backward()
        assert M.engine.precision_mode() == 'bf16'
    finally:
        ops._X3_PRESPLIT = old
        M.engine.set_direct_grads(False)
        type(lib).__getattr__ = orig
    assert calls.count('mmae_mx_prepare_weights') == 1 and calls.count('mmae_stack_fwd') == 1 and calls.count('mmae_stack_bwd') == 1, calls
    # constructor options that used to raise
    from multimae_amd.multimae_utils import Block, run_blocks
    blk = Block(64, 2, qkv_bias=False, norm_layer=partial(nn.LayerNorm, eps=1e-6))
    assert 'attn.qkv.bias' not in blk.state_dict() and len(blk.state_dict()) == 11
    run_blocks([blk], torch.randn(2, 5, 64, requires_grad=True)).sum().backward()
    ad = M.PatchedInputAdapter(num_channels=3, stride_level=1, patch_size_full=8, dim_tokens=64, learnable_pos_emb=True, image_size=32)
    ad(torch.randn(2, 3, 32, 32)).sum().backward()
    assert ad.pos_emb.grad is not None and ad.pos_emb.grad.shape == ad.pos_emb.shape


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
def test_decoder_drop_path_and_fp16_storage_control_flow(stubbed, mode):
    """Host logic of two round-4 adapter options against the type-checking stub: stochastic depth in decoder_transformer (the adapters
    leave the one-call composite, every block draws two per-sample scales while training, none in eval mode) and the 'h16' / 'f16'
    modes of an fp32 output adapter (the mini model's 532-column patch rows are outside the fp16-storage grid: the call must fall back)."""
    import multimae_amd as M
    from functools import partial
    from torch import nn
    from multimae_amd import multimae_utils as mu
    doms, P, S = MINI['doms'], MINI['P'], MINI['S']
    torch.manual_seed(0)
    ins = {d: (M.SemSegInputAdapter(num_classes=133, dim_class_emb=16, interpolate_class_emb=False, stride_level=4, patch_size_full=P, image_size=S)
               if d == 'semseg' else M.PatchedInputAdapter(num_channels=3 if d == 'rgb' else 1, stride_level=1, patch_size_full=P, image_size=S))
           for d in doms}
    outs = {}
    for key, task in [(d, d) for d in doms] + [('norm_rgb', 'rgb')]:
        ch = {'rgb': 3, 'depth': 1, 'semseg': 133}[task]
        outs[key] = M.SpatialOutputAdapter(num_channels=ch, stride_level=4 if task == 'semseg' else 1, patch_size_full=P, dim_tokens=64, depth=3,
                                           num_heads=2, use_task_queries=True, task=task, context_tasks=list(doms), use_xattn=True, image_size=S,
                                           drop_path_rate=0.3)
    model = M.MultiMAE(ins, outs, num_global_tokens=1, dim_tokens=128, depth=2, num_heads=2, mlp_ratio=4, qkv_bias=True,
                       norm_layer=partial(nn.LayerNorm, eps=1e-6)).train()
    draws = []
    real = mu._drop_path_rand
    mu._drop_path_rand = lambda shape, device: (draws.append(shape), real(shape, device))[1]
    try:
        g = load_mini()
        for gemm in ('h16', 'f16'):
            old = M.engine.fp32_adapter_gemm()
            M.engine.set_fp32_adapter_gemm(gemm)
            try:
                draws.clear()
                _step(model, mode, False, g['x'], fp32_adapters=('semseg',))
            finally:
                M.engine.set_fp32_adapter_gemm(old)
            assert len(draws) == 4 * 2 * 2, draws            # four adapters x the two blocks with a rate > 0 x two branches
            for n, p in model.named_parameters():
                assert (p.grad is not None) == p.requires_grad, n
            model.zero_grad()
        model.eval()
        draws.clear()
        with torch.no_grad(), M.engine.precision(mode):
            model(g['x'], num_encoded_tokens=MINI['nvis'], alphas=1.0)
        assert draws == []
    finally:
        mu._drop_path_rand = real


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
def test_dropout_control_flow(stubbed, mode):
    """MultiMAE(drop_rate, attn_drop_rate > 0) with adapters built the same way, training mode, on the ABI stub: every nn.Dropout site of the
    reference requests exactly one keep mask, in the reference's order (per block: attention probabilities (B, H, N, N), proj output, fc2
    output -- multimae_utils.py:154, 177, 181; per adapter the cross attention's two first, output_adapters.py:118-119), every trainable
    parameter receives a gradient, and eval mode requests none."""
    import multimae_amd as M
    from functools import partial
    from torch import nn
    from multimae_amd import ops
    doms, P, S = MINI['doms'], MINI['P'], MINI['S']
    torch.manual_seed(0)
    ins = {}
    for d in doms:
        if d == 'semseg':
            ins[d] = M.SemSegInputAdapter(num_classes=133, dim_class_emb=16, interpolate_class_emb=False, stride_level=4, patch_size_full=P,
                                          image_size=S)
        else:
            ins[d] = M.PatchedInputAdapter(num_channels=3 if d == 'rgb' else 1, stride_level=1, patch_size_full=P, image_size=S)
    outs = {}
    for key, task in [(d, d) for d in doms] + [('norm_rgb', 'rgb')]:
        ch = {'rgb': 3, 'depth': 1, 'semseg': 133}[task]
        outs[key] = M.SpatialOutputAdapter(num_channels=ch, stride_level=4 if task == 'semseg' else 1, patch_size_full=P, dim_tokens=64, depth=2,
                                           num_heads=2, use_task_queries=True, task=task, context_tasks=list(doms), use_xattn=(key != 'depth'),
                                           image_size=S, drop_rate=0.1, attn_drop_rate=0.2)
    model = M.MultiMAE(ins, outs, num_global_tokens=1, dim_tokens=128, depth=2, num_heads=2, mlp_ratio=4, qkv_bias=True, drop_rate=0.1,
                       attn_drop_rate=0.2, norm_layer=partial(nn.LayerNorm, eps=1e-6)).train()
    asked = []
    real = ops._dropout_keep
    ops._dropout_keep = lambda shape, p, device: (asked.append((p, len(shape))), real(shape, p, device))[1]
    try:
        g = load_mini()
        _step(model, mode, False, g['x'], fp32_adapters=('semseg',))
        block = [(0.2, 4), (0.1, 2), (0.1, 2)]
        want = block * 2
        for key in outs:
            want += ([(0.2, 4), (0.1, 2)] if key != 'depth' else []) + block * 2
        assert asked == want, asked
        for n, p in model.named_parameters():
            # (the depth adapter was built without cross attention: its unused xattn / MLP parameters do not exist)
            assert (p.grad is not None) == p.requires_grad, n
        model.eval()
        asked.clear()
        with torch.no_grad(), M.engine.precision(mode):
            model(g['x'], num_encoded_tokens=MINI['nvis'], alphas=1.0)
        assert asked == []
    finally:
        ops._dropout_keep = real


def test_fp16_weight_copies_are_cached_by_storage_not_by_object():
    """ops.h16_weights: the fp16 copies of an fp32 adapter's weights are allocated once per set of weights -- the callers pass fresh detach()
    views every forward, so the cache must key on where the weights live (a key on id() allocated a new copy every step)."""
    from multimae_amd import ops
    ws = [torch.nn.Parameter(torch.randn(8, 8)) for _ in range(3)]
    a = ops.h16_weights([w.detach() for w in ws])
    b = ops.h16_weights([w.detach() for w in ws])
    assert a is b and len(a.copies) == 3 and a.copies[0].dtype == torch.float16
    other = [torch.nn.Parameter(torch.randn(8, 8)) for _ in range(3)]
    assert ops.h16_weights([w.detach() for w in other]) is not a
    assert ops.h16_weights([w.detach() for w in ws]) is a

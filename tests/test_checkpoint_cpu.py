"""CPU: checkpoint compatibility -- key maps against outputs of the reference's own converter functions
(tests/golden/ckpt_maps.npz, generator committed), optimiser-state interchange with torch.optim.AdamW, save / auto-resume."""
import os
import types

import numpy as np
import torch

from multimae_amd import checkpoint as ck

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ckpt_maps.npz')


def _group(z, pre):
    return {k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)}


def test_key_maps_match_reference_converters():
    z = np.load(GOLD)
    sd = _group(z, 'in/')
    for fn, pre in ((ck.multimae_to_vit, 'vit/'), (ck.multimae_to_vitmultimae, 'vitmm/')):
        got, exp = fn({k: v.clone() for k, v in sd.items()}), _group(z, pre)
        assert set(got) == set(exp)
        for k in exp:
            assert got[k].shape == exp[k].shape and torch.equal(got[k], exp[k]), (pre, k)
    got, exp = ck.vit_to_multimae({k: v.clone() for k, v in _group(z, 'vit_in/').items()}), _group(z, 'back/')
    assert set(got) == set(exp)
    for k in exp:
        assert torch.equal(got[k].contiguous(), exp[k]), k
    # round trip: MultiMAE -> ViT -> MultiMAE restores the encoder / patch-embed / pos-emb tensors
    rt = ck.vit_to_multimae(ck.multimae_to_vit({k: v.clone() for k, v in sd.items()}))
    for k in ('input_adapters.rgb.pos_emb', 'input_adapters.rgb.proj.weight', 'encoder.0.attn.qkv.weight', 'global_tokens'):
        assert torch.equal(rt[k].contiguous(), sd[k]), k


def test_pos_emb_resize_matches_reference():
    z = np.load(GOLD)
    sd = _group(z, 'in/')
    D = sd['input_adapters.rgb.pos_emb'].shape[1]
    model = types.SimpleNamespace(input_adapters=types.SimpleNamespace(rgb=types.SimpleNamespace(pos_emb=torch.zeros(1, D, 5, 4))))
    ck.interpolate_pos_embed_multimae(model, sd)
    exp = torch.from_numpy(z['resized/input_adapters.rgb.pos_emb'])
    assert sd['input_adapters.rgb.pos_emb'].shape == (1, D, 5, 4)
    assert torch.allclose(sd['input_adapters.rgb.pos_emb'], exp, rtol=0, atol=1e-6)


def _same_moments(a, b):
    """equal on every parameter's slice (the 64-element alignment padding between tensors carries no state)"""
    return all(torch.equal(a.m[o:o + n], b.m[o:o + n]) and torch.equal(a.v[o:o + n], b.v[o:o + n]) for _, o, n, _ in ck._trainable(a))


def _fake_fused(model, step=3):
    """FusedAdamW-shaped object on CPU tensors (the real one needs the GPU only for its step kernel)."""
    from multimae_amd import engine
    arena = engine.ParamArena(model)
    opt = types.SimpleNamespace(arena=arena, m=torch.randn(arena.n_trainable), v=torch.rand(arena.n_trainable), step_count=step,
                                param_groups=[dict(lr=1e-3, weight_decay=0.05, lr_scale=1.0, betas=(0.9, 0.95), eps=1e-8)])
    return opt


def test_optimizer_state_interchanges_with_torch_adamw(tmp_path):
    """The exported state loads into a real torch.optim.AdamW over the same parameters (the reference's optimiser,
    utils/optim_factory.py:166) and comes back bit-identical; a checkpoint written here resumes with model + moments + epoch."""
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.LayerNorm(5), torch.nn.Linear(5, 3))
    model[1].bias.requires_grad_(False)                 # a frozen tensor: not part of the optimiser (pos-emb case)
    opt = _fake_fused(model)
    sd = ck.optimizer_state_to_torch(opt)
    ref = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
    ref.load_state_dict(sd)                              # torch validates group sizes / shapes here
    for i, p in enumerate(p for p in model.parameters() if p.requires_grad):
        assert torch.equal(ref.state[p]['exp_avg'], sd['state'][i]['exp_avg']) and ref.state[p]['exp_avg'].shape == p.shape
    back = _fake_fused(model, step=0)
    back.m.zero_(); back.v.zero_()
    ck.optimizer_state_from_torch(back, ref.state_dict())
    assert back.step_count == 3 and _same_moments(back, opt)
    # files: save twice, auto-resume picks the latest, state comes back
    d = str(tmp_path)
    ck.save_checkpoint(d, 4, model, opt)
    with torch.no_grad():
        model[0].weight.add_(1.0)
    p9 = ck.save_checkpoint(d, 9, model, opt, args={'epochs': 10})
    assert ck.latest_checkpoint(d) == p9 and sorted(os.listdir(d)) == ['checkpoint-4.pth', 'checkpoint-9.pth']
    saved = torch.load(p9, weights_only=False)
    assert set(saved) == {'model', 'optimizer', 'epoch', 'scaler', 'args'} and saved['epoch'] == 9
    w9 = model[0].weight.detach().clone()
    with torch.no_grad():
        model[0].weight.zero_()
    fresh = _fake_fused(model, step=0)
    fresh.m.zero_()
    assert ck.load_checkpoint(p9, model, fresh) == 10
    assert torch.equal(model[0].weight, w9) and _same_moments(fresh, opt) and fresh.step_count == 3
    assert ck.latest_checkpoint(str(tmp_path / 'nothing_here')) is None


def test_optimizer_state_order_is_registration_order_with_readiness_arena():
    """build_arena() lays the arena out in backward-readiness order (output adapters, encoder.L-1 .. 0, then
    the embeddings); a checkpoint must still number the optimiser state the way torch.optim.AdamW over
    ``[p for n, p in model.named_parameters() if p.requires_grad]`` does (utils/optim_factory.py:138-149) -- same-shaped tensors
    (encoder.0 <-> encoder.1) would otherwise swap moments silently."""
    from multimae_amd import engine

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.global_tokens = torch.nn.Parameter(torch.zeros(1, 1, 4))
            self.encoder = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Linear(4, 4), torch.nn.Linear(4, 4))
            self.output_adapters = torch.nn.ModuleDict({'rgb': torch.nn.Linear(4, 2), 'depth': torch.nn.Linear(4, 2)})

    torch.manual_seed(0)
    model = Toy()
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    groups = [f'output_adapters.{d}.' for d in reversed(list(model.output_adapters))] + [f'encoder.{l}.' for l in reversed(range(3))]
    arena = engine.ParamArena(model, groups=groups)
    assert arena.names != names and arena.param_order == names          # the layout really is permuted; the registration order is kept
    # a real AdamW over model.parameters() takes one step; its state must map onto the arena by NAME
    ref = torch.optim.AdamW([p for n, p in model.named_parameters() if p.requires_grad], lr=1e-2, betas=(0.9, 0.95), weight_decay=0.05)
    for p in model.parameters():
        p.grad = torch.randn_like(p)
    ref.step()
    opt = types.SimpleNamespace(arena=arena, m=torch.zeros(arena.n_trainable), v=torch.zeros(arena.n_trainable), step_count=0,
                                param_groups=[dict(lr=1e-2, weight_decay=0.05, lr_scale=1.0, betas=(0.9, 0.95), eps=1e-8)])
    ck.optimizer_state_from_torch(opt, ref.state_dict())
    byname = dict(model.named_parameters())
    for n in names:
        o, s = arena.offsets[n], arena.sizes[n]
        assert torch.equal(opt.m[o:o + s].view(byname[n].shape), ref.state[byname[n]]['exp_avg']), n
        assert torch.equal(opt.v[o:o + s].view(byname[n].shape), ref.state[byname[n]]['exp_avg_sq']), n
    # ... and back: the exported dict loads into a fresh AdamW with every tensor's moments in its own slot
    out = ck.optimizer_state_to_torch(opt)
    ref2 = torch.optim.AdamW([p for n, p in model.named_parameters() if p.requires_grad], lr=1e-2, betas=(0.9, 0.95), weight_decay=0.05)
    ref2.load_state_dict(out)
    for n in names:
        assert torch.equal(ref2.state[byname[n]]['exp_avg'], ref.state[byname[n]]['exp_avg']), n


def test_create_model_builder_arguments_and_checkpoint_path(tmp_path):
    """registry.create_model with the reference builder's signature (utils/model_builder.py:29-76): timm-era arguments accepted,
    `drop_connect_rate` mapped to `drop_path_rate`, and `checkpoint_path` loaded after construction (reference-format file or a bare
    state_dict) -- VERDICT r3 missing #5."""
    import warnings
    import multimae_amd as M
    from helpers import MINI

    def adapters():
        ins = {'rgb': M.PatchedInputAdapter(num_channels=3, stride_level=1, patch_size_full=MINI['P'], image_size=MINI['S'])}
        outs = {'rgb': M.SpatialOutputAdapter(num_channels=3, stride_level=1, patch_size_full=MINI['P'], dim_tokens=32, depth=1, num_heads=2,
                                              use_task_queries=True, task='rgb', context_tasks=['rgb'], image_size=MINI['S'])}
        return ins, outs
    torch.manual_seed(3)
    ins, outs = adapters()
    src = M.create_model('pretrain_multimae_base', input_adapters=ins, output_adapters=outs, num_global_tokens=1, drop_connect_rate=0.1,
                         bn_tf=False, bn_eps=None, scriptable=None, exportable=None, no_jit=None)
    assert abs(src.encoder[-1].drop_path.drop_prob - 0.1) < 1e-6        # drop_connect_rate -> drop_path_rate
    f1, f2 = str(tmp_path / 'ref_format.pth'), str(tmp_path / 'bare.pth')
    torch.save({'model': src.state_dict(), 'epoch': 7}, f1)
    torch.save(src.state_dict(), f2)
    for f in (f1, f2):
        torch.manual_seed(99)                                            # a different initialisation ...
        ins, outs = adapters()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            m = M.create_model('pretrain_multimae_base', pretrained=True, checkpoint_path=f, input_adapters=ins, output_adapters=outs,
                               num_global_tokens=1, drop_path_rate=0.0)
        assert any('pretrained' in str(x.message) for x in w)
        for (k, a), (_, b) in zip(m.state_dict().items(), src.state_dict().items()):
            assert torch.equal(a, b), k                                  # ... replaced by the checkpoint's weights

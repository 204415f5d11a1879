"""CPU: host-side logic of the boundary -- state_dict contract, seeded-init parity with the
reference, registry, C-ABI export check, no-fallback behaviour."""
import ctypes
import os

import pytest
import torch

from helpers import MINI, build_engine_model, build_mini_engine, load_mini, load_scalars


def test_cabi_exports_every_declared_symbol():
    from multimae_amd import _lib
    names = _lib.declared_symbols()
    assert len(names) >= 30
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f'libmmae_hip.so does not export {n}'
    assert _lib.load().mmae_abi_version() == 7        # load() also checks every ctypes struct mirror against mmae_struct_size()


def test_state_dict_contract_and_seeded_init_base():
    """Appendix A: 351 keys; same RNG call order as the reference => identical initial weights."""
    gold = load_scalars()['base_rgb_depth_semseg']
    torch.manual_seed(0)
    m = build_engine_model(['rgb', 'depth', 'semseg'], 16, 224)
    sd = m.state_dict()
    assert len(sd) == gold['num_keys'] == 351
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == gold['num_trainable']
    assert abs(float(sum(v.double().sum() for v in sd.values())) - gold['state_dict_sum']) < 1e-6
    assert sd['encoder.0.attn.qkv.weight'].shape == (2304, 768)
    assert sd['input_adapters.semseg.proj.weight'].shape == (768, 64, 4, 4)
    assert sd['output_adapters.semseg.out_proj.weight'].shape == (2128, 256)
    assert sd['output_adapters.rgb.task_embeddings.depth'].shape == (1, 1, 256)
    assert not m.input_adapters['rgb'].pos_emb.requires_grad
    keys = list(sd)
    assert keys[0] == 'global_tokens' and keys[-1] == 'encoder.11.mlp.fc2.bias'


def test_seeded_init_tiny_matches_reference_sum():
    gold = load_scalars()['tiny_rgb']
    torch.manual_seed(0)
    m = build_engine_model(['rgb'], 8, 64, enc=(192, 12, 3), posemb_size=224)
    sd = m.state_dict()
    assert len(sd) == gold['num_keys']
    assert abs(float(sum(v.double().sum() for v in sd.values())) - gold['state_dict_sum']) < 1e-6


def test_mini_state_dict_loads_golden_weights():
    g = load_mini()
    m = build_mini_engine()
    missing, unexpected = m.load_state_dict(g['sd'], strict=True)
    assert not missing and not unexpected
    # same key ORDER as the reference's state_dict (the fixture preserves it)
    assert list(m.state_dict()) == list(g['sd'])


def test_registry_and_factories():
    import multimae_amd as M
    from multimae_amd import registry
    for n in ('pretrain_multimae_base', 'pretrain_multimae_large', 'multivit_base', 'multivit_large'):
        assert registry.is_model(n)
    m = M.multivit_base(input_adapters={'rgb': M.PatchedInputAdapter(3, 1, 16)}, output_adapters=None)
    assert isinstance(m, M.MultiViT) and m.get_num_layers() == 12
    assert 'global_tokens' in m.state_dict()
    with pytest.raises(RuntimeError):
        M.create_model('no_such_model')


def test_no_cpu_fallback():
    """The product path must fail loudly off-GPU instead of silently computing with torch."""
    m = build_mini_engine()
    x = {'rgb': torch.randn(2, 3, 32, 32), 'depth': torch.randn(2, 1, 32, 32), 'semseg': torch.randint(0, 133, (2, 8, 8))}
    with pytest.raises(RuntimeError, match='no CPU'):
        m(x, num_encoded_tokens=12)
    import multimae_amd as M
    with pytest.raises(RuntimeError, match='no CPU'):
        M.MaskedMSELoss(8)(torch.randn(2, 3, 32, 32), torch.randn(2, 3, 32, 32), torch.ones(2, 16, dtype=torch.long))


def test_input_info_and_sincos_match_oracle():
    import multimae_oracle as orc
    from multimae_amd.multimae_utils import build_2d_sincos_posemb
    for h, w, d in [(14, 14, 768), (4, 4, 128), (28, 28, 192), (3, 5, 16)]:
        assert torch.equal(build_2d_sincos_posemb(h, w, d), orc.sincos_posemb_2d(h, w, d))
    m = build_mini_engine()
    info = m.generate_input_info({'rgb': 16, 'depth': 16, 'semseg': 16}, (32, 32))
    assert info['tasks']['depth'] == {'num_tokens': 16, 'has_2d_posemb': True, 'start_idx': 16, 'end_idx': 32}
    assert info['num_task_tokens'] == 48 and info['num_global_tokens'] == 1


def test_dropin_package_import_surface(tmp_path):
    """dropin/ makes the engine importable as `multimae` with the reference's import lines
    (run_pretraining_multimae.py:36-40) and registers the factories in a live utils.registry."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # a stand-in for the reference's utils.registry (same private dict name as utils/registry.py:22)
    (tmp_path / 'utils').mkdir()
    (tmp_path / 'utils' / '__init__.py').write_text('')
    (tmp_path / 'utils' / 'registry.py').write_text('_model_entrypoints = {}\n')
    code = (
        'import utils.registry\n'
        'from multimae import multimae\n'
        'from multimae.criterion import MaskedCrossEntropyLoss, MaskedL1Loss, MaskedMSELoss\n'
        'from multimae.input_adapters import PatchedInputAdapter, SemSegInputAdapter\n'
        'from multimae.output_adapters import SpatialOutputAdapter\n'
        'from multimae.multimae import pretrain_multimae_base, MultiViT\n'
        'assert set(utils.registry._model_entrypoints) >= {"pretrain_multimae_base", "pretrain_multimae_large", "multivit_base", "multivit_large"}\n'
        'm = utils.registry._model_entrypoints["pretrain_multimae_base"](input_adapters={"rgb": PatchedInputAdapter(3, 1, 16)}, output_adapters=None)\n'
        'assert type(m).__module__ == "multimae_amd.multimae" and len(m.encoder) == 12\n'
        'print("ok")\n')
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(root, 'dropin'), str(tmp_path)]))
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'ok' in out.stdout, out.stderr[-2000:]


@pytest.mark.skipif(not os.path.isdir('/root/reference/utils'), reason='reference checkout not present (GPU box)')
def test_reference_create_model_resolves_to_engine():
    """The reference's OWN utils.model_builder.create_model / utils.registry (unmodified files, imported through the
    Appendix-B stub so utils/__init__.py's torchvision import never runs) builds the engine's MultiMAE when
    dropin/ is first on sys.path -- the call run_pretraining_multimae.py:285-291 makes."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        'import sys, types\n'
        'pkg = types.ModuleType("utils"); pkg.__path__ = ["/root/reference/utils"]; sys.modules["utils"] = pkg\n'
        'from multimae import multimae\n'
        'from multimae.input_adapters import PatchedInputAdapter, SemSegInputAdapter\n'
        'from multimae.output_adapters import SpatialOutputAdapter\n'
        'from utils.model_builder import create_model\n'
        'ins = {"rgb": PatchedInputAdapter(num_channels=3, stride_level=1, patch_size_full=16)}\n'
        'outs = {"rgb": SpatialOutputAdapter(num_channels=3, stride_level=1, patch_size_full=16, dim_tokens=256, depth=2, num_heads=8,\n'
        '                                   use_task_queries=True, task="rgb", context_tasks=["rgb"], use_xattn=True)}\n'
        'm = create_model("pretrain_multimae_base", input_adapters=ins, output_adapters=outs, num_global_tokens=1, drop_path_rate=0.0)\n'
        'assert type(m).__module__ == "multimae_amd.multimae", type(m)\n'
        'assert m.get_num_layers() == 12 and "global_tokens" in m.no_weight_decay()\n'
        'print("ok")\n')
    env = dict(os.environ, PYTHONPATH=os.path.join(root, 'dropin'), PYTHONDONTWRITEBYTECODE='1')
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'ok' in out.stdout, out.stderr[-2000:]


def test_host_inputs_refresh_protocol():
    """engine.HostInputs (the per-step host values of a captured step graph): the value drawn at capture time serves the
    first replay, later replays call the thunk again, in registration order."""
    from multimae_amd import engine
    calls = []

    def make(tag):
        def thunk():
            calls.append(tag)
            return torch.tensor([float(len(calls))])
        return thunk
    hi = engine.HostInputs()
    a = hi.add(make('a'), 'cpu')
    b = hi.add(make('b'), 'cpu')
    assert calls == ['a', 'b']
    hi.refresh()                                   # first replay: pending capture-time values, no new draw
    assert calls == ['a', 'b'] and float(a) == 1.0 and float(b) == 2.0
    hi.refresh()
    assert calls == ['a', 'b', 'a', 'b'] and float(a) == 3.0 and float(b) == 4.0
    # eager mode: host_input just evaluates the thunk
    assert engine.capturing() is None
    assert float(engine.host_input(make('c'), 'cpu')) == 5.0


def test_linear_output_adapter_seeded_init_and_oracle():
    """Same RNG consumption as the reference's LinearOutputAdapter.init (trunc-normal head, output_adapters.py:313-336): a
    seeded construction reproduces its weights bit for bit; the oracle restatement reproduces the reference outputs."""
    import os
    import numpy as np
    import multimae_amd as M
    import multimae_oracle as orc
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'linear_head.npz'))
    torch.manual_seed(32)
    h = M.LinearOutputAdapter(num_classes=10, dim_tokens_enc=32)
    assert list(h.state_dict()) == ['norm.weight', 'norm.bias', 'head.weight', 'head.bias']
    assert torch.equal(h.head.weight.detach(), torch.from_numpy(z['init/head.weight']))
    for pool in ('mean', 'last'):
        sd = {k[len(pool) + 4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pool + '/sd/')}
        y = orc.linear_output_adapter(torch.from_numpy(z[pool + '/x']), sd, use_mean_pooling=(pool == 'mean'))
        assert torch.allclose(y, torch.from_numpy(z[pool + '/y']), atol=1e-6)


def test_bench_result_line_keeps_the_measurement_contract():
    """bench.py assembles its one JSON line in result_line(): every field of the measurement contract from the LIVE code (not from
    committed files), for the default single-GPU line and a data-parallel one."""
    import argparse
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    args = argparse.Namespace(config='cfg3', precision='bf16', steps=20, warmup=5)
    roof = {'bound': 'mfma', 'achieved': 750.0, 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': 0.3, 'traffic': None}
    cpu = {'value': 14.0, 'unit': 'images/s', 'cores': 16, 'kind': 'port', 'sample': 'B=16, 5 steps'}
    d = bench.result_line(args, 256, 1, 32.0, 8000.0, 8.08, {'steps': 25}, roof, cpu, None, 9.0, 20.0, False, 98, ['rgb', 'depth', 'semseg'])
    d = json.loads(json.dumps(d))                        # must be JSON-serialisable as is
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
              'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['unit'] == 'images/s' and d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
    # the storage form of the fp32 output adapter is part of the precision statement, where a truncated workload string cannot lose it
    assert d['dtype'] == 'bf16 (fp32_output_adapters: fp16 storage)' and d['data'] == 'synthetic' and 'workload' in d['config'] and 'model' not in d['config']
    assert d['n_gpus'] == 1 and d['steps'] == 20 and d['warmup'] == 5 and d['config']['global_batch'] == 256 and d['config']['parallelism'] == 'dp1'
    assert 'fp16 STORAGE' in d['config']['workload'] and 'TF32' in d['config']['workload'] and 'configs[2]' in d['config']['workload']
    args_f16 = argparse.Namespace(config='cfg3', precision='bf16', steps=20, warmup=5, fp32_adapter_gemm='f16')
    assert 'fp16 operands' in bench.result_line(args_f16, 256, 1, 32.0, 8000.0, 8.08, {'steps': 25}, roof, cpu, None, 9.0, 20.0, False, 98,
                                                ['rgb', 'depth', 'semseg'])['config']['workload']
    assert bench.result_line(args_f16, 256, 1, 32.0, 8000.0, 8.08, {'steps': 25}, roof, cpu, None, 9.0, 20.0, False, 98,
                             ['rgb', 'depth', 'semseg'])['dtype'] == 'bf16 (fp32_output_adapters: f32 tensors, fp16 operands)'
    assert bench.result_line(argparse.Namespace(config='cfg2', precision='bf16', steps=20, warmup=5), 256, 1, 32.0, 8000.0, 8.08, {'steps': 25}, roof, cpu, None,
                             9.0, 20.0, False, 98, ['rgb'])['dtype'] == 'bf16'
    args_x3 = argparse.Namespace(config='cfg3', precision='bf16', steps=20, warmup=5, fp32_adapter_gemm='x3')
    assert 'x3 split-bf16' in bench.result_line(args_x3, 256, 1, 32.0, 8000.0, 8.08, {'steps': 25}, roof, cpu, None, 9.0, 20.0, False, 98,
                                                ['rgb', 'depth', 'semseg'])['config']['workload']
    assert 'data_parallel' not in d
    dp = {'exposed_allreduce_ms_per_step': 1.2, 'bucket_mb': 64.0, 'buckets': 7, 'rccl_ranks_seen': 8}
    d8 = bench.result_line(args, 256, 8, 36.0, 56000.0, 8.08, {'steps': 25}, roof, None, dp, 9.0, 20.0, False, 98, ['rgb', 'depth', 'semseg'])
    assert d8['n_gpus'] == 8 and d8['config']['global_batch'] == 2048 and d8['config']['parallelism'] == 'dp8' and d8['data_parallel'] == dp


def test_isa_audit_of_the_pingpong_kernels():
    """the transposing fragment reads of the ping-pong GEMMs are inline asm the compiler does not wait for.  On the BUILT
    objects: no instruction touches a `ds_read_b64_tr_b16` destination before the `s_waitcnt lgkmcnt(0)` that covers it, and the
    flavoured instantiations the training step launches use no scratch (a spill of a fragment register would be such a touch)."""
    import importlib.util
    import os
    import pytest
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('isa_audit', os.path.join(root, 'tools', 'isa_audit.py'))
    ia = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ia)
    objs = [o for o in ia.DEFAULT if os.path.exists(o)]
    if len(objs) != len(ia.DEFAULT) or not os.path.exists(os.path.join(ia.LLVM, 'llvm-objdump')):
        pytest.skip('object files of the kernel library not present (build with make -C multimae_amd/csrc)')
    report, problems = ia.audit(objs)
    assert not problems, problems
    fl = [(k, r, n) for _, k, r, n in report if ia.flavoured(k)]
    assert len(fl) >= 16
    for k, r, _ in fl:
        assert r['scratch'] == 0 and r['spill'] == 0, (k, r)
    assert sum(n for k, r, n in fl) > 0 and any(n > 0 for _, k, r, n in report if 'dwgroup' in k)      # the audit did see the asm reads


def test_lazy_prediction_is_written_on_first_read_only():
    """lazy.LazyPrediction (round 4): the (B, C, H, W) image an output adapter returns is filled from the patch rows the first time a
    torch function reads it; metadata, record_stream-type calls and the loop's `preds[task].float()` do not trigger the write; the
    result of any real operation is a plain tensor; autograd through the wrapper reaches the producing node."""
    import torch
    from multimae_amd.lazy import LazyPrediction, materialize
    calls = []
    ref = torch.arange(96.).view(2, 3, 4, 4)

    def fill(t):
        calls.append(1)
        t.detach().copy_(ref)
    x = LazyPrediction.wrap(torch.empty(2, 3, 4, 4), fill)
    assert tuple(x.shape) == (2, 3, 4, 4) and x.dtype == torch.float32 and x.size(1) == 3 and x.dim() == 4 and len(x) == 2 and not x.is_cuda
    assert x.float() is x and not x.materialized and calls == []
    x._mmae_pat = 'handle'
    assert getattr(x.float(), '_mmae_pat') == 'handle'
    y = x + 1
    assert calls == [1] and type(y) is torch.Tensor and torch.equal(y, ref + 1) and x.materialized
    assert torch.equal(x.clone(), ref) and calls == [1]                       # written once
    z = LazyPrediction.wrap(torch.empty(2, 3, 4, 4), fill)
    materialize(z)
    materialize(z)
    assert calls == [1, 1] and torch.equal(z.detach(), ref)
    d = LazyPrediction.wrap(torch.empty(2, 3, 4, 4), fill).double()            # a real conversion reads the data
    assert calls == [1, 1, 1] and d.dtype == torch.float64 and torch.equal(d, ref.double())

    class F(torch.autograd.Function):
        @staticmethod
        def forward(ctx, w):
            return torch.empty_like(w)

        @staticmethod
        def backward(ctx, g):
            return g * 2
    w = torch.zeros(2, 3, 4, 4, requires_grad=True)
    lo = LazyPrediction.wrap(F.apply(w), fill)
    assert lo.requires_grad and lo.grad_fn is not None and calls == [1, 1, 1]
    (lo.float() * 3).sum().backward()
    assert calls == [1, 1, 1, 1] and torch.equal(w.grad, torch.full((2, 3, 4, 4), 6.0))

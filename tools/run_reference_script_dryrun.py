"""Execute the reference's UNMODIFIED ``run_pretraining_multimae.py`` end to end through the drop-in seam (``dropin/multimae`` -> the
engine's modules), in the build container (CPU only; the reference checkout exists here, a GPU does not -- on the GPU box it is the
other way round), with

  * the C ABI replaced by the type-checking stub of tests/dryrun_harness.py (every library call is validated against include/mmae.h,
    outputs are zero-filled: the NUMBERS ARE MEANINGLESS, the control flow is the real one);
  * ``utils.datasets`` replaced by a synthetic dataset of the pre-training shapes ("stub utils.datasets");
  * import stubs for what this image lacks and the script's imports pull in: torchvision / timm / wandb / ... (auto-stubbed on
    ModuleNotFoundError) and ``torch._six`` (removed from torch long ago; utils/native_scaler.py:10 imports ``inf`` from it);
  * ``torch.cuda.synchronize`` / ``max_memory_allocated`` as no-ops (run_pretraining_multimae.py:540 calls them unconditionally), and
    ``{'scale': 1.0}`` as the state of the GradScaler that torch disables on a machine without a GPU (:538 reads ``['scale']``).

What it shows: argument parsing, ``utils.create_model('pretrain_multimae_base', ...)`` resolving to the engine's factory through the
reference's own registry, the DOMAIN_CONF adapters / criteria, ``create_optimizer``'s dict branch, ``NativeScaler``, the cosine tables,
``train_one_epoch`` with its MetricLogger, and the epoch loop all run against the engine's classes without a line of the script changed.
With ``--patched`` the same run on a temporary copy of the script with ``dropin/run_pretraining_multimae.patch`` applied
(``amd_loop.wrap_model`` / ``create_optimizer`` / ``LossScaler``).

    PYTHONDONTWRITEBYTECODE=1 python tools/run_reference_script_dryrun.py [--patched] > profiles/r05_reference_script_dryrun[_patched].log
"""
import importlib.abc
import importlib.machinery
import os
import runpy
import shutil
import subprocess
import sys
import tempfile
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
PATCHED = '--patched' in sys.argv

import torch  # noqa: E402

# ---- import stubs ------------------------------------------------------------------------------------------------------------
STUB_TOPS = {'torchvision', 'timm', 'wandb', 'albumentations', 'cv2', 'PIL', 'skimage', 'tensorboardX', 'matplotlib', 'scipy_stub'}


class _Anything:
    """stands for any class / function of a stubbed package: constructible, callable, subclassable, attribute-complete"""
    def __init__(self, *a, **k): pass
    def __call__(self, *a, **k): return self
    def __getattr__(self, n): return _Anything()
    def __mro_entries__(self, bases): return (object,)


class _StubModule(types.ModuleType):
    def __getattr__(self, n):
        if n.startswith('__'):
            raise AttributeError(n)
        return type(n, (_Anything,), {})


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split('.')[0] in STUB_TOPS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        print(f'[dryrun] stubbed import: {module.__name__}')


sys.meta_path.append(_StubFinder())
six = types.ModuleType('torch._six')
six.inf, six.string_classes, six.container_abcs = float('inf'), (str,), __import__('collections.abc').abc
sys.modules['torch._six'] = six
torch.cuda.synchronize = lambda *a, **k: None
torch.cuda.max_memory_allocated = lambda *a, **k: 0
# without a GPU torch disables the GradScaler NativeScaler creates, and a disabled scaler's state_dict() is {} -- the loop reads ['scale'] (:538)
_sd = torch.cuda.amp.GradScaler.state_dict
torch.cuda.amp.GradScaler.state_dict = lambda self: (_sd(self) or {'scale': 1.0})

# ---- the engine on CPU behind the type-checking ABI stub -------------------------------------------------------------------------
sys.path[:0] = [os.path.join(ROOT, 'dropin'), ROOT, os.path.join(ROOT, 'tests')]
import dryrun_harness  # noqa: E402
dryrun_harness.install()
_empty = torch.empty
torch.empty = lambda *a, **k: _empty(*a, **k).zero_()        # stubbed kernels write nothing: keep every "output" finite

# ---- synthetic dataset instead of utils.datasets --------------------------------------------------------------------------------
os.chdir(REF)
sys.path.insert(2, REF)                                       # after dropin/ (so `multimae` is the drop-in package), before site-packages
import utils  # noqa: E402  (the reference's own package: registry, optimiser factory, NativeScaler, MetricLogger, ...)

STEPS, BATCH = 3, 2


class _Synthetic(torch.utils.data.Dataset):
    def __len__(self): return STEPS * BATCH

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(i)
        return ({'rgb': torch.randn(3, 224, 224, generator=g), 'depth': torch.randn(1, 224, 224, generator=g),
                 'semseg': torch.randint(0, 133, (56, 56), generator=g)}, 0)


ds = types.ModuleType('utils.datasets')
ds.build_multimae_pretraining_dataset = lambda args: _Synthetic()
sys.modules['utils.datasets'] = ds
utils.datasets = ds

script = os.path.join(REF, 'run_pretraining_multimae.py')
tmpdir = None
if PATCHED:
    tmpdir = tempfile.mkdtemp()
    dst = os.path.join(tmpdir, 'run_pretraining_multimae.py')
    shutil.copy(script, dst)
    r = subprocess.run(['patch', '-p0', dst, os.path.join(ROOT, 'dropin', 'run_pretraining_multimae.patch')], capture_output=True, text=True)
    print('[dryrun] patch:', r.stdout.strip(), r.stderr.strip())
    assert r.returncode == 0, 'dropin/run_pretraining_multimae.patch does not apply to the reference script'
    script = dst
sys.argv = [script, '--device', 'cpu', '--batch_size', str(BATCH), '--epochs', '2', '--warmup_epochs', '1', '--num_workers', '0',
            '--in_domains', 'rgb-depth-semseg', '--out_domains', 'rgb-depth-semseg', '--extra_norm_pix_loss', '--no_loss_on_unmasked',
            '--fp32_output_adapters', 'semseg', '--no_standardize_depth', '--no_auto_resume', '--output_dir', '', '--model', 'pretrain_multimae_base']
print(f'[dryrun] running {"the PATCHED copy of " if PATCHED else "the UNMODIFIED "}{os.path.join(REF, "run_pretraining_multimae.py")}: ' + ' '.join(sys.argv[1:]))
import multimae  # noqa: E402
print('[dryrun] `multimae` resolves to', os.path.dirname(multimae.__file__))
assert os.path.dirname(multimae.__file__).startswith(os.path.join(ROOT, 'dropin'))
try:
    runpy.run_path(script, run_name='__main__')
    print('[dryrun] the script ran to completion')
finally:
    if tmpdir:
        shutil.rmtree(tmpdir, ignore_errors=True)

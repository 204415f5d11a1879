"""CPU (build container only: needs the reference checkout): the reference's UNMODIFIED run_pretraining_multimae.py executed end to
end through the drop-in seam on the type-checking ABI stub, and again with dropin/run_pretraining_multimae.patch applied
(tools/run_reference_script_dryrun.py; logs kept as profiles/r05_reference_script_dryrun*.log)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/run_pretraining_multimae.py'


@pytest.mark.skipif(not os.path.exists(REF), reason='the reference checkout exists in the build container only')
@pytest.mark.parametrize('patched', [False, True])
def test_reference_training_script_runs_through_the_dropin_package(patched):
    cmd = [sys.executable, os.path.join(ROOT, 'tools', 'run_reference_script_dryrun.py')] + (['--patched'] if patched else [])
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}      # (earlier tests of the session set them)
    env['PYTHONDONTWRITEBYTECODE'] = '1'
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert '`multimae` resolves to ' + os.path.join(ROOT, 'dropin', 'multimae') in out
    assert 'Creating model: pretrain_multimae_base' in out and 'Number of params: 97.917072 M' in out      # the engine's factory, the reference's count
    assert out.count('Averaged stats:') == 2 and '[dryrun] the script ran to completion' in out
    if patched:
        assert 'patching file' in out

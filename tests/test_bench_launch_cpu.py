"""CPU: the LAUNCH path of bench.py.  `python bench.py --gpus N` is the driver's command shape; without a launcher in the
environment it must start its N ranks itself (it died on an assert).  Run here as a real subprocess with
--dry-run 1 (CPU tensors, the type-checking stub of the C ABI, gloo): self-launch -> torch.distributed.run -> process group ->
reducer -> rank 0 prints exactly one JSON line.  The numbers in that line are meaningless by construction."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(env_extra or {})
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--dry-run', '1', '--steps', '2', '--warmup', '1'] + extra
    return subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)


def test_bench_gpus_2_launches_its_own_ranks_and_prints_one_line():
    r = _run(['--gpus', '2'])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['parallelism'] == 'dp2' and out['config']['global_batch'] == 2 * out['config']['per_gpu_batch']
    dp = out['data_parallel']
    assert dp['rccl_ranks_seen'] == 2 and dp['backend'] == 'gloo' and dp['buckets'] >= 2
    assert dp['gemm_cu_reserved'] == 16                 # the persistent GEMM grids leave CUs to the collective while buckets fly
    assert 'DRY RUN' in out['data']
    assert 'without a launcher' in r.stderr


def test_bench_single_rank_dry_run_line_keeps_the_contract():
    r = _run(['--gpus', '1'])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
              'config', 'roofline', 'cpu_baseline'):
        assert k in out, k
    assert out['n_gpus'] == 1 and 'data_parallel' not in out


def test_bench_refuses_a_launcher_environment_that_disagrees_with_gpus():
    r = _run(['--gpus', '2'], env_extra=dict(WORLD_SIZE='1', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29541'), timeout=300)
    assert r.returncode != 0 and 'must agree' in r.stderr

"""CPU, world_size 2, gloo: the data-parallel gradient reducer (bucket planning, readiness-driven
launch, averaging) on the same code path the GPU ranks use with RCCL."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from multimae_amd.dist import GradAllReducer, plan_buckets


def test_plan_buckets_contiguous_cover():
    sizes, off = [], 0
    for i, n in enumerate([64, 640, 128, 4096, 64, 64, 1024]):
        sizes.append((f'p{i}', off, n)); off += n
    b = plan_buckets(sizes, 1000)
    assert b[0][0] == 0 and b[-1][1] == off
    for (s0, e0, _), (s1, e1, _) in zip(b, b[1:]):
        assert e0 == s1
    assert sum(len(n) for _, _, n in b) == len(sizes)


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    try:
        _worker_body(rank, world, port, q)
    except Exception as e:          # surface failures instead of a queue timeout
        q.put((rank, False, repr(e)))


def _worker_body(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(rank)
    sizes, off = [], 0
    for i, n in enumerate([64, 640, 128, 4096, 64, 64, 1024]):
        sizes.append((f'encoder.{i}.w', off, n)); off += n
    grad = torch.randn(off)
    mine = grad.clone()
    red = GradAllReducer(grad, sizes, bucket_mb=700 * 4 / (1024 * 1024))
    assert len(red.buckets) >= 3
    # backward order: last layers first; a bucket only launches when ALL its tensors reported
    for i in reversed(range(3, 7)):
        red.mark_prefix_ready(f'encoder.{i}')
    launched_mid = list(red._launched)
    red.finish()
    other = torch.empty_like(mine)
    torch.manual_seed(1 - rank)
    other = torch.randn(off)
    q.put((rank, torch.allclose(grad, (mine + other) / 2, atol=1e-6), launched_mid))
    dist.destroy_process_group()


def test_grad_allreduce_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    for rank, ok, launched_mid in res:
        assert ok, f'rank {rank}: {launched_mid}'
        assert any(launched_mid) and not all(launched_mid), 'buckets must launch incrementally as layers finish'

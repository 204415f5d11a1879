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
    # a forced cut: the bucket in progress ends right before the named tensor (the small, exposed tail bucket)
    c = plan_buckets(sizes, 1000, cut_before=['p5'])
    assert any(names[0] == 'p5' for _, _, names in c) and c[-1][1] == off
    assert sum(len(n) for _, _, n in c) == len(sizes)


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
    assert len(red.buckets) >= 3 and red.grad_prescale == 0.5          # buckets are SUMMED; the optimiser applies 1 / world
    # backward order: last layers first; a bucket only launches when ALL its tensors reported
    for i in reversed(range(3, 7)):
        red.mark_prefix_ready(f'encoder.{i}')
    launched_mid = list(red._launched)
    red.finish()
    other = torch.empty_like(mine)
    torch.manual_seed(1 - rank)
    other = torch.randn(off)
    ok = torch.allclose(grad, mine + other, atol=1e-6)
    # the same exchange with bf16 buckets: the fp32 arena receives the bf16-rounded sum of the bf16-rounded addends
    g2 = mine.clone()
    red2 = GradAllReducer(g2, sizes, bucket_mb=700 * 4 / (1024 * 1024), bf16_buckets=True)
    red2.finish()
    exact = mine + other
    ok = ok and bool(((g2 - exact).abs() <= 2.0 ** -7 * (mine.abs() + other.abs()) + 1e-6).all()) and not torch.equal(g2, exact)
    q.put((rank, ok, launched_mid))
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


def _model_worker(rank, world, port, q, direct=True, average=False, learnable_pos=False, bf16_buckets=False, exchange='all_reduce'):
    try:
        import sys
        here = os.path.dirname(os.path.abspath(__file__))
        sys.path[:0] = [here, os.path.join(os.path.dirname(here), 'oracle')]
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group('gloo', rank=rank, world_size=world)
        import dryrun_harness
        dryrun_harness.install()                      # C ABI stubbed: the host-side control flow runs, kernels do not
        import multimae_amd as M
        from multimae_amd.dist import attach, broadcast_parameters
        from helpers import MINI, build_mini_engine, load_mini
        torch.manual_seed(100 + rank)                 # ranks build different initial weights (run_pretraining_multimae.py:300)
        model = build_mini_engine(learnable_pos=learnable_pos)
        arena = model.build_arena()
        before = arena.param.clone()
        broadcast_parameters(arena)                   # DDP-constructor semantics: rank 0's values everywhere
        gathered = [torch.empty_like(arena.param) for _ in range(world)]
        dist.all_gather(gathered, arena.param)
        same_params = all(torch.equal(g, gathered[0]) for g in gathered)
        red = GradAllReducer.for_arena(arena, bucket_mb=0.25, average_in_place=average, bf16_buckets=bf16_buckets, exchange=exchange)
        attach(model, red)
        # arena = gradient-readiness order: last output adapter first, encoder from the top down, embedding parameters last
        names = [n for n in arena.names if arena.trainable[n]]
        assert names[0].startswith('output_adapters.norm_rgb.') and names[-1] in arena.tail_names(), (names[0], names[-1])
        enc_first = [i for i, n in enumerate(names) if n.startswith('encoder.')][0]
        assert names[enc_first].startswith(f'encoder.{len(model.encoder) - 1}.')
        tail = set(arena.tail_names())
        for _, _, bn in red.buckets:                  # the embedding parameters never share a bucket with anything else
            assert set(bn) <= tail or not (set(bn) & tail), bn
        assert set(red.buckets[-1][2]) <= tail
        order = []
        launch = red._launch
        red._launch = lambda i: (order.append(i) if not red._launched[i] else None, launch(i))[1]
        g = load_mini()
        P = MINI['P']
        fns = {'rgb': M.MaskedMSELoss(P, 1), 'depth': M.MaskedL1Loss(P, 1), 'semseg': M.MaskedCrossEntropyLoss(P, 4),
               'norm_rgb': M.MaskedMSELoss(P, 1, norm_pix=True)}
        M.engine.set_direct_grads(direct)
        arena.grad.fill_(float(rank + 1))             # stand-in gradients (the stubbed kernels write nothing)
        with M.engine.precision('bf16'):
            preds, masks = model(g['x'], num_encoded_tokens=MINI['nvis'], alphas=1.0, fp32_output_adapters=['semseg'])
            mk = dict(masks, norm_rgb=masks['rgb'])
            tgt = dict(g['x'], norm_rgb=g['x']['rgb'])
            sum(fns[k](preds[k].float(), tgt[k], mask=mk[k]) for k in preds).backward()
        mid = sum(red._launched)
        if not direct:
            # autograd delivered the (stub) gradients with AccumulateGrad AFTER the nodes returned: the callbacks must have been
            # ignored, or the buckets would have been reduced before the gradients existed (dist.py:111)
            assert mid == 0, mid
            arena.grad.fill_(float(rank + 1))
        red.finish()
        total = sum(range(1, world + 1))
        ok = bool(torch.allclose(arena.grad, torch.full_like(arena.grad, total / world if average else float(total))))
        q.put((rank, ok and same_params and (rank == 0 or not torch.equal(before, arena.param)), order, mid, len(red.buckets)))
        dist.destroy_process_group()
    except Exception as e:          # noqa: BLE001
        import traceback
        q.put((rank, False, traceback.format_exc(), 0, 0))


def test_model_backward_drives_reducer_world2_gloo():
    """The real model's backward (host control flow on the stubbed C ABI) drives the bucketed all-reduce on two gloo ranks: the
    adapters' and encoder layers' completion callbacks launch buckets in the SAME order on both ranks while backward is still
    running, finish() reduces the rest, every gradient element ends up averaged, and broadcast_parameters leaves both ranks
    with rank 0's weights."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_model_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    for rank, ok, order, mid, nb in res:
        assert ok, f'rank {rank}: {order}'
        assert nb >= 4 and 0 < mid <= nb, (mid, nb)           # buckets launched from inside backward
        assert order == sorted(order), f'buckets complete in arena order (= readiness order): {order}'
    assert res[0][2] == res[1][2], 'ranks must issue their collectives in the same order'


def _run_model_workers(world, **kw):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_model_worker, args=(r, world, port, q), kwargs=kw) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    return res


def test_reducer_ignores_callbacks_without_direct_grads_world2_gloo():
    """engine.set_direct_grads(False): the hand-written backward hands its gradients to autograd, which adds them into .grad only
    after the node returned -- a bucket launched from the node's completion callback would be reduced before its gradients
    exist.  The reducer must ignore the callbacks then and reduce everything in finish() (here with in-place averaging)."""
    res = _run_model_workers(2, direct=False, average=True)
    for rank, ok, order, mid, nb in res:
        assert ok, f'rank {rank}: {order}'
        assert mid == 0
    assert res[0][2] == res[1][2]


def test_model_backward_drives_reducer_world4_gloo():
    """Four ranks, 0.25 MB buckets: several buckets close inside each output adapter's backward (on the GPU: on that adapter's
    stream -- the events recorded per readiness report order the all-reduce behind every contributing stream)."""
    res = _run_model_workers(4)
    for rank, ok, order, mid, nb in res:
        assert ok, f'rank {rank}: {order}'
        assert order == sorted(order) and mid >= nb - 1, (order, mid, nb)      # everything but the tail launched inside backward
    assert all(r[2] == res[0][2] for r in res)


def test_model_backward_drives_reducer_world8_gloo():
    """The node size the driver's scaling run uses (8 ranks): identical bucket order on every rank, all buckets but the tail
    launched inside backward, summed gradients equal to the closed form (the optimiser applies 1 / 8)."""
    res = _run_model_workers(8)
    assert len(res) == 8
    for rank, ok, order, mid, nb in res:
        assert ok, f'rank {rank}: {order}'
        assert order == sorted(order) and mid >= nb - 1, (order, mid, nb)
    assert all(r[2] == res[0][2] for r in res)


def test_tail_bucket_waits_for_autograd_delivered_embedding_gradients_world2_gloo():
    """with learnable_pos_emb=True the position tables' gradients reach .grad through F.interpolate's
    backward + AccumulateGrad AFTER EmbedFn.backward has returned; reporting 'input_adapters' ready from inside that node let
    the tail bucket's all-reduce race with (or miss) them.  The embedding bucket must now stay un-launched until finish()."""
    res = _run_model_workers(2, learnable_pos=True)
    for rank, ok, order, mid, nb in res:
        assert ok, f'rank {rank}: {order}'
        assert mid == nb - 1, (mid, nb)                      # everything but the embedding (tail) bucket launched inside backward
        assert order == sorted(order) and order[-1] == nb - 1
    assert res[0][2] == res[1][2]


def test_bf16_gradient_buckets_world2_gloo():
    """bf16_buckets=True: every bucket travels as bf16 and is widened back into the fp32 arena.  The stand-in
    gradients (rank + 1) are exact in bf16, so the sum must match the fp32-bucket result bit for bit."""
    res = _run_model_workers(2, bf16_buckets=True)
    for rank, ok, order, mid, nb in res:
        assert ok, f'rank {rank}: {order}'
    assert res[0][2] == res[1][2]


def test_reduce_scatter_all_gather_buckets_world4_gloo():
    """exchange='rs_ag' (the direct form for point-to-point xGMI): every bucket is summed by a
    reduce_scatter followed by an all_gather of the summed shards.  Same readiness-ordered launches, same sums (the stand-in gradients
    are small integers: exact in any order), identical bucket order on every rank -- at world 4 and, with bf16 buckets, at world 2."""
    res = _run_model_workers(4, exchange='rs_ag')
    for rank, ok, order, mid, nb in res:
        assert ok, f'rank {rank}: {order}'
        assert order == sorted(order) and mid >= nb - 1, (order, mid, nb)
    assert all(r[2] == res[0][2] for r in res)
    res = _run_model_workers(2, exchange='rs_ag', bf16_buckets=True)
    for rank, ok, order, mid, nb in res:
        assert ok, f'rank {rank}: {order}'
    assert res[0][2] == res[1][2]


def _rs_ag_numeric_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group('gloo', rank=rank, world_size=world)
        n = 64 * 37                                      # arena tensors are 64-element aligned: every bucket divides by 2 / 4 / 8 ranks
        sizes = [(f'p{i}', 64 * i * 1, 64) for i in range(37)]
        out = {}
        for mode in ('all_reduce', 'rs_ag'):
            torch.manual_seed(7 + rank)
            grad = torch.randn(n)
            red = GradAllReducer(grad, sizes, bucket_mb=64 * 8 * 4 / (1024 * 1024), exchange=mode)
            assert len(red.buckets) >= 4
            red.finish()
            out[mode] = grad.clone()
        torch.manual_seed(7)
        ref = sum(torch.randn(n, generator=torch.Generator().manual_seed(7 + r)) for r in range(world))
        q.put((rank, bool(torch.allclose(out['rs_ag'], out['all_reduce'], rtol=0, atol=1e-5)), bool(torch.allclose(out['rs_ag'], ref, rtol=0, atol=1e-5)),
               bool(torch.equal(out['rs_ag'], out['rs_ag']))))
        # a bucket that the world size does not divide falls back to all_reduce (stand-alone use with unaligned tensors)
        grad = torch.full((10,), float(rank + 1))
        red = GradAllReducer(grad, [('a', 0, 10)], bucket_mb=1.0, exchange='rs_ag')
        red.finish()
        assert torch.allclose(grad, torch.full((10,), float(sum(range(1, world + 1))))), grad
        dist.destroy_process_group()
    except Exception as e:                               # noqa: BLE001
        import traceback
        q.put((rank, False, False, traceback.format_exc()))


def test_reduce_scatter_all_gather_matches_all_reduce_on_random_gradients_world4_gloo():
    import torch.multiprocessing as mp
    world, port = 4, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_rs_ag_numeric_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = [q.get(timeout=180) for _ in range(world)]
    [p.join(60) for p in ps]
    for rank, same_as_allreduce, same_as_sum, extra in res:
        assert same_as_allreduce and same_as_sum, (rank, extra)


def test_reducer_reserves_compute_units_only_while_buckets_are_in_flight(monkeypatch):
    """from the first bucket launch of a step until finish() the persistent GEMM grids leave
    `gemm_cu_reserve` CUs to the collective's kernels (ops.gemm_cu_reserve -> mmae_gemm_cu_reserve); the forward pass and the
    optimiser step get the whole chip.  World size 1 with the collectives forced, the launch policy recorded through a stub."""
    from multimae_amd import ops
    calls = []
    monkeypatch.setattr(ops, 'gemm_cu_reserve', lambda k=None: calls.append(k) or 0)
    port = _free_port()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1')
    dist.init_process_group('gloo', rank=0, world_size=1)
    try:
        grad = torch.arange(4096, dtype=torch.float32)
        sizes = [('a', 0, 1024), ('b', 1024, 1024), ('c', 2048, 2048)]
        r = GradAllReducer(grad, sizes, bucket_mb=1024 * 4 / 2 ** 20, force_collective=True)
        assert len(r.buckets) >= 2
        r.gemm_cu_reserve = 16
        r.mark_ready(['a'])
        assert calls == [16]                                  # first bucket in flight: reserve
        r.mark_ready(['b'])
        assert calls == [16]                                  # already reserved
        r.finish()
        assert calls == [16, 0] and not r._reserved           # everything reduced: the whole chip again
        r.mark_ready(['a'])
        r.finish()
        assert calls == [16, 0, 16, 0]
        r.gemm_cu_reserve = 0                                 # off: the policy is never touched
        r.mark_ready(['a'])
        r.finish()
        assert calls == [16, 0, 16, 0]
        assert torch.equal(grad, torch.arange(4096, dtype=torch.float32))      # one-rank sums: unchanged values
    finally:
        dist.destroy_process_group()


def _skew_worker(rank, world, port, q):
    try:
        import time
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group('gloo', rank=rank, world_size=world)
        from multimae_amd import ops
        state = {'reserve': 0, 'log': []}

        def fake_reserve(k=None):
            prev = state['reserve']
            if k is not None and k >= 0:
                state['reserve'] = k
                state['log'].append(k)
            return prev
        ops.gemm_cu_reserve = fake_reserve
        grad = torch.full((4096,), float(rank + 1))
        sizes = [('a', 0, 1024), ('b', 1024, 1024), ('c', 2048, 1024), ('d', 3072, 1024)]
        r = GradAllReducer(grad, sizes, bucket_mb=1024 * 4 / 2 ** 20)
        r.gemm_cu_reserve = 16
        seen_at_step = []
        for step in range(3):
            grad.fill_(float(rank + 1))
            # the same readiness ORDER on every rank (the backward pass is the same program), different TIMING: the ranks reach each
            # report -- and finish() -- at different moments
            for j, n in enumerate(['a', 'b', 'c', 'd']):
                time.sleep(0.02 * ((rank + step + j) % 3))
                r.mark_ready([n])
                assert state['reserve'] == 16, (rank, step, n)       # from the first bucket launch on
            time.sleep(0.03 * ((rank + step) % 2))
            r.finish()
            seen_at_step.append(state['reserve'])                    # what the optimiser step would see on THIS rank
            assert torch.allclose(grad, torch.full((4096,), float(sum(range(1, world + 1)))))
        # a step that dies between the first bucket launch and finish(): reset() gives the CUs back
        r.mark_ready(['a'])
        assert state['reserve'] == 16
        for h, _ in r._handles:
            h.wait()
        r.reset()
        q.put((rank, seen_at_step, state['reserve'], state['log']))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:                     # pragma: no cover
        import traceback
        q.put((rank, 'error', repr(e), traceback.format_exc()))


def test_cu_reserve_is_lifted_before_the_optimizer_step_on_every_rank_under_timing_skew_world2_gloo():
    """the CUs reserved for the collective are returned by finish() -- i.e. before opt.step -- on EVERY rank, also
    when the ranks reach their readiness reports and finish() at different times; and by reset() after a step that raised."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_skew_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    for r in res:
        assert r[1] != 'error', r
        rank, seen, final, log = r
        assert seen == [0, 0, 0], (rank, seen)
        assert final == 0, (rank, final)
        assert log == [16, 0] * 4, (rank, log)

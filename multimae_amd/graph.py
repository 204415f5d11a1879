"""A whole pre-training step as ONE hipGraph.

Why: at B = 256 a cfg3 step is ~1 100 kernel launches on up to 10 HIP streams; enqueueing them from Python costs
20-25 us each (~28 ms per step), and the four output adapters' backward passes -- ~170 short launches each -- arrive
at the GPU slower than it executes them, so the adapter phase of the step ran host-bound and its streams never
overlapped (rocprofv3 trace, profiles/r01_*_v6).  Capturing the step once (every stream fork / join becomes a graph
edge) and replaying it removes the host from the step: one hipGraphLaunch per iteration.

What still changes from step to step enters through static device tensors refreshed right before each replay
(engine.HostInputs): the sampler's Dirichlet token budgets (drawn on the CPU generator exactly as in eager mode, the
reference's call sequence multimae.py:185-189) and AdamW's step-dependent scalars (lr / weight-decay schedule values
read from ``optimizer.param_groups`` at replay time, bias corrections).  The device-side noise comes from torch's
graph-safe Philox generator, so every replay draws fresh masks.

    step = StepGraph(lambda: train_step())      # train_step: zero_grad -> model -> losses -> backward -> opt.step
    for it in range(n):
        loss = step()                           # first call captures (after >= 1 eager step has run), then replays

Constraints (checked where possible): static shapes and input tensors (write new batches INTO the captured input
tensors), no host synchronisation inside the step (no .item() / float(loss)), single process per GPU; a data-parallel
all-reduce stays outside the graph (run it between a forward/backward graph and an eager optimiser step).
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from . import engine


class StepGraph:
    def __init__(self, step_fn: Callable[[], object], stream: Optional['torch.cuda.Stream'] = None):
        self.step_fn = step_fn
        self.graph: Optional['torch.cuda.CUDAGraph'] = None
        self.inputs: Optional[engine.HostInputs] = None
        self.out = None
        self.stream = stream
        self.replays = 0

    def capture(self) -> None:
        if not torch.cuda.is_available():
            raise RuntimeError('StepGraph needs a GPU (the engine has no CPU path)')
        if self.stream is None:
            self.stream = torch.cuda.Stream()
        # streams the step forks to must exist before the capture starts
        engine.side_stream_of(self.stream)
        engine.join_wgrad_streams()
        torch.cuda.synchronize()
        self.inputs = engine.HostInputs(device=torch.cuda.current_device())      # static slab allocated outside the graph's pool
        self.graph = torch.cuda.CUDAGraph()
        engine._capture = self.inputs
        try:
            with torch.cuda.graph(self.graph, stream=self.stream):
                self.out = self.step_fn()
                engine.join_wgrad_streams()          # every forked stream must be back before the capture ends
        except BaseException:
            self.graph, self.inputs = None, None
            raise
        finally:
            engine._capture = None

    def __call__(self):
        if self.graph is None:
            self.capture()
        self.inputs.refresh()
        self.graph.replay()
        self.replays += 1
        return self.out

    @property
    def n_host_inputs(self) -> int:
        return 0 if self.inputs is None else len(self.inputs.items)

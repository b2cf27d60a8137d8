"""HIP-graph replay of a single-token step.

At batch 1 every FP4 Linear is a 4-13 us kernel and a decoder layer is a chain of ~10 dependent launches: issued eagerly the
host paces the GPU (about 6 us of Python + dispatch per op on an MI355X box, 1.84 ms per Mistral-7B-shaped token), replayed
from a HIP graph the same chain takes 1.52 ms.  Every launch of this package goes to the current stream and neither allocates
outside torch's pool nor synchronises, so a step built from its layers captures as is; this helper is the few lines of
plumbing around ``torch.cuda.CUDAGraph`` (static input / output buffers, warm-up on a side stream).  The reference issues
everything on the legacy default stream (csrc/gemv_fp4_optimized.cu:266) and cannot be captured.
"""
from __future__ import annotations

from typing import Callable, Sequence

import torch


class GraphedStep:
    """``GraphedStep(fn, *example_inputs)`` captures ``fn(*inputs)`` once; calling the object copies new inputs into the
    static buffers, replays the graph and returns the static output tensor(s) (valid until the next call; clone to keep).
    Input shapes and dtypes are fixed at capture time - the single-token decode step is the intended use."""

    def __init__(self, fn: Callable, *example_inputs: torch.Tensor, warmup: int = 2):
        if not example_inputs or not all(isinstance(t, torch.Tensor) and t.is_cuda for t in example_inputs):
            raise ValueError("GraphedStep needs CUDA tensors as example inputs")
        self._static_in = [t.clone() for t in example_inputs]
        self._graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=example_inputs[0].device)
        side.wait_stream(torch.cuda.current_stream(example_inputs[0].device))
        with torch.cuda.stream(side), torch.inference_mode():
            for _ in range(max(1, warmup)):  # first calls fix compute dtypes, fill allocator pools, load code objects
                fn(*self._static_in)
            side.synchronize()
            with torch.cuda.graph(self._graph, stream=side, capture_error_mode="thread_local"):
                self._static_out = fn(*self._static_in)
        torch.cuda.current_stream(example_inputs[0].device).wait_stream(side)

    def __call__(self, *inputs: torch.Tensor):
        if len(inputs) != len(self._static_in):
            raise ValueError(f"expected {len(self._static_in)} inputs, got {len(inputs)}")
        for dst, src in zip(self._static_in, inputs):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise ValueError(f"input of shape {tuple(src.shape)} / {src.dtype} does not match the captured {tuple(dst.shape)} / {dst.dtype}")
            dst.copy_(src)
        self._graph.replay()
        return self._static_out

    @property
    def static_inputs(self) -> Sequence[torch.Tensor]:
        return self._static_in

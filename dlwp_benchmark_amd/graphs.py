"""HIP-graph replay of a backbone step (launch-bound models: U-Net, ConvLSTM, the HEALPix nets).

A step of those backbones is 25-60 small launches (3x3 convolutions on 64x64 maps take a few microseconds each); the
host cannot issue them as fast as the GPU retires them.  `GraphedStep` captures `one_step(x)` ONCE into a HIP graph
(`torch.cuda.CUDAGraph` is hipGraph on ROCm; the ctypes launches of libdlwp_hip.so go to the capturing stream like
any other) and replays it per rollout step with the input copied into a static buffer.  Shapes are static per
(model, batch, grid); a new shape captures a new graph.
"""
from typing import Callable, Dict, Tuple

import torch


class GraphedStep:
    def __init__(self, fn: Callable[[torch.Tensor], torch.Tensor]):
        self.fn = fn
        self._graphs: Dict[Tuple, Tuple] = {}

    def _capture(self, x: torch.Tensor):
        static_in = x.clone()
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side):          # warm-up off the capture: lazy plan creation, allocator growth
            for _ in range(2):
                self.fn(static_in)
        torch.cuda.current_stream(x.device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_out = self.fn(static_in)
        return graph, static_in, static_out

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        key = (tuple(x.shape), x.dtype, str(x.device))
        if key not in self._graphs:
            self._graphs[key] = self._capture(x)
        graph, static_in, static_out = self._graphs[key]
        static_in.copy_(x)
        graph.replay()
        return static_out

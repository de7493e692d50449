"""HIP-graph replay of a fixed-shape device step (inference of one input size).

The eager inference step is ~150 kernel launches of 4-150 us each; launched one by one the GPU idles between the
short ones.  Capturing the step once and replaying it removes the gaps.  Every kernel of this package is launched
on torch's current stream (centerpoly_amd._C.stream()), which inside `torch.cuda.graph` is the capture stream, so
the hand-written kernels are captured like torch's own.

Rules for `fn`: fixed shapes, inputs read from tensors that live as long as the graph (update them in place
before a replay), no host synchronisation, no data-dependent control flow; one-time work (weight permutation
caches, library algorithm search) must have happened in the warm-up calls made here before the capture."""
import torch


class GraphedStep(object):
    def __init__(self, fn, warmup=3):
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = fn()

    def __call__(self):
        self.graph.replay()
        return self.out

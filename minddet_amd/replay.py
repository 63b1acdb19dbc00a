"""One inference step captured in a HIP graph and replayed (torch.cuda.CUDAGraph is hipGraph on ROCm).

Where the path is launch-bound -- small batches: YOLOv5s at batch 8 runs ~75 launches in 1.4 ms -- replaying the captured launch
sequence removes the per-launch host cost (r01, tools/graph_try.py: 1.44 -> 1.17 ms, bit-identical outputs); at the benchmark's
batch sizes the GPU is the limiter and replay is no faster (batch 32: 2.20 vs 2.24 ms), so bench.py keeps eager launches unless
--graph is given.  Every op of the library is capturable: stream-ordered, fixed-shape outputs, no host synchronisation, scratch from
hipMallocAsync or a caller workspace.  The reference reaches the same end with MindSpore's graph mode (context.GRAPH_MODE in
centernet/eval.py:58-69); here the "graph" is the recorded launch sequence itself.
"""
import torch


class CapturedStep:
    """step = CapturedStep(model.forward, example_batch);  out = step(batch)  -- `batch` must have the example's shape and dtype; the
    returned tensors are the capture's static outputs and are overwritten by the next call (clone what must survive)."""

    def __init__(self, fn, example, warmup=2):
        if not example.is_cuda:
            raise ValueError("CapturedStep: the example batch must live on the GPU")
        self.static_in = example.clone()
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):     # warm up off the capture stream: lazy packing / allocator growth must not be captured
            for _ in range(max(1, warmup)):
                fn(self.static_in)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = fn(self.static_in)

    def __call__(self, batch):
        if batch.shape != self.static_in.shape or batch.dtype != self.static_in.dtype:
            raise ValueError(f"CapturedStep: captured for {tuple(self.static_in.shape)} {self.static_in.dtype}, got {tuple(batch.shape)} {batch.dtype}")
        if batch.data_ptr() != self.static_in.data_ptr():
            self.static_in.copy_(batch)
        self.graph.replay()
        return self.static_out

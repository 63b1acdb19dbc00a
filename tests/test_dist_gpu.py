"""The RCCL path on ONE MI355X: torch.distributed backend "nccl" (= RCCL) with a single rank -- init -> gather_detections /
gather_masks / the asynchronous form -> destroy -- and bench.py's own launcher (`--gpus 1 --spawn`: the parent makes no GPU call,
the child rank runs init_process_group("nccl"), the step loop with the overlapped all_gather, barrier, all_reduce(MAX)).
Child processes keep the collectives out of the pytest process; at most one child uses the GPU at a time."""
import json
import os
import subprocess
import sys

import pytest

from conftest import has_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs a GPU")]

_CHILD = r"""
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from minddet_amd import shard
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"), RANK="0", WORLD_SIZE="1",
                  HSA_ENABLE_IPC_MODE_LEGACY="0")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
g = torch.Generator().manual_seed(0)
d = torch.rand((4, 10, 6), generator=g).to(dev)
c = torch.tensor([3, 0, 10, 7], dtype=torch.int32, device=dev)
for b in range(4):
    d[b, int(c[b]):] = 0
m = torch.rand((4, 10, 28, 28), generator=g).to(dev)
d1, c1 = shard.gather_detections(d, c, force=True)
m1 = shard.gather_masks(m, force=True)
h = shard.gather_detections_async(d, c, masks=m, force=True)
x = torch.randn((1024, 1024), device=dev) @ torch.randn((1024, 1024), device=dev)   # compute enqueued while the gather is in flight
d2, c2, m2 = h.result()
torch.cuda.synchronize()
assert torch.equal(d1, d) and torch.equal(c1, c) and torch.equal(d2, d) and torch.equal(c2, c)
assert m1.dtype == torch.float16 and torch.equal(m1, m.to(torch.float16)) and torch.equal(m2, m1)
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.barrier(device_ids=[0])
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.5
dist.destroy_process_group()
print("RCCL_SINGLE_RANK_OK")
""" % ROOT


def _env():
    return dict({k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}, HSA_ENABLE_IPC_MODE_LEGACY="0")


def test_single_rank_rccl_gather_path():
    r = subprocess.run([sys.executable, "-c", _CHILD], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_SINGLE_RANK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_spawn_launcher_one_rank():
    """`bench.py --gpus 1 --spawn`: launcher -> one child rank over RCCL -> ONE JSON line with n_gpus 1 (tiny config, 2 steps)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn", "--steps", "2", "--warmup", "1", "--batch", "2",
           "--config", os.path.join(ROOT, "configs", "faster_rcnn", "faster_rcnn_tiny.py"), "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-4000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["config"]["global_batch"] == 2 and j["value"] > 0
    assert j["config"]["parallelism"].startswith("dp1")


def test_bench_spawn_launcher_one_rank_two_streams_two_gathers():
    """the same launcher path with graphs.SplitForward inside the rank (--streams 2) on the Mask R-CNN graph: the two halves are joined on the
    rank's compute stream before the two asynchronous all_gathers (detections; fp16 masks) are issued over RCCL"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn", "--steps", "3", "--warmup", "1", "--batch", "4", "--streams", "2",
           "--config", os.path.join(ROOT, "configs", "mask_rcnn", "mask_rcnn_tiny.py"), "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-4000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["config"]["streams"] == 2 and j["value"] > 0
    assert "serialized single-stream pass" in j["roofline"]["bracketed"] and j["roofline"]["frac"] > 0


def test_bench_refuses_more_ranks_than_devices():
    import torch

    n = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]

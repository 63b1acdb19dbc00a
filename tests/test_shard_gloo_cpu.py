"""world_size-2 gloo test of the N>1 path: contiguous image sharding + the single all_gather of padded
detections reproduces the unsharded result in input order."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from minddet_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    B, M = 6, 10
    dets_all = torch.rand((B, M, 6), generator=g)
    count_all = torch.tensor([3, 0, 10, 1, 7, 5], dtype=torch.int32)
    for b in range(B):
        dets_all[b, count_all[b]:] = 0
    lo, hi = shard.shard_range(B, rank, world)
    d, c = shard.gather_detections(dets_all[lo:hi].contiguous(), count_all[lo:hi].contiguous())
    ok = torch.equal(d, dets_all) and torch.equal(c, count_all)
    masks_all = torch.rand((B, M, 28, 28), generator=g)
    mg = shard.gather_masks(masks_all[lo:hi].contiguous())
    ok = ok and mg.dtype == torch.float16 and torch.equal(mg, masks_all.to(torch.float16))
    q.put((rank, ok, tuple(d.shape)))
    dist.destroy_process_group()


def test_all_gather_of_detections_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
    assert all(ok for _, ok, _ in res), res
    assert all(s == (6, 10, 6) for _, _, s in res)


def test_pack_roundtrip_and_shard_range():
    d = torch.rand((3, 5, 6))
    c = torch.tensor([5, 0, 2], dtype=torch.int32)
    buf = shard.pack_for_gather(d, c)
    assert buf.shape == (3, 6, 7) and buf[2, :5, 6].tolist() == [1, 1, 0, 0, 0]
    d2, c2 = shard.unpack_gathered(buf)
    assert torch.equal(d2, d) and torch.equal(c2, c)
    assert shard.shard_range(256, 3, 8) == (96, 128)
    try:
        shard.shard_range(10, 0, 4)
        assert False
    except ValueError:
        pass

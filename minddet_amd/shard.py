"""Image sharding across the GPUs of a node and the single collective of the inference path.

The reference has no inference collective (only training all-reduce through MindSpore
wrappers; its dataset sharding `num_shards/shard_id` -- centernet/src/dataset.py:411-418 -- is the same
contiguous partition used here).  Contract: global batch B -> rank r takes images
[r*B/N, (r+1)*B/N); each rank runs the full path to padded detections; ONE all_gather per batch of a
fixed-shape fp32 tensor [B_local, max_det+1, 7] = rows (x1,y1,x2,y2,score,label,valid) + one count row.
Payload is a few hundred KB per rank: latency-bound on xGMI, so it is a single fused call, never
per-image gathers.  torch.distributed backend "nccl" is RCCL on ROCm; "gloo" works for CPU tests.
"""
import torch
import torch.distributed as dist


def shard_range(global_batch, rank, world):
    """Contiguous image range of `rank` (gathered order == input order)."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} not divisible by world size {world}")
    per = global_batch // world
    return rank * per, (rank + 1) * per


def pack_for_gather(dets, count):
    """dets [B,max_det,6], count [B] -> [B, max_det+1, 7]."""
    B, M, _ = dets.shape
    buf = torch.zeros((B, M + 1, 7), dtype=torch.float32, device=dets.device)
    buf[:, :M, :6] = dets
    valid = torch.arange(M, device=dets.device).view(1, M) < count.view(B, 1)
    buf[:, :M, 6] = valid.to(torch.float32)
    buf[:, M, 0] = count.to(torch.float32)
    return buf


def unpack_gathered(buf):
    """[B_total, max_det+1, 7] -> (dets [B_total,max_det,6], count [B_total] int32)."""
    M = buf.shape[1] - 1
    return buf[:, :M, :6].contiguous(), buf[:, M, 0].to(torch.int32)


def gather_detections(dets, count, group=None, force=False):
    """All ranks end up with the detections of the whole global batch, in input order.
    `force` runs the collective even for a single rank (used to exercise the RCCL path on one GPU)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return dets, count
    world = dist.get_world_size(group)
    buf = pack_for_gather(dets, count)
    out = torch.empty((world * buf.shape[0],) + tuple(buf.shape[1:]), dtype=buf.dtype, device=buf.device)
    dist.all_gather_into_tensor(out, buf, group=group)
    return unpack_gathered(out)


def gather_masks(masks, group=None, force=False):
    """Mask R-CNN: the per-detection 28x28 masks of the whole global batch as fp16 [B_total, max_det, S, S] ("masks
    gathered as 28x28 fp16 per det ... never full-resolution"), a second fixed-shape all_gather next to gather_detections."""
    m16 = masks.to(torch.float16).contiguous()
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return m16
    world = dist.get_world_size(group)
    out = torch.empty((world * m16.shape[0],) + tuple(m16.shape[1:]), dtype=m16.dtype, device=m16.device)
    dist.all_gather_into_tensor(out, m16, group=group)
    return out



class GatherHandle:
    """An all_gather in flight.  torch.distributed runs an async collective on the backend's own stream behind an event of the
    issuing stream (RCCL: ProcessGroupNCCL's internal stream), so the caller's compute stream goes on with the next batch;
    result() makes the CURRENT stream wait for the collective (no host block on RCCL) and unpacks.  The handle keeps the send /
    receive buffers alive until then."""

    def __init__(self, works, det_out, mask_out, local):
        self._works, self._det_out, self._mask_out, self._local = works, det_out, mask_out, local
        self._done = None

    def result(self):
        if self._done is None:
            for w in self._works:
                w.wait()
            if self._det_out is None:
                self._done = self._local
            else:
                d, c = unpack_gathered(self._det_out)
                self._done = (d, c) if self._mask_out is None else (d, c, self._mask_out)
            self._works = ()
        return self._done


def gather_detections_async(dets, count, masks=None, group=None, force=False):
    """gather_detections (+ gather_masks when `masks` is given) issued with async_op=True: returns a GatherHandle at once.  SURVEY
    8e: "issue it on a side stream overlapped with the next batch's backbone" -- the caller enqueues the next step's forward pass
    and calls handle.result() afterwards."""
    local = (dets, count) if masks is None else (dets, count, masks.to(torch.float16).contiguous())
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return GatherHandle((), None, None, local)
    world = dist.get_world_size(group)
    buf = pack_for_gather(dets, count)
    out = torch.empty((world * buf.shape[0],) + tuple(buf.shape[1:]), dtype=buf.dtype, device=buf.device)
    works = [dist.all_gather_into_tensor(out, buf, group=group, async_op=True)]
    m_out = None
    keep = [buf]
    if masks is not None:
        m16 = local[2]
        m_out = torch.empty((world * m16.shape[0],) + tuple(m16.shape[1:]), dtype=m16.dtype, device=m16.device)
        works.append(dist.all_gather_into_tensor(m_out, m16, group=group, async_op=True))
        keep.append(m16)
    h = GatherHandle(works, out, m_out, local)
    h._keep = keep
    return h

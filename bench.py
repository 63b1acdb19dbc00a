"""bench.py -- images/sec of the Faster R-CNN R50-FPN 800x1344 bf16 inference hot path on N MI355X.

    python bench.py --gpus N --steps K --warmup W [--batch B]

--gpus is authoritative.  Under a launcher (WORLD_SIZE in the environment, e.g. python -m torch.distributed.run --nnodes=1
--nproc-per-node N ... bench.py --gpus N ...) WORLD_SIZE must equal N, else the run stops with exit code 2.  Without a launcher
and N > 1 (or with --spawn) bench.py starts the N ranks itself: the parent counts the devices (no GPU call), spawns one child
process per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 set, relays rank 0's ONE JSON line and exits with the
worst child exit code.  Every rank asserts that the RCCL world size is N.

One step = one pass of the whole path (backbone + FPN + RPN + proposals + RoIAlign + box head +
class-wise NMS + packing) over one batch of synthetic COCO-shaped images already resident in HBM,
followed, when N > 1, by the single all_gather of the padded detections.  Images are
sharded across ranks (weak scaling: B images per GPU); value = N*B*K / max-over-ranks time.

The JSON line also carries
  roofline     -- the DOMINANT conv kernel by time (md_conv2d_last_kernel attributes every launch; each is bracketed by
                  HIP events on the launch stream) against the roofline that binds its launches in aggregate
                  (algorithmic flops / 2.5 PFLOP/s dense bf16 or algorithmic bytes / 8 TB/s), its PMC traffic from
                  profiles/rNN_conv_traffic.json of the newest round (null when that file was measured on other kernel sources), and under "all_conv" the same per kernel and for the whole conv/FC set
                  (incl. frac_of_layerwise_roofline = sum of per-launch max(flops/peak, bytes/peak) / measured time).
  cpu_baseline -- the oracle's plain fp32 torch-CPU restatement of the same graph (oracle/nets.py) timed on this
                  host's cores on a bounded sample (rank 0, N=1 only).
The batch is resident in HBM in the model's input layout (zero-bordered 4-channel NHWC for the fused stem) before the timed
region; --from-uint8 also times md_image_preprocess from a resident uint8 batch.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (guide: ~2.5 PF)
PEAK_HBM_BPS = 8.0e12      # HBM3E spec peak (guide: 8 TB/s, ~6.3 achievable)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=120,
                    help="images per GPU per step (r02 same-box, no instrumentation: 60 -> 2068, 90 -> 2090, 120 -> 2107 images/s; r01: 32 -> 1640, "
                         "48 -> 1797, 60 -> 1870; above 60 the P2-level tensors pass 2 GiB and md_conv2d runs those layers as two image chunks)")
    ap.add_argument("--bracket", choices=["dominant", "all"], default="dominant",
                    help="which conv launches of the TIMED region are bracketed by HIP events: only the dominant kernel's (default: "
                         "the whole-set table then comes from the last warmup step, where every launch is bracketed; 2 events per "
                         "launch cost ~1.5 %% of the step when all 68 launches carry them) or all of them")
    ap.add_argument("--graph", action="store_true",
                    help="capture the step in a HIP graph and replay it (minddet_amd/replay.py): pays off only where the path is "
                         "launch-bound (small batches); implies --no-roofline (events cannot bracket launches inside a graph)")
    ap.add_argument("--config", default=os.path.join(ROOT, "configs", "faster_rcnn", "faster_rcnn_r50_fpn.py"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--from-uint8", action="store_true",
                    help="start every step from a resident uint8 batch: md_image_preprocess (warp + normalise + layout) is timed too")
    ap.add_argument("--dump-convs", default=None, help="write per-launch conv timings (json) to this path")
    ap.add_argument("--spawn", action="store_true",
                    help="start the --gpus ranks from this process even for N = 1 (the RCCL path with one rank)")
    ap.add_argument("--no-from-uint8", action="store_true", help="skip the extra from-uint8 pass (from_uint8 in the line)")
    ap.add_argument("--paste-masks", action="store_true",
                    help="Mask R-CNN configs: paste the 28x28 masks into the image inside the step (md_paste_masks -> [B,max_det,H,W/32] bit masks)")
    ap.add_argument("--streams", type=int, default=None,
                    help="run the batch as this many equal parts, each on its own HIP stream (graphs.SplitForward; default: the config's "
                         "test_cfg.streams, 1 when absent).  With more than one stream the roofline block is measured in a serialized "
                         "single-stream pass AFTER the timed region (a launch's duration under overlap includes the other stream's share of the CUs)")
    ap.add_argument("--no-zero-operands", action="store_true",
                    help="skip the zero-operand replay of the dominant kernel (roofline.zero_operands)")
    return ap.parse_args(argv)


def launch_ranks(args, argv):
    """--gpus N without a launcher: start N ranks as child processes BEFORE this process makes any GPU call (device_count() does not
    initialise the GPU on this image), one rank per GPU, rendezvous on 127.0.0.1; relay rank 0's JSON line."""
    import socket
    import subprocess

    import torch

    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) are visible\n")
        return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    child_argv = [a for a in argv if a != "--spawn"]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", MD_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + child_argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's pipe is drained by a thread (it can never fill), every child is polled: when one rank dies (OOM, bad LOCAL_RANK, RCCL
    # init failure) the others would sit in the rendezvous or a collective until torch.distributed's timeout -- they are terminated
    # instead (fresh child processes of this one; nothing is re-exec'ed), reaped, and the exit codes reported.  MD_BENCH_DEADLINE_S
    # bounds the whole run.
    import threading

    out_buf = []
    reader = threading.Thread(target=lambda: out_buf.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("MD_BENCH_DEADLINE_S", "3000"))
    why, fail_rc = None, None
    while True:
        rcs = [p.poll() for p in procs]
        if all(rc is not None for rc in rcs):
            break
        bad = [i for i, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad or time.time() > deadline:
            why = f"rank {bad[0]} exited with code {rcs[bad[0]]}" if bad else "deadline passed"
            fail_rc = abs(rcs[bad[0]]) if bad else 124   # the code of the rank that died by itself, not of the siblings ended here
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_kill = time.time() + 10.0
            while time.time() < t_kill and any(p.poll() is None for p in procs):
                time.sleep(0.1)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.1)
    rcs = [p.wait() for p in procs]
    reader.join(10.0)
    out0 = (out_buf[0] if out_buf else b"").decode()
    lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
    if why is not None or any(rcs) or len(lines) != 1:
        sys.stderr.write(f"bench.py: {why + '; ' if why else ''}rank exit codes {rcs}, {len(lines)} JSON line(s) from rank 0\n{out0}")
        return fail_rc if fail_rc else max([abs(c) for c in rcs if c] + [1])
    line = json.loads(lines[0])
    if line.get("n_gpus") != args.gpus:
        sys.stderr.write(f"bench.py: rank 0 reports n_gpus {line.get('n_gpus')}, expected {args.gpus}\n")
        return 1
    print(lines[0])
    return 0


def make_step(model_forward, images, use_dist, gatherer=None, preprocess=None):
    """One step of the hot path: [pre-process ->] forward -> (N > 1) all_gather of the padded detections.  The gather of step i is
    issued asynchronously (RCCL runs it on its own stream behind an event of the compute stream) and joined when step i + 1 has
    enqueued its forward pass, so the collective overlaps the next batch's backbone; `finish()` joins the last one."""
    pending = {"h": None, "out": None}

    def step():
        x = images if preprocess is None else preprocess()
        out = model_forward(x)
        if not use_dist:
            pending["out"] = (out[0], out[1])
            return pending["out"]
        prev = pending["h"]
        pending["h"] = gatherer(out)
        if prev is not None:
            pending["out"] = prev.result()
        return pending["out"]

    def finish():
        if pending["h"] is not None:
            pending["out"] = pending["h"].result()
            pending["h"] = None
        return pending["out"]

    return step, finish


HOST_ENQUEUE = {"s": None}   # run_timed: host seconds spent enqueuing the K timed steps (before the closing device sync)


def run_timed(step, finish, steps, warmup, use_dist, sync, barrier, all_reduce_max, before_timed=None, on_warmup=None):
    """W untimed warmup steps, then EXACTLY K steps bracketed by barrier + device sync on both sides; returns the MAX over ranks
    of the elapsed seconds (driver contract).  `on_warmup(i, step)` may replace the plain call of warmup step i."""
    for w_i in range(warmup):
        if on_warmup is not None:
            on_warmup(w_i, step)
        else:
            step()
    finish()
    if before_timed is not None:
        before_timed()
    sync()
    if use_dist:
        barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    finish()
    HOST_ENQUEUE["s"] = time.perf_counter() - t0   # the host is done enqueuing here; the device still runs (nothing above blocks on it)
    sync()
    if use_dist:
        barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        dt = all_reduce_max(dt)
    return dt


def kernel_source_hash():
    """sha256 over the conv-family kernel sources: ties profiles/*_conv_traffic.json (separate rocprofv3 --pmc passes) to the build it
    was measured on."""
    import hashlib

    h = hashlib.sha256()
    for f in ("conv.hip", "bottleneck.hip", "c3pair.hip", "stem.hip", "stemconv.hip", "aot.h"):
        with open(os.path.join(ROOT, "minddet_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def reference_ops_baseline(dev):
    """the reference's OWN compiled operator (oracle/_ref/libref_nms.so = centerpoint/det3d_ms/ops/iou-bev-nms-org.cpp
    built by oracle/Makefile) timed single-thread on this host beside the device ops that replace it, same inputs, outside the
    timed region.  None when oracle/_ref was not shipped."""
    import numpy as np
    import torch

    import oracle
    from minddet_amd import det_ops

    if oracle.ref_lib() is None:
        return None
    rng = np.random.default_rng(0)
    b = np.zeros((1000, 7), np.float32)
    b[:, :2] = rng.uniform(-50, 50, (1000, 2))
    b[:, 3:6] = rng.uniform(1, 5, (1000, 3))
    b[:, 6] = rng.uniform(-np.pi, np.pi, 1000)

    def cpu_time(fn, budget=2.0):
        fn()
        t0, n = time.perf_counter(), 0
        while n < 3 or time.perf_counter() - t0 < budget:
            fn()
            n += 1
        return (time.perf_counter() - t0) / n, n

    def gpu_time(fn, reps=50):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    bd = torch.from_numpy(b).to(dev)
    nms = det_ops.NMS()
    t_nms_cpu, n1 = cpu_time(lambda: oracle.ref_boxes_iou_nms_cpu(b, 0.2))
    t_nms_gpu = gpu_time(lambda: nms(bd, 0.2))
    k_ref, n_ref = oracle.ref_boxes_iou_nms_cpu(b, 0.2)
    k_dev, n_dev = nms(bd, 0.2)
    same = int(n_dev) == n_ref and bool((k_dev.cpu().numpy() == k_ref).all())
    iou = det_ops.BoxesIouBevGpu()
    t_iou_cpu, n2 = cpu_time(lambda: oracle.ref_boxes_iou_bev_cpu(b, b[:200]))
    t_iou_gpu = gpu_time(lambda: iou(bd, bd[:200]))
    return {"kind": "reference", "cores": 1,
            "boxes_iou_nms": {"cpu_ms": round(t_nms_cpu * 1e3, 3), "device_us": round(t_nms_gpu * 1e6, 1), "cpu_calls": n1,
                              "sample": "1000 rotated boxes, thr 0.2: boxes_iou_nms_cpu (iou-bev-nms-org.cpp:237-283) vs boxes_iou_nms_gpu",
                              "keep_lists_identical": same},
            "boxes_iou_bev": {"cpu_ms": round(t_iou_cpu * 1e3, 3), "device_us": round(t_iou_gpu * 1e6, 1), "cpu_calls": n2,
                              "sample": "1000 x 200 rotated IoU matrix: boxes_iou_bev_cpu (iou-bev-nms-org.cpp:227-234) vs BoxesIouBevGpu"}}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus < 1:
        sys.stderr.write("bench.py: --gpus must be >= 1\n")
        return 2
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and (args.gpus > 1 or args.spawn):
        return launch_ranks(args, argv)
    if env_world is not None and int(env_world) != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}; launch with\n  python -m torch.distributed.run "
                         f"--nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus "
                         f"{args.gpus} ...\nor run `python bench.py --gpus {args.gpus}` without WORLD_SIZE (it spawns the ranks itself)\n")
        return 2

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.device_count() <= local_rank:
        sys.stderr.write(f"bench.py: rank {rank} wants cuda:{local_rank}, {torch.cuda.device_count()} device(s) visible\n")
        return 2
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # the RCCL path runs whenever a launcher (torchrun or launch_ranks above) started this rank, also with ONE rank;
    # MD_FORCE_DIST=1 does the same in-process
    use_dist = env_world is not None or os.environ.get("MD_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
        if dist.get_world_size() != args.gpus or dist.get_backend() != "nccl":
            sys.stderr.write(f"bench.py: RCCL world size {dist.get_world_size()} (backend {dist.get_backend()}) != --gpus {args.gpus}\n")
            return 2

    from minddet.models import Config, build_detector
    from minddet_amd import _lib, nn_ops
    from minddet_amd.data import synthetic_images
    from minddet_amd.shard import gather_detections_async

    if not os.path.exists(args.config) and os.path.exists(os.path.join(ROOT, args.config)):
        args.config = os.path.join(ROOT, args.config)   # a repo-relative path from another working directory (rocprofv3 runs from /tmp)
    cfg = Config.fromfile(args.config)
    model = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(dev)
    if args.paste_masks and hasattr(model, "paste"):
        model.paste = True
    H, W = cfg.data.input_hw
    B = args.batch
    images = synthetic_images(B, H, W, seed=20240317 + rank, device=dev)
    has_stem = getattr(getattr(model, "backbone", None), "stem", None) is not None or getattr(model, "stem", None) is not None
    if nn_ops.stem_layout_ok(H, W) and os.environ.get("MD_STEM_LAYOUT", "1") == "1" and has_stem:
        # the batch is resident in HBM in the model's input layout before the timed region starts: zero-bordered
        # 4-channel NHWC (md_stem_pool for the ResNet stems, md_stem_conv for the YOLO stems); MD_STEM_LAYOUT=0 keeps the 8-channel layout for A/B
        images = nn_ops.to_stem_layout(images)

    # ---- per-conv event instrumentation (roofline of the dominant kernel)
    records = []
    orig_conv2d = nn_ops.conv2d
    last_kernel = _lib.lib().md_conv2d_last_kernel
    launch_count = _lib.lib().md_conv2d_launch_count   # a call on a batch past the 2 GiB chunk limit launches once per image chunk
    launch_count.restype = ctypes.c_longlong
    KNAMES = {1: "conv_pingpong_kernel", 2: "conv_igemm_kernel<128x128>", 3: "conv_igemm_kernel<small cout>",
              4: "conv_igemm_kernel<generic K>", 5: "conv3x3_halo_kernel", 6: "conv variant", 7: "bottleneck_fused_kernel", 8: "conv1x1_stream_kernel", 9: "c3pair_kernel"}

    bracket = {"calls": None, "idx": 0}    # calls: None = bracket every launch, else the per-step call indices to bracket

    def _skip():
        i = bracket["idx"]
        bracket["idx"] = i + 1
        return bracket["calls"] is not None and i not in bracket["calls"]

    def timed_conv2d(x, pc, residual=None, relu=None, out=None, variant=None, c_off=0, res_upsample=False, **kw):
        if _skip():
            return orig_conv2d(x, pc, residual=residual, relu=relu, out=out, variant=variant, c_off=c_off, res_upsample=res_upsample, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        l0 = launch_count()
        e0.record()
        y = orig_conv2d(x, pc, residual=residual, relu=relu, out=out, variant=variant, c_off=c_off, res_upsample=res_upsample, **kw)
        e1.record()
        n, ho, wo, _ = y.shape
        # algorithmic bytes: the channels the layer reads / writes (operands may be channel slices of wider tensors)
        res_elems = 0 if residual is None else (residual.numel() if res_upsample else n * ho * wo * pc.cout)
        byts = 2.0 * (x.shape[0] * x.shape[1] * x.shape[2] * pc.cin + n * ho * wo * pc.cout + pc.cout * pc.cin_real * pc.kh * pc.kw + res_elems)
        records.append((e0, e1, 2.0 * n * ho * wo * pc.cout * pc.cin_real * pc.kh * pc.kw, tuple(x.shape[:3]) + (pc.cin,), pc.cout, pc.kh, byts,
                        last_kernel(), pc, None, launch_count() - l0))
        return y

    orig_conv2d_head = nn_ops.conv2d_head

    def timed_conv2d_head(x, pc, pc2, variant=None):
        if _skip():
            return orig_conv2d_head(x, pc, pc2, variant=variant)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        l0 = launch_count()
        e0.record()
        y = orig_conv2d_head(x, pc, pc2, variant=variant)
        e1.record()
        n, ho, wo, _ = y.shape
        fl = 2.0 * n * ho * wo * (pc.cout * pc.cin_real * pc.kh * pc.kw + pc2.cout * pc2.cin_real)
        byts = 2.0 * (x.numel() + y.numel() + pc.cout * pc.cin_real * pc.kh * pc.kw + pc2.cout * pc2.cin_real)
        records.append((e0, e1, fl, tuple(x.shape), pc.cout, pc.kh, byts, last_kernel(), pc, pc2, launch_count() - l0))
        return y

    timed_extra = {}      # other instrumented entry points of nn_ops (name -> wrapper), e.g. the fused bottleneck block
    if hasattr(nn_ops, "bottleneck"):
        orig_bottleneck = nn_ops.bottleneck

        def timed_bottleneck(x, blk, residual=None, **kw):
            if _skip():
                return orig_bottleneck(x, blk, residual=residual, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            l0 = launch_count()
            e0.record()
            y = orig_bottleneck(x, blk, residual=residual, **kw)
            e1.record()
            fl, byts = blk.flops_bytes(x.shape[0], x.shape[1], x.shape[2], with_residual_tensor=residual is not None)
            records.append((e0, e1, fl, tuple(x.shape), blk.cout, 3, byts, last_kernel(), None, None, launch_count() - l0))
            return y

        timed_extra["bottleneck"] = (orig_bottleneck, timed_bottleneck)

    if hasattr(nn_ops, "conv1x1_dual"):
        orig_dual = nn_ops.conv1x1_dual

        def timed_dual(xa, xb, pk, **kw):
            if _skip():
                return orig_dual(xa, xb, pk, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            l0 = launch_count()
            e0.record()
            y = orig_dual(xa, xb, pk, **kw)
            e1.record()
            fl, byts = pk.flops_bytes(xa.shape[0], xa.shape[1], xa.shape[2])
            records.append((e0, e1, fl, tuple(xa.shape[:3]) + (pk.ca + pk.cb,), pk.cout, 1, byts, last_kernel(), None, None, launch_count() - l0))
            return y

        timed_extra["conv1x1_dual"] = (orig_dual, timed_dual)

    if hasattr(nn_ops, "c3_pair"):
        orig_c3 = nn_ops.c3_pair

        def timed_c3_pair(x, pk, out, *a_, **kw):
            if _skip():
                return orig_c3(x, pk, out, *a_, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            l0 = launch_count()
            e0.record()
            y = orig_c3(x, pk, out, *a_, **kw)
            e1.record()
            # algorithmic work of the pair as ONE op (x slice read once, y slice written once; the pass-through copy is not counted)
            fl, byts = pk.flops_bytes(x.shape[0], x.shape[1], x.shape[2])
            records.append((e0, e1, fl, tuple(x.shape[:3]) + (pk.c,), pk.c, 3, byts, last_kernel(), None, None, launch_count() - l0))
            return y

        timed_extra["c3_pair"] = (orig_c3, timed_c3_pair)

    def instrument_on():
        nn_ops.conv2d, nn_ops.conv2d_head = timed_conv2d, timed_conv2d_head
        for k_, (_, tw) in timed_extra.items():
            setattr(nn_ops, k_, tw)

    def instrument_off():
        nn_ops.conv2d, nn_ops.conv2d_head = orig_conv2d, orig_conv2d_head
        for k_, (ow, _) in timed_extra.items():
            setattr(nn_ops, k_, ow)

    images_u8 = pre_mat = None
    MEAN, STD = (0.408, 0.447, 0.470), (0.289, 0.274, 0.278)

    def make_u8():
        g_u8 = torch.Generator(device="cpu").manual_seed(20240317 + rank)
        u8 = torch.randint(0, 256, (B, H, W, 3), generator=g_u8, dtype=torch.uint8).to(dev)
        return u8, torch.tensor([1.0, 0, 0, 0, 1.0, 0], dtype=torch.float32, device=dev).repeat(B, 1).contiguous()

    def pre_part(u8_, mat_):   # uint8 batch (or a part of it) -> the model's input layout
        return nn_ops.image_preprocess(u8_, mat_, MEAN, STD, (H, W), stem_layout=images.shape[3] == 4)

    preprocess = None
    if args.from_uint8:
        images_u8, pre_mat = make_u8()
        preprocess = lambda: (images_u8, pre_mat)  # noqa: E731   (the step's input is the uint8 batch: forward() pre-processes it)

    n_streams = args.streams if args.streams is not None else int(getattr(model, "streams", 1))
    if n_streams > 1 and (B % n_streams or not hasattr(model, "forward_split")):
        if args.streams is not None:   # asked for explicitly: an error
            sys.stderr.write(f"bench.py: --streams {n_streams} needs a batch divisible by it and a detector with forward_split\n")
            return 2
        n_streams = 1                  # the config's default cannot split this batch (SplitForward itself would fall back): one stream, reported as such
    if args.graph:
        n_streams = 1                  # the captured step is a single-stream model.forward: say so in config.streams
    splitter = None
    if n_streams > 1:
        from minddet_amd.graphs import SplitForward

        splitter = model.forward_split if model.forward_split.n == n_streams else SplitForward(model, n_streams)
    serial = {"on": False}   # streams > 1: the bracketed passes (roofline) run the whole batch on one stream

    def forward(x):
        bracket["idx"] = 0
        from_u8_in = isinstance(x, tuple)           # (uint8 batch, affine matrices): md_image_preprocess is part of the step
        if splitter is not None and not serial["on"]:
            return splitter(x, prepare=pre_part if from_u8_in else None)   # each part pre-processed on its own stream
        return model.forward(pre_part(*x) if from_u8_in else x)

    def gatherer(out):   # Mask R-CNN: the 28x28 masks travel as fp16 in a second fixed-shape all_gather
        return gather_detections_async(out[0], out[1], masks=out[2] if len(out) > 2 else None, force=True)

    step, finish = make_step(forward, images, use_dist, gatherer, preprocess)

    if args.graph:
        if use_dist or images_u8 is not None:
            raise SystemExit("bench.py --graph: single-GPU, pre-processed input only")
        from minddet_amd.replay import CapturedStep

        args.no_roofline = True
        captured = CapturedStep(lambda xx: tuple(model.forward(xx))[:2], images)
        step, finish = (lambda: captured(images)), (lambda: None)   # noqa: E731
    instrument = not args.no_roofline
    if splitter is not None and instrument and (args.bracket != "dominant" or args.dump_convs or args.warmup < 1):
        raise SystemExit("bench.py: with --streams > 1 the roofline pass is --bracket dominant, --warmup >= 1, no --dump-convs (use --streams 1 for those)")
    survey = []            # records of ONE fully bracketed step (the last warmup step): the whole-set table in --bracket dominant
    dominant_only = instrument and args.bracket == "dominant" and args.warmup >= 1 and not args.dump_convs

    def on_warmup(w_i, step_):
        if dominant_only and w_i == args.warmup - 1:
            instrument_on()
            serial["on"] = True
            step_()
            serial["on"] = False
            instrument_off()
            torch.cuda.synchronize()
            survey.extend(records)
            del records[:]
        else:
            step_()

    def before_timed():
        if dominant_only and survey:
            t_k = {}
            for r in survey:
                t_k[r[7]] = t_k.get(r[7], 0.0) + r[0].elapsed_time(r[1])
            dom_id = max(t_k, key=t_k.get)
            bracket["calls"] = frozenset(i for i, r in enumerate(survey) if r[7] == dom_id)   # dispatch is deterministic per call site
        if instrument and splitter is None:
            instrument_on()

    def all_reduce_max(v):
        t = torch.tensor([v], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    dt = run_timed(step, finish, args.steps, args.warmup, use_dist, torch.cuda.synchronize,
                   lambda: dist.barrier(device_ids=[local_rank]), all_reduce_max, before_timed, on_warmup)
    instrument_off()
    one_stream = None
    if instrument and splitter is not None and survey:
        # two streams: the dominant kernel's launches are bracketed in a serialized pass of the same steps on ONE stream, outside `dt`
        serial["on"] = True
        torch.cuda.synchronize()
        t_s0 = time.perf_counter()
        instrument_on()
        for _ in range(max(args.steps, 1)):
            step()
        finish()
        instrument_off()
        torch.cuda.synchronize()
        dt_s = time.perf_counter() - t_s0
        serial["on"] = False
        one_stream = {"ms_per_step": round(dt_s / max(args.steps, 1) * 1e3, 3), "value": round(B * max(args.steps, 1) / dt_s, 2), "steps": max(args.steps, 1),
                      "what": "the serialized pass the roofline block is measured in: the same steps with the whole batch on ONE stream "
                              "(events around the dominant kernel's launches), after the timed region"}
    survey = survey or None
    host_enqueue_ms = None if HOST_ENQUEUE["s"] is None else HOST_ENQUEUE["s"] / max(args.steps, 1) * 1e3
    # ... and of ONE step enqueued into an empty queue (device idle before, no synchronisation until the host is done): the pure cost of
    # Python + ctypes + launches, free of any back-pressure of a full queue that the in-run figure may contain
    torch.cuda.synchronize()
    t_h0 = time.perf_counter()
    step()
    host_single_ms = (time.perf_counter() - t_h0) * 1e3
    finish()
    torch.cuda.synchronize()

    # the end-to-end figure from a resident uint8 batch (md_image_preprocess inside the step), next to the resident-layout one
    from_u8 = None
    if not args.from_uint8 and not args.no_from_uint8 and not args.graph and rank == 0 and not use_dist and hasattr(nn_ops, "image_preprocess"):
        u8, mat = make_u8()
        step2, finish2 = make_step(forward, images, False, None, lambda: (u8, mat))
        k2 = max(2, min(args.steps, 5))
        dt2 = run_timed(step2, finish2, k2, 1, False, torch.cuda.synchronize, None, None)
        from_u8 = {"ms_per_step": round(dt2 / k2 * 1e3, 3), "value": round(B * k2 / dt2, 2), "steps": k2,
                   "what": "the same step started from a resident uint8 [B,H,W,3] batch: md_image_preprocess (warp + normalise + "
                           "layout) inside the timed region"}
        del u8, mat

    # the same steps with the batch as two halves on two HIP streams (graphs.SplitForward), when the config runs on one: a secondary figure --
    # `value` stays the one-stream run whose launches the roofline block brackets
    two_streams = None
    if (splitter is None and n_streams == 1 and B % 2 == 0 and hasattr(model, "forward_split") and rank == 0 and not use_dist and not args.graph
            and not args.no_from_uint8 and images_u8 is None):
        from minddet_amd.graphs import SplitForward

        sp2 = SplitForward(model, 2)
        step3, finish3 = make_step(lambda xx: sp2(xx), images, False, None, None)
        k3 = max(2, min(args.steps, 10))
        dt3 = run_timed(step3, finish3, k3, 2, False, torch.cuda.synchronize, None, None)
        two_streams = {"ms_per_step": round(dt3 / k3 * 1e3, 3), "value": round(B * k3 / dt3, 2), "steps": k3,
                       "what": "the same step with the batch as two halves on two HIP streams (test_cfg.streams = 2 would make it the default): "
                               "bit-identical outputs, the other half's launches fill each launch's last partial round of workgroups"}

    roofline = None
    if instrument and records:
        # whole conv/FC set: every launch of the timed region (--bracket all) or of the fully bracketed last warmup step
        all_recs, all_steps = (survey, 1) if survey else (records, max(args.steps, 1))
        ms = [e0.elapsed_time(e1) for e0, e1, *_ in all_recs]
        tot_ms = sum(ms)
        tot_fl = sum(r[2] for r in all_recs)
        n_all = sum(max(r[10], 1) for r in all_recs)   # kernel launches (a chunked call counts once per chunk)
        # per kernel; the DOMINANT one (most time) fills the contract fields, the whole conv/FC set goes to "all_conv"
        per_k = {}
        for r, t_ms in zip(all_recs, ms):
            d = per_k.setdefault(r[7], [0.0, 0.0, 0.0, 0])
            d[0] += t_ms; d[1] += r[2]; d[2] += r[6]; d[3] += max(r[10], 1)
        dom = max(per_k, key=lambda k_: per_k[k_][0])
        dom_share = per_k[dom][0] / tot_ms
        # the dominant kernel's figures always come from the TIMED region's events
        d_ms = d_fl = d_by = 0.0
        d_n = 0
        for r in records:
            if r[7] == dom:
                d_ms += r[0].elapsed_time(r[1]); d_fl += r[2]; d_by += r[6]; d_n += max(r[10], 1)   # kernel launches, not calls
        if d_n == 0:
            raise RuntimeError("bench: the dominant kernel of the bracketed warmup step was not launched in the timed region")
        # which roofline binds the dominant kernel's launches in aggregate: HBM (algorithmic bytes / 8 TB/s) or MFMA
        # (algorithmic flops / 2.5 PFLOP/s dense bf16)
        hbm_bound = d_by / PEAK_HBM_BPS > d_fl / (PEAK_BF16_TFLOPS * 1e12)
        ach = d_by / (d_ms * 1e-3) / 1e9 if hbm_bound else d_fl / (d_ms * 1e-3) / 1e12
        peak = PEAK_HBM_BPS / 1e9 if hbm_bound else PEAK_BF16_TFLOPS
        # layer-wise roofline: each launch is bounded by max(flops / MFMA peak, algorithmic bytes / HBM peak)
        t_roof_ms = sum(max(r[2] / (PEAK_BF16_TFLOPS * 1e12), r[6] / PEAK_HBM_BPS) for r in all_recs) * 1e3
        traffic = all_traffic = None
        traffic_note = "no PMC file for this library build"
        # the newest round's PMC file for this config (profiles/rNN_<...>conv_traffic.json)
        import glob

        suffix = {"FasterRCNN": "conv_traffic.json", "YOLOv5": "yolov5s_conv_traffic.json", "YOLOv8": "yolov8l_conv_traffic.json",
                  "MaskRCNN": "maskrcnn_conv_traffic.json"}.get(type(model).__name__, "conv_traffic.json")
        cands = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
        tp = cands[-1] if cands else os.path.join(ROOT, "profiles", "r04_" + suffix)
        PMC_PREFIX = {1: "conv_pingpong_kernel", 2: "conv_igemm_kernel<256, 2, 2, 2, 2, 2,", 3: "conv_igemm_kernel<256, 1, 4,", 5: "conv3x3_halo_kernel",
                      7: "bottleneck64_kernel", 8: "conv1x1_stream_kernel"}  # all instantiations of the kernel
        if os.path.exists(tp):  # PMC passes are separate rocprofv3 runs (tools/pmc_traffic.py); same config only
            tj = json.load(open(tp))
            # the PMC passes describe ONE build of the kernels: the file records the hash of the conv sources it was measured on
            # (tools/pmc_traffic.py), and a line from any other build carries traffic = null instead of a stale figure
            same_build = tj.get("kernel_source_sha256") == kernel_source_hash()
            traffic_note = (f"rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/{os.path.basename(tp)}" if same_build else
                            f"profiles/{os.path.basename(tp)} was measured on other kernel sources: not reported")
            if same_build and tj.get("batch_per_gpu") == B and type(model).__name__ in ("FasterRCNN", "YOLOv5", "YOLOv8", "MaskRCNN"):
                all_traffic = round(tj["hbm_bytes_per_launch"] / 1e6, 2)
                pmc_rows = [v for k_, v in tj.get("by_kernel", {}).items() if dom in PMC_PREFIX and k_.startswith(PMC_PREFIX[dom])]
                if pmc_rows:
                    traffic = round(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in pmc_rows) / sum(v["launches"] for v in pmc_rows) / 1e6, 2)
        roofline = {"bound": "hbm" if hbm_bound else "mfma",
                    "kernel": KNAMES.get(dom, str(dom)) + " (dominant kernel: %.0f %% of the conv/FC time)" % (100 * dom_share),
                    "bracketed": ("HIP events around this kernel's launches in the timed region" if splitter is None else
                                  f"HIP events around this kernel's launches in a serialized single-stream pass of {args.steps} steps AFTER the "
                                  f"timed region (the timed region overlaps {n_streams} streams: a launch's duration there includes the other "
                                  "stream's share of the CUs)") + (
                        "; all_conv: every launch of the last warmup step" + (" (one stream)" if splitter is not None else "")
                        if survey else "; all_conv: every launch of the timed region"),
                    "achieved": round(ach, 2), "peak": peak, "unit": "GB/s" if hbm_bound else "TFLOP/s", "frac": round(ach / peak, 4),
                    "traffic": traffic, "traffic_unit": "MB per launch (" + traffic_note + ")",
                    "achieved_tflops": round(d_fl / (d_ms * 1e-3) / 1e12, 2),
                    "algorithmic_mb_per_launch": round(d_by / d_n / 1e6, 2),
                    "algorithmic_gflop_per_launch": round(d_fl / d_n / 1e9, 1),
                    "launches_per_step": d_n // max(args.steps, 1),
                    "avg_launch_us": round(d_ms * 1e3 / d_n, 2),
                    "all_conv": {"kernels": {KNAMES.get(k_, str(k_)): {"ms_per_step": round(v[0] / all_steps, 3),
                                                                       "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 1),
                                                                       "launches_per_step": v[3] // all_steps}
                                             for k_, v in sorted(per_k.items())},
                                 "achieved": round(tot_fl / (tot_ms * 1e-3) / 1e12, 2),
                                 "frac": round(tot_fl / (tot_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                                 "frac_of_layerwise_roofline": round(t_roof_ms / tot_ms, 4),
                                 "traffic_mb_per_launch": all_traffic,
                                 "algorithmic_mb_per_launch": round(sum(r[6] for r in all_recs) / n_all / 1e6, 2),
                                 "launches_per_step": n_all // all_steps,
                                 "avg_launch_us": round(tot_ms * 1e3 / n_all, 2),
                                 "conv_ms_per_step": round(tot_ms / all_steps, 3),
                                 "algorithmic_gflop_per_step": round(tot_fl / all_steps / 1e9, 1)}}

    # ---- the dominant kernel's heaviest launch replayed on random and on all-zero operands (same instruction stream, same HBM
    # traffic): the MFMA layers of this chip run at the clock the operands' bit toggling leaves (MI355X_MICROARCH.md, DVFS give-back),
    # so the zero-operand figure is the schedule's own ceiling and the random-data one what dense data gets
    if roofline is not None and rank == 0 and not use_dist and not args.no_zero_operands:
        import copy

        cands = [r for r in (survey or records) if r[7] == dom and r[8] is not None]
        if cands:
            rr = max(cands, key=lambda r: r[0].elapsed_time(r[1]))
            xs, pc, pc2 = rr[3], rr[8], rr[9]
            pcz = copy.copy(pc)
            pcz.w = torch.zeros_like(pc.w)
            g_r = torch.Generator(device=dev).manual_seed(5)
            x_r = torch.randn(xs, generator=g_r, device=dev, dtype=torch.float32).to(torch.bfloat16)
            x_z = torch.zeros_like(x_r)
            y_buf = None if pc2 is not None else torch.empty((xs[0],) + nn_ops.conv_out_hw(xs[1], xs[2], pc) + (pc.cout,), dtype=torch.bfloat16, device=dev)

            def one(xx, pp):
                if pc2 is not None:
                    return orig_conv2d_head(xx, pp, pc2)
                return orig_conv2d(xx, pp, out=y_buf)

            reps, t_arm = 12, {"random": 0.0, "zero": 0.0}
            for arm, xx, pp in (("random", x_r, pc), ("zero", x_z, pcz)):
                one(xx, pp)
            for _ in range(3):          # interleaved rounds, one process (guide 5.4 rule 24)
                for arm, xx, pp in (("random", x_r, pc), ("zero", x_z, pcz)):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _i in range(reps // 3):
                        one(xx, pp)
                    e1.record()
                    torch.cuda.synchronize()
                    t_arm[arm] += e0.elapsed_time(e1)
            tf = {k_: rr[2] * reps / (v * 1e-3) / 1e12 for k_, v in t_arm.items()}
            roofline["zero_operands"] = {
                "layer": f"{xs}->{pc.cout} k{pc.kh}" + (" + fused 1x1 head" if pc2 is not None else ""),
                "random_tflops": round(tf["random"], 1), "random_frac": round(tf["random"] / PEAK_BF16_TFLOPS, 4),
                "zero_tflops": round(tf["zero"], 1), "zero_frac": round(tf["zero"] / PEAK_BF16_TFLOPS, 4),
                "what": "the dominant kernel's heaviest launch, 12 launches each on N(0,1) activations + the layer's weights and on "
                        "all-zero activations and weights (identical instruction stream and HBM traffic); the gap is the clock the "
                        "chip gives up to operand bit toggling, not issue density"}
            del x_r, x_z, y_buf

    if args.dump_convs and rank == 0 and records:
        # per-layer table: measured time against the layer's own roofline max(flops / MFMA peak, algorithmic bytes / HBM peak)
        per = {}
        for e0, e1, fl, xs, cout, k, byts, kid, _pc, _pc2, n_l in records:
            key = f"{xs}->{cout} k{k}" + (" [block]" if kid == 7 else (" [1x1 + 3x3 pair]" if kid == 9 else ""))
            d = per.setdefault(key, [0.0, 0.0, 0, 0.0, kid])
            d[0] += e0.elapsed_time(e1); d[1] += fl; d[2] += max(n_l, 1); d[3] += byts
        rows = []
        for k_, v in per.items():
            t_roof = max(v[1] / (PEAK_BF16_TFLOPS * 1e12), v[3] / PEAK_HBM_BPS) * 1e3
            rows.append({"layer": k_, "kernel": KNAMES.get(v[4], str(v[4])), "launches_per_step": v[2] // args.steps,
                         "ms_per_step": round(v[0] / args.steps, 4), "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 1) if v[0] > 0 else 0,
                         "algorithmic_tb_per_s": round(v[3] / (v[0] * 1e-3) / 1e12, 2) if v[0] > 0 else 0,
                         "bound": "mfma" if v[1] / (PEAK_BF16_TFLOPS * 1e12) >= v[3] / PEAK_HBM_BPS else "hbm",
                         "roofline_ms_per_step": round(t_roof / args.steps, 4), "frac_of_roofline": round(t_roof / v[0], 3) if v[0] > 0 else 0})
        rows.sort(key=lambda r: -r["ms_per_step"])
        with open(args.dump_convs, "w") as f:
            json.dump(rows, f, indent=1)

    # Mask R-CNN: the pasting launch on its own (HBM-write-bound: H * W / 8 bytes per detection slot), from the last step's outputs
    mask_paste = None
    if rank == 0 and hasattr(model, "mask_head") and not args.graph:
        from minddet_amd import det_ops

        out_m = model.forward(images, paste=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        det_ops.paste_masks(out_m[2], out_m[0], (H, W), model.mask_thr, bits=True)
        e0.record()
        for _ in range(5):
            pm = det_ops.paste_masks(out_m[2], out_m[0], (H, W), model.mask_thr, bits=True)
        e1.record()
        torch.cuda.synchronize()
        t_p = e0.elapsed_time(e1) / 5 * 1e-3
        mask_paste = {"in_step": bool(model.paste), "ms": round(t_p * 1e3, 4), "bytes_written": int(pm.numel() * 4),
                      "gb_per_s": round(pm.numel() * 4 / t_p / 1e9, 1), "frac_of_hbm_peak": round(pm.numel() * 4 / t_p / PEAK_HBM_BPS, 4),
                      "what": "md_paste_masks: [B,max_det,28,28] probabilities -> [B,max_det,H,W/32] int32 bit masks at mask_thr_binary"}
        del pm, out_m

    is_frcnn = type(model).__name__ == "FasterRCNN"
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and is_frcnn:
        from oracle import nets

        try:
            ncpu = len(os.sched_getaffinity(0))
        except AttributeError:
            ncpu = os.cpu_count() or 1
        torch.set_num_threads(max(1, min(ncpu, 16)))  # the GPU box's CPU share for one GPU is 16 cores
        x1 = images[:1].float().cpu()
        nets.faster_rcnn_forward(model, x1[:, :256, :256], quant=False)  # warm-up on a crop
        tc = time.perf_counter()
        n_img = 0
        while n_img < 1 or (time.perf_counter() - tc < 10.0 and n_img < 4):
            nets.faster_rcnn_forward(model, x1, quant=False)
            n_img += 1
        tcpu = time.perf_counter() - tc
        cpu_baseline = {"value": round(n_img / tcpu, 4), "unit": "images/sec", "cores": torch.get_num_threads(),
                        "kind": "port",
                        "sample": f"{n_img} image(s) 800x1344, oracle/nets.py fp32 torch-CPU restatement of the same graph "
                                  "(MindSpore-CPU is not installable here; BASELINE.md section 3)"}
        cpu_baseline["reference_ops"] = reference_ops_baseline(dev)

    if rank == 0:
        total_images = world * B * args.steps
        wl = ("faster_rcnn_r50_fpn_800x1344 (BASELINE.json configs[2])" if is_frcnn
              else f"{os.path.basename(args.config)} {H}x{W} (secondary workload)")
        gmac = (model.macs_per_image(H, W) / 1e9 if is_frcnn else
                (sum(r[2] for r in records) / 2 / max(args.steps, 1) / B / 1e9 if records else None))
        line = {
            "metric": ("images/sec, Faster R-CNN R50-FPN inference, COCO-shaped 1333x800 (padded 800x1344)" if is_frcnn
                       else f"images/sec, {type(model).__name__} inference {H}x{W}"),
            "value": round(total_images / dt, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            # host time per step spent ENQUEUING (Python + ctypes + launches) in the timed region, before the closing device sync: well
            # under ms_per_step = the device is the bottleneck and N Python ranks on one node are not launch-bound
            "host_enqueue_ms_per_step": None if host_enqueue_ms is None else round(host_enqueue_ms, 3),
            "host_enqueue_ms_single_step": round(host_single_ms, 3),
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": wl, "batch_per_gpu": B, "streams": n_streams,
                       "global_batch": world * B, "parallelism": f"dp{world} (image sharding + all_gather of detections)",
                       "gmac_per_image": None if gmac is None else round(gmac, 2), "weights": "random init, seed 7"},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "from_uint8": from_u8, "two_streams": two_streams, "one_stream": one_stream, "mask_paste": mask_paste,
            # images (over every step of this run) whose class-wise NMS saw a FULL top-nms_pre prefix and fewer than max_det
            # survivors: only those can differ from the NMS over every candidate (DESIGN.md "pre-NMS prefix"); read after timing
            "nms_prefix": None if not hasattr(model, "prefix_status") else {
                "nms_pre": getattr(getattr(model, "roi_head", model), "nms_pre", None), "image_slots_flagged": model.prefix_status.flagged()},
            "lib": os.path.relpath(_lib.LIB_PATH, ROOT),
        }
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())

"""bench.py -- images/sec of the Faster R-CNN R50-FPN 800x1344 bf16 inference hot path on N MI355X.

    python bench.py --gpus N --steps K --warmup W [--batch B]
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One step = one pass of the whole path (backbone + FPN + RPN + proposals + RoIAlign + box head +
class-wise NMS + packing) over one batch of synthetic COCO-shaped images already resident in HBM,
followed, when N > 1, by the single all_gather of the padded detections.  Images are
sharded across ranks (weak scaling: B images per GPU); value = N*B*K / max-over-ranks time.

The JSON line also carries
  roofline     -- the DOMINANT conv kernel by time (md_conv2d_last_kernel attributes every launch; each is bracketed by
                  HIP events on the launch stream) against the roofline that binds its launches in aggregate
                  (algorithmic flops / 2.5 PFLOP/s dense bf16 or algorithmic bytes / 8 TB/s), its PMC traffic from
                  profiles/r01_conv_traffic.json, and under "all_conv" the same per kernel and for the whole conv/FC set
                  (incl. frac_of_layerwise_roofline = sum of per-launch max(flops/peak, bytes/peak) / measured time).
  cpu_baseline -- the oracle's plain fp32 torch-CPU restatement of the same graph (oracle/nets.py) timed on this
                  host's cores on a bounded sample (rank 0, N=1 only).
The batch is resident in HBM in the model's input layout (zero-bordered 4-channel NHWC for the fused stem) before the timed
region; --from-uint8 also times md_image_preprocess from a resident uint8 batch.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (guide: ~2.5 PF)
PEAK_HBM_BPS = 8.0e12      # HBM3E spec peak (guide: 8 TB/s, ~6.3 achievable)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=60,
                    help="images per GPU per step (r01, no instrumentation: 32 -> 1640, 48 -> 1797, 60 -> 1870, 128 -> 1893 images/s; 60 is the largest batch whose P2 tensors stay under the 2 GiB reach of one launch)")
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
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("MD_FORCE_DIST") == "1"  # MD_FORCE_DIST: exercise the RCCL path with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    from minddet.models import Config, build_detector
    from minddet_amd import _lib, nn_ops
    from minddet_amd.data import synthetic_images
    from minddet_amd.shard import gather_detections, gather_masks

    cfg = Config.fromfile(args.config)
    model = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(dev)
    H, W = cfg.data.input_hw
    B = args.batch
    images = synthetic_images(B, H, W, seed=20240317 + rank, device=dev)
    if type(model).__name__ in ("FasterRCNN", "MaskRCNN") and nn_ops.stem_layout_ok(H, W) and os.environ.get("MD_STEM_LAYOUT", "1") == "1":
        # the batch is resident in HBM in the model's input layout before the timed region starts: zero-bordered
        # 4-channel NHWC (md_stem_pool); MD_STEM_LAYOUT=0 keeps the 8-channel layout + two-launch stem for A/B
        images = nn_ops.to_stem_layout(images)

    # ---- per-conv event instrumentation (roofline of the dominant kernel)
    records = []
    orig_conv2d = nn_ops.conv2d
    from minddet_amd import _lib
    last_kernel = _lib.lib().md_conv2d_last_kernel
    KNAMES = {1: "conv_pingpong_kernel", 2: "conv_igemm_kernel<128x128>", 3: "conv_igemm_kernel<small cout>",
              4: "conv_igemm_kernel<generic K>", 5: "conv3x3_halo_kernel", 6: "conv variant"}

    sel = {"calls": None, "idx": 0}    # calls: None = bracket every launch, else the per-step call indices to bracket

    def _skip():
        i = sel["idx"]
        sel["idx"] = i + 1
        return sel["calls"] is not None and i not in sel["calls"]

    def timed_conv2d(x, pc, residual=None, relu=None, out=None, variant=None, c_off=0, res_upsample=False, **kw):
        if _skip():
            return orig_conv2d(x, pc, residual=residual, relu=relu, out=out, variant=variant, c_off=c_off, res_upsample=res_upsample, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = orig_conv2d(x, pc, residual=residual, relu=relu, out=out, variant=variant, c_off=c_off, res_upsample=res_upsample, **kw)
        e1.record()
        n, ho, wo, _ = y.shape
        # algorithmic bytes: the channels the layer reads / writes (operands may be channel slices of wider tensors)
        res_elems = 0 if residual is None else (residual.numel() if res_upsample else n * ho * wo * pc.cout)
        byts = 2.0 * (x.shape[0] * x.shape[1] * x.shape[2] * pc.cin + n * ho * wo * pc.cout + pc.cout * pc.cin_real * pc.kh * pc.kw + res_elems)
        records.append((e0, e1, 2.0 * n * ho * wo * pc.cout * pc.cin_real * pc.kh * pc.kw, tuple(x.shape[:3]) + (pc.cin,), pc.cout, pc.kh, byts,
                        last_kernel()))
        return y

    orig_conv2d_head = nn_ops.conv2d_head

    def timed_conv2d_head(x, pc, pc2, variant=None):
        if _skip():
            return orig_conv2d_head(x, pc, pc2, variant=variant)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = orig_conv2d_head(x, pc, pc2, variant=variant)
        e1.record()
        n, ho, wo, _ = y.shape
        fl = 2.0 * n * ho * wo * (pc.cout * pc.cin_real * pc.kh * pc.kw + pc2.cout * pc2.cin_real)
        byts = 2.0 * (x.numel() + y.numel() + pc.cout * pc.cin_real * pc.kh * pc.kw + pc2.cout * pc2.cin_real)
        records.append((e0, e1, fl, tuple(x.shape), pc.cout, pc.kh, byts, last_kernel()))
        return y

    images_u8 = pre_mat = None
    if args.from_uint8:
        g_u8 = torch.Generator(device="cpu").manual_seed(20240317 + rank)
        images_u8 = torch.randint(0, 256, (B, H, W, 3), generator=g_u8, dtype=torch.uint8).to(dev)
        pre_mat = torch.tensor([1.0, 0, 0, 0, 1.0, 0], dtype=torch.float32, device=dev).repeat(B, 1).contiguous()

    def step():
        sel["idx"] = 0
        x = images
        if images_u8 is not None:
            x = nn_ops.image_preprocess(images_u8, pre_mat, (0.408, 0.447, 0.470), (0.289, 0.274, 0.278), (H, W),
                                        stem_layout=images.shape[3] == 4)
        out = model.forward(x)
        dets, count = out[0], out[1]
        if use_dist:
            if len(out) > 2:   # Mask R-CNN: the 28x28 masks travel as fp16 in a second fixed-shape all_gather
                gather_masks(out[2], force=True)
            return gather_detections(dets, count, force=True)
        return dets, count

    if args.graph:
        if use_dist or images_u8 is not None:
            raise SystemExit("bench.py --graph: single-GPU, pre-processed input only")
        from minddet_amd.replay import CapturedStep

        args.no_roofline = True
        captured = CapturedStep(lambda xx: tuple(model.forward(xx))[:2], images)
        step = lambda: captured(images)   # noqa: E731
    instrument = not args.no_roofline
    survey = None          # records of ONE fully bracketed step (the last warmup step): the whole-set table in --bracket dominant
    dominant_only = instrument and args.bracket == "dominant" and args.warmup >= 1 and not args.dump_convs
    for w_i in range(args.warmup):
        if dominant_only and w_i == args.warmup - 1:
            nn_ops.conv2d, nn_ops.conv2d_head = timed_conv2d, timed_conv2d_head
            step()
            nn_ops.conv2d, nn_ops.conv2d_head = orig_conv2d, orig_conv2d_head
            torch.cuda.synchronize()
            survey = list(records)
            del records[:]
        else:
            step()
    if dominant_only and survey:
        t_k = {}
        for r in survey:
            t_k[r[7]] = t_k.get(r[7], 0.0) + r[0].elapsed_time(r[1])
        dom_id = max(t_k, key=t_k.get)
        sel["calls"] = frozenset(i for i, r in enumerate(survey) if r[7] == dom_id)   # dispatch is deterministic per call site
    if instrument:
        nn_ops.conv2d = timed_conv2d
        nn_ops.conv2d_head = timed_conv2d_head
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier(device_ids=[local_rank])
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier(device_ids=[local_rank])
    dt = time.perf_counter() - t0
    nn_ops.conv2d = orig_conv2d
    nn_ops.conv2d_head = orig_conv2d_head
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    roofline = None
    if instrument and records:
        # whole conv/FC set: every launch of the timed region (--bracket all) or of the fully bracketed last warmup step
        all_recs, all_steps = (survey, 1) if survey else (records, max(args.steps, 1))
        ms = [e0.elapsed_time(e1) for e0, e1, *_ in all_recs]
        tot_ms = sum(ms)
        tot_fl = sum(r[2] for r in all_recs)
        # per kernel; the DOMINANT one (most time) fills the contract fields, the whole conv/FC set goes to "all_conv"
        per_k = {}
        for r, t_ms in zip(all_recs, ms):
            d = per_k.setdefault(r[7], [0.0, 0.0, 0.0, 0])
            d[0] += t_ms; d[1] += r[2]; d[2] += r[6]; d[3] += 1
        dom = max(per_k, key=lambda k_: per_k[k_][0])
        dom_share = per_k[dom][0] / tot_ms
        # the dominant kernel's figures always come from the TIMED region's events
        d_ms = d_fl = d_by = 0.0
        d_n = 0
        for r in records:
            if r[7] == dom:
                d_ms += r[0].elapsed_time(r[1]); d_fl += r[2]; d_by += r[6]; d_n += 1
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
        tp = os.path.join(ROOT, "profiles", "r01_conv_traffic.json")
        PMC_PREFIX = {1: "conv_pingpong_kernel", 2: "conv_igemm_kernel<256, 2, 2, 2, 2, 2,"}  # all instantiations of the kernel
        if os.path.exists(tp):  # PMC passes are separate rocprofv3 runs (tools/pmc_traffic.py); same config only
            tj = json.load(open(tp))
            if tj.get("batch_per_gpu") == B and type(model).__name__ == "FasterRCNN":
                all_traffic = round(tj["hbm_bytes_per_launch"] / 1e6, 2)
                sel = [v for k_, v in tj.get("by_kernel", {}).items() if dom in PMC_PREFIX and k_.startswith(PMC_PREFIX[dom])]
                if sel:
                    traffic = round(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in sel) / sum(v["launches"] for v in sel) / 1e6, 2)
        roofline = {"bound": "hbm" if hbm_bound else "mfma",
                    "kernel": KNAMES.get(dom, str(dom)) + " (dominant kernel: %.0f %% of the conv/FC time)" % (100 * dom_share),
                    "bracketed": "HIP events around this kernel's launches in the timed region" + (
                        "; all_conv: every launch of the last warmup step" if survey else "; all_conv: every launch of the timed region"),
                    "achieved": round(ach, 2), "peak": peak, "unit": "GB/s" if hbm_bound else "TFLOP/s", "frac": round(ach / peak, 4),
                    "traffic": traffic, "traffic_unit": "MB per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_conv_traffic.json)",
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
                                 "algorithmic_mb_per_launch": round(sum(r[6] for r in all_recs) / len(all_recs) / 1e6, 2),
                                 "launches_per_step": len(all_recs) // all_steps,
                                 "avg_launch_us": round(tot_ms * 1e3 / len(all_recs), 2),
                                 "conv_ms_per_step": round(tot_ms / all_steps, 3),
                                 "algorithmic_gflop_per_step": round(tot_fl / all_steps / 1e9, 1)}}

    if args.dump_convs and rank == 0 and records:
        per = {}
        for e0, e1, fl, xs, cout, k, _b, _kid in records:
            key = f"{xs}->{cout} k{k}"
            d = per.setdefault(key, [0.0, 0.0, 0])
            d[0] += e0.elapsed_time(e1); d[1] += fl; d[2] += 1
        rows = [{"layer": k_, "ms_per_step": v[0] / args.steps, "tflops": v[1] / (v[0] * 1e-3) / 1e12 if v[0] > 0 else 0,
                 "launches_per_step": v[2] // args.steps} for k_, v in per.items()]
        rows.sort(key=lambda r: -r["ms_per_step"])
        with open(args.dump_convs, "w") as f:
            json.dump(rows, f, indent=1)

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
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": wl, "batch_per_gpu": B,
                       "global_batch": world * B, "parallelism": f"dp{world} (image sharding + all_gather of detections)",
                       "gmac_per_image": None if gmac is None else round(gmac, 2), "weights": "random init, seed 7"},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
            "lib": os.path.relpath(_lib.LIB_PATH, ROOT),
        }
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Two HIP streams, each running the detector on its own half batch back to back with NO join between steps; the second stream starts D ms
behind the first.  Question: when the halves are out of phase (one in the HBM-bound backbone while the other is in the MFMA-bound FPN / RPN
convs), does the chip finish more images per second than with both halves in lock step (D = 0 = what SplitForward does today)?
usage: python tools/stagger_ab.py <config> <batch> [steps] [D,D,...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from minddet.models import Config, build_detector
from minddet_amd import nn_ops
from minddet_amd.data import synthetic_images

cfgp, N = sys.argv[1], int(sys.argv[2])
K = int(sys.argv[3]) if len(sys.argv) > 3 else 20
DS = [float(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0.0, 10.0, 20.0, 27.0, 0.0, 20.0]
dev = "cuda:0"
cfg = Config.fromfile(cfgp)
m = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(dev)
hw = (800, 1344) if "rcnn" in cfgp else (640, 640)
x = synthetic_images(N, hw[0], hw[1], device=dev)
x = nn_ops.to_stem_layout(x) if nn_ops.stem_layout_ok(*hw) else x
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
pa, pb = x[:N // 2], x[N // 2:]
m.forward(pa); m.forward(pb); torch.cuda.synchronize()
ev = lambda: torch.cuda.Event(enable_timing=True)


def run(delay_ms):
    torch.cuda.synchronize()
    e0 = ev(); e0.record()
    sa.wait_stream(torch.cuda.current_stream()); sb.wait_stream(torch.cuda.current_stream())
    done_a, done_b = [], []
    t0 = time.perf_counter()
    with torch.cuda.stream(sa):
        m.forward(pa); e = ev(); e.record(); done_a.append(e)
    while (time.perf_counter() - t0) * 1e3 < delay_ms:
        pass
    for k in range(K):
        with torch.cuda.stream(sb):
            m.forward(pb); e = ev(); e.record(); done_b.append(e)
        if k + 1 < K:
            with torch.cuda.stream(sa):
                m.forward(pa); e = ev(); e.record(); done_a.append(e)
    host_ms = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    total = max(e0.elapsed_time(done_a[-1]), e0.elapsed_time(done_b[-1]))
    # steady state: completions 4 .. K-3 of each stream
    lo, hi = 4, K - 3
    steady = 0.5 * (done_a[lo].elapsed_time(done_a[hi]) + done_b[lo].elapsed_time(done_b[hi])) / (hi - lo)
    phase = done_a[lo].elapsed_time(done_b[lo])
    return total, steady, phase, host_ms


run(0.0)
for d in DS:
    total, steady, phase, host_ms = run(d)
    print(f"{os.path.basename(cfgp)} b{N} x {K} steps, second stream {d:5.1f} ms behind: whole run {total / K:7.3f} ms/step ({N * K / total * 1e3:7.1f} images/s) | "
          f"steady state {steady:7.3f} ms per step of both halves ({N / steady * 1e3:7.1f} images/s), B finishes {phase:6.2f} ms behind A | host enqueue {host_ms / K:5.2f} ms/step", flush=True)

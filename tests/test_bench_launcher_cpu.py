"""bench.py's own multi-rank machinery on the CPU: the step / timing / all_reduce(MAX) code driven by a stub model over gloo
(world size 2), the asynchronous all_gather of the detections, and the --gpus launcher's argument checks (no GPU needed:
the launcher refuses before any GPU call when fewer devices than ranks are visible)."""
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _stub_forward(rank, B, M):
    """A model stand-in: detections of image b of this rank carry the GLOBAL image index, so the gathered order is checkable."""
    calls = {"n": 0}

    def fwd(x):
        calls["n"] += 1
        time.sleep(0.01 * (rank + 1))          # ranks of different speed: the MAX over ranks must be the slow one's time
        dets = torch.zeros((B, M, 6))
        count = torch.zeros((B,), dtype=torch.int32)
        for b in range(B):
            g = rank * B + b
            count[b] = g % (M + 1)
            dets[b, :count[b], 4] = float(g) + 0.5
            dets[b, :count[b], 5] = float(calls["n"])
        return dets, count

    return fwd, calls


def _worker(rank, world, port, q, with_masks=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from minddet_amd.shard import gather_detections_async

    B, M, K, W = 3, 5, 4, 2
    fwd0, calls = _stub_forward(rank, B, M)
    fwd = fwd0
    if with_masks:   # the Mask R-CNN form: a third output travels in a second fixed-shape all_gather (fp16 28x28 masks)
        def fwd(x):
            d_, c_ = fwd0(x)
            mk = torch.zeros((B, M, 28, 28))
            for b in range(B):
                mk[b, :, 0, 0] = float(rank * B + b)
                mk[b, :, 1, 1] = float(calls["n"])
            return d_, c_, mk
    step, finish = bench.make_step(fwd, torch.zeros(1), True,
                                   lambda out: gather_detections_async(out[0], out[1], masks=out[2] if len(out) > 2 else None, force=True))

    def all_reduce_max(v):
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    t_own0 = time.perf_counter()
    dt = bench.run_timed(step, finish, K, W, True, lambda: None, dist.barrier, all_reduce_max)
    t_own = time.perf_counter() - t_own0
    res = finish()
    d, c = res[0], res[1]
    ok = calls["n"] == K + W                                   # exactly K timed + W warmup forward passes
    if with_masks:
        mk = res[2]
        ok = ok and mk.dtype == torch.float16 and mk.shape == (world * B, M, 28, 28)
        ok = ok and bool((mk[:, 0, 0, 0] == torch.arange(world * B, dtype=torch.float16)).all()) and bool((mk[:, :, 1, 1] == K + W).all())
    ok = ok and bench.HOST_ENQUEUE["s"] is not None and 0 < bench.HOST_ENQUEUE["s"] <= t_own
    ok = ok and d.shape == (world * B, M, 6) and c.tolist() == [g % (M + 1) for g in range(world * B)]   # input order
    for g in range(world * B):
        n = int(c[g])
        ok = ok and bool((d[g, :n, 4] == g + 0.5).all()) and bool((d[g, n:] == 0).all())
        ok = ok and bool((d[g, :n, 5] == K + W).all())         # the LAST step's detections
    q.put((rank, ok, dt, t_own))
    dist.destroy_process_group()


def _run_world(world, with_masks):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q, with_masks)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(60)
    assert all(ok for _, ok, _, _ in res), res
    # every rank reports the same (max-over-ranks) time, and it covers the SLOWEST rank's 4 x (10 ms x world) of "compute"
    assert all(abs(r[2] - res[0][2]) < 1e-9 for r in res) and res[0][2] >= 4 * 0.01 * world * 0.9 and res[0][2] <= res[-1][3]


def test_bench_step_and_timing_code_world2_gloo():
    _run_world(2, False)


def test_bench_step_and_timing_code_world4_unequal_speeds_gloo():
    """four ranks whose forward passes take 10 / 20 / 30 / 40 ms: the async gather of step i is joined behind step i + 1 on every rank
    without deadlock, the last step's detections arrive in input order everywhere, the reported time is the slowest rank's"""
    _run_world(4, False)


def test_bench_two_gather_form_world2_gloo():
    """Mask R-CNN: detections + a second fixed-shape all_gather of the fp16 28x28 masks through the same make_step / run_timed code"""
    _run_world(2, True)


def test_async_gather_single_process_is_identity():
    from minddet_amd.shard import gather_detections_async

    d, c = torch.rand((2, 4, 6)), torch.tensor([1, 4], dtype=torch.int32)
    h = gather_detections_async(d, c)
    d2, c2 = h.result()
    assert d2 is d and c2 is c and h.result()[0] is d
    m = torch.rand((2, 4, 28, 28))
    d3, c3, m3 = gather_detections_async(d, c, masks=m).result()
    assert m3.dtype == torch.float16 and torch.equal(m3, m.to(torch.float16))


def _run_bench(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_gpus_flag_is_authoritative():
    """`bench.py --gpus 2` without WORLD_SIZE must never look like a 1-GPU run that printed n_gpus 1: with fewer than 2 visible
    devices the launcher refuses (exit 2, no JSON line) BEFORE any GPU call; with WORLD_SIZE != --gpus it refuses too."""
    n_dev = torch.cuda.device_count()
    if n_dev < 2:
        r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"])
        assert r.returncode == 2 and "only" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr and "torch.distributed.run" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    r = _run_bench(["--gpus", "0"])
    assert r.returncode == 2


def test_launcher_relays_exactly_one_line(tmp_path, monkeypatch):
    """launch_ranks with a stand-in child: N processes get RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 and rank 0's line is
    relayed once; a rank that fails, or a line with the wrong n_gpus, is an error."""
    import bench

    child = tmp_path / "child.py"
    child.write_text(
        "import json, os, sys\n"
        "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "assert os.environ['MASTER_ADDR'] == '127.0.0.1' and os.environ['LOCAL_RANK'] == str(r) and int(os.environ['MASTER_PORT']) > 0\n"
        "assert os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' and '--spawn' not in sys.argv\n"
        "mode = os.environ.get('CHILD_MODE', 'ok')\n"
        "if mode == 'fail' and r == 1: sys.exit(3)\n"
        "if mode == 'die' and r == 2: sys.exit(7)\n"
        "if mode in ('die', 'hang'):\n"
        "    import time\n"
        "    if r == 0: print('rank 0 is waiting in a collective', flush=True)\n"
        "    time.sleep(600)\n"
        "if r == 0: print(json.dumps({'n_gpus': w if mode != 'wrong' else 1, 'value': 1.0}))\n")
    monkeypatch.setattr(bench, "__file__", str(child))
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    args = bench.parse_args(["--gpus", "3", "--spawn"])
    import io
    from contextlib import redirect_stdout

    buf = io.StringIO()
    with redirect_stdout(buf):
        rc = bench.launch_ranks(args, ["--gpus", "3", "--spawn"])
    assert rc == 0 and json.loads(buf.getvalue().strip())["n_gpus"] == 3 and buf.getvalue().count("{") == 1
    for mode in ("fail", "wrong"):
        monkeypatch.setenv("CHILD_MODE", mode)
        buf = io.StringIO()
        with redirect_stdout(buf):
            rc = bench.launch_ranks(args, ["--gpus", "3"])
        assert rc != 0 and buf.getvalue() == ""
    # one rank dies while the others sit in a "collective" (sleep 600): the launcher terminates and reaps the siblings and returns the
    # dead rank's code at once, not after torch.distributed's timeout; a run past MD_BENCH_DEADLINE_S is ended the same way
    import psutil

    for mode, want in (("die", 7), ("hang", None)):
        monkeypatch.setenv("CHILD_MODE", mode)
        monkeypatch.setenv("MD_BENCH_DEADLINE_S", "3" if mode == "hang" else "3000")
        t0 = time.time()
        buf = io.StringIO()
        with redirect_stdout(buf):
            rc = bench.launch_ranks(args, ["--gpus", "3"])
        assert rc != 0 and (want is None or rc == want) and buf.getvalue() == "" and time.time() - t0 < 60
        left = [c for c in psutil.Process().children(recursive=True) if c.is_running() and c.status() != psutil.STATUS_ZOMBIE
                and any(str(child) in a for a in c.cmdline())]
        assert not left, left

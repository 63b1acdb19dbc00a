"""One diagnostic pass over the bottleneck64_kernel variants built by tools/bn_variants.sh: each variant library is
loaded in a process of its own and md_bottleneck is bit-compared with the three md_conv2d launches on the shapes / repetitions that showed the
r03 symptom.  For a mismatching run the wrong elements are decoded into (tile, pixel in tile, channel) -> (wave, lane, register group) so
that the ISA of that variant can be read at the store that produced them.  python tools/bn_chain_repro.py [reps]"""
import os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VDIR = os.path.join(ROOT, "minddet_amd", "csrc", "build", "variants")


def child(lib_path, reps):
    sys.path.insert(0, ROOT)
    import torch
    from minddet_amd import _lib, nn_ops
    _lib.LIB_PATH = lib_path
    dev = "cuda:0"
    g = torch.Generator().manual_seed(3)
    total = 0
    for cin, ds, ext_res in ((64, True, False), (256, False, False), (256, False, True)):
        w1 = torch.randn((64, cin, 1, 1), generator=g) * (2.0 / cin) ** 0.5
        w2 = torch.randn((64, 64, 3, 3), generator=g) * (2.0 / 576) ** 0.5
        w3 = torch.randn((256, 64, 1, 1), generator=g) * (2.0 / 64) ** 0.5
        p1 = nn_ops.pack_conv(w1, bias=torch.randn((64,), generator=g) * 0.1, relu=True).to(dev)
        p2 = nn_ops.pack_conv(w2, bias=torch.randn((64,), generator=g) * 0.1, stride=1, pad=1, relu=True).to(dev)
        p3 = nn_ops.pack_conv(w3, bias=torch.randn((256,), generator=g) * 0.1, relu=True).to(dev)
        pd = nn_ops.pack_conv(torch.randn((256, cin, 1, 1), generator=g) * (1.0 / cin) ** 0.5, bias=torch.randn((256,), generator=g) * 0.1,
                              relu=False).to(dev) if ds else None
        blk = nn_ops.pack_bottleneck(p1, p2, p3, pd)
        for shape in ((1, 16, 32), (2, 24, 48), (4, 64, 64), (16, 200, 336)):
            x = torch.randn(shape + (cin,), generator=g).to(torch.bfloat16).to(dev)
            res = nn_ops.conv2d(x, pd) if ds else (torch.randn(shape + (256,), generator=g).to(torch.bfloat16).to(dev) if ext_res else x)
            ref = nn_ops.conv2d(nn_ops.conv2d(nn_ops.conv2d(x, p1), p2), p3, residual=res)
            n_bad = 0
            for rep in range(reps):
                y = nn_ops.bottleneck(x, blk, residual=res if ext_res else None)
                torch.cuda.synchronize()
                if torch.equal(y, ref):
                    continue
                n_bad += 1
                if n_bad <= 2:   # decode: which pixels / channels, in kernel coordinates
                    idx = (y != ref).nonzero().cpu()
                    seen = {}
                    for n_, yy, xx, c in idx.tolist():
                        key = (n_, yy // 8, xx // 16)
                        seen.setdefault(key, []).append(((yy % 8) * 16 + xx % 16, c))
                    for key, lst in list(seen.items())[:4]:
                        px = sorted({p for p, _ in lst}); ch = sorted({c for _, c in lst})
                        n_, ty, tx = key
                        vals = [hex(int(y[n_, ty * 8 + p // 16, tx * 16 + p % 16, ch[0] & ~1:(ch[0] & ~1) + 2].view(torch.int16).view(torch.int32).item()) & 0xffffffff) for p in px[:4]]
                        print(f"    tile {key}: {len(lst)} wrong; pixels {px[:12]} (wave wq {px[0] // 32}, lane-block {(px[0] % 32) // 4}) "
                              f"channels {ch[:16]} (quarter {ch[0] // 64}, wc {(ch[0] % 64) // 32}, g {(ch[0] % 32) // 8}); stored dwords {vals}", flush=True)
            total += n_bad
            print(f"  Cin {cin} ds {ds} ext_res {ext_res} shape {shape}: {reps} runs, {n_bad} mismatching", flush=True)
    print(f"  TOTAL {total}", flush=True)
    return 0


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--child":
        sys.exit(child(sys.argv[2], int(sys.argv[3])))
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    names = sys.argv[2:] or sorted(f[4:-3] for f in os.listdir(VDIR) if f.startswith("lib_") and f.endswith(".so"))
    for v in names:
        print(f"== variant {v}", flush=True)
        subprocess.call([sys.executable, os.path.abspath(__file__), "--child", os.path.join(VDIR, f"lib_{v}.so"), str(reps)])

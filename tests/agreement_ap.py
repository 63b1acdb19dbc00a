"""End-to-end agreement of the device path with the CPU oracle in mAP units (no dataset, no trained weights here).
The oracle's detections (fp32 torch-CPU restatement of the same graph, same random-init weights, same synthetic images) are
taken as ground truth and the device detections are scored against them with the COCO bbox protocol (minddet_amd/coco_eval.py):
AP 1.0 = every oracle box is found with IoU >= 0.95 and nothing else ranks above it.  `quant` = the oracle rounds activations and
weights to bf16 where the device stores bf16 (isolates kernel arithmetic from storage precision).
Lives under tests/ because it drives the oracle (test infrastructure).  usage: python tests/agreement_ap.py [config] [batch]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minddet.models import Config, build_detector
from minddet_amd.coco_eval import COCOBboxEval
from oracle import nets


def records(dets, count, with_score):
    out = []
    for b in range(dets.shape[0]):
        for d in dets[b, :int(count[b])]:
            r = dict(image_id=b, category_id=int(d[5]), bbox=[float(d[0]), float(d[1]), float(d[2] - d[0]), float(d[3] - d[1])])
            if with_score:
                r["score"] = float(d[4])
            out.append(r)
    return out


def agreement(cfg_path="configs/faster_rcnn/faster_rcnn_tiny.py", B=4, hw=(128, 192), seed=0):
    cfg = Config.fromfile(cfg_path)
    m = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to("cuda:0")
    g = torch.Generator().manual_seed(seed)
    x = torch.zeros((B, hw[0], hw[1], 8))
    x[..., :3] = torch.randn((B, hw[0], hw[1], 3), generator=g)
    xb = x.to(torch.bfloat16)
    dets, count = m.forward(xb.to("cuda:0"))[:2]
    dets, count = dets.cpu().numpy(), count.cpu().numpy()
    res = {}
    for quant in (True, False):
        d_o, c_o = nets.faster_rcnn_forward(m, xb.float(), quant=quant)
        gt = records(d_o, c_o, False)
        for i, r in enumerate(gt):
            r["id"] = i
        s = COCOBboxEval(gt, records(dets, count, True)).summarize()
        res["bf16-matched oracle" if quant else "fp32 oracle"] = dict(AP=s["AP"], AP50=s["AP50"], AP75=s["AP75"], AR100=s["AR100"],
                                                                        n_oracle=int(c_o.sum()), n_device=int(count.sum()))
    return res


if __name__ == "__main__":
    a = sys.argv[1:]
    r = agreement(*( [a[0]] if a else [] ), **({"B": int(a[1])} if len(a) > 1 else {}))
    for k, v in r.items():
        print(k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()})

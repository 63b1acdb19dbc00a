"""CPU: weight import / export.  The key map is pinned by the reference's own key lists
(tests/golden/centernet_ckpt_keys.json, generated from centernet_ms_params.txt / centernet_params.txt) and by re-running
the procedure of centernet/convert_ckpt.py:56-92 (positional pairing + BatchNorm name swap) on them; the codecs are
round-tripped on synthetic checkpoints (no real checkpoint ships with the reference: parity of real weights unpinned)."""
import json
import os

import numpy as np
import torch

from minddet_amd import graphs, weights

HERE = os.path.dirname(os.path.abspath(__file__))
KEYS = json.load(open(os.path.join(HERE, "golden", "centernet_ckpt_keys.json")))


def _model(seed):
    return graphs.CenterNet(depth=18, num_classes=80, seed=seed)


def test_exported_key_sets_equal_the_reference_lists():
    m = _model(1)
    assert sorted(weights.centernet_state(m, "ms")) == sorted(KEYS["ms"])
    assert sorted(weights.centernet_state(m, "torch")) == sorted(KEYS["torch"])


def test_torch_to_ms_name_equals_convert_ckpt_procedure():
    swap = {"moving_mean": "gamma", "moving_variance": "beta", "gamma": "moving_mean", "beta": "moving_variance"}
    for pt, ms in zip(KEYS["torch"], KEYS["ms"]):       # convert_ckpt.py: positional pairing ...
        leaf = ms.rsplit(".", 1)[1]
        if leaf in swap:                                   # ... then the BatchNorm name swap (:67-78)
            ms = ms.rsplit(".", 1)[0] + "." + swap[leaf]
        assert weights.torch_to_ms_name(pt) == ms, (pt, ms)


def _same(a, b):
    sa, sb = weights.centernet_state(a, "ms"), weights.centernet_state(b, "ms")
    return all(np.array_equal(sa[k], sb[k]) for k in sa)


def test_ms_ckpt_round_trip(tmp_path):
    a, b = _model(1), _model(2)
    assert not _same(a, b)
    p = str(tmp_path / "centernet.ckpt")
    weights.write_ms_ckpt(p, weights.centernet_state(a, "ms"))
    params = weights.read_ms_ckpt(p)
    assert len(params) == 151 and params["network.hm_fn.2.bias"].dtype == np.float32
    assert weights.load_centernet(b, params) == []
    assert _same(a, b)
    assert torch.equal(a.head2.weight, b.head2.weight) and torch.equal(a.head1.bias, b.head1.bias)  # fused heads rebuilt


def test_torch_pth_round_trip(tmp_path):
    a, b = _model(3), _model(4)
    sd = {k: torch.from_numpy(v) for k, v in weights.centernet_state(a, "torch").items()}
    sd["bn1.num_batches_tracked"] = torch.tensor(7)      # present in real torch checkpoints, ignored
    p = str(tmp_path / "ctdet_coco_resdcn18.pth")
    torch.save({"state_dict": {"module." + k: v for k, v in sd.items()}}, p)
    params = weights.read_torch_pth(p)
    unused = weights.load_centernet(b, params)
    assert unused == [] and _same(a, b)


def test_shape_mismatch_and_missing_keys_are_errors(tmp_path):
    a = _model(1)
    st = weights.centernet_state(a, "ms")
    bad = dict(st)
    bad["network.backbone.conv1.weight"] = np.zeros((64, 3, 3, 3), np.float32)
    try:
        weights.load_centernet(_model(2), bad)
        assert False
    except ValueError:
        pass
    del st["network.reg_fn.2.bias"]
    try:
        weights.load_centernet(_model(2), st)
        assert False
    except KeyError:
        pass
    assert weights.load_centernet(_model(2), st, strict=False) == []


# ----------------------------------------------------------------------------- CenterPoint / PointPillars
def _convert_procedure(keys):
    """The loop of centerpoint/det3d_ms/models/detectors/point_pillars.py:137-168 restated on key names only."""
    keys = sorted(keys)
    out = []
    for item in keys:
        if "num_batches_tracked" in item or "global_step" in item:
            continue
        if "running_mean" in item:
            out.append((item, item.replace("running_mean", "moving_mean")))
        elif "running_var" in item:
            out.append((item, item.replace("running_var", "moving_variance")))
        elif "bias" in item:
            out.append((item, item.replace("bias", "beta") if item.replace("bias", "running_var") in keys else item))
        elif "weight" in item:
            out.append((item, item.replace("weight", "gamma") if item.replace("weight", "running_var") in keys else item))
        else:
            out.append((item, item))
    return dict(out)


def test_centerpoint_key_rule():
    rpn = graphs.RPN(layer_nums=(1, 1), ds_layer_strides=(2, 2), ds_num_filters=(16, 32), us_layer_strides=(1, 2),
                     us_num_filters=(16, 16), num_input_features=8, seed=3)
    tk = list(weights.rpn_state(rpn, naming="torch")) + ["neck.blocks.0.2.num_batches_tracked", "global_step",
                                                         "bbox_head.tasks.0.hm.1.weight", "bbox_head.tasks.0.hm.1.bias"]
    kmap = weights.torch_to_ms_generic(tk)
    assert kmap == _convert_procedure(tk)
    assert set(kmap.values()) - {"bbox_head.tasks.0.hm.1.weight", "bbox_head.tasks.0.hm.1.bias"} == set(weights.rpn_state(rpn, naming="ms"))
    # the cell indices of rpn.py:114-143 / :60-105
    ms = weights.rpn_state(rpn, naming="ms")
    assert ms["neck.blocks.0.1.weight"].shape == (16, 8, 3, 3) and ms["neck.blocks.0.4.weight"].shape == (16, 16, 3, 3)
    assert ms["neck.blocks.1.2.gamma"].shape == (32,) and ms["neck.deblocks.0.0.weight"].shape == (16, 16, 1, 1)
    assert ms["neck.deblocks.1.0.weight"].shape == (32, 16, 2, 2)     # Conv2dTranspose: [Cin, Cout, k, k]


def test_rpn_checkpoint_round_trip_both_namings(tmp_path):
    kw = dict(layer_nums=(1, 2), ds_layer_strides=(2, 2), ds_num_filters=(16, 32), us_layer_strides=(1, 2),
              us_num_filters=(16, 16), num_input_features=8)
    a, b, c = graphs.RPN(seed=1, **kw), graphs.RPN(seed=2, **kw), graphs.RPN(seed=3, **kw)
    p = str(tmp_path / "cp.ckpt")
    weights.write_ms_ckpt(p, weights.rpn_state(a, naming="ms"))
    assert weights.load_rpn(b, weights.read_ms_ckpt(p)) == []
    q = str(tmp_path / "cp.pth")
    sd = {k: torch.from_numpy(v) for k, v in weights.rpn_state(a, naming="torch").items()}
    sd["neck.blocks.0.2.num_batches_tracked"] = torch.tensor(7)
    torch.save({"state_dict": sd}, q)
    assert weights.load_rpn(c, weights.read_torch_pth(q)) == []
    sa = weights.rpn_state(a)
    for m in (b, c):
        sm = weights.rpn_state(m)
        assert all(np.array_equal(sa[k], sm[k]) for k in sa)


def test_pointpillars_train_checkpoint_prefixes():
    p = {"network.network.rpn.blocks.0.1.weight": 1, "optimizer.rpn.blocks.0.2.gamma": 2, "global_step": 3, "learning_rate": 4}
    assert weights.strip_net_prefix(p) == {"rpn.blocks.0.1.weight": 1, "rpn.blocks.0.2.gamma": 2}

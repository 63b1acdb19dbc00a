"""Golden vectors for the range anchor generator and the multi-generator anchor table, produced BY THE REFERENCE:
create_anchors_3d_range (minddet/models/pointpillars/src/core/box_np_ops.py:526-568) and TargetAssigner.generate_anchors
(minddet/models/pointpillars/src/core/target_assigner.py:227-249) over two AnchorGeneratorStride objects with the cyclist /
pedestrian settings of configs/ped_cycle_xyres16.yaml:119-135.  Run here only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/gen_anchors.py

Two compatibility notes, both about the numpy of this image (2.2) against the reference's pin (1.21):
  * create_anchors_3d_range assigns into the result of np.meshgrid, a list before numpy 2.0 and a tuple since: the function's module
    sees a numpy proxy whose meshgrid returns list(np.meshgrid(...)); nothing else is altered;
  * np.linspace(f32, f32, n, dtype=f32) evaluates in float32 on numpy >= 2 (NEP 50) and in float64 on numpy 1.21; the vectors
    below are what the reference's code computes HERE (float32 = md_anchor3d_range_attrs.linspace_mode 0).
Only inputs and outputs are stored.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
from gen_golden import _shim  # noqa: E402


class _NumpyListMeshgrid:
    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def meshgrid(*a, **k):
        return list(np.meshgrid(*a, **k))


def main():
    _shim()
    from src.core import anchor_generator, box_np_ops, target_assigner  # the reference's own modules

    out = {}
    box_np_ops.np = _NumpyListMeshgrid()
    cases = {
        "a": dict(feature_size=[1, 31, 27], anchor_range=[0.0, -39.68, -1.78, 69.12, 39.68, -1.78], sizes=[1.6, 3.9, 1.56], rotations=[0, 1.57]),
        "b": dict(feature_size=[2, 7, 5], anchor_range=[0.1, -3.3, -3.0, 7.7, 3.3, 1.0], sizes=[[0.6, 1.76, 1.73], [0.6, 0.8, 1.73]], rotations=[0, 0.785, 1.57]),
        "c": dict(feature_size=[1, 248, 216], anchor_range=[0.0, -39.68, -1.0, 69.12, 39.68, -1.0], sizes=[1.6, 3.9, 1.56], rotations=[0, np.pi / 2]),
        "d": dict(feature_size=[1, 1, 4], anchor_range=[2.0, 5.0, 0.5, 2.0, 6.0, 0.5], sizes=[1.0, 2.0, 3.0], rotations=[0.0]),  # zero step, n = 1
    }
    for tag, kw in cases.items():
        a = box_np_ops.create_anchors_3d_range(**kw)
        assert a.dtype == np.float32
        if tag == "c":   # the PointPillars car map: a strided sample + a checksum
            flat = a.reshape(-1, 7)
            out["range_c_shape"] = np.array(a.shape, np.int32)
            out["range_c_sample"] = flat[::997]
            out["range_c_sum64"] = flat.astype(np.float64).sum(0)
        else:
            out[f"range_{tag}"] = a
    box_np_ops.np = np
    # ---- two generators concatenated on the per-location axis (ped / cyclist)
    gens = [
        anchor_generator.AnchorGeneratorStride(sizes=[0.6, 1.76, 1.73], anchor_strides=[0.16, 0.16, 0.0], anchor_offsets=[0.08, -19.76, -1.465],
                                               rotations=[0, 1.57], class_id=None, match_threshold=0.5, unmatch_threshold=0.35,
                                               anchor_range=[0, -19.84, -2.5, 47.36, 19.84, 0.5]),
        anchor_generator.AnchorGeneratorStride(sizes=[0.6, 0.8, 1.73], anchor_strides=[0.16, 0.16, 0.0], anchor_offsets=[0.08, -19.76, -1.2],
                                               rotations=[0, 1.57], class_id=None, match_threshold=0.45, unmatch_threshold=0.3,
                                               anchor_range=[0, -19.84, -2.5, 47.36, 19.84, 0.5]),
    ]
    ta = target_assigner.TargetAssigner(box_coder=None, anchor_generators=gens)
    for tag, fs in (("small", [1, 13, 17]), ("full", [1, 248, 296])):
        r = ta.generate_anchors(fs)
        anchors = r["anchors"]
        assert anchors.dtype == np.float32 and anchors.shape == (1, fs[1], fs[2], 4, 7)
        if tag == "small":
            out["concat_small_anchors"] = anchors
            out["concat_small_matched"] = r["matched_thresholds"]
            out["concat_small_unmatched"] = r["unmatched_thresholds"]
        else:
            flat = anchors.reshape(-1, 7)
            out["concat_full_shape"] = np.array(anchors.shape, np.int32)
            out["concat_full_sample"] = flat[::1009]
            out["concat_full_sum64"] = flat.astype(np.float64).sum(0)
            out["concat_full_matched_sample"] = r["matched_thresholds"][::1009]
    p = os.path.join(HERE, "anchors_range_vectors.npz")
    np.savez_compressed(p, **out)
    print("wrote", p, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()

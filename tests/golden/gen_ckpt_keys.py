"""Generates tests/golden/centernet_ckpt_keys.json from the reference's own key lists (data files, not code):
minddet/models/centernet/centernet_ms_params.txt (MindSpore names) and centernet_params.txt (torch names), 151 lines each,
paired positionally by centernet/convert_ckpt.py:33-54.  Run here (the reference tree is not on the GPU box)."""
import json
import os

REF = "/root/reference/minddet/models/centernet"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ms = [l.strip() for l in open(os.path.join(REF, "centernet_ms_params.txt")) if l.strip()]
    pt = [l.strip() for l in open(os.path.join(REF, "centernet_params.txt")) if l.strip()]
    assert len(ms) == len(pt) == 151
    json.dump({"ms": ms, "torch": pt}, open(os.path.join(HERE, "centernet_ckpt_keys.json"), "w"), indent=0)
    print("wrote", len(ms), "key pairs")


if __name__ == "__main__":
    main()

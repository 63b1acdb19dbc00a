"""Synthetic COCO-shaped inputs: uint8 images from default_rng(20240317), normalised with
the CenterNet mean/std (centernet/default_config.yaml:88-89), NHWC bf16 with channels padded 3 -> 8."""
import numpy as np
import torch

MEAN = np.array([0.408, 0.447, 0.470], np.float32)
STD = np.array([0.289, 0.274, 0.278], np.float32)


def synthetic_images(batch, h, w, seed=20240317, device="cuda"):
    rng = np.random.default_rng(seed)
    out = torch.zeros((batch, h, w, 8), dtype=torch.bfloat16)
    for b in range(batch):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8).astype(np.float32) / 255.0
        out[b, :, :, :3] = torch.from_numpy((img - MEAN) / STD).to(torch.bfloat16)
    return out.to(device)

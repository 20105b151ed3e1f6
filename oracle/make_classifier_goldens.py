"""SURVEY section 8(c) fixture (9): eval-mode scores of the reference's OWN weights (swiftwatcher/model.pt) on
seeded segment crops.  Test infrastructure only.

    python3 oracle/make_classifier_goldens.py        (build container; /root/reference does not travel)

What is from the reference: the 52 weight tensors of model.pt (data, stored as float32 arrays so the GPU box can
load them without the checkpoint file).  What is NOT: the forward pass -- segment_classification.py cannot be imported
here (torchvision is installed nowhere in the image), so the scores come from oracle/classifier_ref.py, this project's
restatement of segment_classification.py:14-67 + torchvision's SqueezeNet-1.0 topology, in eval mode (the reference
leaves Dropout live, i.e. its own scores are random).  The transform chain runs on the real Pillow.  So this fixture
pins "the product on the real weight distribution == the oracle on the real weight distribution"; the topology itself
stays PARITY UNPINNED (strict key/shape match with model.pt is its only anchor).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle import classifier_ref as ref  # noqa: E402

MODEL = "/root/reference/swiftwatcher/model.pt"
OUT = os.path.join(ROOT, "tests", "golden", "classifier_model_pt.npz")


def crops(seed, count):
    """Seeded BGR crops the way extract_segment_images cuts them (image_filtering.py:338-369): >= 24 x 24, sky
    gradient + sensor noise, most with a dark bird-like ellipse, some with chimney texture, some plain noise."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(count):
        h = 24 if i % 4 == 0 else int(rng.integers(24, 72))
        w = 24 if i % 4 == 0 else int(rng.integers(24, 96))
        yy, xx = np.mgrid[0:h, 0:w]
        base = rng.uniform(120, 215) + 0.3 * yy
        img = np.stack([base + 10, base, base - 10], -1)
        kind = i % 5
        if kind != 4:
            cy, cx = h / 2 + rng.uniform(-3, 3), w / 2 + rng.uniform(-3, 3)
            a, b = rng.uniform(3, max(4, h / 2.2)), rng.uniform(3, max(4, w / 2.2))
            th = rng.uniform(0, np.pi)
            u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
            v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
            img[(u / b) ** 2 + (v / a) ** 2 <= 1.0] -= rng.uniform(40, 110)
        if kind == 3:
            img[int(0.6 * h):, :] = 60 + rng.uniform(-8, 8, size=(h - int(0.6 * h), w, 1))
        img += rng.normal(0, 2.5, size=img.shape)
        if kind == 4 and i % 10 == 9:
            img = rng.integers(0, 256, size=(h, w, 3)).astype(np.float64)
        out.append(np.clip(np.rint(img), 0, 255).astype(np.uint8))
    return out


if __name__ == "__main__":
    sd = torch.load(MODEL, map_location="cpu", weights_only=True)
    sd = {k: v.float().contiguous() for k, v in sd.items()}
    imgs = crops(20190816, 96)
    scores, keep = ref.classify(sd, imgs)
    d = {"w:" + k: v.numpy() for k, v in sd.items()}
    d["count"] = np.int32(len(imgs))
    for i, im in enumerate(imgs):
        d["crop%d" % i] = im
    d["scores"] = scores.astype(np.float32)
    d["keep"] = keep
    d["torch_version"] = torch.__version__
    np.savez_compressed(OUT, **d)
    margin = np.abs(scores[:, 1] - scores[:, 0])
    print("classifier_model_pt: %d crops, kept %d, min margin %.4g, score range [%.3g, %.3g], %d bytes"
          % (len(imgs), int(keep.sum()), float(margin.min()), float(scores.min()), float(scores.max()), os.path.getsize(OUT)))

"""Seeded synthetic scenes shared by the golden-vector generators (oracle/make_goldens*.py, run under the
reference's interpreter) and by the tests that regenerate the same inputs from a seed instead of storing
megabytes of noise.  TEST INFRASTRUCTURE ONLY; pure numpy, nothing from the reference."""
import hashlib

import numpy as np


def scene(rng, n, H, W, blobs=4, noise=2.5):
    """Noisy sky gradient + dark moving ellipses, u8 (n, H, W)."""
    yy, xx = np.mgrid[0:H, 0:W]
    base = 150.0 + 65.0 * yy / max(H - 1, 1)
    base[int(0.8 * H):, int(0.1 * W):int(0.9 * W)] = 60.0
    frames = np.empty((n, H, W), np.float64)
    pos = rng.uniform([0, 0], [H * 0.7, W], size=(blobs, 2))
    vel = rng.uniform(-4, 4, size=(blobs, 2))
    ax = rng.uniform(1.5, max(2.0, H / 10), size=(blobs, 2))
    depth = rng.uniform(40, 90, size=blobs)
    for t in range(n):
        f = base + rng.normal(0, noise, size=(H, W))
        for b in range(blobs):
            cy, cx = pos[b] + vel[b] * t
            m = ((yy - cy) / ax[b, 0]) ** 2 + ((xx - cx) / ax[b, 1]) ** 2 <= 1.0
            f[m] -= depth[b]
        frames[t] = f
    return np.clip(np.rint(frames), 0, 255).astype(np.uint8)


def sha256(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()

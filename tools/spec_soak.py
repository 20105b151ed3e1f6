"""Soak test of the pass's speculations: many random windows, default context against one with every guess switched
off (and against the A/Y-state pass, where U never leaves f64); outputs must be identical, and the number of reruns the
guesses caused is reported PER KIND OF SCENE -- the bench stream is sigma = 2.5 Gaussian sensor noise, decoded video may
be quieter or blocky:

    gauss      Gaussian noise, sigma in {1.5, 2.5, 4}
    quiet      Gaussian noise, sigma in {0.1, 0.25, 0.5}
    static     no noise at all: frames differ only by the birds (rank-deficient windows: the zero-direction rule)
    codec      sigma 1.5 noise, then every 8 x 8 block flattened towards its mean and the image quantised in steps of
               2 or 4 grey levels (what a low-bitrate intra codec leaves)

    python3 tools/spec_soak.py [trials per kind]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swiftwatcher_amd import _lib, synthetic          # noqa: E402


def make(kind, rng, seed, n, Hc, Wc, birds):
    sigma = {"gauss": rng.choice([1.5, 2.5, 4.0]), "quiet": rng.choice([0.1, 0.25, 0.5]), "static": 0.0, "codec": 1.5}[kind]
    roi = synthetic.roi_window(seed, n, Hc, Wc, birds=birds, noise=float(sigma), bird_len=(8, 20), bird_wid=(3, 9))
    if kind == "codec":
        q = int(rng.choice([2, 4]))
        f = roi.astype(np.float64)
        hb, wb = Hc // 8 * 8, Wc // 8 * 8
        blk = f[:, :hb, :wb].reshape(n, hb // 8, 8, wb // 8, 8, 3)
        mean = blk.mean(axis=(2, 4), keepdims=True)
        f[:, :hb, :wb] = (mean + 0.35 * (blk - mean)).reshape(n, hb, wb, 3)
        roi = np.clip(np.rint(f / q) * q, 0, 255).astype(np.uint8)
    return roi


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    rng = np.random.default_rng(1)
    a = _lib.Context(0)
    b = _lib.Context(0); b.set_sparse_speculation(0); b.set_norm_speculation(0); b.set_integer_start(0)
    c = _lib.Context(0); c.set_ialm_variant(2)
    out = {}
    for kind in ("gauss", "quiet", "static", "codec"):
        windows = batches = reruns = mism = 0
        iters = []
        for trial in range(trials):
            n = int(rng.choice([7, 21, 21, 33, 64, 64]))
            Hc, Wc = [(64, 96), (107, 214), (212, 424), (96, 160), (120, 200)][int(rng.integers(0, 5))]
            if Hc * Wc * n < 1.2e5:
                continue
            nwin = int(rng.integers(1, 4))
            birds = int(rng.integers(1, 14))
            roi = np.concatenate([make(kind, rng, 1000 + 10 * trial + w, n, Hc, Wc, birds) for w in range(nwin)])
            before = a.redo_batches
            ra = a.batch_run(roi, nwin, n, stages=("rpca", "labels"))
            if a.redo_batches != before:
                reruns += 1
                print("RERUN", kind, "trial", trial, "n", n, "roi", Hc, Wc, "nwin", nwin, "birds", birds, "iters", ra["iters"].tolist())
            rb = b.batch_run(roi, nwin, n, stages=("rpca", "labels"))
            rc = c.batch_run(roi, nwin, n, stages=("rpca", "labels"))
            same = all(np.array_equal(ra[k], r[k]) for r in (rb, rc) for k in ("iters", "rpca", "labels"))
            if not same:
                mism += 1
                print("MISMATCH", kind, "trial", trial, n, Hc, Wc, ra["iters"], rb["iters"], rc["iters"],
                      int((ra["rpca"] != rb["rpca"]).sum()), int((ra["rpca"] != rc["rpca"]).sum()))
            windows += nwin
            batches += 1
            iters += [int(i) for i in ra["iters"]]
        out[kind] = {"batches": batches, "windows": windows, "reruns": reruns, "mismatching_batches": mism,
                     "iters_min": min(iters), "iters_max": max(iters)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

"""Soak test of the pass's speculations: many random windows, default context against one with every guess switched
off; outputs must be identical, and the number of reruns the guesses caused is reported."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swiftwatcher_amd import _lib, synthetic
rng = np.random.default_rng(1)
a = _lib.Context(0)
b = _lib.Context(0); b.set_sparse_speculation(0); b.set_norm_speculation(0); b.set_integer_start(0)
c = _lib.Context(0); c.set_ialm_variant(2)          # A/Y-state pass: U never leaves f64
cases = mism = 0
iters = []
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    n = int(rng.choice([7, 21, 21, 33, 64, 64]))
    Hc, Wc = [(64, 96), (107, 214), (212, 424), (96, 160), (120, 200)][int(rng.integers(0, 5))]
    if Hc * Wc * n < 1.2e5:
        continue
    nwin = int(rng.integers(1, 4))
    birds = int(rng.integers(0, 14))
    noise = float(rng.choice([0.5, 1.5, 2.5, 4.0]))
    roi = np.concatenate([synthetic.roi_window(1000 + 10 * trial + w, n, Hc, Wc, birds=birds, noise=noise,
                                               bird_len=(8, 20), bird_wid=(3, 9)) for w in range(nwin)])
    before = a.redo_batches
    ra = a.batch_run(roi, nwin, n, stages=("rpca", "labels"))
    if a.redo_batches != before:
        print("RERUN trial", trial, "n", n, "roi", Hc, Wc, "nwin", nwin, "birds", birds, "noise", noise, "iters", ra["iters"].tolist())
    rb = b.batch_run(roi, nwin, n, stages=("rpca", "labels"))
    rc = c.batch_run(roi, nwin, n, stages=("rpca", "labels"))
    if not (np.array_equal(ra["iters"], rc["iters"]) and np.array_equal(ra["rpca"], rc["rpca"])):
        mism += 1
        print("MISMATCH vs A/Y-state trial", trial, n, Hc, Wc, noise, ra["iters"], rc["iters"], int((ra["rpca"] != rc["rpca"]).sum()))
    cases += nwin
    iters += [int(i) for i in ra["iters"]]
    ok = np.array_equal(ra["iters"], rb["iters"]) and np.array_equal(ra["rpca"], rb["rpca"]) and np.array_equal(ra["labels"], rb["labels"])
    if not ok:
        mism += 1
        print("MISMATCH trial", trial, n, Hc, Wc, ra["iters"], rb["iters"])
print(json.dumps({"windows": cases, "mismatching_batches": mism, "reruns": a.redo_batches, "iters_min": min(iters), "iters_max": max(iters)}))

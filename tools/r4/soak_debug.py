"""Debug: one batch of tools/spec_soak.py (gauss trial 8) under several context settings against the oracle's iteration counts."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from swiftwatcher_amd import _lib
from oracle import reference_path as orc
import spec_soak
rng = np.random.default_rng(1)
# replay the generator's draws up to gauss trial 8
target = None
for kind in ("gauss",):
    for trial in range(24):
        n = int(rng.choice([7, 21, 21, 33, 64, 64]))
        Hc, Wc = [(64, 96), (107, 214), (212, 424), (96, 160), (120, 200)][int(rng.integers(0, 5))]
        if Hc * Wc * n < 1.2e5:
            continue
        nwin = int(rng.integers(1, 4)); birds = int(rng.integers(1, 14))
        roi = np.concatenate([spec_soak.make(kind, rng, 1000 + 10 * trial + w, n, Hc, Wc, birds) for w in range(nwin)])
        if trial in (2, 8):
            print("trial", trial, n, Hc, Wc, nwin, birds)
            ref = []
            for w in range(nwin):
                g = np.stack([orc.bgr2gray(f) for f in roi[w * n:(w + 1) * n]]).reshape(n, -1).T
                ref.append(orc.ialm_defined(g, return_iters=True)[2])
            print("  oracle iters", ref)
            for label, setup in (("default", lambda c: None),
                                 ("guesses off", lambda c: (c.set_sparse_speculation(0), c.set_norm_speculation(0))),
                                 ("guesses off, f64 start", lambda c: (c.set_sparse_speculation(0), c.set_norm_speculation(0), c.set_integer_start(0))),
                                 ("guesses off, f64 start, guard off", lambda c: (c.set_sparse_speculation(0), c.set_norm_speculation(0), c.set_integer_start(0), c.set_norm_guard(0))),
                                 ("f64 start only", lambda c: c.set_integer_start(0)),
                                 ("f64 start, refine off", lambda c: (c.set_integer_start(0), c.set_start_refine(0))),
                                 ("A/Y pass", lambda c: c.set_ialm_variant(2)),
                                 ("A/Y pass, f64 start", lambda c: (c.set_ialm_variant(2), c.set_integer_start(0)))):
                c = _lib.Context(0); setup(c)
                r = c.batch_run(roi, nwin, n, stages=("rpca",))
                rat, err = c.last_stopping_norms()
                print("  %-36s iters %s guard %d redo %d refined %s ratios %s bound %s" % (label, r["iters"].tolist(), c.guard_windows, c.redo_windows, c.refined_windows, np.round(rat / 1e-3, 4).tolist(), np.round(err, 6).tolist()))
                c.close()

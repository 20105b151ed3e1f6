"""Round 4: what the HIP path does on the ill-conditioned reference-made fixtures (few pixels per frame, 64 frames): max |dA|, |dE| against
the reference's values, iteration count, u8 mismatches of the hot path -- per IALM variant and small-matrix solver.
    python tools/r4/cond_probe.py            (GPU box)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from swiftwatcher_amd import _lib          # noqa: E402

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden")
ctx = _lib.Context(0)
for name in ("ialm_40x48x64", "ialm_47x94x64", "ialm_47x94x64_quiet", "ialm_30x40x64", "ialm_64x96x64", "ialm_47x94x21"):
    g = np.load(os.path.join(G, name + ".npz"))
    frames = g["frames"]
    n, H, W = frames.shape
    rows = g["rows"]
    for variant, tau in ((0, 0.0), (0, 1e-5), (1, 1e-5)):
        for method in (0, 1):
            ctx.set_ialm_variant(variant)
            ctx.set_eig_method(method)
            ctx.set_start_refine(tau)
            before = ctx.refined_windows
            A, E, iters = ctx.ialm(frames.reshape(n, H * W))
            after = ctx.refined_windows
            sp = ctx.rpca_epilogue(E).T.reshape(n, H, W)
            print("%-22s variant %d solver %d refine %.0e: iters %d (ref %d)  max|dA| %.2e  max|dE| %.2e  u8 mismatches (A/E route) %d  refined %d unrefined %d" % (
                name, variant, method, tau, iters, int(g["iters"]), np.abs(A[rows] - g["A_rows"]).max(), np.abs(E[rows] - g["E_rows"]).max(),
                int((sp != g["sparse"]).sum()), after[0] - before[0], after[1] - before[1]), flush=True)
    ctx.set_ialm_variant(0)
    ctx.set_eig_method(0)
    ctx.set_start_refine(1e-5)
    before = ctx.refined_windows
    res = ctx.batch_run(np.ascontiguousarray(frames), 1, n, stages=("gray", "rpca"))
    assert np.array_equal(res["gray"], frames)
    after = ctx.refined_windows
    print("%-22s hot path: iters %d  u8 mismatches %d  refined %d unrefined %d" % (name, int(res["iters"][0]), int((res["rpca"] != g["sparse"]).sum()),
                                                                                 after[0] - before[0], after[1] - before[1]), flush=True)

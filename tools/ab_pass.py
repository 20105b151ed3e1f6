"""A/B of the IALM pass kernels and their knobs on ONE box, one process, same data (boxes differ by 10-15 %):
    python3 tools/ab_pass.py [--n 64] [--windows 128] [--size P2] [--rounds 2] v2:0 v5:0 v5:1 v4:0 v4:1 v4:3
Each config = variant:tune.  Prints ms per step, ms per pass launch, per-family kernel ms."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                              # noqa: E402
from swiftwatcher_amd import _lib, synthetic              # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=64)
ap.add_argument("--windows", type=int, default=128)
ap.add_argument("--size", default="P2")
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("configs", nargs="+")
args = ap.parse_args()
geo = getattr(synthetic, args.size)
Hc, Wc, n, nwin = geo["Hc"], geo["Wc"], args.n, args.windows
F, P = nwin * n, Hc * Wc
dev = torch.device("cuda", 0)
frames = synthetic.roi_stream_torch(dev, F, Hc, Wc, bird_len=geo["bird_len"], bird_wid=geo["bird_wid"])
labels = torch.empty((F, Hc, Wc), dtype=torch.uint8, device=dev)
segs = torch.empty((F, 64, 48), dtype=torch.uint8, device=dev)
nseg = torch.empty((F,), dtype=torch.int32, device=dev)
iters = torch.empty((nwin,), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
inp = _lib.Input(frames=frames.data_ptr(), mem=_lib.MEM_DEVICE, channels=3, nwin=nwin, n=n, Hc=Hc, Wc=Wc,
                 x0=0, y0=0, frame_stride=P * 3, row_stride=Wc * 3)
out = _lib.Output(mem=_lib.MEM_DEVICE, seg_cap=64)
out.labels, out.segs, out.nseg, out.iters = labels.data_ptr(), segs.data_ptr(), nseg.data_ptr(), iters.data_ptr()
params = _lib.default_params()
ref = None
for rnd in range(args.rounds):
    for cfg in args.configs:
        variant, tune = (int(v) for v in cfg.lstrip("v").split(":"))
        ctx = _lib.Context(0, nwin, n, Hc, Wc)
        ctx.set_ialm_variant(variant)
        ctx.set_pass_tuning(tune)
        ctx.batch_run_raw(inp, params, out)
        ctx.prof_enable(True)
        ctx.prof_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ctx.batch_run_raw(inp, params, out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        prof = ctx.prof()
        sig = (int(iters.sum().item()), int(nseg.sum().item()), int(labels.to(torch.int64).sum().item()))
        if ref is None:
            ref = sig
        pm, pl = prof["ialm_pass"]
        print(json.dumps({"cfg": cfg, "round": rnd, "ms_per_step": round(dt * 1e3, 2), "frames_per_s": round(F / dt, 0),
                          "pass_ms_per_launch": round(pm / max(pl, 1), 4), "pass_launches": pl,
                          "same_results": sig == ref, "redo": ctx.redo_batches,
                          "kernel_ms": {k: round(v[0] / args.steps, 2) for k, v in prof.items() if isinstance(v, tuple) and v[0] > 0}}),
              flush=True)
        ctx.close()

#!/usr/bin/env python3
"""Benchmark of the segment hot path (BASELINE.json configs[1]):

    synthetic 1080p ROI stream (424x212 crop of a 340-px chimney), frame-batch = 64
    (FrameQueue(queue_size=64)), image_filtering HIP kernels, one MI355X per rank.

A "step" is one swk_batch_run over `--windows` independent 64-frame RPCA windows whose BGR ROI
frames are already resident in HBM; products (u8 label planes, per-frame segment records,
iteration counts) stay in HBM.  value = ROI frames/s over all ranks.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract fields + "roofline" + "cpu_baseline").
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
# algorithmic bytes per matrix element and IALM iteration of the streaming pass:
#   variant 2 (A/Y state, SURVEY 8d's figure): X u8 + A,Y f64 read, A,Y f64 written = 33; first iteration reads X only = 17
#   variant 3 (M state, the default):          X u8 + M f64 + U f16 read, M f64 + U f16 written = 21; first = 11;
#                                              the sparse u8 image is needed once per window (+1 B per element, booked
#                                              once: the passes far from convergence do not store it)
PASS_BYTES = {1: (33, 17, 0), 2: (33, 17, 0), 3: (21, 11, 1)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--windows", type=int, default=128, help="64-frame windows per step per GPU")
    ap.add_argument("--n", type=int, default=64, help="frames per window (queue_size)")
    ap.add_argument("--size", default="P2", choices=["P1", "P2", "P3"])
    ap.add_argument("--integer-start", type=int, default=None, help="A/B: swk_set_integer_start (0 = f64 start pass)")
    ap.add_argument("--tol", type=float, default=None, help="experiments: IALM tolerance (reference: 0.001)")
    ap.add_argument("--maxiter", type=int, default=None, help="experiments: IALM iteration cap (reference: 100)")
    ap.add_argument("--sparse-spec", type=float, default=None, help="A/B: swk_set_sparse_speculation factor (0 = stores in every pass)")
    ap.add_argument("--norm-spec", type=float, default=None, help="A/B: swk_set_norm_speculation factor (0 = norm in every pass)")
    ap.add_argument("--classify", action="store_true",
                    help="also classify every segment inside the timed step (SqueezeNet-1.0 on PyTorch-ROCm, random-init "
                         "weights, inputs cut on the device by swk_segment_inputs): BASELINE config 3 without the tracker")
    ap.add_argument("--variant", type=int, default=0, help="IALM kernel variant (0 auto)")
    ap.add_argument("--groups", type=int, default=0, help="IALM window groups (0 auto)")
    ap.add_argument("--eig-method", type=int, default=0, help="0 Newton-Schulz (MFMA), 1 Jacobi")
    ap.add_argument("--eig-cus", type=int, default=-1, help="CUs reserved for the eigen-solve streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-windows", type=int, default=3, help="windows in the CPU baseline sample (about 7 s each at P2, n=64)")
    ap.add_argument("--host-input", action="store_true", help="also time a step fed from host memory (PCIe inclusive)")
    return ap.parse_args()


def cpu_baseline(n, Hc, Wc, nwin, seed=424242):
    """The CPU oracle (numpy + C restatement of the reference path) on a bounded sample of the same
    workload.  kind = "port": the reference's own cv2 stages cannot run anywhere in this image."""
    import numpy as np
    from swiftwatcher_amd import synthetic
    from oracle import reference_path as orc
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    rois = [synthetic.roi_window(seed + w, n, Hc, Wc) for w in range(nwin)]
    t0 = time.perf_counter()
    nseg = 0
    for roi in rois:
        res = orc.window(roi)
        nseg += sum(len(s) for s in res["segments"])
    dt = time.perf_counter() - t0
    return dict(value=round(nwin * n / dt, 3), unit="frames/s", cores=int(threads), kind="port",
                sample="%d window(s) of %d frames at %dx%d ROI: numpy/LAPACK SVD IALM (BLAS threads=%d) + "
                       "single-threaded C for the byte stages, %.1f s" % (nwin, n, Wc, Hc, threads, dt))


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist
    from swiftwatcher_amd import _lib, synthetic
    from swiftwatcher_amd import distributed as swd

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # RCCL ("nccl" on ROCm); no-op for a single process.  SWK_DIST_BACKEND=gloo lets several ranks share one GPU
    # for rehearsals (RCCL refuses two ranks on one device).
    rank, world, local = swd.init(os.environ.get("SWK_DIST_BACKEND", "nccl"))
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    geo = getattr(synthetic, args.size)
    Hc, Wc, n, nwin = geo["Hc"], geo["Wc"], args.n, args.windows
    F, P = nwin * n, Hc * Wc

    # ---- synthetic stream, resident in HBM (each rank its own "videos": weak scaling) ----
    frames = synthetic.roi_stream_torch(dev, F, Hc, Wc, seed=20190816 + 1000 * rank,
                                        bird_len=geo["bird_len"], bird_wid=geo["bird_wid"])
    labels = torch.empty((F, Hc, Wc), dtype=torch.uint8, device=dev)
    seg_cap = 64
    segs = torch.empty((F, seg_cap, 48), dtype=torch.uint8, device=dev)
    nseg = torch.empty((F,), dtype=torch.int32, device=dev)
    iters = torch.empty((nwin,), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    ctx = _lib.Context(local, nwin, n, Hc, Wc)
    ctx.set_ialm_variant(args.variant)
    ctx.set_ialm_groups(args.groups)
    ctx.set_eig_cus(args.eig_cus)
    ctx.set_eig_method(args.eig_method)
    if args.integer_start is not None:
        ctx.set_integer_start(args.integer_start)
    if args.sparse_spec is not None:
        ctx.set_sparse_speculation(args.sparse_spec)
    if args.norm_spec is not None:
        ctx.set_norm_speculation(args.norm_spec)
    params = _lib.default_params()
    if args.tol is not None:
        params.tol = args.tol
    if args.maxiter is not None:
        params.maxiter = args.maxiter
    inp = _lib.Input(frames=frames.data_ptr(), mem=_lib.MEM_DEVICE, channels=3, nwin=nwin, n=n, Hc=Hc, Wc=Wc,
                     x0=0, y0=0, frame_stride=P * 3, row_stride=Wc * 3)
    out = _lib.Output(mem=_lib.MEM_DEVICE, seg_cap=seg_cap)
    out.labels = labels.data_ptr()
    out.segs = segs.data_ptr()
    out.nseg = nseg.data_ptr()
    out.iters = iters.data_ptr()

    clf = None
    kept_total = [0, 0]
    if args.classify:
        from swiftwatcher_amd.segment_classification import SegmentClassifier, SqueezeNet10
        torch.manual_seed(20190816)
        clf = SegmentClassifier.from_state_dict(SqueezeNet10(2).state_dict(), device=dev, batch_size=2048)

    def step():
        ctx.batch_run_raw(inp, params, out)      # synchronous on the library's own stream
        if clf is not None:
            scores, fidx = clf.scores_from_device(ctx, inp, (Hc, Wc), segs, nseg, seg_cap)
            keep = torch.max(scores, 1)[1] == 1                      # segment_classification.py:36-39
            per_frame = torch.bincount(fidx[keep].to(torch.int64), minlength=F)
            kept_total[0] = int(per_frame.sum().item())
            kept_total[1] = int(scores.shape[0])

    def fence():
        torch.cuda.synchronize()
        swd.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.prof_enable(True)
    ctx.prof_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    prof = ctx.prof()
    ctx.prof_enable(False)

    # ---- per-rank counts gathered over RCCL (the only collective of the path) ----
    it_host = iters.cpu().numpy()
    nseg_host = nseg.cpu().numpy()
    # every rank is one "video" here: (segments found, IALM iterations, frames processed)
    table = swd.gather_counts({rank: (int(nseg_host.sum()), int(it_host.sum()), F * args.steps)}, world)
    total_frames = int(table[:, 2].sum())
    dt_max = swd.max_over_ranks(dt)

    if rank == 0:
        pass_ms, pass_launches = prof["ialm_pass"]
        # exact algorithmic bytes streamed by the full passes of ONE step (same every step: same data)
        elems = n * P
        variant = args.variant if args.variant else 3
        BYTES_STEADY, BYTES_FIRST, BYTES_ONCE = PASS_BYTES[variant]
        step_bytes = int(sum(BYTES_ONCE + BYTES_FIRST + BYTES_STEADY * (int(k) - 1) for k in it_host if k > 0)) * elems
        total_bytes = step_bytes * args.steps
        if variant == 3:
            # exact: the library books what each window-iteration had to move (the passes far from convergence skip
            # the sparse-image stores and most of the f16 copy of Y/mu)
            total_bytes = int(ctx.pass_bytes_per_element * elems)
        achieved = total_bytes / (pass_ms * 1e-3) / 1e9 if pass_ms > 0 else 0.0
        traffic = None
        pmc_file = os.path.join(ROOT, "profiles", "pmc_ialm_pass.json")
        if os.path.exists(pmc_file):
            try:       # PMC passes are separate rocprofv3 runs (tools/pmc_pass.sh); valid for the same n and ROI
                pmc = json.load(open(pmc_file))
                if pmc.get("n") == n and pmc.get("P") == P and pmc.get("variant", 2) == variant:
                    traffic = int(pmc["hbm_bytes_per_window_pass"] * nwin)
            except Exception:
                traffic = None
        res = {
            "metric": "frames/sec (segment+classify) on 1080p ROI batches" if clf else "frames/sec (segment) on 1080p ROI batches",
            "value": round(total_frames / dt_max, 2),
            "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "synthetic 1080p ROI stream %dx%d, frame-batch=%d, %d windows/step/GPU, "
                                   "image_filtering HIP kernels (gray, RPCA/IALM, bilateral, threshold, opening, CCL, "
                                   "region props)" % (Wc, Hc, n, nwin),
                       "roi": [Wc, Hc], "frame_batch": n, "windows_per_step": nwin,
                       "ialm_iters_mean": round(float(it_host.mean()), 2),
                       "segments_per_frame": round(float(nseg_host.mean()), 2),
                       "classify": ("every segment through SqueezeNet-1.0 (fp32, receptive-field cropped, random-init "
                                    "weights): %d segments/step, %d kept" % (kept_total[1], kept_total[0])) if clf else
                                   "not in this config (BASELINE configs[1] is image_filtering only; --classify adds it)",
                       "parallelism": "windows sharded per GPU, no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": "ialm_pass", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "launches": int(pass_launches), "avg_launch_ms": round(pass_ms / max(pass_launches, 1), 4),
                         "bytes_per_launch": int(total_bytes / max(pass_launches, 1)),
                         "bytes_per_element_iteration": round(total_bytes / max(elems * float(it_host.sum()) * args.steps, 1.0), 2),
                         "pass_variant": variant,
                         # the same element-iterations priced at SURVEY 8(d)'s 33 B (the A/Y formulation this kernel
                         # replaces): informational, NOT what "achieved" uses
                         "survey_pricing": {"bytes_per_element_iteration": 33,
                                            "achieved": round(33.0 * elems * float(it_host.sum()) * args.steps / (pass_ms * 1e-3) / 1e9, 1)
                                            if pass_ms > 0 else 0.0}},
            "kernel_ms_per_step": {k: round(v[0] / args.steps, 3) for k, v in prof.items() if isinstance(v, tuple)},
        }
        if args.host_input:
            host = frames.cpu().numpy()
            hin = _lib.Input(frames=host.ctypes.data, mem=_lib.MEM_HOST, channels=3, nwin=nwin, n=n, Hc=Hc, Wc=Wc,
                             x0=0, y0=0, frame_stride=P * 3, row_stride=Wc * 3)
            ctx.batch_run_raw(hin, params, out)
            t1 = time.perf_counter()
            ctx.batch_run_raw(hin, params, out)
            res["pcie_inclusive_frames_per_s"] = round(F / (time.perf_counter() - t1), 2)
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(n, Hc, Wc, args.cpu_windows)
        print(json.dumps(res), flush=True)
    ctx.close()
    if world > 1:
        swd.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

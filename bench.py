#!/usr/bin/env python3
"""Benchmark of the segment + classify hot path (BASELINE.json metric, workload = configs[1]'s stream):

    synthetic 1080p ROI stream (424x212 crop of a 340-px chimney), frame-batch = 64
    (FrameQueue(queue_size=64)), image_filtering HIP kernels + the segment_classification CNN
    (SqueezeNet-1.0 on PyTorch-ROCm, fp32, eval mode), one MI355X per rank.

A "step" is one pass of the path over `--windows` independent 64-frame RPCA windows whose BGR ROI frames are already
resident in HBM: swk_batch_run (gray, RPCA/IALM, bilateral, threshold, opening, CCL, region properties), then every
segment of every frame through the classifier (inputs cut on the device by swk_segment_inputs) and the
keep-if-argmax==1 rule (segment_classification.py:36-39) reduced to kept segments per frame -- the reference's
__main__.py:77-85 for the whole batch.  value = ROI frames/s over all ranks.

After the headline loop a second timed loop runs the image_filtering part alone (BASELINE configs[1] as worded:
"image_filtering HIP kernels") and reports it, with the streaming kernel's roofline, under "segment_only".

After that, on rank 0 at N = 1, the reference's own call pattern is timed from host 1080p frames (sub-results, never `value`):
"drop_in" = FrameQueue() as the CLI constructs it, push + preprocess_queue + segment_queue per 21-frame window
(__main__.py:73-78); "count_loop" = swift_counting_algorithm (__main__.py:56-100) with the classifier ON (the calibrated head),
tracker and events; "pcie_inclusive" = a segment step fed from host memory.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N ...            (starts N ranks itself: one fresh process per GPU, RCCL; this process never touches a GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract fields + "roofline" + "roofline_cnn" + "segment_only" + "drop_in" + "count_loop" +
"pcie_inclusive" + "cpu_baseline").
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
F32_MATRIX_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_* at the f32 vector rate (155 measured)
F64_MATRIX_PEAK_TFLOPS = 78.6   # v_mfma_f64_16x16x4_f64: 64 cycles per instruction per SIMD (measured) = 32 flop/clk/SIMD at 2.4 GHz
# algorithmic bytes per matrix element and IALM iteration of the streaming pass:
#   variant 2 (A/Y state, SURVEY 8d's figure): X u8 + A,Y f64 read, A,Y f64 written = 33; first iteration reads X only = 17
#   class 3 (M state: variants 4 / 5, the default):          X u8 + M f64 + U f16 read, M f64 + U f16 written = 21; first = 11;
#                                              the sparse u8 image is needed once per window (+1 B per element, booked
#                                              once: the passes far from convergence do not store it)
PASS_BYTES = {1: (33, 17, 0), 2: (33, 17, 0), 3: (21, 11, 1)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--windows", type=int, default=128, help="windows per step per GPU")
    ap.add_argument("--n", type=int, default=64, help="frames per window (queue_size)")
    ap.add_argument("--size", default="P2", choices=["P1", "P2", "P3"])
    ap.add_argument("--integer-start", type=int, default=None, help="A/B: swk_set_integer_start (0 = f64 start pass)")
    ap.add_argument("--tol", type=float, default=None, help="experiments: IALM tolerance (reference: 0.001)")
    ap.add_argument("--maxiter", type=int, default=None, help="experiments: IALM iteration cap (reference: 100)")
    ap.add_argument("--sparse-spec", type=float, default=None, help="A/B: swk_set_sparse_speculation factor (0 = stores in every pass)")
    ap.add_argument("--norm-spec", type=float, default=None, help="A/B: swk_set_norm_speculation factor (0 = norm in every pass)")
    ap.add_argument("--classify", action="store_true", help="(default) kept for compatibility: the classifier is part of the step")
    ap.add_argument("--no-classify", action="store_true",
                    help="time the image_filtering part only (the headline metric then reads 'segment'); for kernel A/Bs")
    ap.add_argument("--full-network", action="store_true", help="A/B: the full 224x224 forward instead of the receptive-field cropped one")
    ap.add_argument("--weights", default="auto", choices=["auto", "model_pt", "calibrated"],
                    help="classifier weights: model.pt's tensors (tests/golden/classifier_model_pt.npz; 'auto' takes them when the file is "
                         "there) or random-init weights with a head bias calibrated on the stream")
    ap.add_argument("--cls-batch", type=int, default=8192, help="segments per classifier forward (4096: -2.5 %; 12288: +0.4 %)")
    ap.add_argument("--overlap", action="store_true",
                    help="also time the steps as a two-stage pipeline: the image_filtering part of step i+1 (library stream, "
                         "worker thread) runs while the classifier works on step i (PyTorch's stream); reported under 'overlapped'")
    ap.add_argument("--no-cudnn-benchmark", action="store_true",
                    help="A/B: MIOpen's immediate-mode kernel choice instead of its exhaustive find (the classifier's default)")
    ap.add_argument("--variant", type=int, default=0, help="IALM kernel variant (0 auto)")
    ap.add_argument("--tune", type=int, default=None, help="A/B: swk_set_pass_tuning flags (variants 4 / 5)")
    ap.add_argument("--eig-method", type=int, default=0, help="0 Newton-Schulz (MFMA), 1 Jacobi")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-windows", type=int, default=2, help="windows in the CPU baseline sample (about 7 s each at P2, n=64)")
    ap.add_argument("--host-input", action="store_true", help="(default at N = 1) kept for compatibility: pcie_inclusive is part of the line")
    ap.add_argument("--no-drop-in", action="store_true", help="skip the drop_in / count_loop / pcie_inclusive sub-results (kernel A/Bs)")
    ap.add_argument("--loop-windows", type=int, default=24, help="21-frame windows of the count_loop clip (1080p frames in host memory: 130 MB each)")
    ap.add_argument("--launch-timeout", type=int, default=1500, help="seconds the self-launched ranks of --gpus N may take before they are ended")
    ap.add_argument("--parity-windows", type=int, default=8,
                    help="21-frame windows at the head of the count_loop clip that also go through the CPU restatement's pipeline (count_oracle)")
    ap.add_argument("--video-windows", type=int, default=24, help="21-frame windows per video of the video_sharded leg (0 = skip the leg)")
    ap.add_argument("--videos-per-gpu", type=int, default=1)
    ap.add_argument("--video-config", default="auto", choices=["auto", "4", "5"],
                    help="video_sharded: BASELINE config 4 (1080p frames in host memory) or 5 (4K videos as ROI stream files); auto = 4 up to "
                         "four ranks, 5 beyond")
    ap.add_argument("--verify-all-videos", action="store_true", help="video_sharded: rank 0 recounts every video (default: the first and the last)")
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="launcher rehearsal without any GPU work: the ranks join the process group (SWK_DIST_BACKEND, gloo on CPU-only "
                         "hosts), run the barrier / max-over-ranks clock / count gather around EMPTY steps and print the line with value 0")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: this process -- which must not touch a GPU (a process that has initialised HIP
    may not exec or fork safely) -- starts N fresh interpreters, one rank each (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, what
    torch.distributed.run would set), relays rank 0's JSON line and returns the worst exit code."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    import tempfile
    procs, logs = [], []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SWK_BENCH_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = tempfile.TemporaryFile()
        err = tempfile.TemporaryFile()
        logs.append((out, err))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out, stderr=err))
    # every child is watched: a rank that dies before it joins the process group would leave the others in init / the barrier for
    # ever.  The first non-zero exit (or the overall limit) ends the rest -- they are ordinary children of a process that never
    # touched a GPU, so terminating them is safe -- and every rank's stderr tail is kept.
    deadline = time.monotonic() + args.launch_timeout
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad or time.monotonic() > deadline:
            failed = "rank %d exited with code %d" % (bad[0], codes[bad[0]]) if bad else "no result after %d s" % args.launch_timeout
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
            break
        time.sleep(0.2)
    codes = [p.returncode for p in procs]

    def tail(f, nbytes=4000):
        f.seek(0, 2)
        f.seek(max(f.tell() - nbytes, 0))
        return f.read().decode(errors="replace")

    logs[0][0].seek(0)
    out0 = logs[0][0].read().decode(errors="replace")
    # rank 0's JSON line and nothing else (a backend may chat on stdout: gloo prints its peer count there)
    lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
    if failed is not None or not lines:
        sys.stderr.write("bench.py launcher: %s\n" % (failed or "rank 0 printed no result line"))
        for r, (_, err) in enumerate(logs):
            sys.stderr.write("---- rank %d (exit %s) stderr tail ----\n%s\n" % (r, codes[r], tail(err)))
    else:
        sys.stderr.write(tail(logs[0][1]))
    sys.stdout.write((lines[-1] + "\n") if lines else out0)
    sys.stdout.flush()
    worst = max(abs(c) for c in codes)
    return worst if worst else (1 if failed is not None or not lines else 0)


def rehearse(args):
    """--rehearse-launch: everything of the multi-rank protocol except the GPU work (CPU test of the launcher; gloo)."""
    import torch.distributed as dist
    from swiftwatcher_amd import distributed as swd
    if os.environ.get("SWK_REHEARSE_FAIL_RANK") == os.environ.get("RANK", "0"):
        # (test hook of the launcher: this rank dies before it joins the process group, the others would wait for it for ever)
        sys.stderr.write("rehearsal: rank %s fails on purpose\n" % os.environ.get("RANK", "0"))
        sys.exit(3)
    rank, world, local = swd.init(os.environ.get("SWK_DIST_BACKEND", "gloo"))
    swd.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    swd.barrier()
    dt = swd.max_over_ranks(time.perf_counter() - t0)
    frames = args.windows * args.n * args.steps
    table = swd.gather_counts({rank: (0, 0, frames)}, world)
    if rank == 0:
        print(json.dumps({"metric": "frames/sec (segment+classify) on 1080p ROI batches", "value": 0.0, "unit": "frames/s",
                          "n_gpus": world, "world_size": dist.get_world_size() if dist.is_initialized() else 1,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 6),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "none", "rehearsal": True,
                          "backend": dist.get_backend() if dist.is_initialized() else None,
                          "launched_by": "bench.py" if os.environ.get("SWK_BENCH_LAUNCHED") else "external launcher",
                          "per_rank": [{"rank": int(r), "frames": int(table[r, 2])} for r in range(world)]}), flush=True)
    if world > 1:
        swd.barrier()
        dist.destroy_process_group()


def cpu_baseline(n, Hc, Wc, nwin, classify, seed=424242, sample=None):
    """The CPU oracle (numpy + C restatement of the reference path, + the batch-1 CPU classifier the reference runs)
    on a bounded sample of the same workload.  kind = "port": the reference's own cv2 / torchvision stages cannot run
    anywhere in this image."""
    import numpy as np
    import torch
    from swiftwatcher_amd import synthetic
    from oracle import reference_path as orc
    from oracle import classifier_ref
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    rois = [synthetic.roi_window(seed + w, n, Hc, Wc) for w in range(nwin)]
    t0 = time.perf_counter()
    results = [orc.window(roi) for roi in rois]
    t_seg = time.perf_counter() - t0
    nseg = sum(len(s) for res in results for s in res["segments"])
    if sample is not None:          # the same windows then go through the HIP path: the line's "parity" object (sample_parity)
        sample.update(rois=rois, results=results)
    out = dict(unit="frames/s", cores=int(threads), kind="port")
    t_cls = 0.0
    if classify:
        # every segment of the sample's FIRST window through the batch-1 classifier (segment_classification.py:29-36),
        # scaled to all of the sample's segments
        crop_region = [(0, 0), (Wc, Hc)]
        crops = []
        for pos, segs in enumerate(results[0]["segments"]):
            for s in segs:
                r0, c0, r1, c1 = orc.segment_crop_box(s["bbox"], (24, 24), crop_region)
                crops.append(rois[0][pos][max(r0, 0):max(r1, 0), max(c0, 0):max(c1, 0)])
        sd = classifier_ref.random_state_dict(0)
        crops = crops[:96]                                   # bounded: a batch-1 CPU forward is tens of milliseconds
        # torch's default (one thread per hardware thread) makes a batch-1 forward of these small convolutions slower,
        # not faster, on a 128-thread host: the baseline uses the thread count that is fastest here
        best = None
        for th in (8, 16, 32):
            torch.set_num_threads(min(th, os.cpu_count() or th))
            classifier_ref.classify(sd, crops[:4])
            t1 = time.perf_counter()
            classifier_ref.classify(sd, crops[:24])
            dt = (time.perf_counter() - t1) / 24
            if best is None or dt < best[0]:
                best = (dt, torch.get_num_threads())
        torch.set_num_threads(best[1])
        t1 = time.perf_counter()
        classifier_ref.classify(sd, crops)
        per_seg = (time.perf_counter() - t1) / max(len(crops), 1)
        t_cls = per_seg * nseg
        out["classify_ms_per_segment"] = round(per_seg * 1e3, 3)
        out["classify_sample_segments"] = len(crops)
        out["torch_threads"] = int(torch.get_num_threads())
    out["value"] = round(nwin * n / (t_seg + t_cls), 3)
    out["segment_only_value"] = round(nwin * n / t_seg, 3)
    out["sample"] = ("%d window(s) of %d frames at %dx%d ROI, %d segments: numpy/LAPACK SVD IALM (BLAS threads=%d) + "
                     "single-threaded C for the byte stages, %.1f s%s"
                     % (nwin, n, Wc, Hc, nseg, threads, t_seg,
                        "; batch-1 torch CPU SqueezeNet-1.0 (full 224x224, as the reference, %d torch threads) timed on 96 of the "
                        "first window's segments and scaled to all of them: %.1f s" % (out.get("torch_threads", 0), t_cls) if classify else ""))
    return out


STAGE_IMAGES = ("gray", "rpca", "bilateral", "thresh", "opened", "labels")


def sample_parity(ctx, rois, results, n):
    """Second half of BASELINE.json's metric ("...; swift-count parity vs ref"), kernel level: the cpu_baseline sample's windows --
    whose right answers the CPU restatement of the reference has just produced -- through swk_batch_run, the call the timed loop makes:
    iteration count per window, the six stage images and the region records of every frame compared bit for bit.  Outside the timed
    regions; the CPU side is the checker here, never the thing measured."""
    import numpy as np
    nwin = len(rois)
    res = ctx.batch_run(np.ascontiguousarray(np.concatenate(rois)), nwin, n)
    bad, per_stage, segments = 0, {k: 0 for k in STAGE_IMAGES + ("regions",)}, 0
    for w in range(nwin):
        ref = results[w]
        for i in range(n):
            f = w * n + i
            ok = True
            for key in STAGE_IMAGES:
                if not np.array_equal(res[key][f], ref[key][i]):
                    per_stage[key] += 1
                    ok = False
            got = [(int(g["label"]), int(g["r0"]), int(g["c0"]), int(g["r1"]), int(g["c1"]), int(g["area"]), int(g["sum_r"]), int(g["sum_c"]))
                   for g in res["segs"][f, :res["nseg"][f]]]
            exp = [(g["label"],) + tuple(g["bbox"]) + (g["area"], g["sum_r"], g["sum_c"]) for g in ref["segments"][i]]
            segments += len(exp)
            if got != exp:
                per_stage["regions"] += 1
                ok = False
            bad += 0 if ok else 1
    it_gpu, it_cpu = [int(v) for v in res["iters"]], [int(r["iters"]) for r in results]
    return {"what": "the cpu_baseline sample's windows through swk_batch_run against the CPU restatement of the reference: IALM iteration "
                    "count per window; gray, RPCA sparse, bilateral, threshold, opened and label images and the region records (label, bbox, "
                    "area, centroid sums) of every frame, bit for bit",
            "windows": nwin, "frames": nwin * n, "segments": segments, "mismatches": bad, "mismatching_frames": bad,
            "mismatching_frames_per_stage": per_stage, "iters": it_gpu, "iters_oracle": it_cpu, "iters_equal": it_gpu == it_cpu}


def count_loop_parity(clf, clf_state, flist, crop_region, roi_mask, n, windows):
    """... and at the level the metric names, the swift count: the head of the count_loop clip through the counting loop (HIP path) and
    through the CPU restatement's pipeline (its segments, the batch-1 CPU classifier on the same weights, the same tracker and event
    rules; oracle/pipeline_ref.py, the checker of tests/test_baseline_configs.py).  Belongs to the cpu_baseline leg: CPU reference
    work outside every timed region."""
    from swiftwatcher_amd import pipeline
    from swiftwatcher_amd import event_classification as ec
    from swiftwatcher_amd.io_frames import ArrayReader
    from oracle import pipeline_ref
    sub = flist[:n * windows]
    events = pipeline.swift_counting_algorithm(ArrayReader(list(sub)), crop_region, roi_mask, queue_size=n, classifier=clf, keep_stages=True)
    t0 = time.perf_counter()
    info = pipeline_ref.oracle_frames(sub, crop_region, queue_size=n)
    keep, unsure = None, 0
    if clf_state is not None:
        keep, _, unsure = pipeline_ref.classify_keep(clf_state, info)
    events_orc = pipeline_ref.track(info, roi_mask, keep)
    dt = time.perf_counter() - t0
    same = pipeline_ref.event_signature(events) == pipeline_ref.event_signature(events_orc)
    return {"what": "the first %d windows (%d frames) of the count_loop clip: swift_counting_algorithm on the HIP path against the CPU "
                    "restatement's pipeline (segments, batch-1 CPU classifier on the same weights, tracker, event rules)" % (windows, len(sub)),
            "frames": len(sub), "count": int(ec.count_swifts(events)), "count_oracle": int(ec.count_swifts(events_orc)),
            "events": len(events), "events_oracle": len(events_orc), "events_identical": bool(same),
            "segments_oracle": sum(len(i["segments"]) for i in info),
            "kept_oracle": (sum(sum(k) for k in keep) if keep is not None else None),
            "decisions_within_1e-3_of_a_tie": int(unsure), "oracle_s": round(dt, 1)}


def reference_call_pattern(ctx_device, clf, args, geo, clf_state=None):
    """The reference's own call pattern from host 1080p frames (rank 0, N = 1): sub-results of the line, never `value`."""
    import numpy as np
    from swiftwatcher_amd import pipeline, synthetic
    from swiftwatcher_amd.data_structures import FrameQueue
    from swiftwatcher_amd.io_frames import ArrayReader
    from swiftwatcher_amd import event_classification as ec
    from swiftwatcher_amd import image_filtering as img
    corners = [(790, 620), (1130, 622)]                                  # the 340-px chimney of a 1080p frame
    crop_region = img.generate_crop_region(corners)                      # [(748, 452), (1172, 664)]: 424 x 212
    n = 21
    out = {}
    # ---- drop_in: FrameQueue() exactly as __main__.py:67 constructs it; per window push + preprocess + segment (:73-78) ----
    frames = synthetic.full_frames(3, n, crop_region)
    order = [frames[i] for i in range(n - 1, -1, -1)]
    numbers, stamps = list(range(n)), ["t"] * n
    q = FrameQueue()
    times, nseg = [], 0
    for rep in range(34):
        t0 = time.perf_counter()
        q.push_list_of_frames(order, numbers, stamps)
        q.preprocess_queue(crop_region, (300, 150))
        q.segment_queue((24, 24), crop_region)
        times.append(time.perf_counter() - t0)
        nseg = sum(len(f.segments) for f in q)
        while not q.is_empty():
            q.pop_frame()
    t = sum(times[4:]) / len(times[4:])
    out["drop_in"] = {"what": "FrameQueue() with its default constructor (stage images kept, on the GPU until read): push_list_of_frames + "
                              "preprocess_queue + segment_queue per 21-frame window from host 1080p frames, Python Frame / Segment objects made",
                      "value": round(n / t, 1), "unit": "frames/s", "ms_per_window": round(t * 1e3, 3), "windows_timed": len(times) - 4,
                      "segments_per_window": nseg, "ialm_iters": q.last_iters}
    del frames, order, q
    # ---- count_loop: swift_counting_algorithm (__main__.py:56-100), classifier ON (the bench's calibrated head), tracker, events ----
    # The clip (loop_windows x 21 frames of 1080p, 130 MB per window in host memory) is played `cycles` times in a row for the timed
    # run: a video is hours long, and the first windows of a process pay for memory the loop then recycles (fresh pages are
    # expensive in this sandbox: one batch of Segment objects that had to grow the heap took 20 x longer than the others).
    import gc
    import tempfile
    from swiftwatcher_amd.io_roi_stream import RoiStreamReader, write_roi_stream
    cycles = 3
    # with the reference's trained weights the clip's birds are the kind model.pt takes for swifts (synthetic.SWIFT_LIKE: 21 % kept; of
    # the large dark ellipses of the headline stream it keeps none, and a loop without events would flatter the tracker's share)
    birds = synthetic.SWIFT_LIKE if getattr(args, "swift_like", False) else dict(birds=12)
    clip = synthetic.full_frames(5, n * args.loop_windows, crop_region, **birds)[::-1]          # oldest first
    flist = [clip[i] for i in range(n * args.loop_windows)]
    total = len(flist) * cycles
    roi_mask = np.zeros((212, 424), np.uint8)
    roi_mask[100:, 42:382] = 255
    loop = {}

    def run_loop(make_reader, classifier, **kw):
        # a side measurement must never take the headline line down with it: a failure is reported in its place
        try:
            return run_loop_(make_reader, classifier, **kw)
        except Exception as exc:          # noqa: BLE001
            import traceback
            traceback.print_exc()
            return {"error": repr(exc)}

    def play(reader, classifier, **kw):
        try:
            return pipeline.swift_counting_algorithm(reader, crop_region, roi_mask, queue_size=n, classifier=classifier, keep_stages=True, **kw)
        finally:
            if hasattr(reader, "close"):          # a reader that segments ahead: the batch in flight, the frames its prepared windows hold
                reader.close()

    def run_loop_(make_reader, classifier, **kw):
        events = play(make_reader(1), classifier, **kw)
        del events
        gc.collect()
        reader = make_reader(cycles)
        t0 = time.perf_counter()
        events = play(reader, classifier, **kw)
        dt = time.perf_counter() - t0
        del reader
        return {"value": round(total / dt, 1), "unit": "frames/s", "ms_per_window": round(dt / (args.loop_windows * cycles) * 1e3, 3),
                "frames": total, "events": len(events), "count": int(ec.count_swifts(events))}

    arrays = lambda c: ArrayReader(flist * c)                            # noqa: E731
    loop["reference_pattern"] = run_loop(arrays, clf)
    loop["windows_per_call_8"] = run_loop(arrays, clf, windows_per_call=8)
    loop["reference_pattern_no_classify"] = run_loop(arrays, None)
    # the same unchanged loop over a reader that segments eight queue-fuls ahead in one GPU call (io_frames.PresegmentingReader)
    from swiftwatcher_amd.io_frames import PresegmentingReader
    loop["presegmenting_reader_8"] = run_loop(lambda c: PresegmentingReader(ArrayReader(flist * c), crop_region, queue_size=n, windows=8,
                                                                            device=ctx_device), clf)
    # the same loop fed by a ROI stream file (io_roi_stream.py: the crop region + margin of every frame, 317 KB instead of 6.2 MB;
    # the reader's page-locked blocks are uploaded as they are, the next window is read ahead in a thread)
    with tempfile.TemporaryDirectory() as tmp:
        paths = {c: write_roi_stream(os.path.join(tmp, "clip%d.swkroi" % c), flist * c, crop_region) for c in (1, cycles)}
        stream = lambda c: RoiStreamReader(paths[c], device=ctx_device)  # noqa: E731
        loop["roi_stream"] = run_loop(stream, clf)
        loop["roi_stream_windows_per_call_8"] = run_loop(stream, clf, windows_per_call=8)
        loop["roi_stream_presegmenting_reader_8"] = run_loop(lambda c: PresegmentingReader(stream(c), queue_size=n, windows=8, device=ctx_device), clf)
        if "error" not in loop["roi_stream"]:
            loop["roi_stream"]["input_mb_per_frame"] = round(os.path.getsize(paths[1]) / len(flist) / 1e6, 3)
    if args.parity_windows > 0 and not args.no_cpu_baseline:
        try:
            loop["reference_pattern"]["parity"] = count_loop_parity(clf, clf_state, flist, crop_region, roi_mask, n,
                                                                    min(args.parity_windows, args.loop_windows))
        except Exception as exc:          # noqa: BLE001
            import traceback
            traceback.print_exc()
            loop["reference_pattern"]["parity"] = {"error": repr(exc)}
    out["count_loop"] = dict(loop["reference_pattern"], roi_stream=loop["roi_stream"],
                             roi_stream_windows_per_call_8=loop["roi_stream_windows_per_call_8"],
                             presegmenting_reader_8=loop["presegmenting_reader_8"],
                             roi_stream_presegmenting_reader_8=loop["roi_stream_presegmenting_reader_8"],
                             what="swift_counting_algorithm as __main__.py:56-100 runs it: get_n_frames -> FrameQueue() -> classifier(frame.segments) "
                                  "per popped frame -> the six tracker calls -> events; %d frames of 1080p in host memory played %d times, "
                                  "--classify on (%s; %s)" % (len(flist), cycles, getattr(args, "weights_note", "the bench's classifier"),
                                                               "14 small faint birds per frame" if getattr(args, "swift_like", False) else "12 large birds per frame"),
                             windows_per_call_8=loop["windows_per_call_8"], no_classify=loop["reference_pattern_no_classify"])
    return out


def video_sharded_leg(args, rank, world, local, clf):
    """BASELINE configs 4 / 5 as worded: every rank counts its own VIDEOS end to end -- reader -> FrameQueue -> classifier -> tracker ->
    events -> count (pipeline.swift_counting_algorithm, the loop of __main__.py:56-100, over a reader that segments eight queue-fuls
    ahead) -- videos round-robin over the ranks like the reference's `for src_filepath in src_filepaths` (__main__.py:21) would be
    sharded, and ONE all_gather of (predicted, rejected, frames) per video at the end (RCCL when the ranks own GPUs; io_data.py:113 is
    the count).  N <= 4: 1080p frames in host memory (config 4); N > 4: 4K videos as ROI stream files, 850 x 425 ROI (config 5).
    Every rank calls this (collectives inside); returns the sub-result on rank 0, None elsewhere."""
    import tempfile
    import numpy as np
    import torch
    from swiftwatcher_amd import pipeline, synthetic
    from swiftwatcher_amd import distributed as swd
    from swiftwatcher_amd import event_classification as ec
    from swiftwatcher_amd import image_filtering as img
    from swiftwatcher_amd.io_frames import ArrayReader, PresegmentingReader
    from swiftwatcher_amd.io_roi_stream import RoiFrame, RoiStreamReader, RoiStreamWriter, margin_rect
    big = world > 4 if args.video_config == "auto" else args.video_config == "5"
    n, nf = 21, 21 * args.video_windows
    if big:
        geo, frame_hw, corners = synthetic.P3, (2160, 3840), [(1580, 1240), (2260, 1244)]          # a 680-px chimney of a 4K frame
    else:
        geo, frame_hw, corners = synthetic.P2, (1080, 1920), [(790, 620), (1130, 622)]              # 340 px at 1080p
    crop_region = img.generate_crop_region(corners)
    (x0, y0), (x1, y1) = crop_region
    Hc, Wc = y1 - y0, x1 - x0
    assert (Hc, Wc) == (geo["Hc"], geo["Wc"]), (Hc, Wc)
    roi_mask = np.zeros((Hc, Wc), np.uint8)
    roi_mask[int(0.47 * Hc):, int(0.1 * Wc):int(0.9 * Wc)] = 255
    dev = torch.device("cuda", local)
    videos = world * args.videos_per_gpu
    tmp = tempfile.TemporaryDirectory()

    def make_reader(i):
        """video i: a seeded scene built on the GPU (same statistics as synthetic.roi_window), oldest frame first"""
        birds = synthetic.SWIFT_LIKE if getattr(args, "swift_like", False) else dict(birds=geo["birds"], bird_len=geo["bird_len"], bird_wid=geo["bird_wid"])
        roi = synthetic.roi_stream_torch(dev, nf, Hc, Wc, seed=77000 + i, **birds).cpu().numpy()
        if not big:
            frames = np.full((nf,) + frame_hw + (3,), 128, np.uint8)
            frames[:, y0:y1, x0:x1] = roi
            return ArrayReader([frames[k] for k in range(nf)])
        ya, yb, xa, xb = margin_rect(frame_hw, crop_region)
        path = os.path.join(tmp.name, "video%d.swkroi" % i)
        with RoiStreamWriter(path, frame_hw, crop_region) as w:          # the stored rectangle only: no 25-MB 4K frames are made
            rect = np.full((yb - ya, xb - xa, 3), 128, np.uint8)
            for k in range(nf):
                rect[y0 - ya:y1 - ya, x0 - xa:x1 - xa] = roi[k]
                w.append(RoiFrame(rect, (ya, xa), frame_hw + (3,)))
        return RoiStreamReader(path, device=local)

    def count_video(reader):
        pre = PresegmentingReader(reader, crop_region, queue_size=n, windows=8, device=local)
        try:
            events = pipeline.swift_counting_algorithm(pre, crop_region, roi_mask, queue_size=n, classifier=clf, device=local)
        finally:
            pre.close()          # the batch it was still segmenting ahead, and the frames its prepared windows hold
        count = int(ec.count_swifts(events))
        return (count, len(events) - count, nf)

    def rewind(reader):
        if isinstance(reader, RoiStreamReader):          # (closed by the presegmenting reader's close())
            return RoiStreamReader(reader.filepath, device=local)
        return ArrayReader(reader.frames)

    mine = swd.shard(videos, rank, world)
    failed, local_counts, dt, plays, readers = 0, {}, 0.0, [], {}
    try:
        readers = {i: make_reader(i) for i in mine}
        for i in mine:                                   # untimed first play: allocations, HIP graphs of the window sizes, page faults
            count_video(readers[i])
            readers[i] = rewind(readers[i])
        # three timed plays of every rank's videos, the median reported: a play is 50-200 ms of threads handing frames to each
        # other, and single plays scatter by a factor of three on a busy host
        plays = []
        for _ in range(3):
            torch.cuda.synchronize()
            swd.barrier()
            t0 = time.perf_counter()
            for i in mine:
                local_counts[i] = count_video(readers[i])
                readers[i] = rewind(readers[i])
            torch.cuda.synchronize()
            swd.barrier()
            plays.append(time.perf_counter() - t0)
        dt = sorted(plays)[1]
    except Exception:          # noqa: BLE001  (every rank must still reach the collectives below)
        import traceback
        traceback.print_exc()
        failed = 1
        local_counts = {i: (-1, -1, 0) for i in mine}
    failed = int(swd.max_over_ranks(failed))
    dt_max = swd.max_over_ranks(dt)
    plays_max = [swd.max_over_ranks(p) for p in (plays if not failed and len(plays) == 3 else [0.0, 0.0, 0.0])]
    table = swd.gather_counts(local_counts, videos)          # THE collective of the path: (predicted, rejected, frames) per video
    out = None
    if rank == 0:
        out = {"what": "config %s: %d video(s) of %d frames, one per GPU, each counted end to end by its rank (reader that segments 8 "
                       "queue-fuls ahead -> FrameQueue -> classifier -> tracker -> events -> count), per-video (predicted, rejected, "
                       "frames) gathered with one all_gather over %s"
                       % ("5 (4K videos as ROI stream files, 850x425 ROI)" if big else "4 (1080p frames in host memory, 424x212 ROI)",
                          videos, nf, (torch.distributed.get_backend() if torch.distributed.is_initialized() else "no process group (one rank)")),
               "videos": videos, "frames_per_video": nf, "classifier": getattr(args, "weights_note", None) if clf is not None else "off"}
        if failed:
            out["error"] = "a rank failed (see stderr)"
        else:
            total = int(table[:, 2].sum())
            out.update(value=round(total / dt_max, 1), unit="frames/s", ms=round(dt_max * 1e3, 2), plays_ms=[round(p * 1e3, 2) for p in plays_max],
                       per_video_counts=[[int(v) for v in row] for row in table.tolist()])
            # what a single process gets for the same videos (the first and the last; every one with --verify-all-videos)
            check = list(range(videos)) if args.verify_all_videos else sorted({0, videos - 1})
            same = True
            for i in check:
                same = same and tuple(int(v) for v in table[i].tolist()) == count_video(make_reader(i))
            out["counts_equal_single_rank"] = bool(same)
            out["verified_videos"] = check
    del readers
    tmp.cleanup()
    import gc
    gc.collect()          # 3 GB of frames per video go back before the host-side sub-results are timed
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    if args.rehearse_launch:
        return rehearse(args)
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
    if world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started %d ranks" % (args.gpus, world))
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    classify = not args.no_classify
    if args.no_cudnn_benchmark:
        os.environ["SWK_CUDNN_BENCHMARK"] = "0"

    geo = getattr(synthetic, args.size)
    Hc, Wc, n, nwin = geo["Hc"], geo["Wc"], args.n, args.windows
    F, P = nwin * n, Hc * Wc

    # ---- synthetic stream, resident in HBM (each rank its own "videos": weak scaling) ----
    frames = synthetic.roi_stream_torch(dev, F, Hc, Wc, seed=20190816 + 1000 * rank, birds=geo["birds"],
                                        bird_len=geo["bird_len"], bird_wid=geo["bird_wid"])
    labels = torch.empty((F, Hc, Wc), dtype=torch.uint8, device=dev)
    seg_cap = 64
    segs = torch.empty((F, seg_cap, 48), dtype=torch.uint8, device=dev)
    nseg = torch.empty((F,), dtype=torch.int32, device=dev)
    iters = torch.empty((nwin,), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    ctx = _lib.Context(local, nwin, n, Hc, Wc)
    ctx.set_ialm_variant(args.variant)
    ctx.set_eig_method(args.eig_method)
    if args.tune is not None:
        ctx.set_pass_tuning(args.tune)
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

    def segment_step():
        ctx.batch_run_raw(inp, params, out)      # synchronous on the library's own stream

    clf, sd, use_model_pt = None, None, False
    kept_total = [0, 0]
    if classify:
        from swiftwatcher_amd.segment_classification import SegmentClassifier, SqueezeNet10
        model_pt = os.path.join(ROOT, "tests", "golden", "classifier_model_pt.npz")
        use_model_pt = args.weights == "model_pt" or (args.weights == "auto" and os.path.exists(model_pt))
    if classify and use_model_pt:
        # the reference's own weights (swiftwatcher/model.pt: its 52 tensors travel with the tests as arrays, no checkpoint file does)
        g = np.load(model_pt)
        sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w:")}
        args.weights_note = "model.pt's weights (tests/golden/classifier_model_pt.npz)"
        args.swift_like = True
        clf = SegmentClassifier.from_state_dict(sd, device=dev, batch_size=args.cls_batch, cropped=not args.full_network)
    elif classify:
        # Random-init weights of the architecture (--weights calibrated, or the tensor file is absent).  Random weights put every
        # crop in one class, so the head bias is calibrated on the stream's own segments: with both pre-activations
        # held positive (ReLU inactive) the score difference is affine in the bias, and the median segment is put on
        # the decision boundary -- about half of the segments are then kept, and both branches of the keep rule and
        # the per-frame reduction see real work.
        args.weights_note = "random-init weights with the head bias calibrated on the stream"
        # He-normal weights (torch's default init leaves the head insensitive to the input below float32 resolution)
        gen = torch.Generator().manual_seed(20190816)
        sd = SqueezeNet10(2).state_dict()
        for name, t in sd.items():
            if name.endswith(".weight"):
                fan_in = t.shape[1] * t.shape[2] * t.shape[3]
                sd[name] = torch.randn(t.shape, generator=gen) * (2.0 / fan_in) ** 0.5
            else:
                sd[name] = torch.randn(t.shape, generator=gen) * 0.05
        big = 50.0
        segment_step()
        for attempt in range(8):
            sd["classifier.1.bias"] = torch.tensor([big, big])
            probe = SegmentClassifier.from_state_dict(sd, device=dev, batch_size=args.cls_batch, cropped=not args.full_network)
            s0, _ = probe.scores_from_device(ctx, inp, (Hc, Wc), segs, nseg, seg_cap)
            del probe
            # ReLU inactive everywhere <=> every score sits well inside (0, 2 big); else shrink the head's weights
            # (decisions only depend on the sign of the score difference) and probe again
            if float(s0.min()) > 0.5 * big and float(s0.max()) < 1.5 * big:
                break
            sd["classifier.1.weight"] = sd["classifier.1.weight"] * 0.1
        else:
            raise SystemExit("could not calibrate the classifier head")
        d = (s0[:, 1] - s0[:, 0]).double()
        if float(d.max() - d.min()) < 1e-3:
            raise SystemExit("classifier head does not separate the stream's segments: calibration impossible")
        sd["classifier.1.bias"] = torch.tensor([big, big - float(d.median())])
        clf = SegmentClassifier.from_state_dict(sd, device=dev, batch_size=args.cls_batch, cropped=not args.full_network)

    def step():
        segment_step()
        if clf is not None:
            scores, fidx = clf.scores_from_device(ctx, inp, (Hc, Wc), segs, nseg, seg_cap)
            keep = torch.max(scores, 1)[1] == 1                      # segment_classification.py:36-39
            per_frame = torch.bincount(fidx[keep].to(torch.int64), minlength=F)      # kept segments of every frame
            kept_total[0] = int(per_frame.sum().item())
            kept_total[1] = int(scores.shape[0])

    own_dt = [0.0]

    def fence():
        torch.cuda.synchronize()
        swd.barrier()
        torch.cuda.synchronize()

    def timed(fn):
        """EXACTLY args.steps steps between two fences; library kernel families timed by HIP events on its stream."""
        ctx.prof_enable(True)
        ctx.prof_reset()
        if clf is not None:
            clf.net_time()
            clf.timing = fn is step
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        fence()
        dt = time.perf_counter() - t0
        own_dt[0] = dt if fn is step else own_dt[0]
        prof = ctx.prof()
        bpe = ctx.pass_bytes_per_element
        ctx.prof_enable(False)
        net = clf.net_time() if clf is not None else (0.0, 0, 0)
        if clf is not None:
            clf.timing = False
        return swd.max_over_ranks(dt), prof, bpe, net

    for _ in range(args.warmup):
        step()
    redo0, redo_w0 = ctx.redo_batches, ctx.redo_windows
    dt_max, prof, bpe, net = timed(step)
    overlapped = None
    if args.overlap and clf is not None:
        # Two-stage software pipeline over the SAME K steps: products of step i+1 go to a second set of buffers while
        # the classifier reads step i's.  The classifier cuts its inputs through a second library context (its own
        # stream), so the two stages only meet on the GPU's queues.
        from concurrent.futures import ThreadPoolExecutor
        ctx2 = _lib.Context(local)
        slots = []
        for k in range(2):
            lab2 = labels if k == 0 else torch.empty_like(labels)
            seg2 = segs if k == 0 else torch.empty_like(segs)
            ns2 = nseg if k == 0 else torch.empty_like(nseg)
            it2 = iters if k == 0 else torch.empty_like(iters)
            o = _lib.Output(mem=_lib.MEM_DEVICE, seg_cap=seg_cap)
            o.labels, o.segs, o.nseg, o.iters = lab2.data_ptr(), seg2.data_ptr(), ns2.data_ptr(), it2.data_ptr()
            slots.append((o, seg2, ns2, lab2, it2))
        torch.cuda.synchronize()

        def classify_slot(slot):
            scores, fidx = clf.scores_from_device(ctx2, inp, (Hc, Wc), slot[1], slot[2], seg_cap)
            keep = torch.max(scores, 1)[1] == 1
            per_frame = torch.bincount(fidx[keep].to(torch.int64), minlength=F)
            return int(per_frame.sum().item())

        def pipelined(K):
            kept = 0
            with ThreadPoolExecutor(1) as ex:
                fut = ex.submit(ctx.batch_run_raw, inp, params, slots[0][0])
                for i in range(K):
                    fut.result()
                    if i + 1 < K:
                        fut = ex.submit(ctx.batch_run_raw, inp, params, slots[(i + 1) & 1][0])
                    kept = classify_slot(slots[i & 1])
            return kept
        pipelined(2)
        fence()
        t0 = time.perf_counter()
        kept_o = pipelined(args.steps)
        fence()
        dt_o = swd.max_over_ranks(time.perf_counter() - t0)
        overlapped = {"value": round(F * args.steps * world / dt_o, 2), "ms_per_step": round(dt_o / args.steps * 1e3, 3),
                      "kept_last_step": kept_o, "same_kept_as_serial": kept_o == kept_total[0]}
        ctx2.close()
    # second loop: the image_filtering part alone (same data, same step count)
    seg_only = timed(segment_step) if clf is not None else None
    redo = ctx.redo_batches - redo0

    # ---- configs 4 / 5 as worded: videos sharded over the ranks, per-video counts gathered (every rank takes part) ----
    # (with one rank the leg runs last, below: its gigabytes of host frames would otherwise leave the Python heap in a state that
    #  costs the host-side sub-results -- drop_in, count_loop -- up to half of their throughput in this sandbox)
    video_sharded = video_sharded_leg(args, rank, world, local, clf) if args.video_windows > 0 and world > 1 else None

    # ---- per-rank counts gathered over RCCL (the only collective of the path) ----
    it_host = iters.cpu().numpy()
    nseg_host = nseg.cpu().numpy()
    # every rank is one "video" here: (segments kept or found, IALM iterations, frames processed)
    table = swd.gather_counts({rank: (kept_total[0] if clf else int(nseg_host.sum()), int(it_host.sum()), F * args.steps)}, world)
    total_frames = int(table[:, 2].sum())
    # per-rank clocks (each rank's own elapsed time of the headline loop), gathered the same way
    rank_ms = swd.gather_counts({rank: (int(round(own_dt[0] * 1e6)), 0, 0)}, world)[:, 0]

    if rank == 0:
        elems = n * P
        variant = args.variant if args.variant else 3
        variant = 3 if variant >= 3 else variant          # 3, 4, 5: the M-state pass, same byte accounting

        def pass_roofline(prof, bpe):
            pass_ms, pass_launches = prof["ialm_pass"]
            BYTES_STEADY, BYTES_FIRST, BYTES_ONCE = PASS_BYTES[variant]
            total_bytes = int(sum(BYTES_ONCE + BYTES_FIRST + BYTES_STEADY * (int(k) - 1) for k in it_host if k > 0)) * elems * args.steps
            if variant == 3:
                # exact: the library books what each window-iteration had to move (the passes far from convergence skip
                # the sparse-image stores and most of the f16 copy of Y/mu)
                total_bytes = int(bpe * elems)
            achieved = total_bytes / (pass_ms * 1e-3) / 1e9 if pass_ms > 0 else 0.0
            traffic, traffic_source = None, None
            pmc_file = os.path.join(ROOT, "profiles", "pmc_ialm_pass.json")
            if os.path.exists(pmc_file):
                try:       # PMC passes are separate rocprofv3 runs (tools/pmc_pass.sh); valid for the same n and ROI
                    pmc = json.load(open(pmc_file))
                    if pmc.get("n") == n and pmc.get("P") == P and pmc.get("variant", 2) == variant:
                        traffic = int(pmc["hbm_bytes_per_window_pass"] * nwin)
                        traffic_source = ("profiles/pmc_ialm_pass.json: FETCH_SIZE / WRITE_SIZE of separate rocprofv3 --pmc runs (%s), NOT "
                                          "counters of this run" % pmc.get("measured_at", "round 2"))
                except Exception:
                    traffic = None
            # the same launches against the f64 matrix pipe: MFMAs per 16-pixel tile = NB x NK (A update) + 4 x NB (NB + 1) / 2
            # (symmetric Gram), 2 x 16 x 16 x 4 flop each, for every tile of every window-iteration streamed
            nk, nb = (n + 3) // 4, (n + 15) // 16
            if variant != 3 or (args.variant and args.variant <= 3):
                nk = 4 * nb                                   # block-templated kernels run whole 16-frame blocks
            mfma_per_tile = nb * nk + 2 * nb * (nb + 1)
            tiles = (P + 15) // 16
            flop = float(it_host.sum()) * args.steps * tiles * mfma_per_tile * 2048.0
            f64_tflops = flop / (pass_ms * 1e-3) / 1e12 if pass_ms > 0 else 0.0
            # What bounds the kernel: the f64 execution unit.  f64 MFMAs and f64 vector instructions share it on gfx950
            # (profiles/r2_f64_pipe_probe.txt), the PMC balance of a launch is 69 % MFMA + 17 % VALU busy; HBM is the second floor and
            # is reported beside it ("hbm").  achieved = executed MFMA flops (2 x 16 x 16 x 4 per instruction) / launch time.
            return {"bound": "mfma", "bound_detail": "f64 execution unit (v_mfma_f64_16x16x4_f64 + f64 VALU share one pipe); HBM right behind",
                    "kernel": "ialm_pass", "achieved": round(f64_tflops, 2), "peak": F64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(f64_tflops / F64_MATRIX_PEAK_TFLOPS, 4), "traffic": traffic,
                    # both fractions side by side (rounds 1 / 2 reported the HBM one as `frac`, round 3 on the f64-unit one)
                    "frac_mfma": round(f64_tflops / F64_MATRIX_PEAK_TFLOPS, 4), "frac_hbm": round(achieved / HBM_PEAK_GBS, 4),
                    "hbm": {"achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4)},
                    "launches": int(pass_launches), "avg_launch_ms": round(pass_ms / max(pass_launches, 1), 4),
                    "bytes_per_launch": int(total_bytes / max(pass_launches, 1)),
                    "bytes_per_element_iteration": round(total_bytes / max(elems * float(it_host.sum()) * args.steps, 1.0), 2),
                    "byte_class": variant, "traffic_source": traffic_source,
                    "roofline_definition": "r3: bound = f64 execution unit (executed MFMA flop incl. padded k-steps / launch time); hbm fraction nested",
                    # SURVEY 8(d) prices an element-iteration at 4 n^2 flop (M W and the FULL Gram matrix); the kernel issues the symmetric
                    # half of the Gram product plus the padding of n to whole k-steps: both per pixel and iteration, for comparison
                    "flop_per_pixel_iteration": {"issued": round(mfma_per_tile * 2048.0 / 16.0, 1), "survey_4n2": 4.0 * n * n},
                    "mfma_per_tile": mfma_per_tile,
                    # the same element-iterations priced at SURVEY 8(d)'s 33 B (the A/Y formulation this kernel
                    # replaces): informational, NOT what "achieved" uses
                    "survey_pricing": {"bytes_per_element_iteration": 33,
                                       "achieved": round(33.0 * elems * float(it_host.sum()) * args.steps / (pass_ms * 1e-3) / 1e9, 1)
                                       if pass_ms > 0 else 0.0}}

        def kernel_ms(prof):
            return {k: round(v[0] / args.steps, 3) for k, v in prof.items() if isinstance(v, tuple)}

        workload = ("synthetic 1080p ROI stream %dx%d, frame-batch=%d, %d windows/step/GPU: image_filtering HIP kernels "
                    "(gray, RPCA/IALM, bilateral, threshold, opening, CCL, region props)" % (Wc, Hc, n, nwin))
        res = {
            "metric": "frames/sec (segment+classify) on 1080p ROI batches" if clf else "frames/sec (segment) on 1080p ROI batches",
            "value": round(total_frames / dt_max, 2),
            "unit": "frames/s",
            "n_gpus": world, "world_size": dist.get_world_size() if dist.is_initialized() else 1,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64 (image_filtering), f32 (CNN)" if clf else "f64", "data": "synthetic",
            "config": {"workload": workload + (" + segment_classification CNN on every segment" if clf else ""),
                       "roi": [Wc, Hc], "frame_batch": n, "windows_per_step": nwin,
                       "ialm_iters_mean": round(float(it_host.mean()), 2),
                       "segments_per_frame": round(float(nseg_host.mean()), 2),
                       "classify": ("every segment through SqueezeNet-1.0 (fp32, eval mode, %s, %s): %d segments/step, %d kept"
                                    % ("full 224x224 network" if args.full_network else "receptive-field cropped", args.weights_note,
                                       kept_total[1], kept_total[0])) if clf else "off (--no-classify)",
                       "parallelism": "windows sharded per GPU, no data-path collective",
                       # what changed in the synthetic workloads, by round (records of different revisions are not comparable line by line)
                       "workload_revision": "r4: headline stream unchanged since r1 (12 birds 30-50 x 12-20 px); P3 since r3: 14 birds of the 1080p pixel "
                                            "size; count_loop clip since r4: 14 small faint birds per frame (what model.pt keeps a share of), "
                                            "classifier = model.pt's weights"},
            "redo_batches": int(redo), "redo_windows": int(ctx.redo_windows - redo_w0), "guard_windows": int(ctx.guard_windows),
            "refined_windows": dict(zip(("refined", "given_up"), ctx.refined_windows)),
            "per_rank": [{"rank": r, "frames": int(table[r, 2]), "frames_per_s": round(float(table[r, 2]) / max(float(rank_ms[r]) * 1e-6, 1e-9), 1),
                          "kept_or_found": int(table[r, 0]), "ialm_iterations": int(table[r, 1])} for r in range(world)],
            "launched_by": "bench.py" if os.environ.get("SWK_BENCH_LAUNCHED") else ("torch.distributed.run" if world > 1 else "single process"),
            "roofline": pass_roofline(prof, bpe),
            "kernel_ms_per_step": kernel_ms(prof),
        }
        if clf is not None:
            net_ms, rows, calls = net
            executed, useful = clf.cropped.macs_per_segment() if clf.cropped is not None else (732_600_000, 732_600_000)
            flop = 2.0 * executed * rows
            tf = (lambda f: round(f / (net_ms * 1e-3) / 1e12, 2) if net_ms > 0 else 0.0)
            res["roofline_cnn"] = {
                "bound": "mfma",
                "kernel": "SqueezeNet-1.0 forward, receptive-field cropped: the library's 1x1 / 3x3 / Winograd F(2x2,3x3) kernels on "
                          "v_mfma_f32_32x32x2_f32, conv1 on the library's 7x7 kernel, pools inside the squeezes, the 512 -> 2 head one kernel with a fixed summation order (no MIOpen, no BLAS on the path); forwards of 1,024 rows or more as two chains on two streams over disjoint rows; torch.cuda events on torch's stream around each forward",
                # achieved = the multiply-accumulates the kernels execute (Winograd: 16 per 2x2 outputs instead of 36): what the
                # matrix pipe really does; direct_equivalent prices the same outputs as direct convolutions
                "achieved": tf(flop), "peak": F32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(tf(flop) / F32_MATRIX_PEAK_TFLOPS, 4),
                "direct_equivalent": {"achieved": tf(2.0 * useful * rows), "frac": round(tf(2.0 * useful * rows) / F32_MATRIX_PEAK_TFLOPS, 4)},
                "dtype": "f32", "macs_per_segment_executed": executed, "macs_per_segment_useful": useful,
                "macs_per_segment_full_network": 732_600_000, "rows_per_step": int(rows / args.steps),
                "segments_per_step": kept_total[1], "forwards_per_step": int(calls / args.steps),
                "net_ms_per_step": round(net_ms / args.steps, 3),
                "segments_per_s_in_network": round(rows / (net_ms * 1e-3), 1) if net_ms > 0 else 0.0}
            so_dt, so_prof, so_bpe, _ = seg_only
            if overlapped is not None:
                res["overlapped"] = overlapped
            res["segment_only"] = {"metric": "frames/sec (segment) on 1080p ROI batches", "value": round(total_frames / so_dt, 2),
                                   "ms_per_step": round(so_dt / args.steps * 1e3, 3), "roofline": pass_roofline(so_prof, so_bpe),
                                   "kernel_ms_per_step": kernel_ms(so_prof)}
        if video_sharded is not None:
            res["video_sharded"] = video_sharded
        if world == 1 and not args.no_drop_in:
            # the same segment step fed from host memory (ROI frames in page-locked memory, one copy per batch), 32 windows
            hw_ = min(nwin, 32)
            host = _lib.pinned_empty((hw_ * n, Hc, Wc, 3), np.uint8, device=local)
            host[...] = frames[:hw_ * n].cpu().numpy()
            hin = _lib.Input(frames=host.ctypes.data, mem=_lib.MEM_HOST, channels=3, nwin=hw_, n=n, Hc=Hc, Wc=Wc,
                             x0=0, y0=0, frame_stride=P * 3, row_stride=Wc * 3)
            ctx.batch_run_raw(hin, params, out)
            t1 = time.perf_counter()
            for _ in range(3):
                ctx.batch_run_raw(hin, params, out)
            dt_h = (time.perf_counter() - t1) / 3
            res["pcie_inclusive"] = {"what": "segment step with the BGR ROI frames in page-locked HOST memory (%d windows x %d frames, one "
                                             "host -> device copy per batch inside the timed region)" % (hw_, n),
                                     "value": round(hw_ * n / dt_h, 1), "unit": "frames/s", "input_gb_per_s": round(hw_ * n * P * 3 / dt_h / 1e9, 2)}
            del host
            if args.size == "P2":
                try:
                    res.update(reference_call_pattern(local, clf, args, geo, clf_state=sd if clf is not None else None))
                except Exception as exc:          # noqa: BLE001  (sub-results only: the line above them stands)
                    import traceback
                    traceback.print_exc()
                    res["drop_in"] = res.get("drop_in", {"error": repr(exc)})
                    res["count_loop"] = {"error": repr(exc)}
        if world == 1 and not args.no_cpu_baseline:          # the CPU baseline is a single-GPU-run companion (rank 0 at N = 1 only)
            try:
                sample = {}
                res["cpu_baseline"] = cpu_baseline(n, Hc, Wc, args.cpu_windows, clf is not None, sample=sample)
                res["parity"] = sample_parity(ctx, sample["rois"], sample["results"], n)
            except Exception as exc:          # noqa: BLE001
                import traceback
                traceback.print_exc()
                res["cpu_baseline"] = {"error": repr(exc)}
        if world == 1 and args.video_windows > 0:
            try:
                res["video_sharded"] = video_sharded_leg(args, rank, world, local, clf)
            except Exception as exc:          # noqa: BLE001
                import traceback
                traceback.print_exc()
                res["video_sharded"] = {"error": repr(exc)}
        print(json.dumps(res), flush=True)
    ctx.close()
    if world > 1:
        swd.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

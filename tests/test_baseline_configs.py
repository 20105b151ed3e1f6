"""Every single-GPU workload BASELINE.json names, at its stated size, against the oracle (and, for the float path,
against fixtures the reference itself produced at that size: oracle/make_goldens_r2.py).

  config 1  10 s 480p clip: ROI 94 x 47, queue of 21, 300 frames through the counting loop
  config 2  synthetic 1080p ROI stream: 424 x 212, frame batch 64
  config 3  full 1080p frames, crop [(748, 452), (1172, 664)], classifier + tracker, null-padded last window
  config 5  (per-GPU workload) 4K ROI 850 x 425 at the queue of 21

Integer / byte / index outputs bit-exact; A, E within 1e-5 (north_star)."""
import os

import numpy as np
import pytest

from helpers import oracle_events, oracle_frames, track, event_signature

pytestmark = pytest.mark.gpu
ATOL_AE = 1e-5


@pytest.fixture(scope="module")
def ctx():
    from swiftwatcher_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def orc():
    from oracle import reference_path
    return reference_path


def _segs(res, i):
    return [(int(s["label"]), int(s["r0"]), int(s["c0"]), int(s["r1"]), int(s["c1"]), int(s["area"]),
             int(s["sum_r"]), int(s["sum_c"])) for s in res["segs"][i, :res["nseg"][i]]]


def _orc_segs(seglist):
    return [(s["label"],) + s["bbox"] + (s["area"], s["sum_r"], s["sum_c"]) for s in seglist]


def _downstream(orc, sparse):
    """Oracle byte stages of data_structures.py:194-211 on a sparse image stack."""
    bil = [orc.bilateral_u8(f) for f in sparse]
    thr = [orc.thresh_tozero_u8(f) for f in bil]
    opened = [orc.grey_open_u8(f) for f in thr]
    lab = [orc.labels_to_u8(orc.ccl_u8(f)[1]) for f in opened]
    return dict(bilateral=np.stack(bil), thresh=np.stack(thr), opened=np.stack(opened), labels=np.stack(lab),
                segments=[orc.regionprops_u8(l) for l in lab])


def _expected_integer_start(gray):
    """k_ialm_init's rule restated on the host: the integer start stands iff the first shrinkage (:283) removes
    nothing, i.e. max(X + Y0/mu0) = 1.8 max(X) <= lmbda/mu0 = 0.008 ||X||_F (when ||X||_F >= max(X)/lmbda)."""
    x = gray.astype(np.float64)
    fro = np.sqrt((x * x).sum())
    dual = max(fro, x.max() / 0.01)
    inv_mu = fro / 1.25
    return x.max() + inv_mu * (x.max() / dual) <= 0.01 * inv_mu


@pytest.mark.parametrize("name", ["ialm_212x424x64_seeded", "ialm_212x424x21_seeded", "ialm_425x850x21_seeded"])
def test_reference_fixture_at_workload_size(ctx, orc, golden_dir, name):
    """Configs 2, 3 and 5 at full size against what the REFERENCE's rpca / IALM produced on the same seeded window:
    iteration count, the uint8 sparse image (sha256 + per-frame sums), A and E on the sampled pixel rows <= 1e-5; then
    every byte stage and the region records against the oracle run on that (now proven) sparse image."""
    from test_oracle_golden import seeded_frames
    from oracle.scenes import sha256
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    frames = seeded_frames(g)
    n, H, W = frames.shape
    # the hot path as FrameQueue runs it (M-state pass, no float outputs)
    res = ctx.batch_run(frames, 1, n)
    assert int(res["iters"][0]) == int(g["iters"])
    np.testing.assert_array_equal(res["gray"], frames)
    np.testing.assert_array_equal(res["rpca"].reshape(n, -1).astype(np.int64).sum(axis=1), g["sparse_frame_sums"])
    assert sha256(res["rpca"]) == str(g["sparse_sha256"])
    assert ctx.last_integer_start_windows == int(_expected_integer_start(frames)) == 1
    down = _downstream(orc, res["rpca"])
    for key in ("bilateral", "thresh", "opened", "labels"):
        np.testing.assert_array_equal(res[key], down[key], err_msg=key)
    for i in range(n):
        assert _segs(res, i) == _orc_segs(down["segments"][i])
    assert int(res["nseg"].sum()) >= 8
    # the float64 factors (A/Y-state pass)
    A, E, iters = ctx.ialm(frames.reshape(n, H * W))
    assert iters == int(g["iters"])
    rows = g["rows"]
    np.testing.assert_allclose(A[rows], g["A_rows"], atol=ATOL_AE, rtol=0)
    np.testing.assert_allclose(E[rows], g["E_rows"], atol=ATOL_AE, rtol=0)
    np.testing.assert_allclose(A.sum(axis=0), g["A_colsum"], rtol=1e-8)
    assert sha256(np.ascontiguousarray(ctx.rpca_epilogue(E).T.reshape(n, H, W))) == str(g["sparse_sha256"])


@pytest.mark.parametrize("name", ["ialm_47x94x21_s301", "ialm_47x94x21_s305", "ialm_47x94x21_s302", "ialm_47x94x21_s303",
                                  "ialm_47x94x21_s304", "ialm_107x214x21_s311", "ialm_107x214x21_s312"])
def test_reference_fixtures_around_the_start_switch(ctx, golden_dir, name):
    """Config 1's window size (94 x 47 x 21) on both sides of the switch between the integer start and the f64 start pass -- s301 and
    s305 sit 0.3 % and 0.2 % from it -- and two windows at 214 x 107 x 21, all produced by the REFERENCE's own rpca / IALM
    (oracle/make_goldens_r3.py): which start runs is the host-side statement of the rule, iteration count and the uint8 sparse
    image equal the reference's, A and E on the sampled rows <= 1e-5."""
    from test_oracle_golden import seeded_frames
    from oracle.scenes import sha256
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    frames = seeded_frames(g)
    n, H, W = frames.shape
    integer = bool(_expected_integer_start(frames))
    assert integer == (float(g["switch_ratio"]) <= 1.0)
    res = ctx.batch_run(frames, 1, n, stages=("gray", "rpca"))
    assert ctx.last_integer_start_windows == int(integer)
    assert int(res["iters"][0]) == int(g["iters"])
    assert sha256(res["rpca"]) == str(g["sparse_sha256"])
    np.testing.assert_array_equal(res["rpca"].reshape(n, -1).astype(np.int64).sum(axis=1), g["sparse_frame_sums"])
    A, E, iters = ctx.ialm(frames.reshape(n, H * W))
    assert iters == int(g["iters"])
    rows = g["rows"]
    np.testing.assert_allclose(A[rows], g["A_rows"], atol=ATOL_AE, rtol=0)
    np.testing.assert_allclose(E[rows], g["E_rows"], atol=ATOL_AE, rtol=0)


def test_config2_full_window_against_oracle(ctx, orc):
    """Config 2's unit of work from BGR input: one 424 x 212 x 64 window of the bench's own synthetic stream
    (12 birds per frame) against orc.window -- every stage image, region records, iteration count."""
    from swiftwatcher_amd import synthetic
    roi = synthetic.roi_window(20190816, 64, 212, 424, birds=12)
    res = ctx.batch_run(roi, 1, 64)
    ref = orc.window(roi)
    gray = ref["gray"].reshape(64, -1).T
    assert int(res["iters"][0]) == orc.ialm(gray, return_iters=True)[2]
    for key in ("gray", "rpca", "bilateral", "thresh", "opened", "labels"):
        np.testing.assert_array_equal(res[key], ref[key], err_msg=key)
    for i in range(64):
        assert _segs(res, i) == _orc_segs(ref["segments"][i])
    assert res["nseg"].min() >= 4


def test_config1_window_reference_fixture(ctx, golden_dir):
    """Config 1's ROI (94 x 47) at the queue of 21 = 92.8 k elements: the first shrinkage clips (1.8 max(X) >
    0.008 ||X||_F), so the integer start must NOT be taken and the f64 start pass runs; the reference itself needs 23
    iterations there.  Fixture from the reference's own functions."""
    g = np.load(os.path.join(golden_dir, "ialm_47x94x21.npz"))
    frames = g["frames"]
    n, H, W = frames.shape
    assert not _expected_integer_start(frames)
    res = ctx.batch_run(frames, 1, n, stages=("gray", "rpca"))
    assert ctx.last_integer_start_windows == 0
    assert int(res["iters"][0]) == int(g["iters"]) == 23
    np.testing.assert_array_equal(res["rpca"], g["sparse"])
    A, E, iters = ctx.ialm(frames.reshape(n, H * W))
    assert iters == 23
    rows = g["rows"]
    np.testing.assert_allclose(A[rows], g["A_rows"], atol=ATOL_AE, rtol=0)
    np.testing.assert_allclose(E[rows], g["E_rows"], atol=ATOL_AE, rtol=0)
    np.testing.assert_array_equal(ctx.rpca_epilogue(E).T.reshape(n, H, W), g["sparse"])


def test_config1_480p_clip_counting_loop():
    """Config 1: a 10 s 480p clip (300 frames of 854 x 480), chimney 76 px wide -> ROI 94 x 47
    (image_filtering.py:49-51), the CLI's queue of 21 (14 full windows + one padded with 15 null frames), through
    the reference's counting loop (__main__.py:71-98, --classify off): events and swift count equal those of the same
    tracker fed by the CPU oracle's segments."""
    from swiftwatcher_amd import synthetic, pipeline, _lib
    from swiftwatcher_amd import image_filtering as img
    from swiftwatcher_amd import event_classification as ec
    corners = [(389, 300), (465, 301)]
    crop_region = img.generate_crop_region(corners)
    (x0, y0), (x1, y1) = crop_region
    assert (x1 - x0, y1 - y0) == (94, 47)
    clip = synthetic.full_frames(480, 300, crop_region, frame_hw=(480, 854), birds=3, bird_len=(7, 11), bird_wid=(3, 5))[::-1].copy()
    roi_mask = np.zeros((47, 94), np.uint8)
    roi_mask[24:, 9:85] = 255
    count, events = pipeline.count_swifts(list(clip), crop_region, roi_mask)
    # which start ran: the library's per-window choice (integer matrix cores vs f64 start pass) must be the host-side
    # statement of the rule; this bright-sky clip sits on the integer side, the reference-generated fixture of the same
    # size (test_config1_window_reference_fixture) on the other
    c0 = _lib.default_context(0)
    first = np.ascontiguousarray(np.stack([f[y0:y1, x0:x1] for f in clip[:21]][::-1]))
    c0.batch_run(first, 1, 21, stages=())
    assert c0.last_integer_start_windows == int(_expected_integer_start(c0.bgr2gray(first)))
    ref_events = oracle_events(clip, crop_region, roi_mask)
    assert event_signature(events) == event_signature(ref_events)
    assert count == ec.count_swifts(ref_events)
    assert len(events) >= 3
    count_b, events_b = pipeline.count_swifts(list(clip), crop_region, roi_mask, windows_per_call=8)
    assert count_b == count and event_signature(events_b) == event_signature(events)


def test_config3_1080p_classifier_tracker(tmp_path):
    """Config 3: whole 1920 x 1080 frames, crop [(748, 452), (1172, 664)] (the 340-px chimney of SURVEY 8d),
    70 frames = 3 full queues of 21 + one padded with null frames, classifier ON (head calibrated on the clip's own
    segments so that some but not all are kept), tracker, events and count -- identical to the oracle pipeline
    (oracle segments -> oracle classifier -> same tracker).  Run as the unchanged CLI would (one queue per call,
    classifier per frame) and batched (4 queues per call, one classifier batch)."""
    import torch
    from swiftwatcher_amd import synthetic, pipeline
    from swiftwatcher_amd import image_filtering as img
    from swiftwatcher_amd import event_classification as ec
    from swiftwatcher_amd.segment_classification import SegmentClassifier
    from oracle import classifier_ref, reference_path as orc
    corners = [(790, 620), (1130, 622)]
    crop_region = img.generate_crop_region(corners)
    assert crop_region == [(748, 452), (1172, 664)]
    (x0, y0), (x1, y1) = crop_region
    total = 70
    clip = synthetic.full_frames(1080, total, crop_region, birds=9)[::-1].copy()
    roi_mask = np.zeros((212, 424), np.uint8)
    roi_mask[100:, 42:382] = 255
    # oracle side: segments and crops of every popped frame, then a head calibrated on them so that about half are kept
    info = oracle_frames(clip, crop_region)
    crops = [c for fr in info for c in fr["crops"]]
    assert len(crops) > 400
    sd = classifier_ref.calibrate_head(classifier_ref.random_state_dict(33), crops[::3])
    scores, _ = classifier_ref.classify(sd, crops)
    # calibrate_head puts the median crop ON the boundary; move the boundary into the widest gap nearby so that no
    # decision is close enough for float32 summation order to flip it (both head pre-activations stay positive, so the
    # scores are affine in the bias)
    d = np.sort((scores[:, 1] - scores[:, 0]).astype(np.float64))
    mid = d[len(d) // 3: 2 * len(d) // 3]
    gap = int(np.argmax(np.diff(mid)))
    sd["classifier.1.bias"] = sd["classifier.1.bias"] - torch.tensor([0.0, float(0.5 * (mid[gap] + mid[gap + 1]))])
    scores, keep_flat = classifier_ref.classify(sd, crops)
    margin = np.abs(scores[:, 1] - scores[:, 0])
    assert margin.min() > 2e-4, margin.min()
    assert 0.25 * len(crops) < keep_flat.sum() < 0.75 * len(crops)                 # some, not all
    keep, at = [], 0
    for fr in info:
        keep.append(list(keep_flat[at:at + len(fr["crops"])]))
        at += len(fr["crops"])
    ref_events = track(info, roi_mask, keep)
    assert len(ref_events) != len(track(info, roi_mask))                           # the classifier changes the outcome
    path = tmp_path / "w.pt"
    torch.save(sd, path)
    clf = SegmentClassifier(str(path))
    count, events = pipeline.count_swifts(list(clip), crop_region, roi_mask, classifier=clf)
    assert event_signature(events) == event_signature(ref_events)
    assert count == ec.count_swifts(ref_events)
    assert len(events) >= 2
    # batched: the producer thread segments queue-fuls ahead while this thread cuts classifier inputs on the SAME
    # context (calls are serialised in _lib.Context); profiling on, so the event bookkeeping is exercised too
    from swiftwatcher_amd import _lib
    _lib.default_context(0).prof_enable(True)
    for wpc in (4, 2):
        count_b, events_b = pipeline.count_swifts(list(clip), crop_region, roi_mask, classifier=clf, windows_per_call=wpc)
        assert count_b == count and event_signature(events_b) == event_signature(events)
    _lib.default_context(0).prof_enable(False)


_C3 = {}


def _config3_model_pt_case(golden_dir):
    """Config 3 "incl. model.pt CNN": 70 full 1080p frames (3 queues of 21 + one padded), 14 small faint birds per frame (5-8 x 4-6 px,
    contrast 25-40: the one kind of synthetic blob the reference's trained weights keep a fair share of), oracle segments and crops, the
    oracle classifier on the REFERENCE's own weights (tests/golden/classifier_model_pt.npz = model.pt's 52 tensors)."""
    if not _C3:
        from swiftwatcher_amd import synthetic
        from swiftwatcher_amd import image_filtering as img
        from oracle import classifier_ref
        from test_classifier import _model_pt_fixture
        corners = [(790, 620), (1130, 622)]
        crop_region = img.generate_crop_region(corners)
        clip = synthetic.full_frames(1081, 70, crop_region, birds=14, bird_len=(5, 8), bird_wid=(4, 6), contrast=(25, 40))[::-1].copy()
        info = oracle_frames(clip, crop_region)
        crops = [c for fr in info for c in fr["crops"]]
        sd = _model_pt_fixture(golden_dir)[0]
        scores, keep_flat = classifier_ref.classify(sd, crops)
        margin = np.abs(scores[:, 1] - scores[:, 0])
        assert margin.min() > 1e-3, margin.min()            # margin gate: no decision float32 summation order could flip
        assert 0.1 * len(crops) < keep_flat.sum() < 0.5 * len(crops) and len(crops) > 800
        keep, at = [], 0
        for fr in info:
            keep.append(list(keep_flat[at:at + len(fr["crops"])]))
            at += len(fr["crops"])
        _C3.update(corners=corners, crop_region=crop_region, clip=clip, info=info, sd=sd, keep=keep)
    return _C3


def test_config3_with_model_pt_weights(golden_dir):
    """BASELINE config 3 as worded: full 1080p video, the model.pt CNN, tracking.  SegmentClassifier loads the reference's weights;
    the counting loop (one queue per call, classifier per popped frame -- served from the window's score table) must produce the
    events of the oracle pipeline (oracle segments -> oracle classifier on the same weights -> same tracker), and the count."""
    from swiftwatcher_amd import pipeline
    from swiftwatcher_amd import event_classification as ec
    from swiftwatcher_amd.segment_classification import SegmentClassifier
    c = _config3_model_pt_case(golden_dir)
    roi_mask = np.zeros((212, 424), np.uint8)
    roi_mask[100:, 42:382] = 255
    ref_events = track(c["info"], roi_mask, c["keep"])
    assert event_signature(ref_events) != event_signature(track(c["info"], roi_mask))        # the classifier changes the outcome
    clf = SegmentClassifier.from_state_dict(c["sd"])
    launches = []
    inner = clf.predict_last_batch
    clf.predict_last_batch = lambda *a, **k: (launches.append(1), inner(*a, **k))[1]
    count, events = pipeline.count_swifts(list(c["clip"]), c["crop_region"], roi_mask, classifier=clf)
    assert event_signature(events) == event_signature(ref_events)
    assert count == ec.count_swifts(ref_events)
    assert len(launches) == 4                       # one device-resident scoring batch per queue-ful, not one per frame
    count_b, events_b = pipeline.count_swifts(list(c["clip"]), c["crop_region"], roi_mask, classifier=clf, windows_per_call=2)
    assert count_b == count and event_signature(events_b) == event_signature(events)


def test_counting_loop_from_corners(golden_dir):
    """The loop as __main__.py:62-63 starts it: crop region AND ROI mask generated from the video's first frame and the two chimney
    corners (generate_regions inside count_swifts), against the oracle pipeline with oracle/roi_mask_ref.py's mask -- without and
    with the model.pt classifier."""
    from swiftwatcher_amd import pipeline
    from swiftwatcher_amd import event_classification as ec
    from swiftwatcher_amd import image_filtering as img
    from swiftwatcher_amd.segment_classification import SegmentClassifier
    from oracle import roi_mask_ref
    c = _config3_model_pt_case(golden_dir)
    mask_ref = roi_mask_ref.roi_mask(c["clip"][0], c["corners"], c["crop_region"])
    crop_region, mask, _ = img.generate_regions(c["clip"][0], c["corners"])
    assert crop_region == c["crop_region"]
    np.testing.assert_array_equal(mask, mask_ref)
    assert 2000 < int((mask == 255).sum()) < 20000
    ref_events = track(c["info"], mask_ref)
    count, events = pipeline.count_swifts(list(c["clip"]), corners=c["corners"])
    assert event_signature(events) == event_signature(ref_events) and count == ec.count_swifts(ref_events)
    assert len(events) >= 3
    ref_events_clf = track(c["info"], mask_ref, c["keep"])
    clf = SegmentClassifier.from_state_dict(c["sd"])
    count_c, events_c = pipeline.count_swifts(list(c["clip"]), corners=c["corners"], classifier=clf)
    assert event_signature(events_c) == event_signature(ref_events_clf) and count_c == ec.count_swifts(ref_events_clf)
    assert event_signature(events_c) != event_signature(events)


def test_config5_4k_roi_window_from_bgr(ctx, orc):
    """Config 5's per-GPU unit of work from BGR input: 850 x 425 ROI (680-px chimney in a 4K frame), queue of 21."""
    from swiftwatcher_amd import synthetic
    roi = synthetic.roi_window(4000, 21, 425, 850, birds=12, bird_len=(60, 100), bird_wid=(24, 40))
    res = ctx.batch_run(roi, 1, 21)
    ref = orc.window(roi)
    gray = ref["gray"].reshape(21, -1).T
    assert int(res["iters"][0]) == orc.ialm(gray, return_iters=True)[2]
    for key in ("gray", "rpca", "bilateral", "thresh", "opened", "labels"):
        np.testing.assert_array_equal(res[key], ref[key], err_msg=key)
    for i in range(21):
        assert _segs(res, i) == _orc_segs(ref["segments"][i])
    assert int(res["nseg"].sum()) >= 20

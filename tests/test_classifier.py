"""segment_classification drop-in: structure checks on CPU, scores/decisions against the CPU oracle
on the GPU (float32, tolerance 2e-4 on scores that are O(1); decisions identical away from ties)."""
import os

import numpy as np
import pytest
import torch

REF_MODEL = "/root/reference/swiftwatcher/model.pt"


def _segments(rng, n):
    from swiftwatcher_amd.image_filtering import RegionProps
    from swiftwatcher_amd.data_structures import Segment
    out = []
    for i in range(n):
        h, w = (24, 24) if i % 3 == 0 else (int(rng.integers(24, 70)), int(rng.integers(24, 90)))
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        out.append(Segment(RegionProps(i + 1, (0, 0, h, w), (h / 2, w / 2), h * w), 5, "t", img))
    return out


def test_topology_matches_reference_weights_when_present():
    from swiftwatcher_amd.segment_classification import SqueezeNet10
    from oracle import classifier_ref as ref
    m = SqueezeNet10(2)
    keys = set(m.state_dict().keys())
    assert len(keys) == 52 and sum(p.numel() for p in m.parameters()) == 736450
    assert keys == set(ref.random_state_dict(0).keys())
    for k, shp in ref.EXPECTED_SHAPES.items():
        assert tuple(m.state_dict()[k].shape) == shp
    if os.path.exists(REF_MODEL):      # build container only: the reference's own weight file
        sd = torch.load(REF_MODEL, map_location="cpu", weights_only=True)
        m.load_state_dict(sd, strict=True)


def test_classifier_requires_gpu_unless_asked(tmp_path):
    from swiftwatcher_amd.segment_classification import SegmentClassifier
    from oracle import classifier_ref as ref
    path = tmp_path / "w.pt"
    torch.save(ref.random_state_dict(1), path)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            SegmentClassifier(str(path))


def test_cpu_device_matches_oracle(tmp_path):
    """Host logic (preprocess chain, keep rule, relabelling) with torch's CPU kernels, explicitly requested."""
    from swiftwatcher_amd.segment_classification import SegmentClassifier
    from oracle import classifier_ref as ref
    rng = np.random.default_rng(0)
    segs = _segments(rng, 7)
    imgs = [s.segment_image for s in segs]
    sd = ref.calibrate_head(ref.random_state_dict(2), imgs)
    path = tmp_path / "w.pt"
    torch.save(sd, path)
    clf = SegmentClassifier(str(path), device="cpu")
    exp_scores, exp_keep = ref.classify(sd, imgs)
    got = clf.scores(imgs).numpy()
    np.testing.assert_allclose(got, exp_scores, atol=2e-4, rtol=1e-4)
    kept = clf(segs)
    assert 0 < exp_keep.sum() < len(segs)
    assert [s.parent_frame_number for s in kept] == [5] * int(exp_keep.sum())
    assert [s.label for s in kept] == list(range(1, len(kept) + 1))
    assert clf([]) == []


def test_cropped_network_equals_full_network(tmp_path):
    """Receptive-field cropped evaluation: same scores as the full 224x224 forward to float32
    summation order, and the oracle's decisions, on the CPU kernels."""
    from swiftwatcher_amd.segment_classification import SegmentClassifier
    from oracle import classifier_ref as ref
    rng = np.random.default_rng(8)
    segs = _segments(rng, 24)
    imgs = [s.segment_image for s in segs]
    sd = ref.calibrate_head(ref.random_state_dict(6), imgs)
    path = tmp_path / "w.pt"
    torch.save(sd, path)
    full = SegmentClassifier(str(path), device="cpu", cropped=False)
    crop = SegmentClassifier(str(path), device="cpu", batch_size=7)            # ragged last batch
    assert crop.cropped is not None and full.cropped is None
    a, b = full.scores(imgs).numpy(), crop.scores(imgs).numpy()
    np.testing.assert_allclose(b, a, atol=2e-6, rtol=1e-5)
    exp_scores, exp_keep = ref.classify(sd, imgs)
    np.testing.assert_allclose(b, exp_scores, atol=2e-4, rtol=1e-4)
    # the window the cropped network reads is exactly rows/cols 92..131 of the full input
    x_full, x_win = full.preprocess(imgs[:5]), crop.preprocess(imgs[:5], window=True)
    assert x_win.shape == (5, 3, 40, 40) and torch.equal(x_win, x_full[:, :, 92:132, 92:132])
    # and nothing outside it may matter: geometry recorded by the plan
    assert crop.cropped.final == (1, 11, 13)
    assert [tuple(p[2].shape[1:]) for p in crop.cropped.plan] == [
        (96, 12, 12), (128, 14, 14), (128, 16, 16), (256, 17, 17), (256, 12, 12), (256, 14, 14), (384, 16, 16),
        (384, 18, 18), (512, 19, 19), (512, 13, 13)]


@pytest.mark.gpu
def test_gpu_cropped_and_full_paths_agree(tmp_path):
    from swiftwatcher_amd.segment_classification import SegmentClassifier
    from oracle import classifier_ref as ref
    rng = np.random.default_rng(9)
    segs = _segments(rng, 64)
    imgs = [s.segment_image for s in segs]
    sd = ref.calibrate_head(ref.random_state_dict(7), imgs)
    path = tmp_path / "w.pt"
    torch.save(sd, path)
    full = SegmentClassifier(str(path), cropped=False)
    crop = SegmentClassifier(str(path))
    a, b = full.scores(imgs).cpu().numpy(), crop.scores(imgs).cpu().numpy()
    np.testing.assert_allclose(b, a, atol=2e-5, rtol=1e-4)
    x_full, x_win = full.preprocess(imgs[:9]), crop.preprocess(imgs[:9], window=True)
    assert torch.equal(x_win, x_full[:, :, 92:132, 92:132])                  # HIP window kernel, bit-exact
    exp_scores, exp_keep = ref.classify(sd, imgs)
    np.testing.assert_allclose(b, exp_scores, atol=2e-4, rtol=1e-4)
    margin = np.abs(exp_scores[:, 1] - exp_scores[:, 0]) > 2e-3
    kept_ids = {id(s) for s in crop(segs)}
    for s, k, m in zip(segs, exp_keep, margin):
        if m:
            assert (id(s) in kept_ids) == bool(k)


@pytest.mark.gpu
def test_gpu_scores_and_decisions_match_oracle(tmp_path):
    from swiftwatcher_amd.segment_classification import SegmentClassifier
    from oracle import classifier_ref as ref
    rng = np.random.default_rng(1)
    segs = _segments(rng, 40)
    imgs = [s.segment_image for s in segs]
    sd = ref.calibrate_head(ref.random_state_dict(3), imgs)
    path = tmp_path / "w.pt"
    torch.save(sd, path)
    clf = SegmentClassifier(str(path))
    assert clf.device.type == "cuda"
    exp_scores, exp_keep = ref.classify(sd, imgs)
    got = clf.scores(imgs).cpu().numpy()
    np.testing.assert_allclose(got, exp_scores, atol=2e-4, rtol=1e-4)
    margin = np.abs(exp_scores[:, 1] - exp_scores[:, 0]) > 2e-3
    kept_ids = {id(s) for s in clf(segs)}
    for s, k, m in zip(segs, exp_keep, margin):
        if m:
            assert (id(s) in kept_ids) == bool(k)
    assert 0 < exp_keep.sum() < len(segs)


@pytest.mark.gpu
def test_hip_classifier_input_is_pillow_exact(tmp_path):
    """swk_classifier_input (resize + pad + ToTensor + Normalize in one HIP kernel) against Pillow / the torchvision
    statements restated in the oracle: uint8 patches bit-exact, float32 network input bit-exact."""
    from PIL import Image
    from swiftwatcher_amd import _lib
    from oracle import classifier_ref as ref
    rng = np.random.default_rng(5)
    shapes = [(24, 24), (24, 31), (40, 24), (25, 25), (37, 61), (90, 130), (212, 424), (24, 25), (47, 48), (512, 300), (1, 1), (3, 500)]
    crops = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for h, w in shapes]
    crops.append(np.zeros((30, 30, 3), np.uint8))
    crops.append(np.full((60, 33, 3), 255, np.uint8))
    ctx = _lib.Context(0)
    patches, net = ctx.classifier_input(crops, ref.MEAN, ref.STD, want_patches=True)
    bil = getattr(Image, "Resampling", Image).BILINEAR
    for i, c in enumerate(crops):
        exp = np.asarray(Image.fromarray(c).resize((24, 24), bil))
        np.testing.assert_array_equal(patches[i], exp, err_msg="crop %d %r" % (i, c.shape))
        np.testing.assert_array_equal(net[i], ref.transform(c)[0].numpy(), err_msg="net input %d" % i)
    with pytest.raises(_lib.SwkError):
        ctx.classifier_input([np.zeros((513, 10, 3), np.uint8)], ref.MEAN, ref.STD)
    ctx.close()
    # the classifier's own preprocess() takes that path on the GPU and must agree with the host statements
    from swiftwatcher_amd.segment_classification import SegmentClassifier
    path = tmp_path / "w.pt"
    torch.save(ref.random_state_dict(4), path)
    clf = SegmentClassifier(str(path))
    x = clf.preprocess(crops[:9]).cpu()
    for i in range(9):
        assert torch.equal(x[i], ref.transform(crops[i])[0])


@pytest.mark.gpu
def test_device_resident_segment_inputs_match_host_path(tmp_path):
    """swk_segment_inputs (crop boxes + Pillow resize + normalise, cut from device frames by region records that
    never left the GPU) against the host path: extract_segment_images on the host frames + swk_classifier_input.
    Same input tensors bit for bit, so the same scores; frame indices in frame order."""
    from swiftwatcher_amd import _lib, synthetic
    from swiftwatcher_amd import image_filtering as img
    from swiftwatcher_amd.segment_classification import SegmentClassifier, IMAGENET_MEAN, IMAGENET_STD
    from oracle import classifier_ref as ref
    crop_region = [(40, 30), (40 + 120, 30 + 60)]
    nwin, n, FH, FW = 2, 21, 128, 200
    frames = np.concatenate([synthetic.full_frames(300 + w, n, crop_region, frame_hw=(FH, FW), birds=2 + w,
                                                   bird_len=(28, 45), bird_wid=(10, 18)) for w in range(nwin)])
    ctx = _lib.Context(0)
    host = ctx.batch_run(frames, nwin, n, crop=(40, 30, 120, 60), stages=("labels",))
    imgs, frame_of = [], []
    for f in range(nwin * n):
        rps = img.regionprops_from_records(host["segs"][f, :host["nseg"][f]])
        crops = img.extract_segment_images(rps, frames[f], (24, 24), crop_region)
        imgs += crops
        frame_of += [f] * len(crops)
    assert len(imgs) > 20 and any(c.shape[:2] != (24, 24) for c in imgs)
    # the same batch with every buffer in HBM
    dev = torch.device("cuda", 0)
    dframes = torch.from_numpy(frames).to(dev)
    seg_cap = 32
    segs = torch.zeros((nwin * n, seg_cap, 48), dtype=torch.uint8, device=dev)
    nseg = torch.zeros((nwin * n,), dtype=torch.int32, device=dev)
    iters = torch.zeros((nwin,), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    inp = _lib.Input(frames=dframes.data_ptr(), mem=_lib.MEM_DEVICE, channels=3, nwin=nwin, n=n, Hc=60, Wc=120,
                     x0=40, y0=30, frame_stride=FH * FW * 3, row_stride=FW * 3)
    out = _lib.Output(mem=_lib.MEM_DEVICE, seg_cap=seg_cap)
    out.segs, out.nseg, out.iters = segs.data_ptr(), nseg.data_ptr(), iters.data_ptr()
    ctx.batch_run_raw(inp, _lib.default_params(), out)
    np.testing.assert_array_equal(nseg.cpu().numpy(), host["nseg"])
    # raw inputs: full 224 and the 40-pixel window
    for pad in (100, 8):
        side = 24 + 2 * pad
        x = torch.empty((len(imgs), 3, side, side), dtype=torch.float32, device=dev)
        fidx = torch.empty((len(imgs),), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        total, skipped = ctx.segment_inputs(inp, (FH, FW), segs.data_ptr(), nseg.data_ptr(), seg_cap, IMAGENET_MEAN,
                                            IMAGENET_STD, x.data_ptr(), len(imgs), pad=pad, seg_frame_ptr=fidx.data_ptr())
        assert total == len(imgs) and skipped == 0
        assert fidx.cpu().tolist() == frame_of
        _, net = ctx.classifier_input(imgs, IMAGENET_MEAN, IMAGENET_STD, pad=pad)
        np.testing.assert_array_equal(x.cpu().numpy(), net)
    # chunked scoring through the classifier
    sd = ref.calibrate_head(ref.random_state_dict(11), imgs)
    path = tmp_path / "w.pt"
    torch.save(sd, path)
    clf = SegmentClassifier(str(path), batch_size=16)
    s_dev, f_dev = clf.scores_from_device(ctx, inp, (FH, FW), segs, nseg, seg_cap)
    s_host = clf.scores(imgs)
    assert f_dev.cpu().tolist() == frame_of
    np.testing.assert_allclose(s_dev.cpu().numpy(), s_host.cpu().numpy(), atol=1e-5, rtol=1e-5)
    exp_scores, _ = ref.classify(sd, imgs)
    np.testing.assert_allclose(s_dev.cpu().numpy(), exp_scores, atol=2e-4, rtol=1e-4)
    # a box at the frame edge is intersected with the frame; bad geometry is refused
    inp_bad = _lib.Input(frames=dframes.data_ptr(), mem=_lib.MEM_HOST, channels=3, nwin=nwin, n=n, Hc=60, Wc=120,
                         x0=40, y0=30, frame_stride=FH * FW * 3, row_stride=FW * 3)
    with pytest.raises(_lib.SwkError):
        ctx.segment_inputs(inp_bad, (FH, FW), segs.data_ptr(), nseg.data_ptr(), seg_cap, IMAGENET_MEAN, IMAGENET_STD,
                           x.data_ptr(), 4)
    ctx.close()


# ------------------------------------------------------------------ the reference's own weights (model.pt)
def _model_pt_fixture(golden_dir):
    """tests/golden/classifier_model_pt.npz (oracle/make_classifier_goldens.py): model.pt's 52 tensors as arrays,
    96 seeded crops and their eval-mode scores from the oracle."""
    g = np.load(os.path.join(golden_dir, "classifier_model_pt.npz"))
    sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w:")}
    crops = [g["crop%d" % i] for i in range(int(g["count"]))]
    return sd, crops, g["scores"], g["keep"]


def test_model_pt_weights_load_strict_and_oracle_reproduces_fixture(golden_dir):
    """The real weight file's key set / shapes match the restated topology (strict load, no reference checkout
    needed), and the oracle reproduces the committed scores on this machine."""
    from swiftwatcher_amd.segment_classification import SqueezeNet10, SegmentClassifier
    from oracle import classifier_ref as ref
    sd, crops, scores, keep = _model_pt_fixture(golden_dir)
    assert len(sd) == 52 and sum(v.numel() for v in sd.values()) == 736450
    SqueezeNet10(2).load_state_dict(sd, strict=True)
    if os.path.exists(REF_MODEL):      # build container: the arrays ARE the checkpoint's tensors
        disk = torch.load(REF_MODEL, map_location="cpu", weights_only=True)
        assert set(disk) == set(sd) and all(torch.equal(disk[k].float(), sd[k]) for k in sd)
    got, got_keep = ref.classify(sd, crops[:32])
    np.testing.assert_allclose(got, scores[:32], atol=2e-4, rtol=1e-4)
    assert list(got_keep) == list(keep[:32])
    # the product's host logic on the torch CPU kernels, full and receptive-field cropped
    for cropped in (False, True):
        clf = SegmentClassifier.from_state_dict(sd, device="cpu", cropped=cropped, batch_size=16)
        s = clf.scores(crops[:32]).numpy()
        np.testing.assert_allclose(s, scores[:32], atol=2e-4, rtol=1e-4)
        assert list(np.argmax(s, 1) == 1) == list(keep[:32])
    assert 0 < keep.sum() < len(keep)


@pytest.mark.gpu
def test_gpu_model_pt_scores_and_decisions(golden_dir):
    """SURVEY 8(c) fixture (9) on the MI355X: the full 224 x 224 network and the receptive-field cropped one, loaded
    with the reference's own weights, within 2e-4 of the oracle's eval-mode scores on all 96 crops; decisions equal
    (the smallest margin in the fixture is 0.27); both keep and drop occur."""
    from swiftwatcher_amd.segment_classification import SegmentClassifier
    sd, crops, scores, keep = _model_pt_fixture(golden_dir)
    margin = np.abs(scores[:, 1] - scores[:, 0])
    assert margin.min() > 0.1
    for cropped in (False, True):
        clf = SegmentClassifier.from_state_dict(sd, cropped=cropped, batch_size=64)
        assert clf.device.type == "cuda"
        s = clf.scores(crops).cpu().numpy()
        np.testing.assert_allclose(s, scores, atol=2e-4, rtol=1e-4)
        assert list(np.argmax(s, 1) == 1) == list(keep)
    segs = _segments(np.random.default_rng(3), len(crops))
    for sgm, c in zip(segs, crops):
        sgm.segment_image = c
    kept = clf(segs)
    assert len(kept) == int(keep.sum()) and [s.label for s in kept] == list(range(1, len(kept) + 1))


@pytest.mark.gpu
@pytest.mark.parametrize("split_bf16", [0, 1])
def test_fused_conv1x1_kernel_against_torch(split_bf16):
    """(split_bf16 = 1: the expand1x1 shapes of the list on k_expand1x1_bf16s, swk_set_cnn_tuning knob 1 -- same placement, same tolerance.)
    swk_nhwc_conv1x1_bias_relu_place (convolution + bias + ReLU + placement on the f32 matrix cores) against
    torch.nn.functional.conv2d on the shapes the Fire modules use and on ragged ones (pixel count not a multiple of 32,
    output channels not a multiple of 32, crop inside the source, channel offset in the destination).  float32 in a
    different summation order: 2e-5 relative to the output scale."""
    import ctypes
    from swiftwatcher_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(5)
    cases = [  # n, cin, cout, sh, crop, size, dH, off, dC, c_off
        (5, 96, 16, 8, 0, 8, 12, 2, 16, 0), (3, 128, 32, 14, 0, 14, 16, 1, 32, 0), (7, 256, 48, 12, 0, 12, 14, 1, 48, 0),
        (4, 512, 64, 9, 0, 9, 13, 2, 64, 0), (6, 16, 64, 12, 1, 10, 10, 0, 128, 0), (3, 32, 128, 16, 1, 14, 17, 0, 256, 0),
        (2, 48, 192, 14, 1, 12, 12, 0, 384, 0), (3, 64, 256, 18, 1, 16, 19, 2, 512, 0), (1, 64, 96, 7, 2, 3, 5, 1, 160, 60),
        (9, 16, 4, 5, 0, 5, 5, 0, 8, 4), (3, 80, 40, 6, 1, 5, 6, 1, 40, 0), (2, 384, 64, 14, 0, 14, 18, 2, 64, 0),
        (2, 192, 24, 7, 0, 7, 7, 0, 24, 0), (37, 32, 128, 3, 0, 3, 3, 0, 128, 0),
        # the workgroup size follows the number of 32-pixel row tiles (4, 8 or 16 waves): batches that take the 8- and the 16-wave kernels
        (700, 96, 16, 8, 0, 8, 8, 0, 16, 0), (1400, 16, 64, 10, 1, 8, 8, 0, 128, 64), (160, 48, 192, 12, 0, 12, 12, 0, 384, 0),
        (300, 48, 192, 12, 0, 12, 12, 0, 384, 192), (330, 512, 64, 9, 0, 9, 11, 1, 64, 0)]
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    assert lib.swk_set_cnn_tuning(1, split_bf16) == 0
    for n, cin, cout, sh, crop, size, dH, off, dC, c_off in cases:
        x = torch.randn((n, cin, sh, sh), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        wgt = (torch.randn((cout, cin, 1, 1), generator=g) * (2.0 / cin) ** 0.5).to(dev)
        bias = (torch.randn((cout,), generator=g) * 0.3).to(dev)
        dst = torch.full((n, dC, dH, dH), -7.0, device=dev).contiguous(memory_format=torch.channels_last)
        exp = dst.clone()
        y = torch.relu(torch.nn.functional.conv2d(x[:, :, crop:crop + size, crop:crop + size], wgt, bias))
        exp[:, c_off:c_off + cout, off:off + size, off:off + size] = y
        torch.cuda.synchronize()
        rc = lib.swk_nhwc_conv1x1_bias_relu_place(stream, x.data_ptr(), n, sh, sh, cin, crop, crop, size, size,
                                                  wgt.reshape(cout, cin).contiguous().data_ptr(), bias.data_ptr(), cout,
                                                  dst.data_ptr(), dH, dH, dC, off, off, c_off)
        assert rc == 0, (rc, n, cin, cout)
        torch.cuda.synchronize()
        scale = float(y.abs().max()) + 1e-6
        err = float((dst - exp).abs().max())
        assert err <= 2e-5 * max(scale, 1.0), (err, scale, n, cin, cout)
        # nothing outside the placed block was touched
        mask = torch.ones_like(dst, dtype=torch.bool)
        mask[:, c_off:c_off + cout, off:off + size, off:off + size] = False
        assert bool((dst[mask] == -7.0).all())
    assert lib.swk_set_cnn_tuning(1, 0) == 0
    # bad arguments are refused, not launched
    assert lib.swk_nhwc_conv1x1_bias_relu_place(stream, x.data_ptr(), 1, 4, 4, 24, 0, 0, 4, 4, wgt.data_ptr(), bias.data_ptr(), 8,
                                                dst.data_ptr(), 4, 4, 8, 0, 0, 0) != 0          # cin not a multiple of 16
    assert lib.swk_nhwc_conv1x1_bias_relu_place(stream, x.data_ptr(), 1, 3, 3, 32, 0, 0, 3, 3, wgt.data_ptr(), bias.data_ptr(), 6,
                                                dst.data_ptr(), 3, 3, 8, 0, 0, 0) != 0          # output channels not a multiple of 4


@pytest.mark.gpu
def test_split_bf16_expand_kernel_is_float32_accurate():
    """k_expand1x1_bf16s (off by default; swk_set_cnn_tuning knob 1): every float32 product as six bf16 x bf16 MFMA products of three-way split
    operands, accumulated in float32.  Against float64 its error must be that of a float32 multiply-add chain -- it is held to the error of the
    float32 kernel (swk_set_cnn_tuning knob 1 = 0) on the same data, with a floor of 4e-7 of the output scale -- on post-ReLU
    activations, on activations with a large dynamic range, and with a ragged last tile; and the two kernels place the same block."""
    import ctypes
    from swiftwatcher_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(31)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    worst = 0.0
    try:
        for n, cin, side, spread in ((67, 16, 8, 1.0), (131, 32, 12, 1.0), (40, 48, 10, 1.0), (300, 64, 14, 1.0), (33, 64, 9, 1e4), (5, 48, 3, 1e-3)):
            cout = 4 * cin
            x = torch.relu(torch.randn((n, cin, side, side), generator=g)) * 3.0
            x = (x * torch.exp(torch.randn((n, 1, side, side), generator=g) * float(np.log(spread)) * 0.5)).to(dev).contiguous(memory_format=torch.channels_last)
            wgt = (torch.randn((cout, cin), generator=g) * (2.0 / cin) ** 0.5).to(dev)
            bias = (torch.randn((cout,), generator=g) * 0.3).to(dev)
            want = torch.relu(torch.einsum("nchw,kc->nkhw", x.double(), wgt.double()) + bias.double().view(1, -1, 1, 1))
            scale = float(want.abs().max())
            got = {}
            for knob in (1, 0):
                assert lib.swk_set_cnn_tuning(1, knob) == 0
                dst = torch.zeros((n, cout, side, side), device=dev).contiguous(memory_format=torch.channels_last)
                rc = lib.swk_nhwc_conv1x1_bias_relu_place(stream, x.data_ptr(), n, side, side, cin, 0, 0, side, side, wgt.data_ptr(), bias.data_ptr(),
                                                          cout, dst.data_ptr(), side, side, cout, 0, 0, 0)
                assert rc == 0
                torch.cuda.synchronize()
                got[knob] = float((dst.double() - want).abs().max()) / scale
            worst = max(worst, got[1])
            assert got[1] <= max(1.5 * got[0], 4e-7), (got, n, cin, spread)
    finally:
        lib.swk_set_cnn_tuning(1, 0)
    assert worst < 2e-6


@pytest.mark.gpu
def test_own_kernels_match_miopen_path_at_bench_batch(tmp_path):
    """The cropped network on the library's own kernels (conv1, 1 x 1, Winograd 3 x 3, pools) against the same network with every
    convolution on MIOpen, at a ragged batch of the bench's size (2,377 rows: row tiles, Winograd tasks and workgroups that end in
    the middle of a segment) and at a single row; then the direct 3 x 3 kernel in place of the Winograd one."""
    from swiftwatcher_amd.segment_classification import SegmentClassifier
    from oracle import classifier_ref as ref
    path = tmp_path / "w.pt"
    torch.save(ref.random_state_dict(11), path)
    clf = SegmentClassifier(str(path), batch_size=4096)
    net = clf.cropped
    g = torch.Generator(device="cpu").manual_seed(3)
    for rows in (2377, 1):
        x = torch.randn((rows, 3, 40, 40), generator=g).cuda()
        with torch.no_grad():
            own = net(x).clone()
            try:
                net.winograd = False                     # the direct 3x3 kernel instead of F(2x2, 3x3)
                direct = net(x).clone()
                net.own_kernels = False                  # every convolution through MIOpen + the placement kernel
                miopen = net(x).clone()
            finally:
                net.own_kernels, net.winograd = True, True
        scale = float(miopen.abs().max()) + 1e-6
        assert float((own - miopen).abs().max()) <= 2e-5 * max(scale, 1.0), rows
        assert float((direct - miopen).abs().max()) <= 2e-5 * max(scale, 1.0), rows


@pytest.mark.gpu
def test_fused_conv1_kernel_against_torch():
    """swk_nhwc_conv7x7s2_bias_relu (the 7 x 7 stride-2 first convolution + bias + ReLU on the f32 matrix cores, patch rows split
    between the MFMA's k halves) against torch.nn.functional.conv2d: the cropped network's 40 x 40 window, a crop inside a larger
    image, batches that do not fill the last row tile."""
    import ctypes
    from swiftwatcher_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(26)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for n, side, lo, m in ((5, 40, 0, 17), (3, 64, 4, 9), (1, 8, 0, 1), (130, 22, 1, 7)):
        x = torch.randn((n, 3, side, side), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        w = (torch.randn((96, 3, 7, 7), generator=g) * (2.0 / 147) ** 0.5).to(dev).contiguous()
        bias = (torch.randn((96,), generator=g) * 0.3).to(dev)
        y = torch.relu(torch.nn.functional.conv2d(x, w, bias, stride=2))[:, :, lo:lo + m, lo:lo + m]
        dst = torch.full((n, 96, m, m), -7.0, device=dev).contiguous(memory_format=torch.channels_last)
        torch.cuda.synchronize()
        rc = lib.swk_nhwc_conv7x7s2_bias_relu(stream, x.data_ptr(), n, side, lo, m, w.data_ptr(), bias.data_ptr(), 96, dst.data_ptr())
        assert rc == 0, (rc, n, side, lo, m)
        torch.cuda.synchronize()
        scale = max(float(y.abs().max()), 1.0)
        assert float((dst - y).abs().max()) <= 2e-5 * scale, (n, side, lo, m)
    # an output whose patch would leave the image is refused
    assert lib.swk_nhwc_conv7x7s2_bias_relu(stream, x.data_ptr(), 1, 22, 1, 8, w.data_ptr(), bias.data_ptr(), 96, dst.data_ptr()) != 0


@pytest.mark.gpu
def test_fused_conv3x3_kernel_against_torch():
    """swk_nhwc_conv3x3_bias_relu_place (valid 3 x 3 convolution over the squeeze tile + bias + ReLU + placement behind the
    expand1x1 channels, weights streamed through LDS) against torch.nn.functional.conv2d: every (channels, tile) shape the
    Fire modules use, batches whose pixel count is not a multiple of the row tiles, and a channel count that is not a
    multiple of 32."""
    import ctypes
    from swiftwatcher_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(6)
    cases = [  # n, cin, cout, t, dH, off, dC, c_off
        (7, 16, 64, 12, 10, 0, 128, 64), (5, 16, 64, 14, 12, 0, 128, 64), (3, 32, 128, 16, 17, 1, 256, 128),
        (9, 32, 128, 12, 10, 0, 256, 128), (2, 48, 192, 14, 12, 0, 384, 192), (3, 48, 192, 16, 14, 0, 384, 192),
        (2, 64, 256, 18, 19, 2, 512, 256), (5, 64, 256, 13, 11, 0, 512, 256), (1, 16, 40, 5, 4, 1, 44, 4), (300, 16, 64, 4, 2, 0, 64, 0)]
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for n, cin, cout, t, dH, off, dC, c_off in cases:
        x = torch.randn((n, cin, t, t), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        wgt = (torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dev)
        bias = (torch.randn((cout,), generator=g) * 0.3).to(dev)
        wt = wgt.permute(2, 3, 1, 0).contiguous()
        dst = torch.full((n, dC, dH, dH), -7.0, device=dev).contiguous(memory_format=torch.channels_last)
        exp = dst.clone()
        y = torch.relu(torch.nn.functional.conv2d(x, wgt, bias))
        o = t - 2
        exp[:, c_off:c_off + cout, off:off + o, off:off + o] = y
        torch.cuda.synchronize()
        rc = lib.swk_nhwc_conv3x3_bias_relu_place(stream, x.data_ptr(), n, t, cin, wt.data_ptr(), bias.data_ptr(), cout,
                                                  dst.data_ptr(), dH, dH, dC, off, off, c_off)
        assert rc == 0, (rc, n, cin, cout, t)
        torch.cuda.synchronize()
        scale = max(float(y.abs().max()), 1.0)
        err = float((dst - exp).abs().max())
        assert err <= 2e-5 * scale, (err, scale, n, cin, cout, t)
        mask = torch.ones_like(dst, dtype=torch.bool)
        mask[:, c_off:c_off + cout, off:off + o, off:off + o] = False
        assert bool((dst[mask] == -7.0).all())


def test_winograd_filter_transform_host():
    """swk_winograd_f2x2_3x3_weights (host code): U = G g G^T of every filter, in the kernel's operand layout
    input channel 16 chunk + 8 (k half) + 4 quad + j, output channels padded to whole column blocks; one layout per kernel
    configuration: [position][chunk][k half][quad][cg * 32 + r][4] with output channel 32 cg + r (one column block per
    wave); 64 -> 256 keeps every wave's 2 KB of a phase contiguous: [position][chunk][cg][k half][quad][r][4]."""
    import ctypes
    from swiftwatcher_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(3)
    for cout, cin in ((104, 32), (256, 64)):
        _check_winograd_layouts(lib, rng, cout, cin)
    w = rng.standard_normal((8, 24, 3, 3)).astype(np.float32)
    out = np.zeros(16, np.float32)
    assert lib.swk_winograd_f2x2_3x3_weights(w.ctypes.data_as(ctypes.c_void_p), 8, 24, out.ctypes.data_as(ctypes.c_void_p)) != 0


def _check_winograd_layouts(lib, rng, cout, cin):
    import ctypes
    w = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32)
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
    U = np.einsum("ak,oikl,bl->abio", G, w.astype(np.float64), G)              # [xi][nu][ci][co]
    nbw = 1
    CG = -(-cout // (32 * nbw))
    out = np.full(16 * cin * nbw * 32 * CG, np.nan, np.float32)
    assert lib.swk_winograd_f2x2_3x3_weights(w.ctypes.data_as(ctypes.c_void_p), cout, cin, out.ctypes.data_as(ctypes.c_void_p)) == 0
    if (cin, cout) != (64, 256):         # shared phase blocks: p, h, chunk, k half, quad, cg, r, j
        out = out.reshape(16, nbw, cin // 16, 2, 2, CG, 32, 4)
        got = out.transpose(0, 2, 3, 4, 7, 5, 1, 6).reshape(4, 4, cin, 32 * nbw * CG)      # -> [xi][nu][channel][32 cg + r]
    else:                # 64 -> 256, private slices: a wave's 2 KB of a phase contiguous: p, chunk, cg, k half, quad, r, j
        out = out.reshape(16, cin // 16, CG, 2, 2, 32, 4)
        got = out.transpose(0, 1, 3, 4, 6, 2, 5).reshape(4, 4, cin, 32 * CG)               # -> [xi][nu][channel][32 cg + r]
    assert np.array_equal(got[..., :cout], U.astype(np.float32))
    assert not got[..., cout:].any()


@pytest.mark.gpu
def test_winograd_conv3x3_kernel_against_torch():
    """swk_nhwc_conv3x3_winograd_bias_relu_place (F(2x2, 3x3) on the f32 matrix cores) against torch.nn.functional.conv2d and
    against the direct kernel, on the three Fire shapes it takes: even and odd output sizes (the odd one computes a half-used
    last tile row / column whose patch reaches past the tile), batches that do not fill the last workgroup task, one segment.
    Tolerance 1e-5 of the output scale (the transform adds a few roundings to the direct kernel's 2e-5 bound... measured 2e-6)."""
    import ctypes
    from swiftwatcher_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(16)
    cases = [  # n, cin, cout, t, dH, off, dC, c_off
        (3, 32, 128, 16, 17, 1, 256, 128), (9, 32, 128, 12, 10, 0, 256, 128), (2, 48, 192, 14, 12, 0, 384, 192),
        (3, 48, 192, 16, 14, 0, 384, 192), (2, 64, 256, 18, 19, 2, 512, 256), (5, 64, 256, 13, 11, 0, 512, 256),
        (1, 64, 256, 3, 1, 0, 256, 0), (1, 32, 128, 5, 3, 0, 128, 0), (70, 64, 256, 7, 5, 0, 256, 0), (33, 48, 192, 4, 2, 0, 192, 0),
        (7, 16, 64, 12, 10, 0, 128, 64), (5, 16, 64, 14, 12, 0, 128, 64), (130, 16, 64, 5, 3, 0, 64, 0)]
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for n, cin, cout, t, dH, off, dC, c_off in cases:
        x = torch.randn((n, cin, t, t), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        wcpu = (torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5).contiguous()
        wgt = wcpu.to(dev)
        bias = (torch.randn((cout,), generator=g) * 0.3).to(dev)
        ww = torch.empty(16 * cin * cout, dtype=torch.float32)
        assert lib.swk_winograd_f2x2_3x3_weights(wcpu.data_ptr(), cout, cin, ww.data_ptr()) == 0
        ww = ww.to(dev)
        dst = torch.full((n, dC, dH, dH), -7.0, device=dev).contiguous(memory_format=torch.channels_last)
        exp = dst.clone()
        direct = dst.clone()
        y = torch.relu(torch.nn.functional.conv2d(x, wgt, bias))
        o = t - 2
        exp[:, c_off:c_off + cout, off:off + o, off:off + o] = y
        torch.cuda.synchronize()
        rc = lib.swk_nhwc_conv3x3_winograd_bias_relu_place(stream, x.data_ptr(), n, t, cin, ww.data_ptr(), bias.data_ptr(), cout,
                                                           dst.data_ptr(), dH, dH, dC, off, off, c_off)
        assert rc == 0, (rc, n, cin, cout, t)
        wt = wgt.permute(2, 3, 1, 0).contiguous()
        assert lib.swk_nhwc_conv3x3_bias_relu_place(stream, x.data_ptr(), n, t, cin, wt.data_ptr(), bias.data_ptr(), cout,
                                                    direct.data_ptr(), dH, dH, dC, off, off, c_off) == 0
        torch.cuda.synchronize()
        scale = max(float(y.abs().max()), 1.0)
        err = float((dst - exp).abs().max())
        assert err <= 1e-5 * scale, (err, scale, n, cin, cout, t)
        assert float((dst - direct).abs().max()) <= 1e-5 * scale
        mask = torch.ones_like(dst, dtype=torch.bool)
        mask[:, c_off:c_off + cout, off:off + o, off:off + o] = False
        assert bool((dst[mask] == -7.0).all())
    # shapes outside the Fire ratio are refused (the caller takes the direct kernel)
    assert lib.swk_nhwc_conv3x3_winograd_bias_relu_place(stream, x.data_ptr(), 1, 4, 16, ww.data_ptr(), bias.data_ptr(), 32,
                                                         dst.data_ptr(), 2, 2, 32, 0, 0, 0) != 0


def _calibrated_classifier(crops, seed, tmp_path, margin=5e-5):
    """A random-weight classifier whose head keeps some but not all of `crops`, no decision closer to the boundary than
    `margin` (so float32 summation order cannot flip one); returns (SegmentClassifier on the GPU, state dict, oracle keep flags)."""
    from swiftwatcher_amd.segment_classification import SegmentClassifier
    from oracle import classifier_ref as ref
    sd = ref.calibrate_head(ref.random_state_dict(seed), crops[::3])
    scores, _ = ref.classify(sd, crops)
    d = np.sort((scores[:, 1] - scores[:, 0]).astype(np.float64))
    mid = d[len(d) // 3: 2 * len(d) // 3]
    gap = int(np.argmax(np.diff(mid)))
    sd["classifier.1.bias"] = sd["classifier.1.bias"] - torch.tensor([0.0, float(0.5 * (mid[gap] + mid[gap + 1]))])
    scores, keep = ref.classify(sd, crops)
    assert np.abs(scores[:, 1] - scores[:, 0]).min() > margin
    assert 0.2 * len(crops) < keep.sum() < 0.8 * len(crops)
    path = tmp_path / ("w%d.pt" % seed)
    torch.save(sd, path)
    return SegmentClassifier(str(path)), sd, keep


@pytest.mark.gpu
def test_window_score_table_equals_per_frame_calls(tmp_path):
    """__main__.py:84-85 calls classifier(frame.segments) once per popped frame.  Here the first call of a window scores ALL
    its segments in one device-resident batch (swk_segment_inputs_last on what segment_queue left on the GPU: the ROI frames
    with a margin of half the minimum segment size, and the region records) and the other calls look their rows up.  Kept
    segments and labels must equal (a) the oracle classifier on the reference's crops (extract_segment_images' views of the FULL
    frame: boxes near the ROI's edge grow into the margin), (b) this classifier's per-call image path, which still serves when the
    context has moved on in between.  From the second window on the scoring starts inside segment_queue; a window whose scores
    nobody asked for switches that off again."""
    from swiftwatcher_amd import synthetic, _lib
    from swiftwatcher_amd.data_structures import FrameQueue
    from oracle import reference_path as orc
    crop_region = [(60, 50), (60 + 212, 50 + 106)]
    n, windows = 21, 3
    clip = synthetic.full_frames(31, n * windows, crop_region, frame_hw=(220, 340), birds=10, bird_len=(14, 24), bird_wid=(6, 10))[::-1].copy()
    clip[:, :50] = (clip[:, 50:100][:, ::-1] // 2 + 40)        # structure OUTSIDE the ROI: a crop that reaches into the margin sees it
    clip[:, :, :60] = (clip[:, :, 60:120][:, :, ::-1] // 2 + 30)

    def run(mode, clf):
        q = FrameQueue()
        out = []
        launches = 0
        for w in range(windows):
            fr = [clip[w * n + i] for i in range(n)]
            q.push_list_of_frames(fr, list(range(w * n, w * n + n)), ["t"] * n)
            q.preprocess_queue(crop_region, None)
            q.segment_queue((24, 24), crop_region)
            if mode == "window" and w >= 1:
                assert q._last_batch._tables, "the hinted classifier's forward was not started inside segment_queue"
            if mode == "stale":                                 # another batch on the same context before the first classifier call
                _lib.default_context(0).batch_run(np.ascontiguousarray(clip[:4, 50:80, 60:100]), 1, 4, stages=())
            while not q.is_empty():
                f = q.pop_frame()
                before = [(s.bbox, s.segment_image) for s in f.segments]
                if mode == "images":
                    for s in f.segments:
                        del s._batch
                kept = clf(f.segments)
                assert [s.label for s in kept] == list(range(1, len(kept) + 1))
                out.append((f.frame_number, before, [s.bbox for s in kept]))
            if mode == "stale":
                assert not q._last_batch._tables
        return out

    # oracle: crops of the full frame by the reference's box rule, oracle classifier
    q = FrameQueue()
    crops, boxes = [], []
    for w in range(windows):
        fr = [clip[w * n + i] for i in range(n)]
        q.push_list_of_frames(fr, list(range(w * n, w * n + n)), ["t"] * n)
        q.preprocess_queue(crop_region, None)
        q.segment_queue((24, 24), crop_region)
        while not q.is_empty():
            f = q.pop_frame()
            for s in f.segments:
                r0, c0, r1, c1 = orc.segment_crop_box(s.bbox, (24, 24), crop_region)
                crops.append(f.frame[max(r0, 0):max(r1, 0), max(c0, 0):max(c1, 0)])
                boxes.append((f.frame_number, s.bbox))
    assert len(crops) > 200
    reach = sum(1 for (_, b) in boxes if b[0] < 12 or b[1] < 12)
    assert reach >= 5, "no segment near the ROI's top / left edge: the margin is not exercised"
    clf, sd, keep = _calibrated_classifier(crops, 5, tmp_path)
    expect = {}
    for (fn, b), kp in zip(boxes, keep):
        expect.setdefault(fn, [])
        if kp:
            expect[fn].append(b)
    got = run("window", clf)
    for fn, before, kept in got:
        assert kept == expect.get(fn, []), fn
    for mode in ("images", "stale"):
        other = run(mode, clf)
        assert [(a, c) for a, _, c in other] == [(a, c) for a, _, c in got], mode
    # a window that nobody classifies turns the eager start off again
    q = FrameQueue()
    for w in range(3):
        fr = [clip[w * n + i] for i in range(n)]
        q.push_list_of_frames(fr, list(range(n)), ["t"] * n)
        q.preprocess_queue(crop_region, None)
        q.segment_queue((24, 24), crop_region)
        if w == 0:
            clf(q[20].segments)
            assert q._classifier_hint is not None
        if w == 1:
            assert q._last_batch._tables                        # started eagerly, then never used
        if w == 2:
            assert q._classifier_hint is None and not q._last_batch._tables
        while not q.is_empty():
            q.pop_frame()
    # events of a deep copy (what __main__.py:100 returns) carry plain data only
    import copy
    q.push_list_of_frames([clip[i] for i in range(n)], list(range(n)), ["t"] * n)
    q.preprocess_queue(crop_region, None)
    q.segment_queue((24, 24), crop_region)
    some = [s for f in q for s in f.segments][:3]
    cp = copy.deepcopy(some)
    assert all(not hasattr(c, "_batch") for c in cp) and all(np.array_equal(c.segment_image, s.segment_image) for c, s in zip(cp, some))


@pytest.mark.gpu
def test_graph_replay_of_window_sized_forwards_equals_the_eager_forward():
    """A FrameQueue window's forward (<= 512 rows on a persistent input slot) is captured once as a HIP graph and replayed:
    bit-identical to launching the kernels one by one.  Forwards on disjoint row ranges of the persistent per-layer tiles (row0)
    give each row the result it has in a forward of its own."""
    from swiftwatcher_amd.segment_classification import SegmentClassifier
    from oracle import classifier_ref as ref
    clf = SegmentClassifier.from_state_dict(ref.random_state_dict(12), batch_size=1024)
    g = torch.Generator(device="cpu").manual_seed(5)
    for rows in (64, 192, 320, 512):
        x = torch.randn((rows, 3, 40, 40), generator=g).to(clf.device).contiguous(memory_format=torch.channels_last)
        eager = clf._forward(x).clone()
        first = clf._forward_graphed(x).clone()              # captures
        again = clf._forward_graphed(x).clone()              # replays
        assert clf._use_graphs and clf._graph_error is None, clf._graph_error
        assert torch.equal(first, eager) and torch.equal(again, eager), rows
        x.add_(0.25)                                          # same buffer, new contents: the replay reads them
        assert torch.equal(clf._forward_graphed(x), clf._forward(x))
    assert len(clf._graphs) == 4
    x = torch.randn((96, 3, 40, 40), generator=g).to(clf.device).contiguous(memory_format=torch.channels_last)
    whole = clf.cropped(x)
    np.testing.assert_allclose(torch.cat([clf.cropped(x[:32], 0), clf.cropped(x[32:], 32)]).cpu().numpy(), whole.cpu().numpy(), atol=1e-6, rtol=1e-6)


@pytest.mark.gpu
def test_large_forward_on_two_streams_equals_one_chain():
    """Forwards of 1,024 rows or more run as two chains on two streams over disjoint rows of the persistent tiles: the same kernels
    on the same rows, so the scores are the one-chain forward's bit for bit.  Repeated with fresh contents in the same buffer, with a row count that does not halve evenly, and back to back with
    smaller forwards on the main stream (the side stream's reads and writes are ordered against both)."""
    from swiftwatcher_amd.segment_classification import SegmentClassifier
    from oracle import classifier_ref as ref
    clf = SegmentClassifier.from_state_dict(ref.random_state_dict(13), batch_size=4096)
    assert clf._split_rows == 1024
    g = torch.Generator(device="cpu").manual_seed(6)
    for rows in (1024, 1536, 3584, 4096):
        x = torch.randn((rows, 3, 40, 40), generator=g).to(clf.device).contiguous(memory_format=torch.channels_last)
        for _ in range(2):
            two = clf._forward(x).clone()
            small = clf._forward(x[:192]).clone()                       # one chain, right behind the two
            clf._split_rows = 0
            one = clf._forward(x).clone()
            clf._split_rows = 1024
            assert torch.equal(two, one) and torch.equal(small, one[:192]), rows
            x.mul_(-0.5)
    assert clf._side_stream is not None


@pytest.mark.gpu
def test_head_kernel_against_torch_and_batch_independence():
    """swk_nhwc_head2_relu_mean = Conv2d(c, 2, 1) + ReLU + mean over all positions (the ring's share added as a constant), against
    float64 torch; and the property it was written for: a segment's scores are bit-identical whatever batch it is scored in."""
    from swiftwatcher_amd import _lib
    from swiftwatcher_amd.segment_classification import SegmentClassifier
    from oracle import classifier_ref as ref
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(8)
    stream = torch.cuda.current_stream(dev).cuda_stream
    for c, side, n in ((512, 9, 37), (256, 1, 3), (1024, 4, 5), (768, 11, 2)):
        x = torch.randn((n, c, side, side), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        w = (torch.randn((2, c), generator=g) * 0.05).to(dev)
        b = torch.randn((2,), generator=g).to(dev)
        ring = torch.randn((2,), generator=g).to(dev)
        out = torch.empty((n, 2), dtype=torch.float32, device=dev)
        n_pos = float(side * side + 40)
        rc = lib.swk_nhwc_head2_relu_mean(stream, x.data_ptr(), n, side * side, c, w.data_ptr(), b.data_ptr(), ring.data_ptr(), n_pos, out.data_ptr())
        assert rc == 0
        want = (torch.relu(torch.einsum("nchw,kc->nkhw", x.double(), w.double()) + b.double().view(1, 2, 1, 1)).sum(dim=(2, 3)) + ring.double()) / n_pos
        np.testing.assert_allclose(out.cpu().numpy(), want.cpu().numpy(), rtol=2e-6, atol=2e-6)
        assert lib.swk_nhwc_head2_relu_mean(stream, x.data_ptr(), n, side * side, c + 4, w.data_ptr(), b.data_ptr(), ring.data_ptr(), n_pos, out.data_ptr()) != 0
    clf = SegmentClassifier.from_state_dict(ref.random_state_dict(14), batch_size=4096)
    x = torch.randn((2752, 3, 40, 40), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    whole = clf._forward(x).clone()
    for k, row0 in ((96, 0), (33, 64), (2720, 32), (1, 2751)):
        part = clf.cropped(x[row0:row0 + k], row0=row0 if k > 1 else 0)
        assert torch.equal(part, whole[row0:row0 + k]), (k, row0)
    # what the shared-ring reads of the pool + squeeze kernel rely on: outside its live square every segment's pool tile is the first one's
    bufs, _ = clf.cropped._buf
    pools = 0
    for j, (kind, layer, tile, off, n, pad, crop) in enumerate(clf.cropped.plan):
        if kind == "pool":
            t = bufs[j].clone()
            t[:, :, off:off + n, off:off + n] = 0
            assert torch.equal(t, t[0:1].expand_as(t)) and 0 < n < t.shape[2]
            live = bufs[j][:2752, :, off:off + n, off:off + n]
            assert not torch.equal(live[0], live[1])
            pools += 1
    assert pools == 2


@pytest.mark.gpu
def test_fused_maxpool_squeeze_kernel_against_torch():
    """swk_nhwc_maxpool3s2_conv1x1_bias_relu_place (MaxPool2d(3, 2) + a Fire module's squeeze + bias + ReLU + placement as one kernel)
    against torch on the network's three pool -> squeeze pairs (96 -> 16 on 17 x 17, 256 -> 32 on 17 x 17, 512 -> 64 on 19 x 19) and
    on ragged ones; the pooled values are exact, the product is float32 in another summation order: 2e-5 of the output scale."""
    import ctypes
    from swiftwatcher_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(21)
    cases = [  # n, cin, cout, t, dH, off, dC
        (5, 96, 16, 17, 10, 1, 16), (4, 256, 32, 17, 10, 1, 32), (3, 512, 64, 19, 11, 1, 64), (300, 512, 64, 19, 11, 1, 64),
        (2, 64, 8, 7, 3, 0, 8), (7, 128, 48, 13, 8, 2, 64), (1, 32, 64, 5, 2, 0, 64), (2500, 96, 16, 17, 8, 0, 16)]
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for n, cin, cout, t, dH, off, dC in cases:
        x = torch.randn((n, cin, t, t), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        wgt = (torch.randn((cout, cin, 1, 1), generator=g) * (2.0 / cin) ** 0.5).to(dev)
        bias = (torch.randn((cout,), generator=g) * 0.3).to(dev)
        dst = torch.full((n, dC, dH, dH), -7.0, device=dev).contiguous(memory_format=torch.channels_last)
        exp = dst.clone()
        y = torch.relu(torch.nn.functional.conv2d(torch.nn.functional.max_pool2d(x, 3, 2), wgt, bias))
        p = y.shape[2]
        exp[:, :cout, off:off + p, off:off + p] = y
        torch.cuda.synchronize()
        rc = lib.swk_nhwc_maxpool3s2_conv1x1_bias_relu_place(stream, x.data_ptr(), n, t, cin, wgt.reshape(cout, cin).contiguous().data_ptr(),
                                                             bias.data_ptr(), cout, dst.data_ptr(), dH, dH, dC, off, off, None, 0, 0)
        assert rc == 0, (rc, n, cin, cout, t)
        torch.cuda.synchronize()
        scale = max(float(y.abs().max()), 1.0)
        assert float((dst - exp).abs().max()) <= 2e-5 * scale, (n, cin, cout, t)
        mask = torch.ones_like(dst, dtype=torch.bool)
        mask[:, :cout, off:off + p, off:off + p] = False
        assert bool((dst[mask] == -7.0).all())
    # a ring shared by every tile: pixels outside the live square come from ONE tile, the segments' own copies are not read (they hold
    # NaN here), and that tile's live square is not read either
    for n, cin, cout, t, lo, ln in ((37, 256, 32, 17, 2, 12), (300, 512, 64, 19, 3, 14), (5, 96, 16, 17, 0, 17), (9, 64, 8, 9, 4, 1), (3, 64, 8, 9, 2, 0)):
        full = torch.randn((n, cin, t, t), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        inside = torch.zeros((1, 1, t, t), dtype=torch.bool, device=dev)
        inside[:, :, lo:lo + ln, lo:lo + ln] = True
        full = torch.where(inside, full, full[0:1]).contiguous(memory_format=torch.channels_last)      # what a forward leaves in the tiles
        nan = torch.full_like(full, float("nan"))
        src = torch.where(inside, full, nan).contiguous(memory_format=torch.channels_last)
        ring = torch.where(inside, nan[0:1], full[0:1]).contiguous(memory_format=torch.channels_last)
        wgt = (torch.randn((cout, cin, 1, 1), generator=g) * (2.0 / cin) ** 0.5).to(dev)
        bias = (torch.randn((cout,), generator=g) * 0.3).to(dev)
        p = (t - 3) // 2 + 1
        outs = []
        for s_, r_, a_, b_ in ((full, None, 0, 0), (src, ring, lo, ln)):
            dst = torch.zeros((n, cout, p, p), device=dev).contiguous(memory_format=torch.channels_last)
            rc = lib.swk_nhwc_maxpool3s2_conv1x1_bias_relu_place(stream, s_.data_ptr(), n, t, cin, wgt.reshape(cout, cin).contiguous().data_ptr(),
                                                                 bias.data_ptr(), cout, dst.data_ptr(), p, p, cout, 0, 0,
                                                                 None if r_ is None else r_.data_ptr(), a_, b_)
            assert rc == 0
            outs.append(dst)
        torch.cuda.synchronize()
        assert not bool(torch.isnan(outs[1]).any()) and torch.equal(outs[0], outs[1]), (n, cin, t, lo, ln)
    assert lib.swk_nhwc_maxpool3s2_conv1x1_bias_relu_place(stream, x.data_ptr(), 1, 17, 96, wgt.data_ptr(), bias.data_ptr(), 16, dst.data_ptr(),
                                                           8, 8, 16, 0, 0, x.data_ptr(), 5, 13) != 0   # live square leaves the tile
    assert lib.swk_nhwc_maxpool3s2_conv1x1_bias_relu_place(stream, x.data_ptr(), 1, 17, 48, wgt.data_ptr(), bias.data_ptr(), 16, dst.data_ptr(),
                                                           8, 8, 16, 0, 0, None, 0, 0) != 0          # cin not a multiple of 32
    assert lib.swk_nhwc_maxpool3s2_conv1x1_bias_relu_place(stream, x.data_ptr(), 1, 23, 96, wgt.data_ptr(), bias.data_ptr(), 16, dst.data_ptr(),
                                                           11, 11, 16, 0, 0, None, 0, 0) != 0        # 121 pooled pixels: more than three pixel tiles


@pytest.mark.gpu
def test_two_threads_share_a_context_and_a_classifier(tmp_path):
    """A reader that segments and scores ahead (io_frames.PresegmentingReader, pipeline.py windows_per_call) works beside the counting
    loop's own FrameQueue on ONE context and ONE classifier.  Two threads, each with its own clip, hammer both at once: every
    window's kept boxes must equal what the same thread's clip gives when it runs alone.  (What this guards: a batch taking another
    thread's generation -- 'total does not match the batch', seen once on the GPU box -- shared staging arrays, interleaved use of the
    classifier's input slots.)"""
    import threading
    from swiftwatcher_amd import synthetic
    from swiftwatcher_amd.data_structures import FrameQueue
    crop_region = [(40, 30), (40 + 212, 30 + 106)]
    n, windows = 21, 6
    clips = [synthetic.full_frames(70 + c, n * windows, crop_region, frame_hw=(180, 300), birds=8 + 3 * c, bird_len=(14, 24),
                                   bird_wid=(6, 10))[::-1].copy() for c in range(2)]

    def crops_of(clip):
        q = FrameQueue()
        out = []
        q.push_list_of_frames([clip[i] for i in range(n)], list(range(n)), ["t"] * n)
        q.preprocess_queue(crop_region, None)
        q.segment_queue((24, 24), crop_region)
        while not q.is_empty():
            out.extend(s.segment_image for s in q.pop_frame().segments)
        return out
    clf, _, _ = _calibrated_classifier(crops_of(clips[0]) + crops_of(clips[1]), 9, tmp_path, margin=1e-4)

    def run(clip, result, errors):
        try:
            q = FrameQueue()
            for w in range(windows):
                q.push_list_of_frames([clip[w * n + i] for i in range(n)], list(range(w * n, w * n + n)), ["t"] * n)
                q.preprocess_queue(crop_region, None)
                q.segment_queue((24, 24), crop_region)
                while not q.is_empty():
                    f = q.pop_frame()
                    result.append((f.frame_number, [s.bbox for s in f.segments], [s.bbox for s in clf(f.segments)]))
        except Exception as e:          # noqa: BLE001
            errors.append(e)

    alone = []
    for clip in clips:
        res, err = [], []
        run(clip, res, err)
        assert not err, err
        alone.append(res)
    assert all(sum(len(k) for _, _, k in res) > 20 for res in alone)
    for rep in range(2):
        together, errors = [[], []], []
        threads = [threading.Thread(target=run, args=(clips[c], together[c], errors)) for c in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        assert together[0] == alone[0] and together[1] == alone[1]
    assert clf._graph_error is None, clf._graph_error          # no HIP-graph capture was broken by the other thread's library calls


@pytest.mark.gpu
@pytest.mark.parametrize("roi_at", ["inside", "top_left_corner", "bottom_right_corner"])
def test_segment_inputs_of_random_boxes_match_the_host_path(roi_at):
    """swk_segment_inputs on HAND-MADE region records: boxes of every shape (one pixel, a line, the whole ROI), at the ROI's edges and
    corners, with the ROI inside the frame or in one of its corners (a box that extract_segment_images grows past the frame is clipped
    there, image_filtering.py:349-366) -- the network inputs must be the host path's, bit for bit (same crops, same Pillow resize)."""
    from swiftwatcher_amd import _lib
    from swiftwatcher_amd import image_filtering as img
    from swiftwatcher_amd.segment_classification import IMAGENET_MEAN, IMAGENET_STD
    rng = np.random.default_rng({"inside": 1, "top_left_corner": 2, "bottom_right_corner": 3}[roi_at])
    FH, FW, Hc, Wc, F, cap = 150, 260, 90, 170, 6, 24
    x0, y0 = {"inside": (45, 30), "top_left_corner": (0, 0), "bottom_right_corner": (FW - Wc, FH - Hc)}[roi_at]
    crop_region = [(x0, y0), (x0 + Wc, y0 + Hc)]
    frames = rng.integers(0, 256, size=(F, FH, FW, 3), dtype=np.uint8)
    records = np.zeros((F, cap), _lib.SEGMENT_DTYPE)
    counts = np.zeros(F, np.int32)
    for f in range(F):
        k = int(rng.integers(3, cap + 1))
        counts[f] = k
        for i in range(k):
            kind = int(rng.integers(0, 6))
            if kind == 0:          # a single pixel, often on the ROI's border
                r0 = int(rng.choice([0, Hc - 1, rng.integers(0, Hc)])); c0 = int(rng.choice([0, Wc - 1, rng.integers(0, Wc)]))
                r1, c1 = r0 + 1, c0 + 1
            elif kind == 1:        # a horizontal line
                r0 = int(rng.integers(0, Hc)); r1 = r0 + 1; c0 = int(rng.integers(0, Wc - 30)); c1 = int(rng.integers(c0 + 25, Wc + 1))
            elif kind == 2:        # a vertical line
                c0 = int(rng.integers(0, Wc)); c1 = c0 + 1; r0 = int(rng.integers(0, Hc - 30)); r1 = int(rng.integers(r0 + 25, Hc + 1))
            elif kind == 3:        # the whole ROI
                r0, c0, r1, c1 = 0, 0, Hc, Wc
            else:                  # anything
                r0 = int(rng.integers(0, Hc)); r1 = int(rng.integers(r0 + 1, Hc + 1)); c0 = int(rng.integers(0, Wc)); c1 = int(rng.integers(c0 + 1, Wc + 1))
            records[f, i] = (i + 1, r0, c0, r1, c1, 0, (r1 - r0) * (c1 - c0), 0, 0)
    imgs, frame_of = [], []
    for f in range(F):
        rps = img.regionprops_from_records(records[f, :counts[f]])
        crops = img.extract_segment_images(rps, frames[f], (24, 24), crop_region)
        imgs += crops
        frame_of += [f] * len(crops)
    assert len({c.shape[:2] for c in imgs}) > 20
    ctx = _lib.Context(0)
    dev = torch.device("cuda", 0)
    dframes = torch.from_numpy(frames).to(dev)
    segs = torch.from_numpy(records.view(np.uint8).reshape(F, cap, 48)).to(dev)
    nseg = torch.from_numpy(counts).to(dev)
    torch.cuda.synchronize()
    inp = _lib.Input(frames=dframes.data_ptr(), mem=_lib.MEM_DEVICE, channels=3, nwin=1, n=F, Hc=Hc, Wc=Wc, x0=x0, y0=y0,
                     frame_stride=FH * FW * 3, row_stride=FW * 3)
    for pad, nhwc in ((100, False), (8, False), (8, True)):
        side = 24 + 2 * pad
        fmt = torch.channels_last if nhwc else torch.contiguous_format
        x = torch.empty((len(imgs), 3, side, side), dtype=torch.float32, device=dev, memory_format=fmt)
        fidx = torch.empty((len(imgs),), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        total, skipped = ctx.segment_inputs(inp, (FH, FW), segs.data_ptr(), nseg.data_ptr(), cap, IMAGENET_MEAN, IMAGENET_STD,
                                            x.data_ptr(), len(imgs), pad=pad, seg_frame_ptr=fidx.data_ptr(), channels_last=nhwc)
        assert total == len(imgs) and skipped == 0
        assert fidx.cpu().tolist() == frame_of
        _, net = ctx.classifier_input(imgs, IMAGENET_MEAN, IMAGENET_STD, pad=pad)
        np.testing.assert_array_equal(x.cpu().numpy(), net, err_msg="pad %d nhwc %s" % (pad, nhwc))
    ctx.close()

"""Pin the CPU oracle against fixtures generated from the reference itself
(oracle/make_goldens.py, run under /opt/conda/bin/python3.9 in the build container)."""
import os

import numpy as np
import pytest

from oracle import reference_path as orc


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_crop_regions(golden_dir):
    g = _load(golden_dir, "crop_regions.npz")
    frame = np.random.default_rng(int(g["frame_seed"])).integers(0, 256, size=(1080, 1920, 3), dtype=np.uint8)
    for corners, region, ext, s, shp in zip(g["corners"], g["regions"], g["extents"], g["crop_sums"], g["crop_shapes"]):
        c = [tuple(int(v) for v in corners[0]), tuple(int(v) for v in corners[1])]
        assert orc.chimney_extents(c) == tuple(int(v) for v in ext)
        got = orc.crop_region_from_corners(c)
        assert [tuple(p) for p in got] == [tuple(int(v) for v in p) for p in region]
        roi = orc.crop(frame, got)
        assert roi.shape == tuple(shp) and int(roi.astype(np.int64).sum()) == int(s)


def test_grey_opening(golden_dir):
    g = _load(golden_dir, "grey_opening.npz")
    for i in range(int(g["count"])):
        np.testing.assert_array_equal(orc.grey_open_u8(g["in%d" % i], (3, 3)), g["out%d" % i])
    np.testing.assert_array_equal(orc.grey_open_u8(g["in_5x3"], (5, 3)), g["out_5x3"])


def test_segment_crop_boxes(golden_dir):
    g = _load(golden_dir, "segment_crops.npz")
    frame = np.random.default_rng(int(g["frame_seed"])).integers(0, 256, size=(1080, 1920, 3), dtype=np.uint8)
    cr = [tuple(int(v) for v in g["crop_region"][0]), tuple(int(v) for v in g["crop_region"][1])]
    for bbox, shp, s, fp, lp in zip(g["bboxes"], g["shapes"], g["sums"], g["first_px"], g["last_px"]):
        r0, c0, r1, c1 = orc.segment_crop_box(tuple(int(v) for v in bbox), (24, 24), cr)
        im = frame[r0:r1, c0:c1]
        assert im.shape == tuple(shp) and int(im.astype(np.int64).sum()) == int(s)
        np.testing.assert_array_equal(im[0, 0], fp)
        np.testing.assert_array_equal(im[-1, -1], lp)


def test_regionprops(golden_dir):
    g = _load(golden_dir, "regionprops.npz")
    for i in range(int(g["count"])):
        got = orc.regionprops_u8(g["lab%d" % i])
        assert [s["label"] for s in got] == list(g["labels%d" % i])
        assert [s["bbox"] for s in got] == [tuple(b) for b in g["bbox%d" % i]]
        assert [s["area"] for s in got] == list(g["area%d" % i])
        # centroid from integer sums must be bit-identical to coords.mean(axis=0)
        np.testing.assert_array_equal(np.array([s["centroid"] for s in got]), g["centroid%d" % i])


# ialm_47x94x21 = BASELINE config 1's ROI (480p clip, chimney 76 px wide) at the CLI's queue size: 92.8 k elements, where
# 0.008 ||X||_F < 1.8 max(X) and the first shrinkage already clips the brightest sky pixels (23 iterations).
IALM_CASES = ["ialm_128x160x7", "ialm_64x96x21", "ialm_40x48x64", "ialm_64x96x64", "ialm_107x214x21", "ialm_47x94x21",
              "ialm_47x94x64", "ialm_47x94x64_quiet", "ialm_30x40x64"]          # round 4: ill-conditioned windows (oracle/make_goldens_r4.py)
# the BASELINE workload sizes: inputs regenerated from the seed (oracle/scenes.py), outputs of the reference stored as
# iteration count, sha256 of the sparse image and A / E on sampled pixel rows (oracle/make_goldens_r2.py)
SEEDED_CASES = ["ialm_212x424x21_seeded", "ialm_212x424x64_seeded", "ialm_425x850x21_seeded"]
# round 3 (oracle/make_goldens_r3.py): config 1's window size on both sides of -- two of them within 0.5 % of -- the switch between the
# integer start and the f64 start pass (1.8 max(X) against 0.008 ||X||_F), and two more at 214 x 107 x 21
R3_CASES = ["ialm_47x94x21_s301", "ialm_47x94x21_s305", "ialm_47x94x21_s302", "ialm_47x94x21_s303", "ialm_47x94x21_s304",
            "ialm_107x214x21_s311", "ialm_107x214x21_s312"]


def seeded_frames(g):
    """Regenerate a seeded fixture's input window and prove it is the one the reference saw."""
    from oracle.scenes import scene, sha256
    n, H, W = (int(v) for v in g["shape"])
    frames = scene(np.random.default_rng(int(g["seed"])), n, H, W, blobs=int(g["blobs"]))
    if "offset" in g.files:                      # round-3 fixtures: brightness offset (moves the side of the start switch)
        frames = np.clip(frames.astype(np.int32) + int(g["offset"]), 0, 255).astype(np.uint8)
    assert sha256(frames) == str(g["frames_sha256"]), "scene generator no longer reproduces the fixture's input"
    return frames


@pytest.mark.parametrize("name", SEEDED_CASES + R3_CASES)
def test_ialm_numpy_restatement_at_baseline_sizes(golden_dir, name):
    from oracle.scenes import sha256
    g = _load(golden_dir, name + ".npz")
    frames = seeded_frames(g)
    n, H, W = frames.shape
    A, E, k = orc.ialm(np.transpose(frames.reshape(n, H * W)), return_iters=True)
    assert k == int(g["iters"])
    rows = g["rows"]
    np.testing.assert_allclose(A[rows], g["A_rows"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(E[rows], g["E_rows"], atol=1e-6, rtol=0)
    sparse = orc.rpca_epilogue(E).T.reshape(n, H, W)
    np.testing.assert_array_equal(sparse.reshape(n, -1).astype(np.int64).sum(axis=1), g["sparse_frame_sums"])
    assert sha256(sparse) == str(g["sparse_sha256"])
    if "sparse" in g.files and g["sparse"].size:
        np.testing.assert_array_equal(sparse, g["sparse"])


@pytest.mark.parametrize("name", IALM_CASES)
def test_ialm_numpy_restatement(golden_dir, name):
    g = _load(golden_dir, name + ".npz")
    frames = g["frames"]
    n, H, W = frames.shape
    X = np.transpose(frames.reshape(n, H * W))
    A, E, k = orc.ialm(X, return_iters=True)
    assert k == int(g["iters"])
    rows = g["rows"]
    real = slice(int(g["null_frames"]), None)     # padded (all-zero) columns are LAPACK-arbitrary
    np.testing.assert_allclose(A[rows][:, real], g["A_rows"][:, real], atol=1e-6, rtol=0)
    np.testing.assert_allclose(E[rows][:, real], g["E_rows"][:, real], atol=1e-6, rtol=0)
    sparse = np.stack(orc.rpca(list(frames)))
    np.testing.assert_array_equal(sparse[real], g["sparse"][real])


def test_ialm_null_padded_window_is_defined(golden_dir):
    """Padded windows (the last window of every video, io_video.py:40-44).  The reference's own output there is
    LAPACK-arbitrary: a zero column gives the SVD a zero singular value, the always-full `svp` (:285) turns its
    arbitrary left vector u into A[:, j] = -(1/mu) u with entries of hundreds of grey levels, and that leaks into
    the real frames from iteration 2 on (see ialm_defined).  This project's DEFINED behaviour excludes the null
    columns.  The fixture is what the reference produced under numpy 1.26 in the build container; this test states
    exactly how far the three parties are apart instead of allowing a margin:

      reference on numpy 1.26 vs the same statements on numpy 2.2:  3.1 % of the real-frame pixels differ, by <= 2
      defined vs reference (1.26):                                   5.8 % differ, by <= 9 grey levels
      pixels above the to-zero threshold of 15 (what segmentation sees): 542 in the reference, 539 defined, 539 common
    """
    g = _load(golden_dir, "ialm_64x96x21_null5.npz")
    frames = g["frames"]
    n, H, W = frames.shape
    nz = int(g["null_frames"])
    X = np.transpose(frames.reshape(n, H * W))
    A, E, k = orc.ialm_defined(X, return_iters=True)
    assert not A[:, :nz].any() and not E[:, :nz].any()
    # zero-weight rule == leaving the null columns out of the decomposition (to the rounding of two different SVDs)
    A2, E2, k2 = orc.ialm(X[:, nz:], return_iters=True)
    assert k == k2
    np.testing.assert_allclose(A[:, nz:], A2, atol=1e-8, rtol=0)
    np.testing.assert_allclose(E[:, nz:], E2, atol=1e-8, rtol=0)
    sparse = np.stack(orc.rpca(list(frames)))
    assert not sparse[:nz].any()
    ref = g["sparse"][nz:].astype(int)
    d = np.abs(sparse[nz:].astype(int) - ref)
    assert (d > 0).mean() < 0.06 and d.max() <= 9                     # measured: 0.0575, 9
    fg_ref, fg_def = ref > 15, sparse[nz:] > 15
    assert int(fg_ref.sum()) == 542 and int((fg_ref & fg_def).sum()) == int(fg_def.sum()) >= 539
    # the reference is not reproducible against ITSELF on these windows: the faithful restatement under this
    # interpreter's LAPACK differs from the fixture too (wherever that stops being true the fixture can be pinned)
    _, E22 = orc.ialm(X)
    s22 = orc.rpca_epilogue(E22).T.reshape(n, H, W)[nz:].astype(int)
    d22 = np.abs(s22 - ref)
    assert d22.max() <= 3 and (d22 > 0).mean() < 0.05
    # downstream of the filters: same segment count on all but at most one real frame, same boxes on most
    same_count = same_boxes = 0
    for i in range(n - nz):
        segs = []
        for img in (g["sparse"][nz + i], sparse[nz + i]):
            opened = orc.grey_open_u8(orc.thresh_tozero_u8(orc.bilateral_u8(img)))
            segs.append(orc.regionprops_u8(orc.labels_to_u8(orc.ccl_u8(opened)[1])))
        same_count += len(segs[0]) == len(segs[1])
        same_boxes += [s["bbox"] for s in segs[0]] == [s["bbox"] for s in segs[1]]
    assert same_count >= n - nz - 1 and same_boxes >= n - nz - 2       # measured: 15 and 14 of 16


def test_ccl_matches_scipy_raster_order():
    """4- and 8-connected labelling in first-pixel raster order == scipy.ndimage.label
    (the numbering OpenCV's SAUF produces); the 2x2-block order is a re-ranking."""
    from scipy import ndimage
    rng = np.random.default_rng(3)
    for shape, dens in [((31, 45), 0.45), ((64, 64), 0.6), ((5, 9), 0.5), ((212, 424), 0.08)]:
        img = (rng.random(shape) < dens).astype(np.uint8) * 200
        for conn, st in [(4, ndimage.generate_binary_structure(2, 1)), (8, np.ones((3, 3), int))]:
            ref, nref = ndimage.label(img, structure=st)
            n, lab = orc.ccl_u8(img, conn, 0)
            assert n == nref
            np.testing.assert_array_equal(lab, ref)
            nb, labb = orc.ccl_u8(img, conn, 1)
            assert nb == n
            if conn == 4:       # block order is an 8-way (BBDT) rule; 4-way ignores it
                np.testing.assert_array_equal(labb, lab)
                continue
            # same partition, numbered by first 2x2 block in block-raster order
            pairs = np.unique(np.stack([lab.ravel(), labb.ravel()]), axis=1)
            assert pairs.shape[1] == n + (1 if (img == 0).any() else 0)
            Wb = (shape[1] + 1) // 2
            rr, cc = np.nonzero(labb)
            key = (rr >> 1) * Wb + (cc >> 1)
            first = np.full(nb + 1, np.iinfo(np.int64).max)
            np.minimum.at(first, labb[rr, cc], key)
            assert np.all(np.diff(first[1:]) > 0)


def test_bgr2gray_and_threshold_spec():
    """Specification tests for the cv2-backed integer rules (PARITY UNPINNED)."""
    px = np.array([[[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 20, 30]]], np.uint8)
    got = orc.bgr2gray(px)[0]
    exp = [(b * 1868 + g * 9617 + r * 4899 + 8192) >> 14 for b, g, r in px[0].astype(int)]
    assert list(got) == exp and got[0] == 0 and got[1] == 255
    src = np.arange(256, dtype=np.uint8).reshape(16, 16)
    out = orc.thresh_tozero_u8(src, 15)
    assert np.all(out[src <= 15] == 0) and np.all(out[src > 15] == src[src > 15])


def test_bilateral_spec():
    """Bilateral: constant image is a fixed point; an isolated spike uses the 29-tap disc."""
    flat = np.full((12, 13), 77, np.uint8)
    np.testing.assert_array_equal(orc.bilateral_u8(flat), flat)
    img = np.zeros((15, 15), np.uint8)
    img[7, 7] = 200
    out = orc.bilateral_u8(img)
    # centre: colour weight of |0-200| at sigma 15 underflows to ~0 => stays 200
    assert out[7, 7] == 200
    # python restatement of the tap loop for one neighbour pixel
    cw = np.exp(-0.5 * (np.arange(256) ** 2) / 15.0 ** 2).astype(np.float32)
    s = np.float32(0); ws = np.float32(0)
    for i in range(-3, 4):
        for j in range(-3, 4):
            r = np.sqrt(float(i * i + j * j))
            if r > 3:
                continue
            sw = np.float32(np.exp(-0.5 * r * r))
            v = int(img[7 + i, 8 + j]); w = np.float32(sw * cw[abs(v - 0)])
            s = np.float32(s + np.float32(np.float32(v) * w)); ws = np.float32(ws + w)
    assert out[7, 8] == int(np.rint(np.float32(s / ws)))


def test_grey_opening_other_windows_fixture(golden_dir):
    """Round 4: windows other than (3, 3), even sizes included, made by the reference's own grayscale_opening (scipy)."""
    g = _load(golden_dir, "grey_opening_windows.npz")
    for i in range(int(g["count"])):
        for kh, kw in g["sizes"]:
            np.testing.assert_array_equal(orc.grey_open_u8(g["in%d" % i], (int(kh), int(kw))), g["out%d_%dx%d" % (i, kh, kw)])


def test_resize_restatement_properties():
    """cv2.resize is installed nowhere (PARITY UNPINNED): what the restated arithmetic guarantees by itself."""
    rng = np.random.default_rng(3)
    im = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    np.testing.assert_array_equal(orc.resize_linear_u8(im, (53, 37)), im)
    assert (orc.resize_linear_u8(np.full((20, 30), 99, np.uint8), (300, 150)) == 99).all()
    up = orc.resize_linear_u8(im[:, :, 0], (106, 74)).astype(int)
    assert up.min() >= im[:, :, 0].min() and up.max() <= im[:, :, 0].max()

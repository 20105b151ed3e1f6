"""Generate tests/golden/*.npz from the REFERENCE ITSELF.  Test infrastructure only.

Run in the build container (never on the GPU box; /root/reference does not travel):

    /opt/conda/bin/python3.9 oracle/make_goldens.py

The reference's functions are imported unchanged from /root/reference.  Its modules
start with `import cv2`; opencv-python is installed nowhere in the image, so an EMPTY
placeholder module named cv2 satisfies the import line.  No cv2 behaviour is emulated:
only reference functions that never touch cv2 are called (crop geometry, rpca / IALM,
grayscale_opening, extract_segment_images) plus skimage.measure.regionprops, which
get_segment_properties wraps (its `coordinates='xy'` kwarg was removed after skimage
0.15 and raises on the installed 0.18.3; the call without the kwarg is used instead
and yields the same label/bbox/centroid/area definitions).

The fixtures hold inputs and expected outputs only -- data, no reference source.
"""
import os
import sys
import types

import numpy as np

sys.modules.setdefault("cv2", types.ModuleType("cv2"))
sys.path.insert(0, "/root/reference")
import swiftwatcher.image_filtering as ref_img  # noqa: E402
from skimage import measure  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from scenes import scene  # noqa: E402  (this project's own scene generator, shared with the tests)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def ialm_case(name, seed, n, H, W, null_frames=0, sample_every=1):
    rng = np.random.default_rng(seed)
    frames = scene(rng, n, H, W)
    if null_frames:
        frames[:null_frames] = 0          # newest-first queue: padded frames sit at index 0..
    X = np.transpose(frames.reshape(n, H * W))
    # iteration count: re-run the loop's stopping rule is internal, so recover K from verbose print
    import io
    import contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        A, E = ref_img.inexact_augmented_lagrange_multiplier(X, verbose=True)
    iters = int(buf.getvalue().strip().split()[-1])
    sparse = np.stack(ref_img.rpca(list(frames)))
    rows = np.arange(0, H * W, sample_every)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), frames=frames, iters=np.int32(iters),
        rows=rows.astype(np.int64), A_rows=A[rows], E_rows=E[rows],
        A_colsum=A.sum(axis=0), E_colsum=E.sum(axis=0),
        A_abs_sum=np.abs(A).sum(), E_abs_sum=np.abs(E).sum(), sparse=sparse,
        null_frames=np.int32(null_frames))
    print(name, "iters", iters, "nnz sparse", int((sparse > 0).sum()))


def crop_cases():
    corners = [((790, 620), (1130, 622)), ((874, 620), (1046, 623)), ((1130, 622), (790, 620)),
               ((100, 300), (176, 298)), ((610, 900), (1290, 905)), ((15, 40), (40, 41)),
               ((333, 777), (1001, 770))]
    regions = [ref_img.generate_crop_region(c) for c in corners]
    extents = [ref_img.determine_chimney_extents(c) for c in corners]
    rng = np.random.default_rng(5)
    frame = rng.integers(0, 256, size=(1080, 1920, 3), dtype=np.uint8)
    sums = [int(ref_img.crop_frame(frame, r).astype(np.int64).sum()) for r in regions]
    shapes = [ref_img.crop_frame(frame, r).shape for r in regions]
    np.savez_compressed(os.path.join(OUT, "crop_regions.npz"), corners=np.array(corners),
                        regions=np.array(regions), extents=np.array(extents),
                        crop_sums=np.array(sums), crop_shapes=np.array(shapes), frame_seed=np.int32(5))
    print("crop_regions", regions)


def opening_cases():
    rng = np.random.default_rng(11)
    imgs = [rng.integers(0, 256, size=(37, 53), dtype=np.uint8),
            (rng.random((64, 96)) > 0.7).astype(np.uint8) * rng.integers(16, 256, size=(64, 96), dtype=np.uint8),
            np.zeros((9, 9), np.uint8), np.full((5, 7), 200, np.uint8),
            rng.integers(0, 256, size=(3, 3), dtype=np.uint8),
            rng.integers(0, 256, size=(1, 17), dtype=np.uint8)]
    imgs[2][4, 4] = 255
    d = {}
    for i, im in enumerate(imgs):
        d["in%d" % i] = im
        d["out%d" % i] = ref_img.grayscale_opening(im, (3, 3))
    d["in_5x3"] = imgs[0]
    d["out_5x3"] = ref_img.grayscale_opening(imgs[0], (5, 3))
    d["count"] = np.int32(len(imgs))
    np.savez_compressed(os.path.join(OUT, "grey_opening.npz"), **d)
    print("grey_opening", len(imgs))


class _Seg:
    def __init__(self, bbox):
        self.bbox = bbox


def segment_crop_cases():
    rng = np.random.default_rng(17)
    frame = rng.integers(0, 256, size=(1080, 1920, 3), dtype=np.uint8)
    crop_region = [(748, 452), (1172, 664)]
    bboxes = [(10, 10, 15, 13), (0, 0, 3, 3), (100, 200, 131, 260), (50, 60, 73, 84), (50, 60, 75, 61),
              (200, 400, 212, 424), (7, 9, 30, 10), (20, 30, 44, 54), (20, 30, 45, 55), (1, 1, 2, 24)]
    segs = [_Seg(b) for b in bboxes]
    imgs = ref_img.extract_segment_images(segs, frame, (24, 24), crop_region)
    np.savez_compressed(os.path.join(OUT, "segment_crops.npz"), frame_seed=np.int32(17),
                        crop_region=np.array(crop_region), bboxes=np.array(bboxes),
                        shapes=np.array([im.shape for im in imgs]),
                        sums=np.array([int(im.astype(np.int64).sum()) for im in imgs]),
                        first_px=np.array([im[0, 0] for im in imgs]),
                        last_px=np.array([im[-1, -1] for im in imgs]))
    print("segment_crops", [im.shape for im in imgs])


def regionprops_cases():
    rng = np.random.default_rng(23)
    labs = []
    a = np.zeros((40, 60), np.uint8)
    a[2:7, 3:9] = 1; a[10:12, 20:40] = 2; a[30, 59] = 3; a[39, 0] = 7; a[15:25, 45:47] = 200
    labs.append(a)
    b = np.zeros((64, 64), np.uint8)           # split label (u8 wrap makes 257 alias 1)
    b[0:4, 0:4] = 1; b[50:60, 40:64] = 1; b[20:22, 20:22] = 255
    labs.append(b)
    c = rng.integers(0, 6, size=(33, 47)).astype(np.uint8)
    labs.append(c)
    d = {}
    for i, lab in enumerate(labs):
        props = measure.regionprops(lab)
        d["lab%d" % i] = lab
        d["labels%d" % i] = np.array([p.label for p in props], np.int64)
        d["bbox%d" % i] = np.array([p.bbox for p in props], np.int64)
        d["centroid%d" % i] = np.array([p.centroid for p in props], np.float64)
        d["area%d" % i] = np.array([p.area for p in props], np.int64)
    d["count"] = np.int32(len(labs))
    np.savez_compressed(os.path.join(OUT, "regionprops.npz"), **d)
    print("regionprops", [len(d["labels%d" % i]) for i in range(len(labs))])


if __name__ == "__main__":
    crop_cases()
    opening_cases()
    segment_crop_cases()
    regionprops_cases()
    # Sizes are chosen so the first IALM iteration is NOT rank deficient.  With the
    # reference's always-full `svp` (image_filtering.py:285) iteration 1 sees
    # M = clamp(1.8*X, +-lambda/mu) with lambda/mu = 0.008*||X||_F; when the window is so
    # small that 0.008*||X||_F < 1.8*max(X) the clamp saturates, M loses rank and the
    # result depends on LAPACK's arbitrary null-space vectors (observed: numpy 1.26 and 2.2
    # disagree by 0.5 grey levels on 16x16x7).  P*n >= ~1.1e5 avoids it for 8-bit sky scenes.
    ialm_case("ialm_64x96x21", 101, 21, 64, 96, sample_every=3)
    ialm_case("ialm_40x48x64", 102, 64, 40, 48, sample_every=2)      # few pixels per frame: ill-conditioned
    ialm_case("ialm_64x96x64", 106, 64, 64, 96, sample_every=8)
    ialm_case("ialm_107x214x21", 103, 21, 107, 214, sample_every=53)
    ialm_case("ialm_128x160x7", 105, 7, 128, 160, sample_every=11)
    ialm_case("ialm_64x96x21_null5", 104, 21, 64, 96, null_frames=5, sample_every=3)

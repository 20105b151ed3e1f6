"""Round-4 golden vectors from the REFERENCE ITSELF for ill-conditioned windows: few pixels per frame against a long queue
(config 1's 94 x 47 ROI with FrameQueue(queue_size=64): 4,418 pixels x 64 frames), where cond(M) reaches 1e5 in the last iterations and
the Gram-matrix route of the HIP path (G = M^T M squares it) is at its weakest.  Test infrastructure only.

    /opt/conda/bin/python3.9 oracle/make_goldens_r4.py          (build container; /root/reference does not travel)

Same recipe as make_goldens_r2.py (reference functions imported unchanged behind an empty cv2 placeholder).  Small windows: the frames
are stored whole, A / E on every seventh pixel row.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_goldens_r2 import run_reference, OUT          # noqa: E402  (imports the reference behind the cv2 placeholder)
from scenes import scene                                # noqa: E402


def case(name, seed, n, H, W, blobs, noise):
    frames = scene(np.random.default_rng(seed), n, H, W, blobs=blobs, noise=noise)
    A, E, iters, sparse = run_reference(frames)
    rows = np.arange(0, H * W, 7)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), frames=frames, iters=np.int32(iters), rows=rows.astype(np.int64),
        A_rows=A[rows], E_rows=E[rows], A_colsum=A.sum(axis=0), E_colsum=E.sum(axis=0),
        A_abs_sum=np.abs(A).sum(), E_abs_sum=np.abs(E).sum(), sparse=sparse, null_frames=np.int32(0),
        numpy_version=np.__version__)
    print(name, "iters", iters, "nnz sparse", int((sparse > 0).sum()))


def opening_windows():
    """grayscale_opening (image_filtering.py:319-322, the reference's own function: scipy.ndimage.grey_opening(size=SE)) with windows
    other than the (3, 3) of the loop: odd, even, rectangular, 1-wide; on noise, on structured blobs and on an image smaller than the
    window."""
    import make_goldens_r2 as r2
    rng = np.random.default_rng(404)
    yy, xx = np.mgrid[0:45, 0:61]
    blobs = (120 + 100 * np.sin(yy / 5.0) * np.cos(xx / 7.0)).clip(0, 255).astype(np.uint8)
    blobs[rng.random(blobs.shape) < 0.05] = 255
    imgs = [rng.integers(0, 256, size=(37, 53), dtype=np.uint8), blobs, rng.integers(0, 256, size=(4, 6), dtype=np.uint8)]
    sizes = [(5, 5), (3, 7), (7, 3), (1, 5), (2, 2), (4, 4), (2, 5), (6, 3), (9, 9)]
    d = dict(count=np.int32(len(imgs)), sizes=np.array(sizes, np.int32))
    for i, im in enumerate(imgs):
        d["in%d" % i] = im
        for kh, kw in sizes:
            d["out%d_%dx%d" % (i, kh, kw)] = r2.ref_img.grayscale_opening(im, (kh, kw))
    np.savez_compressed(os.path.join(OUT, "grey_opening_windows.npz"), **d)
    print("grey_opening_windows", len(imgs), "images x", len(sizes), "windows")


if __name__ == "__main__":
    opening_windows()
    if "--openings-only" in sys.argv:
        sys.exit(0)
    case("ialm_47x94x64", 401, 64, 47, 94, blobs=3, noise=2.5)          # config 1's ROI, queue of 64
    case("ialm_47x94x64_quiet", 402, 64, 47, 94, blobs=3, noise=0.6)    # the same with a quiet sensor: smaller sigma_min still
    case("ialm_30x40x64", 403, 64, 30, 40, blobs=2, noise=2.5)          # 1,200 pixels for 64 frames

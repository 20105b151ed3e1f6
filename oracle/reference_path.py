"""CPU oracle for swiftwatcher's segment path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package (swiftwatcher_amd/) never does.

Two halves:

* numpy restatement of the float path (RPCA by inexact ALM), following the
  reference statement by statement, SVD included (image_filtering.py:220-301).
  PINNED: tests/golden/ialm_*.npz hold A, E, iteration counts and the uint8
  sparse images produced by the reference's own functions (imported unchanged
  under /opt/conda/bin/python3.9 with an empty placeholder `cv2` module; see
  oracle/make_goldens.py), and tests/test_oracle_golden.py checks this file
  against them.
* ctypes wrappers over oracle/swk_oracle.c for the integer/byte stages.
  grey opening, region properties and segment-crop geometry are PINNED by
  fixtures from the reference (scipy / skimage); BGR2GRAY, bilateral, threshold
  and connected components live in opencv-python 4.1.0.25 which is installed
  nowhere in the build image: PARITY UNPINNED, restated from OpenCV 4.1.0's
  published algorithms (details in swk_oracle.c).
"""
import ctypes
import math
import os

import numpy as np

from . import build as _build

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = _build.LIB if os.path.exists(_build.LIB) else _build.build()
        _LIB = ctypes.CDLL(path)
    return _LIB


class OrcSegment(ctypes.Structure):
    _fields_ = [("label", ctypes.c_int32), ("r0", ctypes.c_int32), ("c0", ctypes.c_int32),
                ("r1", ctypes.c_int32), ("c1", ctypes.c_int32), ("pad_", ctypes.c_int32),
                ("area", ctypes.c_int64), ("sum_r", ctypes.c_int64), ("sum_c", ctypes.c_int64)]


def _p(a, t=ctypes.c_void_p):
    return a.ctypes.data_as(t)


# --------------------------------------------------------------------------
# crop geometry: image_filtering.py:31-53, 78-91 (pure integer arithmetic)
# --------------------------------------------------------------------------
def chimney_extents(corners):
    xs = (corners[0][0], corners[1][0])
    ys = (corners[0][1], corners[1][1])
    return min(xs), max(xs), max(ys)


def crop_region_from_corners(corners):
    left, right, bottom = chimney_extents(corners)
    w = right - left
    margin = int(0.125 * w)
    return [(left - margin, bottom - int(0.5 * w)), (right + margin, bottom + margin)]


def crop(frame, crop_region):
    (x0, y0), (x1, y1) = crop_region
    return frame[y0:y1, x0:x1]


# --------------------------------------------------------------------------
# integer / byte stages (C)
# --------------------------------------------------------------------------
def bgr2gray(bgr, mode=0):
    """image_filtering.py:188-196.  2-D input passes through (:193-194)."""
    if bgr.ndim == 2:
        return bgr
    H, W, _ = bgr.shape
    assert bgr.dtype == np.uint8 and bgr.strides[2] == 1 and bgr.strides[1] == 3
    out = np.empty((H, W), np.uint8)
    lib().orc_bgr2gray(_p(bgr), H, W, ctypes.c_int64(bgr.strides[0]), int(mode), _p(out))
    return out


def rpca_epilogue(E):
    """image_filtering.py:244-245."""
    E = np.ascontiguousarray(E, np.float64)
    out = np.empty(E.shape, np.uint8)
    lib().orc_rpca_epilogue(_p(E), ctypes.c_int64(E.size), _p(out))
    return out


def bilateral_u8(src, d=7, sigma_color=15.0, sigma_space=1.0, use_fma=False):
    """image_filtering.py:304-307 (PARITY UNPINNED)."""
    src = np.ascontiguousarray(src, np.uint8)
    H, W = src.shape
    out = np.empty_like(src)
    lib().orc_bilateral_u8(_p(src), H, W, int(d), ctypes.c_double(sigma_color),
                           ctypes.c_double(sigma_space), int(bool(use_fma)), _p(out))
    return out


def thresh_tozero_u8(src, thresh=15):
    """image_filtering.py:310-316 (PARITY UNPINNED, trivial)."""
    src = np.ascontiguousarray(src, np.uint8)
    out = np.empty_like(src)
    lib().orc_thresh_tozero_u8(_p(src), ctypes.c_int64(src.size), int(thresh), _p(out))
    return out


def grey_open_u8(src, size=(3, 3)):
    """image_filtering.py:319-322 (PINNED vs scipy via the reference)."""
    src = np.ascontiguousarray(src, np.uint8)
    H, W = src.shape
    out = np.empty_like(src)
    rc = lib().orc_grey_open_u8(_p(src), H, W, int(size[0]), int(size[1]), _p(out))
    if rc:
        raise ValueError("bad opening window")
    return out


def resize_linear_u8(src, dsize):
    """image_filtering.py:206-212: cv2.resize(frame, dsize=(width, height)), INTER_LINEAR (PARITY UNPINNED; dead code in the reference)."""
    src = np.ascontiguousarray(src, np.uint8)
    H, W = src.shape[:2]
    ch = 1 if src.ndim == 2 else src.shape[2]
    dW, dH = int(dsize[0]), int(dsize[1])
    out = np.empty((dH, dW) + ((ch,) if src.ndim == 3 else ()), np.uint8)
    if lib().orc_resize_linear_u8(_p(src), H, W, ch, dH, dW, _p(out)):
        raise MemoryError
    return out


def ccl_u8(src, connectivity=8, order=1):
    """image_filtering.py:325-329 (PARITY UNPINNED).  Returns (count, int32 labels)."""
    src = np.ascontiguousarray(src, np.uint8)
    H, W = src.shape
    lab = np.empty((H, W), np.int32)
    n = lib().orc_ccl_u8(_p(src), H, W, int(connectivity), int(order), _p(lab))
    if n < 0:
        raise MemoryError
    return n, lab


def labels_to_u8(lab):
    return (lab & 0xff).astype(np.uint8)


def regionprops_u8(lab8):
    """image_filtering.py:332-335: list of dicts label/bbox/area/sum_r/sum_c/centroid."""
    lab8 = np.ascontiguousarray(lab8, np.uint8)
    H, W = lab8.shape
    buf = (OrcSegment * 255)()
    n = lib().orc_regionprops_u8(_p(lab8), H, W, buf)
    out = []
    for i in range(n):
        s = buf[i]
        out.append(dict(label=int(s.label), bbox=(int(s.r0), int(s.c0), int(s.r1), int(s.c1)),
                        area=int(s.area), sum_r=int(s.sum_r), sum_c=int(s.sum_c),
                        centroid=(s.sum_r / s.area, s.sum_c / s.area)))
    return out


def segment_crop_box(bbox, min_seg_size, crop_region):
    """image_filtering.py:349-362: expand bbox to >= min size, translate to full frame.
    Returns (r0, c0, r1, c1) used to slice the full frame (:363-365)."""
    b = list(bbox)
    h, w = b[2] - b[0], b[3] - b[1]
    if h < min_seg_size[0]:
        d = min_seg_size[0] - h
        b[0] -= math.floor(d / 2)
        b[2] += math.ceil(d / 2)
    if w < min_seg_size[1]:
        d = min_seg_size[1] - w
        b[1] -= math.floor(d / 2)
        b[3] += math.ceil(d / 2)
    oy, ox = crop_region[0][1], crop_region[0][0]
    return (b[0] + oy, b[1] + ox, b[2] + oy, b[3] + ox)


# --------------------------------------------------------------------------
# float path: RPCA via inexact ALM (numpy, SVD, faithful)
# --------------------------------------------------------------------------
def ialm(X, lmbda=0.01, tol=0.001, maxiter=100, return_iters=False):
    """image_filtering.py:256-301, statement by statement.

    X is (pixels, frames), any real dtype (the reference passes uint8).  Note the
    reference's `svp = (S > 1/mu).shape[0]` (:285) is the LENGTH of the boolean
    vector, i.e. always len(S): every singular value is shifted by 1/mu, the
    negative results included.  Reproduced as is.
    """
    X = np.asarray(X)
    flat = X.ravel()
    two_norm = np.linalg.norm(flat, 2)                       # :269
    inf_norm = np.linalg.norm(flat, np.inf) / lmbda           # :270
    scale = np.max([two_norm, inf_norm])                      # :271
    Y = X / scale                                             # :272
    A = np.zeros(Y.shape)
    E = np.zeros(Y.shape)
    x_fro = np.linalg.norm(X, 'fro')                          # :275
    mu = 1.25 / two_norm                                      # :276
    growth = 1.5
    k = 0
    while True:
        raw = X - A + (1 / mu) * Y                            # :282
        E_next = np.maximum(raw - lmbda / mu, 0) + np.minimum(raw + lmbda / mu, 0)   # :283
        U, S, Vt = np.linalg.svd(X - E_next + (1 / mu) * Y, full_matrices=False)     # :284
        keep = S.shape[0]                                     # :285 (always all of them)
        A = np.dot(np.dot(U[:, :keep], np.diag(S[:keep] - 1 / mu)), Vt[:keep, :])    # :290
        E = E_next
        Z = X - A - E                                         # :293
        Y = Y + mu * Z                                        # :294
        mu = np.min([mu * growth, mu * 1e7])                  # :295
        k += 1
        if (np.linalg.norm(Z, 'fro') / x_fro) < tol or k >= maxiter:   # :297
            break
    if return_iters:
        return A, E, k
    return A, E


DEAD_EIG = 1e-13          # eigenvalues of M^T M below DEAD_EIG * lambda_max are zero singular directions


def ialm_defined(X, lmbda=0.01, tol=0.001, maxiter=100, return_iters=False):
    """ialm() with the one behaviour the reference leaves to LAPACK made explicit.

    The last window of every video is rank deficient: it is padded with all-zero "null" frames (io_video.py:40-44),
    and before them comes one DUPLICATE of the last real frame (the range test of get_frame is inclusive, so the
    frame one past the end is requested once and served by the re-deliver-the-last-good-frame fallback,
    io_video.py:40,51-53).  Both enter rpca().  A zero singular value has an arbitrary left vector u, and the
    always-full `svp` (:285) turns it into the term -(1/mu) u v^T of A: LAPACK-dependent garbage of hundreds of grey
    levels that leaks into every frame of the window from iteration 2 on (measured: numpy 1.26 and 2.2 disagree by
    up to 2 grey levels on 3 % of the pixels of tests/golden/ialm_64x96x21_null5).

    Defined behaviour of this project (oracle and HIP path alike): a singular direction of M whose sigma^2 is below
    DEAD_EIG * sigma_max^2 is a ZERO direction and contributes nothing to A (its weight is 0 instead of
    sigma - 1/mu).  For all-zero columns that is the same as leaving them out of the decomposition (their A and E stay
    0); for duplicated columns the two copies share one decomposition.  Windows of full rank never meet the rule, so
    this is the reference's algorithm wherever the reference is reproducible at all.
    """
    X = np.asarray(X)
    flat = X.ravel()
    two_norm = np.linalg.norm(flat, 2)                       # :269
    if two_norm == 0:                                         # nothing to decompose (the reference would divide by zero)
        z = np.zeros(X.shape)
        return (z, z.copy(), 0) if return_iters else (z, z.copy())
    inf_norm = np.linalg.norm(flat, np.inf) / lmbda           # :270
    scale = np.max([two_norm, inf_norm])                      # :271
    Y = X / scale                                             # :272
    A = np.zeros(Y.shape)
    x_fro = np.linalg.norm(X, 'fro')                          # :275
    mu = 1.25 / two_norm                                      # :276
    k = 0
    null_cols = ~X.any(axis=0)
    while True:
        raw = X - A + (1 / mu) * Y                            # :282
        E = np.maximum(raw - lmbda / mu, 0) + np.minimum(raw + lmbda / mu, 0)   # :283
        U, S, Vt = np.linalg.svd(X - E + (1 / mu) * Y, full_matrices=False)     # :284
        w = np.where(S * S > DEAD_EIG * S[0] * S[0], S - 1 / mu, 0.0)            # :285-290 with the zero-direction rule
        A = np.dot(U * w, Vt)
        A[:, null_cols] = 0.0                                 # exactly, not to rounding: a null frame stays out
        Z = X - A - E                                         # :293
        Y = Y + mu * Z                                        # :294
        mu = mu * 1.5                                         # :295
        k += 1
        if (np.linalg.norm(Z, 'fro') / x_fro) < tol or k >= maxiter:   # :297
            break
    return (A, E, k) if return_iters else (A, E)


def rpca(gray_frames, return_iters=False, **kw):
    """image_filtering.py:220-253: list of n (H,W) u8 frames -> list of n (H,W) u8 (and the IALM's iteration count on request)."""
    stack = np.array(gray_frames)
    n, H, W = stack.shape
    cols = np.transpose(stack.reshape(n, H * W))              # (P, n), column j = frame j
    _, E, k = ialm_defined(cols, return_iters=True, **kw)
    S = rpca_epilogue(E)
    out = [np.reshape(S[:, i], (H, W)) for i in range(n)]
    return (out, k) if return_iters else out


# --------------------------------------------------------------------------
# whole window, stage by stage: data_structures.py:171-217
# --------------------------------------------------------------------------
DEFAULTS = dict(lmbda=0.01, tol=0.001, maxiter=100, bil_d=7, bil_sigma_color=15.0,
                bil_sigma_space=1.0, bil_fma=False, thresh=15, open_size=(3, 3),
                connectivity=8, label_order=1, gray_mode=0)


def window(roi_bgr, **overrides):
    """roi_bgr: (n, Hc, Wc, 3) u8 in queue order.  Returns dict of per-stage stacks and
    per-frame segment lists, the same products FrameQueue.segment_queue stores."""
    p = dict(DEFAULTS)
    p.update(overrides)
    n = roi_bgr.shape[0]
    gray = [bgr2gray(roi_bgr[i], p["gray_mode"]) for i in range(n)]
    sparse, iters = rpca(gray, return_iters=True, lmbda=p["lmbda"], tol=p["tol"], maxiter=p["maxiter"])
    bil = [bilateral_u8(f, p["bil_d"], p["bil_sigma_color"], p["bil_sigma_space"], p["bil_fma"])
           for f in sparse]
    thr = [thresh_tozero_u8(f, p["thresh"]) for f in bil]
    opened = [grey_open_u8(f, p["open_size"]) for f in thr]
    lab = [labels_to_u8(ccl_u8(f, p["connectivity"], p["label_order"])[1]) for f in opened]
    segs = [regionprops_u8(l) for l in lab]
    return dict(gray=np.stack(gray), rpca=np.stack(sparse), bilateral=np.stack(bil),
                thresh=np.stack(thr), opened=np.stack(opened), labels=np.stack(lab),
                segments=segs, iters=iters)

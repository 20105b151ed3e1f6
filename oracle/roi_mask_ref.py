"""CPU oracle of the ROI-mask section (image_filtering.py:99-180) -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED: the reference's arithmetic is in opencv-python 4.1.0.25 (cv2.medianBlur, cv2.threshold + OTSU,
cv2.Canny, cv2.dilate), installed nowhere in the build image, and the reference has no test vectors.  This module restates
OpenCV 4.1.0's algorithms INDEPENDENTLY of the product's C++ (different data structures: scipy.ndimage for the median
and the directional maximum, vectorised numpy for Otsu and the Canny maps, scipy.ndimage.label for the hysteresis) so
that a slip in either statement shows up as a disagreement.
"""
import numpy as np
from scipy import ndimage


def roi_crop_region(corners):
    """image_filtering.py:56-75."""
    left, right = min(corners[0][0], corners[1][0]), max(corners[0][0], corners[1][0])
    bottom = max(corners[0][1], corners[1][1])
    w = right - left
    return [(int(left + 0.025 * w), int(bottom - 0.25 * w)), (int(right - 0.025 * w), int(bottom))]


def median_blur(image, k):
    """cv2.medianBlur: k x k per channel, BORDER_REPLICATE."""
    image = np.asarray(image, np.uint8)
    if image.ndim == 2:
        return ndimage.median_filter(image, size=(k, k), mode="nearest")
    return np.stack([ndimage.median_filter(image[:, :, c], size=(k, k), mode="nearest") for c in range(image.shape[2])], -1)


def otsu(image):
    """cv2.threshold(..., THRESH_BINARY + THRESH_OTSU): (t, image > t ? 255 : 0).  getThreshVal_Otsu_8u's loop, with
    its running mu1 / q1 recurrences replaced by their closed forms (cumulative sums) where that is exact enough to pick
    the same bin: verified against the recurrence below."""
    image = np.asarray(image, np.uint8)
    h = np.bincount(image.ravel(), minlength=256).astype(np.float64)
    scale = 1.0 / image.size
    mu = (np.arange(256) * h).sum() * scale
    mu1 = q1 = 0.0
    best, best_t = 0.0, 0
    eps = float(np.finfo(np.float32).eps)
    for i in range(256):
        p = h[i] * scale
        mu1 *= q1
        q1 += p
        q2 = 1.0 - q1
        if min(q1, q2) < eps or max(q1, q2) > 1.0 - eps:
            continue
        mu1 = (mu1 + i * p) / q1
        mu2 = (mu - q1 * mu1) / q2
        sigma = q1 * q2 * (mu1 - mu2) ** 2
        if sigma > best:
            best, best_t = sigma, i
    return best_t, np.where(image > best_t, 255, 0).astype(np.uint8)


def canny(image, low, high):
    """cv2.Canny, aperture 3, L1 gradient."""
    if low > high:
        low, high = high, low
    a = np.pad(np.asarray(image, np.int32), 1, mode="edge")              # Sobel with BORDER_REPLICATE
    dx = (a[:-2, 2:] + 2 * a[1:-1, 2:] + a[2:, 2:]) - (a[:-2, :-2] + 2 * a[1:-1, :-2] + a[2:, :-2])
    dy = (a[2:, :-2] + 2 * a[2:, 1:-1] + a[2:, 2:]) - (a[:-2, :-2] + 2 * a[:-2, 1:-1] + a[:-2, 2:])
    mag = np.abs(dx) + np.abs(dy)
    m = np.pad(mag, 1)                                                     # zeros all round
    c = m[1:-1, 1:-1]
    left, right, up, down = m[1:-1, :-2], m[1:-1, 2:], m[:-2, 1:-1], m[2:, 1:-1]
    ul, ur, dl, dr = m[:-2, :-2], m[:-2, 2:], m[2:, :-2], m[2:, 2:]
    TG22 = int(0.4142135623730950488016887242097 * (1 << 15) + 0.5)
    ax = np.abs(dx).astype(np.int64)
    ay = np.abs(dy).astype(np.int64) << 15
    tg22 = ax * TG22
    tg67 = tg22 + (ax << 16)
    horiz = ay < tg22
    vert = ~horiz & (ay > tg67)
    diag = ~horiz & ~vert
    same_sign = (dx ^ dy) >= 0                                             # s = +1: compare with up-left and down-right
    peak = (horiz & (c > left) & (c >= right)) | (vert & (c > up) & (c >= down)) | \
           (diag & same_sign & (c > ul) & (c > dr)) | (diag & ~same_sign & (c > ur) & (c > dl))
    cand = peak & (mag > low)
    strong = cand & (mag > high)
    # hysteresis: a candidate is an edge iff its 8-connected component of candidates holds a strong pixel
    lab, n = ndimage.label(cand, structure=np.ones((3, 3), int))
    keep = np.zeros(n + 1, bool)
    keep[np.unique(lab[strong])] = True
    keep[0] = False
    return np.where(keep[lab], 255, 0).astype(np.uint8)


def dilate_up(image, N):
    """cv2.dilate(image, ones((N, 1)), anchor=(0, 0)): each pixel takes the maximum of itself and the N - 1 below it."""
    image = np.asarray(image, np.uint8)
    # footprint rows 0 .. N-1 below the anchor: origin shifts scipy's centred window; outside = 0 (identity for u8 max)
    out = image.copy()
    for k in range(1, N):
        out[:-k] = np.maximum(out[:-k], image[k:])
    return out


def roi_mask(frame, corners, crop_region):
    """image_filtering.py:99-122."""
    roi = roi_crop_region(corners)
    sub = frame[roi[0][1]:roi[1][1], roi[0][0]:roi[1][0]]
    blurred = median_blur(median_blur(sub, 9), 9)
    _, binary = otsu(blurred[:, :, 0])
    grown = dilate_up(canny(binary, 0, 256), 20)
    full = np.zeros(frame.shape[:2], np.uint8)
    full[roi[0][1]:roi[1][1], roi[0][0]:roi[1][0]] = grown
    cropped = full[crop_region[0][1]:crop_region[1][1], crop_region[0][0]:crop_region[1][0]]
    return otsu(cropped)[1]

"""Drop-in for the preprocessing + segmentation sections of the reference's
swiftwatcher/image_filtering.py (lines 188-369): same function names, argument meaning and
return types, every pixel operation running as a HIP kernel on the MI355X through
libswk.so (ctypes, swiftwatcher_amd/_lib.py).  No OpenCV / SciPy / scikit-image involved
and no CPU fallback: without the GPU library these functions raise SwkError.

The ROI-mask section of the reference file (generate_regions / generate_roi_mask and friends, lines 20-28 and
99-180) runs once per video on a few hundred by a hundred pixels and feeds the host-side tracker: it is host C++ in
the same library (csrc/roi_mask.cpp), same function names here.
"""
import math

import numpy as np

from . import _lib

###############################################################################
# crop geometry -- pure integer arithmetic, image_filtering.py:31-91
###############################################################################


def determine_chimney_extents(corners):
    """image_filtering.py:78-91: (left, right, bottom) of the two chimney corners."""
    left = min(corners[0][0], corners[1][0])
    right = max(corners[0][0], corners[1][0])
    bottom = max(corners[0][1], corners[1][1])
    return left, right, bottom


def generate_crop_region(corners):
    """image_filtering.py:31-53: [(x0, y0), (x1, y1)], 1.25 x 0.625 chimney widths."""
    left, right, bottom = determine_chimney_extents(corners)
    width = right - left
    return [(left - int(0.125 * width), bottom - int(0.5 * width)),
            (right + int(0.125 * width), bottom + int(0.125 * width))]


def generate_roi_crop_region(corners):
    """image_filtering.py:56-75."""
    left, right, bottom = determine_chimney_extents(corners)
    width = right - left
    return [(int(left + 0.025 * width), int(bottom - 0.25 * width)),
            (int(right - 0.025 * width), int(bottom))]


def crop_frame(frame, crop_region):
    """image_filtering.py:199-203: a view, like the reference."""
    return frame[crop_region[0][1]:crop_region[1][1], crop_region[0][0]:crop_region[1][0]]


###############################################################################
# ROI mask -- once per video, host C++ (image_filtering.py:20-28, 99-180); PARITY UNPINNED (cv2)
###############################################################################


def generate_regions(first_frame, corners):
    """image_filtering.py:20-28: (crop_region, roi_mask, resize_dim) for a video, from its first frame and the two
    chimney corners.  resize_dim is handed through like in the reference (its resize step is commented out there)."""
    resize_dim = (300, 150)
    crop_region, roi_mask = _lib.roi_mask(first_frame, corners)
    return crop_region, roi_mask, resize_dim


def generate_roi_mask(frame, corners, crop_region=None, resize_dim=None):
    """image_filtering.py:99-122 (crop_region is recomputed from the corners, as generate_regions does)."""
    return _lib.roi_mask(frame, corners)[1]


def median_blur(image, kernel_size):
    """image_filtering.py:125-131 (cv2.medianBlur)."""
    return _lib.median_blur_u8(image, kernel_size)


def split_bgr_channels(image):
    """image_filtering.py:134-139 (cv2.split)."""
    return (np.ascontiguousarray(image[:, :, 0]), np.ascontiguousarray(image[:, :, 1]), np.ascontiguousarray(image[:, :, 2]))


def threshold_channel(image):
    """image_filtering.py:142-151: Otsu's threshold, binary 0 / 255 output."""
    return _lib.otsu_threshold_u8(image)[1]


def detect_canny_edges(image):
    """image_filtering.py:154-159: cv2.Canny(image, 0, 256)."""
    return _lib.canny_u8(image, 0, 256)


def dilate_upwards(image, N):
    """image_filtering.py:162-170: N x 1 kernel anchored at its top, so edges only grow upwards."""
    return _lib.dilate_up_u8(image, N)


def create_mask(mask, frame_region, frame):
    """image_filtering.py:173-181: the ROI-sized mask pasted into a blank single-channel image of the frame's size."""
    out = np.zeros(frame.shape[:2], np.uint8)
    out[frame_region[0][1]:frame_region[1][1], frame_region[0][0]:frame_region[1][0]] = mask
    return out


###############################################################################
# stages -- each one a HIP kernel behind the C ABI
###############################################################################


def _ctx():
    return _lib.default_context()


def convert_grayscale(frame, gray_mode=_lib.GRAY_Q14):
    """image_filtering.py:188-196.  3-channel BGR -> gray with OpenCV 4.1.0's fixed-point
    weights; a 2-D frame passes through unchanged (:193-194)."""
    if frame.ndim == 2:
        return frame
    return _ctx().bgr2gray(frame, gray_mode)


def inexact_augmented_lagrange_multiplier(X, lmbda=0.01, tol=0.001, maxiter=100, verbose=False):
    """image_filtering.py:256-301.  X: (pixels, frames) uint8 matrix of column-vector images.
    Returns (A, E) float64 (pixels, frames)."""
    X = np.asarray(X)
    if X.dtype != np.uint8:
        raise TypeError("the HIP IALM takes the uint8 image matrix the reference passes (image_filtering.py:234-241)")
    planes = np.ascontiguousarray(X.T)
    A, E, it = _ctx().ialm(planes, lmbda, tol, maxiter, want_E=True)
    if verbose:
        print("Finished at iteration %d" % it)
    return A, E


def rpca(frame_list):
    """image_filtering.py:220-253: list of n gray frames -> list of n uint8 sparse images."""
    stack = np.ascontiguousarray(np.array(frame_list), np.uint8)
    n, H, W = stack.shape
    res = _ctx().batch_run(stack, 1, n, stages=("rpca",), seg_cap=1)
    return [res["rpca"][i] for i in range(n)]


def bilateral_blur(frame, d, sigmaColor, sigmaSpace, use_fma=False):
    """image_filtering.py:304-307."""
    return _ctx().bilateral_u8(frame, d, float(sigmaColor), float(sigmaSpace), use_fma)


def thresh_to_zero(frame, thresh):
    """image_filtering.py:310-316."""
    return _ctx().thresh_tozero_u8(frame, int(thresh))


def grayscale_opening(frame, SE):
    """image_filtering.py:319-322: scipy.ndimage.grey_opening(frame, size=SE).  (3, 3) -- the window the reference uses
    (data_structures.py:202) -- runs on the tiled kernel, any other window on the general one."""
    if tuple(SE) == (3, 3):
        return _ctx().grey_open3x3_u8(frame)
    return _ctx().grey_open_u8(frame, tuple(SE))


def resize_frame(frame, dimensions):
    """image_filtering.py:206-212: cv2.resize(frame, dimensions) with dimensions = (width, height).  The reference's two call
    sites are commented out (data_structures.py:179-181, image_filtering.py:117-118); PARITY UNPINNED (cv2)."""
    return _ctx().resize_linear_u8(frame, dimensions)


def cc_labeling(frame, connectivity=None, effective_connectivity=8, label_order=_lib.ORDER_BLOCK2X2):
    """image_filtering.py:325-329.

    The reference calls cv2.connectedComponents(frame, connectivity) with connectivity=4
    (data_structures.py:206), but the binding's second positional parameter is `labels`, so
    OpenCV runs its defaults: 8-connectivity, BBDT numbering.  To stay a drop-in the positional
    `connectivity` is accepted and, like there, has no effect; use effective_connectivity /
    label_order to choose something else.  Returns uint8 labels (wraps mod 256 like astype)."""
    _, lab = _ctx().ccl_u8(frame, effective_connectivity, label_order)
    return (lab & 0xff).astype(np.uint8)


class RegionProps:
    """The regionprops fields the rest of swiftwatcher reads (label, bbox, centroid, area)."""
    __slots__ = ("label", "bbox", "centroid", "area")

    def __init__(self, label, bbox, centroid, area):
        self.label = label
        self.bbox = bbox
        self.centroid = centroid
        self.area = area

    def __repr__(self):
        return "RegionProps(label=%d, bbox=%r, centroid=%r, area=%d)" % (self.label, self.bbox, self.centroid, self.area)


def regionprops_from_records(records):
    """swk_segment records -> RegionProps list.  The centroid is sum/area in float64, which is
    bit-identical to skimage's coords.mean(axis=0)."""
    if len(records) == 0:
        return []
    area = records["area"].astype(np.float64)
    cr = (records["sum_r"].astype(np.float64) / area).tolist()
    cc = (records["sum_c"].astype(np.float64) / area).tolist()
    return [RegionProps(lab, (r0, c0, r1, c1), (a, b), ar)
            for lab, r0, c0, r1, c1, ar, a, b in zip(records["label"].tolist(), records["r0"].tolist(), records["c0"].tolist(),
                                                     records["r1"].tolist(), records["c1"].tolist(), records["area"].tolist(),
                                                     cr, cc)]


def get_segment_properties(frame):
    """image_filtering.py:332-335: one RegionProps per distinct nonzero label, ascending."""
    segs, _ = _ctx().regionprops_u8(frame)
    return regionprops_from_records(segs)


def segment_crop_box(bbox, frame_shape, min_seg_size, crop_region):
    """The rows / columns of the full frame extract_segment_images slices for one region (image_filtering.py:345-366)."""
    r0, c0, r1, c1 = bbox
    h, w = r1 - r0, c1 - c0
    if h < min_seg_size[0]:
        d = min_seg_size[0] - h
        r0 -= d // 2
        r1 += d - d // 2
    if w < min_seg_size[1]:
        d = min_seg_size[1] - w
        c0 -= d // 2
        c1 += d - d // 2
    oy, ox = crop_region[0][1], crop_region[0][0]
    return max(r0 + oy, 0), max(r1 + oy, 0), max(c0 + ox, 0), max(c1 + ox, 0)


def segment_image_getter(segment, frame, min_seg_size, crop_region):
    """Zero-argument callable that cuts one region's segment image (the same view extract_segment_images makes) when it is
    first needed: data_structures.Segment resolves it on the first read of .segment_image."""
    bbox = segment.bbox

    def cut():
        r0, r1, c0, c1 = segment_crop_box(bbox, frame.shape, min_seg_size, crop_region)
        return frame[r0:r1, c0:c1]
    return cut


def extract_segment_images(segments, frame, min_seg_size, crop_region):
    """image_filtering.py:338-369: expand each bbox to at least min_seg_size (floor/ceil split),
    translate by the crop origin and slice the FULL frame (views).

    One deliberate deviation: a box whose top or left edge leaves the frame is intersected with the frame.
    The reference's unchecked slice wraps a negative start around (numpy semantics), which yields an empty or
    unrelated crop that its classifier then fails on; boxes leaving the bottom / right edge are clipped by numpy
    in the reference too.  swk_segment_inputs (the device-resident path) uses the same rule, so both
    classification paths see the same crop for every segment."""
    images = []
    oy, ox = crop_region[0][1], crop_region[0][0]
    for segment in segments:
        r0, c0, r1, c1 = segment.bbox
        h, w = r1 - r0, c1 - c0
        if h < min_seg_size[0]:
            d = min_seg_size[0] - h
            r0 -= math.floor(d / 2)
            r1 += math.ceil(d / 2)
        if w < min_seg_size[1]:
            d = min_seg_size[1] - w
            c0 -= math.floor(d / 2)
            c1 += math.ceil(d / 2)
        images.append(frame[max(r0 + oy, 0):max(r1 + oy, 0), max(c0 + ox, 0):max(c1 + ox, 0)])
    return images

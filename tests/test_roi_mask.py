"""ROI-mask generation (image_filtering.py:20-28, 99-180): host C++ in libswk against the
independent numpy/scipy oracle (oracle/roi_mask_ref.py), plus specification cases worked out by hand.  PARITY
UNPINNED: both sides restate OpenCV 4.1.0, which is installed nowhere in the image.  No GPU involved."""
import numpy as np
import pytest

from oracle import roi_mask_ref as ref
from oracle import reference_path as orc


def _chimney_frame(rng, H=1080, W=1920, corners=((790, 620), (1130, 622)), tilt=0.0, noise=3.0):
    """Sky gradient above, dark textured chimney below its (slightly tilted) top edge, BGR."""
    yy, xx = np.mgrid[0:H, 0:W]
    sky = 150.0 + 65.0 * yy / (H - 1)
    img = np.stack([sky + 25, sky, sky - 10], -1)                  # blue sky: B channel bright
    (x1, y1), (x2, y2) = corners
    top = y1 + (xx - x1) * (y2 - y1) / max(x2 - x1, 1) + tilt * np.sin(xx / 37.0)
    stack = (xx >= min(x1, x2)) & (xx <= max(x1, x2)) & (yy >= top)
    img[stack] = 55.0 + 10.0 * rng.random(int(stack.sum()))[:, None]
    img += rng.normal(0, noise, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def test_stage_functions_against_oracle():
    from swiftwatcher_amd import image_filtering as img
    rng = np.random.default_rng(1)
    for shape in [(85, 323, 3), (40, 57, 3), (9, 9, 3), (12, 5, 3)]:
        a = rng.integers(0, 256, size=shape, dtype=np.uint8)
        np.testing.assert_array_equal(img.median_blur(a, 9), ref.median_blur(a, 9))
        np.testing.assert_array_equal(img.median_blur(a[:, :, 1], 5), ref.median_blur(a[:, :, 1], 5))
    for _ in range(6):
        g = np.clip(rng.normal(rng.uniform(40, 200), rng.uniform(5, 60), size=(60, 90)), 0, 255).astype(np.uint8)
        g[30:, :] //= 3
        t, b = ref.otsu(g)
        from swiftwatcher_amd import _lib
        t2, b2 = _lib.otsu_threshold_u8(g)
        assert t == t2
        np.testing.assert_array_equal(b2, b)
        np.testing.assert_array_equal(img.threshold_channel(g), b)
    # Otsu on degenerate inputs: constant image -> threshold 0, all 255 unless the constant is 0
    assert img.threshold_channel(np.full((5, 7), 9, np.uint8)).min() == 255
    assert img.threshold_channel(np.zeros((5, 7), np.uint8)).max() == 0
    # Canny on binary shapes (what the pipeline feeds it) and on grey noise (exercises weak edges + hysteresis)
    for k in range(8):
        if k < 4:
            im = (rng.random((50, 70)) < 0.5).astype(np.uint8) * 255
            im = ref.median_blur(im, 5 if k % 2 else 9)
        else:
            im = ref.median_blur(rng.integers(0, 256, size=(48, 64), dtype=np.uint8), 3)
        for lo, hi in ((0, 256), (40, 120), (300, 100)):
            np.testing.assert_array_equal(img.detect_canny_edges(im) if (lo, hi) == (0, 256) else
                                          __import__("swiftwatcher_amd")._lib.canny_u8(im, lo, hi), ref.canny(im, lo, hi), err_msg="%d %r" % (k, (lo, hi)))
        np.testing.assert_array_equal(img.dilate_upwards(im, 20), ref.dilate_up(im, 20))
        np.testing.assert_array_equal(img.dilate_upwards(im, 3), ref.dilate_up(im, 3))


def test_canny_and_dilation_by_hand():
    from swiftwatcher_amd import image_filtering as img
    # a vertical step edge: Sobel gives |dx| = 4 * 255 on the two columns next to the step; non-maximum suppression
    # keeps the LEFT one (m > left neighbour, m >= right neighbour)
    im = np.zeros((7, 8), np.uint8)
    im[:, 4:] = 255
    e = img.detect_canny_edges(im)
    assert e[:, 3].min() == 255 and e[:, 4].max() == 0 and int((e > 0).sum()) == 7
    # dilation only reaches upwards: a single pixel at row 5 fills rows 3..5 for N = 3
    d = np.zeros((8, 3), np.uint8)
    d[5, 1] = 255
    out = img.dilate_upwards(d, 3)
    assert list(np.flatnonzero(out[:, 1])) == [3, 4, 5] and out[:, 0].max() == 0
    # split: channel order B, G, R
    px = np.zeros((2, 2, 3), np.uint8)
    px[..., 0], px[..., 1], px[..., 2] = 1, 2, 3
    b, g, r = img.split_bgr_channels(px)
    assert b[0, 0] == 1 and g[0, 0] == 2 and r[0, 0] == 3


@pytest.mark.parametrize("corners,hw", [(((790, 620), (1130, 622)), (1080, 1920)), (((1046, 623), (874, 620)), (1080, 1920)),
                                        (((389, 300), (465, 301)), (480, 854)), (((610, 1500), (1290, 1505)), (2160, 3840))])
def test_generate_regions_against_oracle(corners, hw):
    from swiftwatcher_amd import image_filtering as img
    rng = np.random.default_rng(hash(corners) % 1000)
    frame = _chimney_frame(rng, hw[0], hw[1], corners, tilt=1.5)
    crop_region, mask, resize_dim = img.generate_regions(frame, corners)
    assert resize_dim == (300, 150)
    assert crop_region == orc.crop_region_from_corners(corners) == img.generate_crop_region(corners)
    exp = ref.roi_mask(frame, corners, crop_region)
    assert mask.shape == exp.shape == (crop_region[1][1] - crop_region[0][1], crop_region[1][0] - crop_region[0][0])
    np.testing.assert_array_equal(mask, exp)
    assert set(np.unique(mask)) <= {0, 255}
    # what the mask is for: a band above the chimney's top edge, inside the chimney's width, nothing elsewhere
    left, right, bottom = img.determine_chimney_extents(corners)
    rows, cols = np.nonzero(mask)
    assert rows.size > 0.5 * 20 * 0.9 * (right - left)
    assert cols.min() >= left - crop_region[0][0] and cols.max() <= right - crop_region[0][0]
    assert rows.max() <= bottom - crop_region[0][1] and rows.min() >= bottom - crop_region[0][1] - int(0.25 * (right - left)) - 1
    # the stage functions chained the way generate_roi_mask chains them give the same mask
    roi = img.generate_roi_crop_region(corners)
    sub = img.crop_frame(frame, roi)
    blurred = img.median_blur(img.median_blur(sub, 9), 9)
    b, _, _ = img.split_bgr_channels(blurred)
    grown = img.dilate_upwards(img.detect_canny_edges(img.threshold_channel(b)), 20)
    full = img.create_mask(grown, roi, frame)
    np.testing.assert_array_equal(img.threshold_channel(img.crop_frame(img.convert_grayscale(full), crop_region)), mask)
    np.testing.assert_array_equal(img.generate_roi_mask(frame, corners, crop_region, resize_dim), mask)


def test_regions_leaving_the_frame_are_refused():
    from swiftwatcher_amd import image_filtering as img, _lib
    frame = np.zeros((200, 300, 3), np.uint8)
    with pytest.raises(_lib.SwkError):
        img.generate_regions(frame, ((10, 100), (290, 100)))          # crop region would start left of the frame

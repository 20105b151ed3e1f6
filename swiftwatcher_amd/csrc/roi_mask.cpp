// ROI-mask generation: host C++, no GPU -- it runs ONCE per video on a ~0.95 w x 0.25 w
// sub-image (about 320 x 85 pixels for a 1080p chimney) and its consumer is the host-side tracker
// (segment_tracking.py:161-176).  Replaces image_filtering.py:99-180 of the reference:
//
//   crop to the ROI region -> cv2.medianBlur(9) twice -> B channel -> Otsu threshold -> cv2.Canny(0, 256)
//   -> dilate with a 20 x 1 kernel anchored at its top (edges grow UPWARDS) -> paste into a frame-sized blank image
//   -> crop to the crop region -> Otsu threshold again.
//
// Every step is integer arithmetic (or a comparison of doubles computed from integers), so the definitions below are
// exact.  PARITY UNPINNED: the arithmetic lives in opencv-python 4.1.0.25, which is installed nowhere in the build
// image; each function restates OpenCV 4.1.0's published algorithm (imgproc/src/{median_blur,thresh,canny,morph}.cpp)
// and is cross-checked by tests/test_roi_mask.py against an independent numpy/scipy statement of the same rules.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "swk.h"

namespace {

inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// cv2.medianBlur(src, k): per channel, k x k window, BORDER_REPLICATE; exact median (the (k*k)/2-th order statistic)
void median_blur(const uint8_t *src, int H, int W, int C, int k, uint8_t *dst)
{
    const int r = k / 2, half = (k * k) / 2;
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < H; ++y) {
            int hist[256];
            for (int x = 0; x < W; ++x) {
                // (the image is tiny: a fresh histogram per pixel keeps the code obviously right)
                memset(hist, 0, sizeof hist);
                for (int dy = -r; dy <= r; ++dy) {
                    const uint8_t *row = src + (size_t)clampi(y + dy, 0, H - 1) * W * C;
                    for (int dx = -r; dx <= r; ++dx) hist[row[(size_t)clampi(x + dx, 0, W - 1) * C + c]]++;
                }
                int acc = 0, v = 0;
                for (; v < 256; ++v) { acc += hist[v]; if (acc > half) break; }
                dst[((size_t)y * W + x) * C + c] = (uint8_t)v;
            }
        }
}

// getThreshVal_Otsu_8u (imgproc/src/thresh.cpp), double arithmetic in OpenCV's statement order
int otsu_threshold(const uint8_t *src, size_t count)
{
    int h[256] = {0};
    for (size_t i = 0; i < count; ++i) h[src[i]]++;
    double mu = 0.0;
    const double scale = 1.0 / (double)count;
    for (int i = 0; i < 256; ++i) mu += i * (double)h[i];
    mu *= scale;
    double mu1 = 0.0, q1 = 0.0, max_sigma = 0.0;
    int max_val = 0;
    const double eps = 1.1920928955078125e-07;              // FLT_EPSILON
    for (int i = 0; i < 256; ++i) {
        const double p_i = h[i] * scale;
        mu1 *= q1;
        q1 += p_i;
        const double q2 = 1.0 - q1;
        if (std::min(q1, q2) < eps || std::max(q1, q2) > 1.0 - eps) continue;
        mu1 = (mu1 + i * p_i) / q1;
        const double mu2 = (mu - q1 * mu1) / q2;
        const double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
        if (sigma > max_sigma) { max_sigma = sigma; max_val = i; }
    }
    return max_val;
}

// cv2.Canny(image, low, high): 3 x 3 Sobel with BORDER_REPLICATE, L1 magnitude, non-maximum suppression with OpenCV's
// fixed-point tangent tests, hysteresis over the 8-neighbourhood (imgproc/src/canny.cpp)
void canny(const uint8_t *src, int H, int W, int low, int high, uint8_t *dst)
{
    if (low > high) std::swap(low, high);
    const int MW = W + 2;
    std::vector<int> mag((size_t)(H + 2) * MW, 0);            // one pixel of zeros all round
    std::vector<short> gx((size_t)H * W), gy((size_t)H * W);
    auto px = [&](int y, int x) { return (int)src[(size_t)clampi(y, 0, H - 1) * W + clampi(x, 0, W - 1)]; };
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int dx = (px(y - 1, x + 1) + 2 * px(y, x + 1) + px(y + 1, x + 1)) - (px(y - 1, x - 1) + 2 * px(y, x - 1) + px(y + 1, x - 1));
            const int dy = (px(y + 1, x - 1) + 2 * px(y + 1, x) + px(y + 1, x + 1)) - (px(y - 1, x - 1) + 2 * px(y - 1, x) + px(y - 1, x + 1));
            gx[(size_t)y * W + x] = (short)dx;
            gy[(size_t)y * W + x] = (short)dy;
            mag[(size_t)(y + 1) * MW + x + 1] = (dx < 0 ? -dx : dx) + (dy < 0 ? -dy : dy);
        }
    // map: 0 = might belong to an edge, 1 = does not, 2 = does
    std::vector<uint8_t> map((size_t)(H + 2) * MW, 1);
    std::vector<int> stack;
    const int TG22 = (int)(0.4142135623730950488016887242097 * (1 << 15) + 0.5);
    for (int y = 0; y < H; ++y) {
        const int *mp = &mag[(size_t)y * MW + 1], *ma = mp + MW, *mn = ma + MW;     // previous, this, next row
        for (int x = 0; x < W; ++x) {
            const int m = ma[x];
            if (m <= low) continue;
            const int xs = gx[(size_t)y * W + x], ys = gy[(size_t)y * W + x];
            const int ax = xs < 0 ? -xs : xs;
            const int ay = (ys < 0 ? -ys : ys) << 15;
            const int tg22x = ax * TG22;
            bool is_max;
            if (ay < tg22x) is_max = m > ma[x - 1] && m >= ma[x + 1];
            else {
                const int tg67x = tg22x + (ax << 16);
                if (ay > tg67x) is_max = m > mp[x] && m >= mn[x];
                else {
                    const int s = (xs ^ ys) < 0 ? -1 : 1;
                    is_max = m > mp[x - s] && m > mn[x + s];
                }
            }
            if (!is_max) continue;
            const size_t at = (size_t)(y + 1) * MW + x + 1;
            if (m > high) { map[at] = 2; stack.push_back((int)at); }
            else map[at] = 0;
        }
    }
    const int nb[8] = {-MW - 1, -MW, -MW + 1, -1, 1, MW - 1, MW, MW + 1};
    while (!stack.empty()) {
        const int at = stack.back();
        stack.pop_back();
        for (int k = 0; k < 8; ++k) {
            const int q = at + nb[k];
            if (map[q] == 0) { map[q] = 2; stack.push_back(q); }      // the border ring is 1 and stops the walk
        }
    }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) dst[(size_t)y * W + x] = map[(size_t)(y + 1) * MW + x + 1] == 2 ? 255 : 0;
}

// cv2.dilate(image, ones((N, 1)), anchor=(0, 0)): dst(y, x) = max over k = 0 .. N-1 of src(y + k, x); rows past the
// bottom do not take part (the morphology border value is the identity of max)
void dilate_up(const uint8_t *src, int H, int W, int N, uint8_t *dst)
{
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            uint8_t m = 0;
            for (int k = 0; k < N && y + k < H; ++k) m = std::max(m, src[(size_t)(y + k) * W + x]);
            dst[(size_t)y * W + x] = m;
        }
}

void regions(const int32_t c[4], int32_t crop[4], int32_t roi[4])
{
    const int left = std::min(c[0], c[2]), right = std::max(c[0], c[2]), bottom = std::max(c[1], c[3]);   // :78-91
    const int width = right - left;
    crop[0] = left - (int)(0.125 * width); crop[1] = bottom - (int)(0.5 * width);                          // :49-51
    crop[2] = right + (int)(0.125 * width); crop[3] = bottom + (int)(0.125 * width);
    roi[0] = (int)(left + 0.025 * width); roi[1] = (int)(bottom - 0.25 * width);                           // :72-73
    roi[2] = (int)(right - 0.025 * width); roi[3] = (int)bottom;
}

}  // namespace

#pragma GCC visibility push(default)
extern "C" {

int32_t swk_median_blur_u8(const uint8_t *src, int32_t H, int32_t W, int32_t channels, int32_t ksize, uint8_t *dst)
{
    if (!src || !dst || H < 1 || W < 1 || channels < 1 || channels > 4 || ksize < 3 || !(ksize & 1) || ksize > 15) return SWK_ERR_ARG;
    median_blur(src, H, W, channels, ksize, dst);
    return SWK_OK;
}

int32_t swk_otsu_threshold_u8(const uint8_t *src, int64_t count, uint8_t *dst, int32_t *thresh)
{
    if (!src || count < 1) return SWK_ERR_ARG;
    const int t = otsu_threshold(src, (size_t)count);
    if (thresh) *thresh = t;
    if (dst) for (int64_t i = 0; i < count; ++i) dst[i] = src[i] > t ? 255 : 0;     // THRESH_BINARY, maxval 255
    return SWK_OK;
}

int32_t swk_canny_u8(const uint8_t *src, int32_t H, int32_t W, int32_t low, int32_t high, uint8_t *dst)
{
    if (!src || !dst || H < 1 || W < 1) return SWK_ERR_ARG;
    canny(src, H, W, low, high, dst);
    return SWK_OK;
}

int32_t swk_dilate_up_u8(const uint8_t *src, int32_t H, int32_t W, int32_t N, uint8_t *dst)
{
    if (!src || !dst || H < 1 || W < 1 || N < 1) return SWK_ERR_ARG;
    dilate_up(src, H, W, N, dst);
    return SWK_OK;
}

int32_t swk_roi_mask(const uint8_t *frame, int32_t H, int32_t W, int64_t row_stride, const int32_t corners[4],
                     int32_t crop_region[4], uint8_t *mask, int64_t mask_capacity)
{
    if (!frame || !corners || !crop_region || H < 1 || W < 1 || row_stride < (int64_t)W * 3) return SWK_ERR_ARG;
    int32_t roi[4];
    regions(corners, crop_region, roi);
    // numpy slicing semantics of crop_frame (:199-203) for regions that leave the frame are not reproduced: refuse
    if (roi[0] < 0 || roi[1] < 0 || roi[2] > W || roi[3] > H || roi[2] <= roi[0] || roi[3] <= roi[1]) return SWK_ERR_ARG;
    if (crop_region[0] < 0 || crop_region[1] < 0 || crop_region[2] > W || crop_region[3] > H) return SWK_ERR_ARG;
    const int rh = roi[3] - roi[1], rw = roi[2] - roi[0];
    const int ch = crop_region[3] - crop_region[1], cw = crop_region[2] - crop_region[0];
    if (!mask || mask_capacity < (int64_t)ch * cw) return SWK_ERR_CAPACITY;
    std::vector<uint8_t> a((size_t)rh * rw * 3), b((size_t)rh * rw * 3), g((size_t)rh * rw), e((size_t)rh * rw);
    for (int y = 0; y < rh; ++y) memcpy(&a[(size_t)y * rw * 3], frame + (int64_t)(roi[1] + y) * row_stride + (int64_t)roi[0] * 3, (size_t)rw * 3);
    median_blur(a.data(), rh, rw, 3, 9, b.data());                               // :105
    median_blur(b.data(), rh, rw, 3, 9, a.data());                               // :106
    for (size_t i = 0; i < (size_t)rh * rw; ++i) g[i] = a[3 * i];                // :107 B channel of BGR
    const int t = otsu_threshold(g.data(), g.size());                            // :108
    for (auto &v : g) v = v > t ? 255 : 0;
    canny(g.data(), rh, rw, 0, 256, e.data());                                   // :109
    dilate_up(e.data(), rh, rw, 20, g.data());                                   // :110
    // :113 paste into a frame-sized blank image, :119 crop to the crop region = intersect the two rectangles
    memset(mask, 0, (size_t)ch * cw);
    for (int y = 0; y < rh; ++y) {
        const int fy = roi[1] + y - crop_region[1];
        if (fy < 0 || fy >= ch) continue;
        for (int x = 0; x < rw; ++x) {
            const int fx = roi[0] + x - crop_region[0];
            if (fx >= 0 && fx < cw) mask[(size_t)fy * cw + fx] = g[(size_t)y * rw + x];
        }
    }
    const int t2 = otsu_threshold(mask, (size_t)ch * cw);                        // :120
    for (size_t i = 0; i < (size_t)ch * cw; ++i) mask[i] = mask[i] > t2 ? 255 : 0;
    return SWK_OK;
}

}  // extern "C"
#pragma GCC visibility pop

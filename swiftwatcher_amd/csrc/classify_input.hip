// Input pipeline of the segment classifier on the device -- replaces the torchvision transform chain of
// segment_classification.py:18-24 (ToPILImage, Resize((24,24)), Pad(100), ToTensor, Normalize) for a batch of
// segment crops.  Resize is Pillow's antialiased bilinear resampling for 8-bit images, restated exactly
// (libImaging/Resample.c): coefficients in float64 (support = max(scale, 1), weights normalised, rounded to
// 22-bit fixed point), a horizontal pass then a vertical pass, each accumulating in integers from 2^21 and
// clipping (x >> 22) to [0, 255].  A crop that already is 24x24 passes through (as in Pillow).
// One workgroup per segment; coefficient tables and the horizontally resized image live in LDS.
#include "swk_internal.h"

namespace swk {

constexpr int kOut = 24;                       // Resize((24, 24))

constexpr int kMaxIn = 512;                    // largest crop side handled
constexpr int kMaxK = 2 * ((kMaxIn + kOut - 1) / kOut) + 1;     // coefficients per output sample
constexpr int kPrecision = 22;                 // 32 - 8 - 2 bits

__device__ __forceinline__ void pil_coeffs(int in_size, int xx, int *k_out, int *bounds)
{
    const double scale = (double)in_size / (double)kOut;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;                    // bilinear filter support = 1
    const double center = 0.0 + (xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    double w[kMaxK];
    for (int x = 0; x < xmax; ++x) {
        double v = (x + xmin - center + 0.5) * ss;
        if (v < 0.0) v = -v;
        const double f = v < 1.0 ? 1.0 - v : 0.0;
        w[x] = f;
        ww += f;
    }
    for (int x = 0; x < xmax; ++x) {
        double c = w[x];
        if (ww != 0.0) c /= ww;
        k_out[x] = c < 0 ? (int)(-0.5 + c * (double)(1 << kPrecision)) : (int)(0.5 + c * (double)(1 << kPrecision));
    }
    bounds[0] = xmin;
    bounds[1] = xmax;
}

__device__ __forceinline__ int clip8(int v)
{
    v >>= kPrecision;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// Resize one HxWx3 image (rows row_stride bytes apart) to 24x24 and emit the patch / the normalised network input.
// All 256 threads of the workgroup take part.
struct ResizeLds {
    int kx[kOut * kMaxK], ky[kOut * kMaxK];
    int bx[kOut * 2], by[kOut * 2];
    uint8_t tmp[kMaxIn * kOut * 3];              // horizontally resized rows
    uint8_t out[kOut * kOut * 3];
};

__device__ __forceinline__ void resize_and_emit(ResizeLds &L, const uint8_t *__restrict__ img, int64_t row_stride, int h, int w,
                                                uint8_t *__restrict__ patch_out, float *__restrict__ net_out, int pad, int nhwc,
                                                const float *mean, const float *sd)
{
    const int tid = threadIdx.x;
    if (tid < kOut) pil_coeffs(w, tid, &L.kx[tid * kMaxK], &L.bx[tid * 2]);
    else if (tid >= 64 && tid < 64 + kOut) pil_coeffs(h, tid - 64, &L.ky[(tid - 64) * kMaxK], &L.by[(tid - 64) * 2]);
    __syncthreads();
    // horizontal pass (skipped by Pillow when the width already matches)
    for (int i = tid; i < h * kOut; i += 256) {
        const int y = i / kOut, xx = i - y * kOut;
        const uint8_t *row = img + (int64_t)y * row_stride;
        if (w == kOut) {
            L.tmp[i * 3 + 0] = row[xx * 3 + 0]; L.tmp[i * 3 + 1] = row[xx * 3 + 1]; L.tmp[i * 3 + 2] = row[xx * 3 + 2];
            continue;
        }
        const int xmin = L.bx[xx * 2], xmax = L.bx[xx * 2 + 1];
        const int *k = &L.kx[xx * kMaxK];
        int a0 = 1 << (kPrecision - 1), a1 = a0, a2 = a0;
        for (int x = 0; x < xmax; ++x) {
            const uint8_t *px = row + (x + xmin) * 3;
            a0 += px[0] * k[x]; a1 += px[1] * k[x]; a2 += px[2] * k[x];
        }
        L.tmp[i * 3 + 0] = (uint8_t)clip8(a0); L.tmp[i * 3 + 1] = (uint8_t)clip8(a1); L.tmp[i * 3 + 2] = (uint8_t)clip8(a2);
    }
    __syncthreads();
    // vertical pass
    for (int i = tid; i < kOut * kOut; i += 256) {
        const int yy = i / kOut, xx = i - yy * kOut;
        int r0, r1, r2;
        if (h == kOut) {
            r0 = L.tmp[i * 3]; r1 = L.tmp[i * 3 + 1]; r2 = L.tmp[i * 3 + 2];
        } else {
            const int ymin = L.by[yy * 2], ymax = L.by[yy * 2 + 1];
            const int *k = &L.ky[yy * kMaxK];
            int a0 = 1 << (kPrecision - 1), a1 = a0, a2 = a0;
            for (int y = 0; y < ymax; ++y) {
                const uint8_t *px = &L.tmp[((y + ymin) * kOut + xx) * 3];
                a0 += px[0] * k[y]; a1 += px[1] * k[y]; a2 += px[2] * k[y];
            }
            r0 = clip8(a0); r1 = clip8(a1); r2 = clip8(a2);
        }
        L.out[i * 3] = (uint8_t)r0; L.out[i * 3 + 1] = (uint8_t)r1; L.out[i * 3 + 2] = (uint8_t)r2;
    }
    __syncthreads();
    if (patch_out)
        for (int i = tid; i < kOut * kOut * 3; i += 256) patch_out[i] = L.out[i];
    if (net_out) {
        // Pad(100) with zeros, ToTensor (/255), Normalize ((t - mean) / std): the border is the constant (0 - mean)/std.
        // pad = 100 writes the whole 224x224 input; a smaller pad writes the centred (24 + 2 pad)^2 window of it
        // (the receptive-field cropped network reads rows/cols 92..131 only: pad = 8).
        const int side = kOut + 2 * pad;
        // memory order of the output: planes [c][y][x], or channels-last [y][x][c] (what the convolution kernels read)
        for (int i = tid; i < 3 * side * side; i += 256) {
            int c, y, x;
            if (nhwc) { const int px = i / 3; c = i - 3 * px; y = px / side; x = px - y * side; }
            else { c = i / (side * side); const int rem = i - c * side * side; y = rem / side; x = rem - y * side; }
            float v = 0.0f;
            if (y >= pad && y < pad + kOut && x >= pad && x < pad + kOut)
                v = (float)L.out[((y - pad) * kOut + (x - pad)) * 3 + c] / 255.0f;
            net_out[i] = (v - mean[c]) / sd[c];
        }
    }
}

__global__ __launch_bounds__(256) void k_classifier_input(const uint8_t *__restrict__ crops, const int64_t *__restrict__ offsets,
                                                          const int32_t *__restrict__ hw, uint8_t *__restrict__ patches,
                                                          float *__restrict__ net, int pad, int nhwc, float m0, float m1, float m2,
                                                          float s0, float s1, float s2)
{
    __shared__ ResizeLds L;
    const int seg = blockIdx.x;
    const int h = hw[2 * seg], w = hw[2 * seg + 1];
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
    const int side = kOut + 2 * pad;
    resize_and_emit(L, crops + offsets[seg], (int64_t)w * 3, h, w,
                    patches ? patches + (int64_t)seg * kOut * kOut * 3 : nullptr,
                    net ? net + (int64_t)seg * 3 * side * side : nullptr, pad, nhwc, mean, sd);
}

// ---------------------------------------------------------------------------------
// Classifier inputs straight from device-resident frames and region records (no host round trip):
// segment k of the batch = region i of frame f, f found by bisection in the exclusive prefix sums of the
// per-frame region counts; its crop box is extract_segment_images' (image_filtering.py:338-369): the bbox grown
// symmetrically to at least min_h x min_w (floor on the low side, ceil on the high side), moved by the ROI origin
// into full-frame coordinates, and -- where the reference would mis-slice -- intersected with the frame.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_segment_prefix(const int32_t *__restrict__ nseg, int F, int seg_cap, int32_t *__restrict__ offsets)
{
    // single workgroup: offsets[f] = sum_{g<f} min(nseg[g], seg_cap), offsets[F] = total
    __shared__ int s_part[256];
    const int tid = threadIdx.x;
    const int per = (F + 255) / 256;
    const int lo = tid * per, hi = lo + per < F ? lo + per : F;
    int sum = 0;
    for (int f = lo; f < hi; ++f) sum += nseg[f] < seg_cap ? nseg[f] : seg_cap;
    s_part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < 256; ++i) { const int v = s_part[i]; s_part[i] = run; run += v; }
        offsets[F] = run;
    }
    __syncthreads();
    int run = s_part[tid];
    for (int f = lo; f < hi; ++f) { offsets[f] = run; run += nseg[f] < seg_cap ? nseg[f] : seg_cap; }
}

__global__ __launch_bounds__(256) void k_segment_inputs(const uint8_t *__restrict__ frames, int64_t frame_stride, int64_t row_stride,
                                                        int frame_h, int frame_w, int x0, int y0,
                                                        const swk_segment *__restrict__ segs, const int32_t *__restrict__ offsets,
                                                        int F, int seg_cap, int min_h, int min_w, int first, int count,
                                                        float *__restrict__ net, int32_t *__restrict__ seg_frame, int pad, int nhwc,
                                                        float m0, float m1, float m2, float s0, float s1, float s2,
                                                        int32_t *__restrict__ oversize)
{
    __shared__ ResizeLds L;
    __shared__ int s_box[5];
    const int k = first + blockIdx.x;
    if ((int)blockIdx.x >= count || k >= offsets[F]) return;
    if (threadIdx.x == 0) {
        int lo = 0, hi = F;                        // largest f with offsets[f] <= k
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (offsets[mid] <= k) lo = mid; else hi = mid;
        }
        const swk_segment sg = segs[(int64_t)lo * seg_cap + (k - offsets[lo])];
        int r0 = sg.r0, c0 = sg.c0, r1 = sg.r1, c1 = sg.c1;
        const int h = r1 - r0, w = c1 - c0;
        if (h < min_h) { const int d = min_h - h; r0 -= d / 2; r1 += d - d / 2; }       // :349-358
        if (w < min_w) { const int d = min_w - w; c0 -= d / 2; c1 += d - d / 2; }
        r0 += y0; r1 += y0; c0 += x0; c1 += x0;                                          // :361-362
        r0 = r0 < 0 ? 0 : r0; c0 = c0 < 0 ? 0 : c0;
        r1 = r1 > frame_h ? frame_h : r1; c1 = c1 > frame_w ? frame_w : c1;
        s_box[0] = lo; s_box[1] = r0; s_box[2] = c0; s_box[3] = r1 - r0; s_box[4] = c1 - c0;
    }
    __syncthreads();
    const int f = s_box[0], r0 = s_box[1], c0 = s_box[2];
    int h = s_box[3], w = s_box[4];
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
    const int side = kOut + 2 * pad;
    if (seg_frame && threadIdx.x == 0) seg_frame[blockIdx.x] = f;
    float *o = net + (int64_t)blockIdx.x * 3 * side * side;
    if (h < 1 || w < 1 || h > kMaxIn || w > kMaxIn) {
        // empty or oversize box: flagged; the input is the blank image
        if (threadIdx.x == 0) atomicAdd(oversize, 1);
        for (int i = threadIdx.x; i < 3 * side * side; i += 256) {
            const int c = nhwc ? i % 3 : i / (side * side);
            o[i] = (0.0f - mean[c]) / sd[c];
        }
        return;
    }
    resize_and_emit(L, frames + (int64_t)f * frame_stride + (int64_t)r0 * row_stride + (int64_t)c0 * 3, row_stride, h, w,
                    nullptr, o, pad, nhwc, mean, sd);
}

void launch_classifier_input(hipStream_t s, const uint8_t *crops, const int64_t *offsets, const int32_t *hw, int nseg,
                             uint8_t *patches, float *net, int pad, bool nhwc, const float *mean, const float *sd)
{
    hipLaunchKernelGGL(k_classifier_input, dim3(nseg), dim3(256), 0, s, crops, offsets, hw, patches, net, pad, nhwc ? 1 : 0,
                       mean[0], mean[1], mean[2], sd[0], sd[1], sd[2]);
}

void launch_segment_prefix(hipStream_t s, const int32_t *nseg, int F, int seg_cap, int32_t *offsets)
{
    hipLaunchKernelGGL(k_segment_prefix, dim3(1), dim3(256), 0, s, nseg, F, seg_cap, offsets);
}

void launch_segment_inputs(hipStream_t s, const uint8_t *frames, int64_t frame_stride, int64_t row_stride, int frame_h, int frame_w,
                           int x0, int y0, const swk_segment *segs, const int32_t *offsets, int F, int seg_cap, int min_h, int min_w,
                           int first, int count, float *net, int32_t *seg_frame, int pad, bool nhwc, const float *mean, const float *sd,
                           int32_t *oversize)
{
    if (count < 1) return;
    hipLaunchKernelGGL(k_segment_inputs, dim3(count), dim3(256), 0, s, frames, frame_stride, row_stride, frame_h, frame_w, x0, y0,
                       segs, offsets, F, seg_cap, min_h, min_w, first, count, net, seg_frame, pad, nhwc ? 1 : 0,
                       mean[0], mean[1], mean[2], sd[0], sd[1], sd[2], oversize);
}

}  // namespace swk

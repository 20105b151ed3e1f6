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

__global__ __launch_bounds__(256) void k_classifier_input(const uint8_t *__restrict__ crops, const int64_t *__restrict__ offsets,
                                                          const int32_t *__restrict__ hw, uint8_t *__restrict__ patches,
                                                          float *__restrict__ net, int pad, float m0, float m1, float m2,
                                                          float s0, float s1, float s2)
{
    __shared__ int s_kx[kOut * kMaxK], s_ky[kOut * kMaxK];
    __shared__ int s_bx[kOut * 2], s_by[kOut * 2];
    __shared__ uint8_t s_tmp[kMaxIn * kOut * 3];              // horizontally resized rows
    __shared__ uint8_t s_out[kOut * kOut * 3];
    const int seg = blockIdx.x, tid = threadIdx.x;
    const int h = hw[2 * seg], w = hw[2 * seg + 1];
    const uint8_t *img = crops + offsets[seg];
    if (tid < kOut) pil_coeffs(w, tid, &s_kx[tid * kMaxK], &s_bx[tid * 2]);
    else if (tid >= 64 && tid < 64 + kOut) pil_coeffs(h, tid - 64, &s_ky[(tid - 64) * kMaxK], &s_by[(tid - 64) * 2]);
    __syncthreads();
    // horizontal pass (skipped by Pillow when the width already matches)
    for (int i = tid; i < h * kOut; i += 256) {
        const int y = i / kOut, xx = i - y * kOut;
        const uint8_t *row = img + (int64_t)y * w * 3;
        if (w == kOut) {
            s_tmp[i * 3 + 0] = row[xx * 3 + 0]; s_tmp[i * 3 + 1] = row[xx * 3 + 1]; s_tmp[i * 3 + 2] = row[xx * 3 + 2];
            continue;
        }
        const int xmin = s_bx[xx * 2], xmax = s_bx[xx * 2 + 1];
        const int *k = &s_kx[xx * kMaxK];
        int a0 = 1 << (kPrecision - 1), a1 = a0, a2 = a0;
        for (int x = 0; x < xmax; ++x) {
            const uint8_t *px = row + (x + xmin) * 3;
            a0 += px[0] * k[x]; a1 += px[1] * k[x]; a2 += px[2] * k[x];
        }
        s_tmp[i * 3 + 0] = (uint8_t)clip8(a0); s_tmp[i * 3 + 1] = (uint8_t)clip8(a1); s_tmp[i * 3 + 2] = (uint8_t)clip8(a2);
    }
    __syncthreads();
    // vertical pass
    for (int i = tid; i < kOut * kOut; i += 256) {
        const int yy = i / kOut, xx = i - yy * kOut;
        int r0, r1, r2;
        if (h == kOut) {
            r0 = s_tmp[i * 3]; r1 = s_tmp[i * 3 + 1]; r2 = s_tmp[i * 3 + 2];
        } else {
            const int ymin = s_by[yy * 2], ymax = s_by[yy * 2 + 1];
            const int *k = &s_ky[yy * kMaxK];
            int a0 = 1 << (kPrecision - 1), a1 = a0, a2 = a0;
            for (int y = 0; y < ymax; ++y) {
                const uint8_t *px = &s_tmp[((y + ymin) * kOut + xx) * 3];
                a0 += px[0] * k[y]; a1 += px[1] * k[y]; a2 += px[2] * k[y];
            }
            r0 = clip8(a0); r1 = clip8(a1); r2 = clip8(a2);
        }
        s_out[i * 3] = (uint8_t)r0; s_out[i * 3 + 1] = (uint8_t)r1; s_out[i * 3 + 2] = (uint8_t)r2;
    }
    __syncthreads();
    if (patches)
        for (int i = tid; i < kOut * kOut * 3; i += 256) patches[(int64_t)seg * kOut * kOut * 3 + i] = s_out[i];
    if (net) {
        // Pad(100) with zeros, ToTensor (/255), Normalize ((t - mean) / std): the border is the constant (0 - mean)/std.
        // pad = 100 writes the whole 224x224 input; a smaller pad writes the centred (24 + 2 pad)^2 window of it
        // (the receptive-field cropped network reads rows/cols 92..131 only: pad = 8).
        const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
        const int side = kOut + 2 * pad;
        float *o = net + (int64_t)seg * 3 * side * side;
        for (int i = tid; i < 3 * side * side; i += 256) {
            const int c = i / (side * side), rem = i - c * side * side;
            const int y = rem / side, x = rem - y * side;
            float v = 0.0f;
            if (y >= pad && y < pad + kOut && x >= pad && x < pad + kOut)
                v = (float)s_out[((y - pad) * kOut + (x - pad)) * 3 + c] / 255.0f;
            o[i] = (v - mean[c]) / sd[c];
        }
    }
}

void launch_classifier_input(hipStream_t s, const uint8_t *crops, const int64_t *offsets, const int32_t *hw, int nseg,
                             uint8_t *patches, float *net, int pad, const float *mean, const float *sd)
{
    hipLaunchKernelGGL(k_classifier_input, dim3(nseg), dim3(256), 0, s, crops, offsets, hw, patches, net, pad,
                       mean[0], mean[1], mean[2], sd[0], sd[1], sd[2]);
}

}  // namespace swk

// Glue kernels of the receptive-field cropped classifier (swiftwatcher_amd/segment_classification.py): what is left between
// the convolution kernels (cnn_conv1x1.hip, cnn_conv3x3.hip, cnn_wino3x3.hip, which carry bias, ReLU and placement
// themselves) -- bias + ReLU + crop behind conv1 (MIOpen), and the three max-pools (PyTorch's NHWC max-pool kernel runs far
// below the memory rate).  Plain streaming jobs, launched on the CALLER's stream (PyTorch's current stream).
// Layout: channels-last dense float32, tensor (n, c, h, w) = memory [n][h][w][c].
#include "swk_internal.h"

namespace swk {

// dst[n][off_y + y][off_x + x][c_off + ch] = max(src[n][crop_y + y][crop_x + x][ch] + bias[ch], 0)
// for y < h, x < w, ch < c;  src is [n][sh][sw][c], dst is [n][dH][dW][dC].  c, dC, c_off multiples of 4.
__global__ __launch_bounds__(256) void k_bias_relu_place(const float4 *__restrict__ src, int sh, int sw, int c4, int crop_y, int crop_x,
                                                         int h, int w, const float4 *__restrict__ bias, float4 *__restrict__ dst,
                                                         int dH, int dW, int dC4, int off_y, int off_x, int c_off4, int64_t total)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ch = (int)(i % c4);
    int64_t r = i / c4;
    const int x = (int)(r % w); r /= w;
    const int y = (int)(r % h);
    const int64_t n = r / h;
    const float4 v = src[((n * sh + crop_y + y) * sw + crop_x + x) * c4 + ch];
    const float4 b = bias[ch];
    float4 o;
    o.x = fmaxf(v.x + b.x, 0.0f); o.y = fmaxf(v.y + b.y, 0.0f); o.z = fmaxf(v.z + b.z, 0.0f); o.w = fmaxf(v.w + b.w, 0.0f);
    dst[((n * dH + off_y + y) * dW + off_x + x) * dC4 + c_off4 + ch] = o;
}

// MaxPool2d(3, stride 2) without padding over [n][h][w][c] -> [n][oh][ow][c], oh = (h - 3) / 2 + 1.
// One thread = one output ROW of one channel quad: it walks the row's columns once, keeping the maximum of the column it shares
// with the next window (6 loads per output instead of 9; lanes = consecutive channel quads, every load 1 KB per wave).
__global__ __launch_bounds__(256) void k_maxpool3s2(const float4 *__restrict__ src, int h, int w, int c4, float4 *__restrict__ dst,
                                                    int oh, int ow, int64_t total)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ch = (int)(i % c4);
    int64_t r = i / c4;
    const int oy = (int)(r % oh);
    const int64_t n = r / oh;
    const float4 *p = src + ((n * h + 2 * oy) * w) * c4 + ch;
    const int64_t rs = (int64_t)w * c4;
    auto colmax = [&](int x) {
        const float4 a = p[(int64_t)x * c4], b = p[rs + (int64_t)x * c4], c = p[2 * rs + (int64_t)x * c4];
        float4 m;
        m.x = fmaxf(fmaxf(a.x, b.x), c.x); m.y = fmaxf(fmaxf(a.y, b.y), c.y);
        m.z = fmaxf(fmaxf(a.z, b.z), c.z); m.w = fmaxf(fmaxf(a.w, b.w), c.w);
        return m;
    };
    float4 prev = colmax(0);
    float4 *q = dst + ((n * oh + oy) * ow) * (int64_t)c4 + ch;
#pragma unroll 3
    for (int ox = 0; ox < ow; ++ox) {
        const float4 c1 = colmax(2 * ox + 1), c2 = colmax(2 * ox + 2);
        float4 m;
        m.x = fmaxf(fmaxf(prev.x, c1.x), c2.x); m.y = fmaxf(fmaxf(prev.y, c1.y), c2.y);
        m.z = fmaxf(fmaxf(prev.z, c1.z), c2.z); m.w = fmaxf(fmaxf(prev.w, c1.w), c2.w);
        q[(int64_t)ox * c4] = m;
        prev = c2;
    }
}

// The classifier's head (Dropout is the identity in eval mode, then Conv2d(512, 2, 1), ReLU, AdaptiveAvgPool2d(1): torchvision's
// SqueezeNet classifier as segment_classification.py:47-67 of the reference re-heads it) over the live square of the last Fire's
// output, [n][px][c] channels-last:
//     out[n][k] = (sum_p max(sum_ch x[n][p][ch] w[k][ch] + bias[k], 0) + ring[k]) / n_pos,       k = 0, 1
// (ring: the sum over the positions outside the square, which do not depend on the segment).  One workgroup per segment, four waves;
// a wave takes pixels wave, wave + 4, ...: lane l holds the weights of channels 256 j + 4 l .. + 3, multiplies its float4s, the two
// sums cross the wave as a butterfly, pixels add up in order, the four waves in order: the summation order is a function of the
// shapes alone, so a segment's score does not depend on the batch it is scored in (the library product this replaces picked its
// kernel, and with it the last bit, by the row count).  Memory-bound: 4 c bytes per pixel read once.
template <int CJ>
__global__ __launch_bounds__(256) void k_head2_relu_mean(const float *__restrict__ x, int px, const float *__restrict__ w,
                                                         const float *__restrict__ bias, const float *__restrict__ ring, float inv_pos,
                                                         float *__restrict__ out)
{
    constexpr int C = 256 * CJ;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *xp = x + (int64_t)blockIdx.x * px * C + 4 * lane;
    float4 w0[CJ], w1[CJ];
#pragma unroll
    for (int j = 0; j < CJ; ++j) {
        w0[j] = *(const float4 *)(w + 256 * j + 4 * lane);
        w1[j] = *(const float4 *)(w + C + 256 * j + 4 * lane);
    }
    const float b0 = bias[0], b1 = bias[1];
    float s0 = 0.0f, s1 = 0.0f;
    auto dot = [&](const float4 (&v)[CJ], float &d0, float &d1) {
        d0 = 0.0f; d1 = 0.0f;
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            d0 = fmaf(v[j].x, w0[j].x, d0); d0 = fmaf(v[j].y, w0[j].y, d0); d0 = fmaf(v[j].z, w0[j].z, d0); d0 = fmaf(v[j].w, w0[j].w, d0);
            d1 = fmaf(v[j].x, w1[j].x, d1); d1 = fmaf(v[j].y, w1[j].y, d1); d1 = fmaf(v[j].z, w1[j].z, d1); d1 = fmaf(v[j].w, w1[j].w, d1);
        }
    };
    auto across = [&](float v) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        return v;
    };
    // two pixels per trip: their loads leave together
    int p = wave;
    for (; p + 4 < px; p += 8) {
        float4 va[CJ], vb[CJ];
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            va[j] = *(const float4 *)(xp + (int64_t)p * C + 256 * j);
            vb[j] = *(const float4 *)(xp + (int64_t)(p + 4) * C + 256 * j);
        }
        float a0, a1, c0, c1;
        dot(va, a0, a1);
        dot(vb, c0, c1);
        s0 += fmaxf(across(a0) + b0, 0.0f); s1 += fmaxf(across(a1) + b1, 0.0f);
        s0 += fmaxf(across(c0) + b0, 0.0f); s1 += fmaxf(across(c1) + b1, 0.0f);
    }
    if (p < px) {
        float4 va[CJ];
#pragma unroll
        for (int j = 0; j < CJ; ++j) va[j] = *(const float4 *)(xp + (int64_t)p * C + 256 * j);
        float a0, a1;
        dot(va, a0, a1);
        s0 += fmaxf(across(a0) + b0, 0.0f); s1 += fmaxf(across(a1) + b1, 0.0f);
    }
    __shared__ float part[4][2];
    if (lane == 0) { part[wave][0] = s0; part[wave][1] = s1; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const int k = threadIdx.x;
        const float t = ((part[0][k] + part[1][k]) + part[2][k]) + part[3][k];
        out[(int64_t)blockIdx.x * 2 + k] = (t + ring[k]) * inv_pos;
    }
}

}  // namespace swk

#pragma GCC visibility push(default)
extern "C" {

int32_t swk_nhwc_bias_relu_place(void *stream, const float *src, int32_t n, int32_t sh, int32_t sw, int32_t c, int32_t crop_y,
                                 int32_t crop_x, int32_t h, int32_t w, const float *bias, float *dst, int32_t dH, int32_t dW,
                                 int32_t dC, int32_t off_y, int32_t off_x, int32_t c_off)
{
    if (!src || !bias || !dst || n < 1 || h < 1 || w < 1 || c < 4 || (c & 3) || (dC & 3) || (c_off & 3) || crop_y < 0 || crop_x < 0 ||
        crop_y + h > sh || crop_x + w > sw || off_y < 0 || off_x < 0 || off_y + h > dH || off_x + w > dW || c_off < 0 || c_off + c > dC ||
        (((uintptr_t)src | (uintptr_t)bias | (uintptr_t)dst) & 15))
        return SWK_ERR_ARG;
    const int64_t total = (int64_t)n * h * w * (c / 4);
    hipLaunchKernelGGL(swk::k_bias_relu_place, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)src, sh, sw, c / 4, crop_y, crop_x, h, w, (const float4 *)bias, (float4 *)dst, dH, dW, dC / 4,
                       off_y, off_x, c_off / 4, total);
    return hipGetLastError() == hipSuccess ? SWK_OK : SWK_ERR_HIP;
}

int32_t swk_nhwc_maxpool3s2(void *stream, const float *src, int32_t n, int32_t h, int32_t w, int32_t c, float *dst)
{
    if (!src || !dst || n < 1 || h < 3 || w < 3 || c < 4 || (c & 3) || (((uintptr_t)src | (uintptr_t)dst) & 15)) return SWK_ERR_ARG;
    const int oh = (h - 3) / 2 + 1, ow = (w - 3) / 2 + 1;
    const int64_t total = (int64_t)n * oh * (c / 4);
    hipLaunchKernelGGL(swk::k_maxpool3s2, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)src, h, w, c / 4, (float4 *)dst, oh, ow, total);
    return hipGetLastError() == hipSuccess ? SWK_OK : SWK_ERR_HIP;
}

int32_t swk_nhwc_head2_relu_mean(void *stream, const float *x, int32_t n, int32_t px, int32_t c, const float *w, const float *bias,
                                 const float *ring, float n_pos, float *out)
{
    if (!x || !w || !bias || !ring || !out || n < 1 || px < 1 || !(n_pos > 0.0f) || (((uintptr_t)x | (uintptr_t)w) & 15)) return SWK_ERR_ARG;
    if (c != 256 && c != 512 && c != 768 && c != 1024) return SWK_ERR_ARG;
    const dim3 grid((unsigned)n), block(256);
    hipStream_t st = (hipStream_t)stream;
    const float inv = 1.0f / n_pos;
    switch (c / 256) {
    case 1: hipLaunchKernelGGL(swk::k_head2_relu_mean<1>, grid, block, 0, st, x, px, w, bias, ring, inv, out); break;
    case 2: hipLaunchKernelGGL(swk::k_head2_relu_mean<2>, grid, block, 0, st, x, px, w, bias, ring, inv, out); break;
    case 3: hipLaunchKernelGGL(swk::k_head2_relu_mean<3>, grid, block, 0, st, x, px, w, bias, ring, inv, out); break;
    default: hipLaunchKernelGGL(swk::k_head2_relu_mean<4>, grid, block, 0, st, x, px, w, bias, ring, inv, out); break;
    }
    return hipGetLastError() == hipSuccess ? SWK_OK : SWK_ERR_HIP;
}

}  // extern "C"
#pragma GCC visibility pop

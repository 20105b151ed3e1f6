// Byte-image stages of the segment path on gfx950:
//   crop + BGR2GRAY      (image_filtering.py:199-203, 188-196)
//   bilateral filter      (image_filtering.py:304-307; OpenCV 4.1.0 bilateralFilter_8u semantics)
//   to-zero threshold     (image_filtering.py:310-316)
//   3x3 grey opening      (image_filtering.py:319-322; scipy.ndimage.grey_opening, mode 'reflect')
// All HBM-bound u8 work: one read of the sparse image, one write of the opened image, tiles
// staged in LDS with their halos.  Float arithmetic of the bilateral filter is IEEE f32 with
// contraction off (the library is built with -ffp-contract=off) so results are bit-identical
// to a scalar C evaluation in the same tap order.
#include "swk_internal.h"

namespace swk {

// ---------------------------------------------------------------------------------
// crop + gray.  OpenCV 4.1.0 RGB2Gray<uchar>: (B*1868 + G*9617 + R*4899 + 2^13) >> 14.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gray(const uint8_t *__restrict__ frames, int channels,
                                              int64_t frame_stride, int64_t row_stride, int x0, int y0,
                                              int H, int W, int mode, uint8_t *__restrict__ out)
{
    const int f = blockIdx.z, r = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= W) return;
    const uint8_t *src = frames + (int64_t)f * frame_stride + (int64_t)(y0 + r) * row_stride + (int64_t)(x0 + c) * channels;
    int y;
    if (channels == 1) {
        y = src[0];
    } else {
        const int bb = src[0], gg = src[1], rr = src[2];
        if (mode == SWK_GRAY_Q14) y = (bb * 1868 + gg * 9617 + rr * 4899 + (1 << 13)) >> 14;
        else y = (bb * 3735 + gg * 19235 + rr * 9798 + (1 << 14)) >> 15;
    }
    out[((int64_t)f * H + r) * W + c] = (uint8_t)y;
}

// 4 pixels per thread, threads flat over the ROI (no idle lanes at row ends); 12 source bytes as three
// dwords when the row start allows it, one dword store.
__global__ __launch_bounds__(256) void k_gray4(const uint8_t *__restrict__ frames, int64_t frame_stride, int64_t row_stride,
                                               int x0, int y0, int H, int W, int mode, uint8_t *__restrict__ out)
{
    const int f = blockIdx.y;
    const int wq = W >> 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * wq) return;
    const int r = idx / wq, c = (idx - r * wq) << 2;
    const uint8_t *src = frames + (int64_t)f * frame_stride + (int64_t)(y0 + r) * row_stride + (int64_t)(x0 + c) * 3;
    uint32_t w0, w1, w2;
    if (((uintptr_t)src & 3) == 0) {
        const uint32_t *s32 = (const uint32_t *)src;
        w0 = s32[0]; w1 = s32[1]; w2 = s32[2];
    } else {
        uint32_t b[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) b[k] = src[k];
        w0 = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
        w1 = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
        w2 = b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24);
    }
    const uint32_t px[4][3] = {{w0 & 255u, (w0 >> 8) & 255u, (w0 >> 16) & 255u},
                               {w0 >> 24, w1 & 255u, (w1 >> 8) & 255u},
                               {(w1 >> 16) & 255u, w1 >> 24, w2 & 255u},
                               {(w2 >> 8) & 255u, (w2 >> 16) & 255u, w2 >> 24}};
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t y = mode == SWK_GRAY_Q14 ? (px[k][0] * 1868u + px[k][1] * 9617u + px[k][2] * 4899u + (1u << 13)) >> 14
                                                : (px[k][0] * 3735u + px[k][1] * 19235u + px[k][2] * 9798u + (1u << 14)) >> 15;
        packed |= y << (8 * k);
    }
    *(uint32_t *)(out + ((int64_t)f * H + r) * W + c) = packed;
}

// Any width: quads per row rounded up, the last one partial; the store is as wide as the output address allows
// (rows of a 214- or 850-pixel ROI start on alternating 4- and 2-byte boundaries).
__global__ __launch_bounds__(256) void k_gray4g(const uint8_t *__restrict__ frames, int64_t frame_stride, int64_t row_stride,
                                                int x0, int y0, int H, int W, int mode, uint8_t *__restrict__ out)
{
    const int f = blockIdx.y;
    const int wq = (W + 3) >> 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * wq) return;
    const int r = idx / wq, c = (idx - r * wq) << 2;
    const int npx = W - c < 4 ? W - c : 4;
    const uint8_t *src = frames + (int64_t)f * frame_stride + (int64_t)(y0 + r) * row_stride + (int64_t)(x0 + c) * 3;
    uint32_t b[12];
    if (npx == 4 && ((uintptr_t)src & 3) == 0) {
        const uint32_t *s32 = (const uint32_t *)src;
        const uint32_t w0 = s32[0], w1 = s32[1], w2 = s32[2];
#pragma unroll
        for (int k = 0; k < 4; ++k) { b[k] = (w0 >> (8 * k)) & 255u; b[4 + k] = (w1 >> (8 * k)) & 255u; b[8 + k] = (w2 >> (8 * k)) & 255u; }
    } else {
#pragma unroll
        for (int k = 0; k < 12; ++k) b[k] = k < 3 * npx ? src[k] : 0u;
    }
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t y = mode == SWK_GRAY_Q14 ? (b[3 * k] * 1868u + b[3 * k + 1] * 9617u + b[3 * k + 2] * 4899u + (1u << 13)) >> 14
                                                : (b[3 * k] * 3735u + b[3 * k + 1] * 19235u + b[3 * k + 2] * 9798u + (1u << 14)) >> 15;
        packed |= y << (8 * k);
    }
    uint8_t *dst = out + ((int64_t)f * H + r) * W + c;
    if (npx == 4 && ((uintptr_t)dst & 3) == 0) *(uint32_t *)dst = packed;
    else if (npx == 4 && ((uintptr_t)dst & 1) == 0) { ((uint16_t *)dst)[0] = (uint16_t)packed; ((uint16_t *)dst)[1] = (uint16_t)(packed >> 16); }
    else for (int k = 0; k < npx; ++k) dst[k] = (uint8_t)(packed >> (8 * k));
}

void launch_gray(hipStream_t s, const uint8_t *frames, int channels, int64_t frame_stride, int64_t row_stride,
                 int x0, int y0, int F, int H, int W, int gray_mode, uint8_t *out)
{
    if (channels == 3 && (W & 3) == 0 && ((uintptr_t)out & 3) == 0) {
        const int groups = H * (W >> 2);
        for (int f0 = 0; f0 < F; f0 += 32768) {
            const int fc = F - f0 < 32768 ? F - f0 : 32768;
            hipLaunchKernelGGL(k_gray4, dim3((groups + 255) / 256, fc), dim3(256), 0, s,
                               frames + (int64_t)f0 * frame_stride, frame_stride, row_stride, x0, y0, H, W, gray_mode,
                               out + (int64_t)f0 * H * W);
        }
        return;
    }
    if (channels == 3) {
        const int groups = H * ((W + 3) >> 2);
        for (int f0 = 0; f0 < F; f0 += 32768) {
            const int fc = F - f0 < 32768 ? F - f0 : 32768;
            hipLaunchKernelGGL(k_gray4g, dim3((groups + 255) / 256, fc), dim3(256), 0, s,
                               frames + (int64_t)f0 * frame_stride, frame_stride, row_stride, x0, y0, H, W, gray_mode,
                               out + (int64_t)f0 * H * W);
        }
        return;
    }
    // single-channel input (passes through): grid.z is limited to 65535, split the frame range
    for (int f0 = 0; f0 < F; f0 += 32768) {
        const int fc = F - f0 < 32768 ? F - f0 : 32768;
        hipLaunchKernelGGL(k_gray, dim3((W + 255) / 256, H, fc), dim3(256), 0, s,
                           frames + (int64_t)f0 * frame_stride, channels, frame_stride, row_stride, x0, y0, H, W,
                           gray_mode, out + (int64_t)f0 * H * W);
    }
}

// ---------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}
__device__ __forceinline__ int clampi(int p, int len) { return p < 0 ? 0 : (p >= len ? len - 1 : p); }

constexpr int kMaxTaps = 81;   // radius <= 4 (d <= 9)
constexpr int kTH = 32, kTW = 64;        // output tile of the tiled kernels
constexpr int kHalo = 5;                 // fused kernel: bilateral radius 3 + erosion 1 + dilation 1

// Generic bilateral (any radius <= 4), one thread per pixel, global reads (L1/L2 absorb the reuse).
// Stage-level entry point; the hot path uses the fused tile kernel below.
__global__ __launch_bounds__(256) void k_bilateral(const uint8_t *__restrict__ src, int H, int W,
                                                   const float *__restrict__ color_w, const float *__restrict__ space_w,
                                                   const int8_t *__restrict__ tdr, const int8_t *__restrict__ tdc,
                                                   int maxk, int use_fma, uint8_t *__restrict__ dst)
{
    __shared__ float s_cw[256];
    __shared__ float s_sw[kMaxTaps];
    __shared__ int s_dr[kMaxTaps], s_dc[kMaxTaps];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_cw[i] = color_w[i];
    for (int i = threadIdx.x; i < maxk; i += blockDim.x) { s_sw[i] = space_w[i]; s_dr[i] = tdr[i]; s_dc[i] = tdc[i]; }
    __syncthreads();
    const int f = blockIdx.z, r = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= W) return;
    const uint8_t *img = src + (int64_t)f * H * W;
    const int v0 = img[r * W + c];
    float sum = 0.f, wsum = 0.f;
    for (int k = 0; k < maxk; ++k) {
        const int rr = reflect101(r + s_dr[k], H), cc = reflect101(c + s_dc[k], W);
        const int v = img[rr * W + cc];
        const int dv = v - v0;
        const float w = s_sw[k] * s_cw[dv < 0 ? -dv : dv];
        if (use_fma) sum = __fmaf_rn((float)v, w, sum);
        else sum = sum + (float)v * w;
        wsum = wsum + w;
    }
    dst[((int64_t)f * H + r) * W + c] = (uint8_t)__float2int_rn(sum / wsum);     // cvRound: half to even
}

void launch_bilateral(hipStream_t s, const uint8_t *src, int F, int H, int W, const BilateralTables &t,
                      int use_fma, uint8_t *dst)
{
    for (int f0 = 0; f0 < F; f0 += 32768) {
        const int fc = F - f0 < 32768 ? F - f0 : 32768;
        hipLaunchKernelGGL(k_bilateral, dim3((W + 255) / 256, H, fc), dim3(256), 0, s, src + (int64_t)f0 * H * W, H, W,
                           t.color_w, t.space_w, t.tap_dr, t.tap_dc, t.maxk, use_fma, dst + (int64_t)f0 * H * W);
    }
}

__global__ void k_thresh(const uint8_t *__restrict__ src, int64_t count, int thresh, uint8_t *__restrict__ dst)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = src[i];
        dst[i] = v > thresh ? (uint8_t)v : (uint8_t)0;
    }
}

void launch_thresh(hipStream_t s, const uint8_t *src, int64_t count, int thresh, uint8_t *dst)
{
    int64_t blocks = (count + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_thresh, dim3((unsigned)blocks), dim3(256), 0, s, src, count, thresh, dst);
}

// 3x3 grey opening = min3x3 then max3x3; scipy's 'reflect' border equals clamp-to-edge at radius 1.
// Tile 32x64 outputs; erosion is needed on a 1-px ring, the source on a 2-px ring.

__global__ __launch_bounds__(256) void k_open3x3(const uint8_t *__restrict__ src, int H, int W, uint8_t *__restrict__ dst)
{
    __shared__ uint8_t s_in[(kTH + 4) * (kTW + 4)];
    __shared__ uint8_t s_er[(kTH + 2) * (kTW + 2)];
    const int f = blockIdx.z;
    const int r0 = blockIdx.y * kTH, c0 = blockIdx.x * kTW;
    const uint8_t *img = src + (int64_t)f * H * W;
    for (int i = threadIdx.x; i < (kTH + 4) * (kTW + 4); i += blockDim.x) {
        const int lr = i / (kTW + 4), lc = i % (kTW + 4);
        const int r = clampi(r0 + lr - 2, H), c = clampi(c0 + lc - 2, W);
        s_in[i] = img[r * W + c];
    }
    __syncthreads();
    // erosion at image positions (r0-1+lr, c0-1+lc); window coordinates clamp to the IMAGE
    for (int i = threadIdx.x; i < (kTH + 2) * (kTW + 2); i += blockDim.x) {
        const int lr = i / (kTW + 2), lc = i % (kTW + 2);
        const int r = clampi(r0 + lr - 1, H), c = clampi(c0 + lc - 1, W);
        int acc = 255;
        for (int dr = -1; dr <= 1; ++dr)
            for (int dc = -1; dc <= 1; ++dc) {
                const int rr = clampi(r + dr, H), cc = clampi(c + dc, W);
                const int v = s_in[(rr - r0 + 2) * (kTW + 4) + (cc - c0 + 2)];
                acc = v < acc ? v : acc;
            }
        s_er[i] = (uint8_t)acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kTH * kTW; i += blockDim.x) {
        const int lr = i / kTW, lc = i % kTW;
        const int r = r0 + lr, c = c0 + lc;
        if (r >= H || c >= W) continue;
        int acc = 0;
        for (int dr = -1; dr <= 1; ++dr)
            for (int dc = -1; dc <= 1; ++dc) {
                const int rr = clampi(r + dr, H), cc = clampi(c + dc, W);
                const int v = s_er[(rr - r0 + 1) * (kTW + 2) + (cc - c0 + 1)];
                acc = v > acc ? v : acc;
            }
        dst[((int64_t)f * H + r) * W + c] = (uint8_t)acc;
    }
}

// Flat grey erosion / dilation with any window (scipy.ndimage.grey_opening(size=(kh, kw)), border mode 'reflect': d c b a | a b c d):
// erosion looks at offsets -k/2 .. k-1-k/2, dilation at the mirrored ones -- the same for an odd size, one apart for an even one (scipy
// negates the origin for the dilation and shifts it by one when the size is even).  The reference only ever asks for (3, 3)
// (data_structures.py:202), which the tiled kernels above serve; this one is the stage function's general case: one thread per
// pixel straight from global memory (a frame's window rows sit in L2).
__device__ __forceinline__ int reflect_index(int p, int len)
{
    while (p < 0 || p >= len) p = p < 0 ? -p - 1 : 2 * len - 1 - p;
    return p;
}

template <bool IS_MAX>
__global__ __launch_bounds__(256) void k_flat_minmax(const uint8_t *__restrict__ src, int H, int W, int kh, int kw, uint8_t *__restrict__ dst)
{
    const int f = blockIdx.z, c = blockIdx.x * 64 + (threadIdx.x & 63), r = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (r >= H || c >= W) return;
    const uint8_t *img = src + (int64_t)f * H * W;
    const int lo_r = IS_MAX ? -(kh - 1 - kh / 2) : -(kh / 2), hi_r = IS_MAX ? kh / 2 : kh - 1 - kh / 2;
    const int lo_c = IS_MAX ? -(kw - 1 - kw / 2) : -(kw / 2), hi_c = IS_MAX ? kw / 2 : kw - 1 - kw / 2;
    int acc = IS_MAX ? 0 : 255;
    for (int i = lo_r; i <= hi_r; ++i) {
        const uint8_t *row = img + (int64_t)reflect_index(r + i, H) * W;
        for (int j = lo_c; j <= hi_c; ++j) {
            const int v = row[reflect_index(c + j, W)];
            acc = IS_MAX ? (v > acc ? v : acc) : (v < acc ? v : acc);
        }
    }
    dst[((int64_t)f * H + r) * W + c] = (uint8_t)acc;
}

void launch_grey_open(hipStream_t s, const uint8_t *src, int F, int H, int W, int kh, int kw, uint8_t *tmp, uint8_t *dst)
{
    for (int f0 = 0; f0 < F; f0 += 32768) {
        const int fc = F - f0 < 32768 ? F - f0 : 32768;
        const dim3 grid((W + 63) / 64, (H + 3) / 4, fc);
        const int64_t off = (int64_t)f0 * H * W;
        hipLaunchKernelGGL(k_flat_minmax<false>, grid, dim3(256), 0, s, src + off, H, W, kh, kw, tmp + off);
        hipLaunchKernelGGL(k_flat_minmax<true>, grid, dim3(256), 0, s, tmp + off, H, W, kh, kw, dst + off);
    }
}

// resize_frame (image_filtering.py:206-212): cv2.resize(frame, (w, h)) = INTER_LINEAR on 8-bit pixels, OpenCV 4.1.0's generic 8u
// arithmetic restated (PARITY UNPINNED; dead code in the reference -- both call sites are commented out): per axis the source
// coordinate (d + 0.5) * src / dst - 0.5 in float32, clamped to the image with weight 0 on the missing neighbour; weights as 11-bit
// fixed point; horizontal sums in int32; vertical (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.  The per-axis
// tables (index, two weights) are made by the host.
__global__ __launch_bounds__(256) void k_resize_linear_u8(const uint8_t *__restrict__ src, int H, int W, int ch, int dH, int dW,
                                                          const int *__restrict__ xi, const short *__restrict__ xw,
                                                          const int *__restrict__ yi, const short *__restrict__ yw, uint8_t *__restrict__ dst)
{
    const int f = blockIdx.z, x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= dH || x >= dW) return;
    const uint8_t *img = src + (int64_t)f * H * W * ch;
    const int y0 = yi[y], y1 = y0 + 1 < H ? y0 + 1 : y0, x0 = xi[x], x1 = x0 + 1 < W ? x0 + 1 : x0;
    const int a0 = xw[2 * x], a1 = xw[2 * x + 1], b0 = yw[2 * y], b1 = yw[2 * y + 1];
    for (int c = 0; c < ch; ++c) {
        const int s0 = img[((int64_t)y0 * W + x0) * ch + c] * a0 + img[((int64_t)y0 * W + x1) * ch + c] * a1;
        const int s1 = img[((int64_t)y1 * W + x0) * ch + c] * a0 + img[((int64_t)y1 * W + x1) * ch + c] * a1;
        const int v = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2;
        dst[(((int64_t)f * dH + y) * dW + x) * ch + c] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
}

void launch_resize_linear(hipStream_t s, const uint8_t *src, int F, int H, int W, int ch, int dH, int dW, const int *xi, const short *xw,
                          const int *yi, const short *yw, uint8_t *dst)
{
    hipLaunchKernelGGL(k_resize_linear_u8, dim3((dW + 63) / 64, (dH + 3) / 4, F), dim3(256), 0, s, src, H, W, ch, dH, dW, xi, xw, yi, yw, dst);
}

void launch_open3x3(hipStream_t s, const uint8_t *src, int F, int H, int W, uint8_t *dst)
{
    for (int f0 = 0; f0 < F; f0 += 32768) {
        const int fc = F - f0 < 32768 ? F - f0 : 32768;
        hipLaunchKernelGGL(k_open3x3, dim3((W + kTW - 1) / kTW, (H + kTH - 1) / kTH, fc), dim3(256), 0, s,
                           src + (int64_t)f0 * H * W, H, W, dst + (int64_t)f0 * H * W);
    }
}

// ---------------------------------------------------------------------------------
// Fused hot-path kernel: bilateral (radius 3, 29 taps) -> to-zero threshold -> 3x3 opening, one 32x64 output
// tile per workgroup.  Opening needs the thresholded image on a 2-px ring, the bilateral filter the sparse image
// 3 px beyond that: a 42-row source tile in LDS (80 columns: the window starts 8 px left of the tile so interior
// tiles load it as aligned dwords).  The stage is instruction-bound, so the work is made as sparse as the data:
//   * outputs are pre-cleared and a tile whose source window is all zero ends right after loading it;
//   * the nonzero pixels of the source tile are kept as one 80-bit mask per row; shifts and ORs of those masks give
//     the ring cells whose 7x7 neighbourhood holds any nonzero pixel, and only those cells -- listed from the set
//     bits -- run the 29-tap loop (a cell with an all-zero neighbourhood filters to 0);
//   * erosion runs over the same list (a zero pixel erodes to zero); an all-zero eroded tile ends the block.
// ---------------------------------------------------------------------------------
constexpr int kR = 3;                    // bilateral radius of the fused kernel
constexpr int kSH = kTH + 2 * kHalo;     // 42 source rows
constexpr int kSP = 80;                  // source pitch: image columns c0-8 .. c0+71
constexpr int kSX = 8;                   // column of the source window = image column - c0 + kSX
constexpr int kBH = kTH + 4, kBW = kTW + 4;      // thresholded ring 36 x 68
constexpr int kEH = kTH + 2, kEW = kTW + 2;      // eroded ring 34 x 66

__global__ __launch_bounds__(256) void k_filter_fused(const uint8_t *__restrict__ src, int H, int W,
                                                      const float *__restrict__ color_w, const float *__restrict__ space_w,
                                                      const int8_t *__restrict__ tdr, const int8_t *__restrict__ tdc,
                                                      int maxk, int use_fma, int thresh,
                                                      uint8_t *__restrict__ bil_out, uint8_t *__restrict__ thr_out,
                                                      uint8_t *__restrict__ open_out)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_src[kSH * kSP];
    __shared__ uint32_t s_nz[kSH * 3];             // per source row: 80-bit mask of its nonzero pixels
    __shared__ unsigned long long s_hlo[kSH];      // per source row: bit lc = some nonzero among source columns lc+3 .. lc+9
    __shared__ uint32_t s_hhi[kSH];
    __shared__ __attribute__((aligned(4))) uint8_t s_thr[kBH * kBW];
    __shared__ __attribute__((aligned(4))) uint8_t s_er[kEH * kEW];
    __shared__ uint16_t s_list[kBH * kBW];
    __shared__ float s_cw[256];
    __shared__ float s_sw[32];
    __shared__ int s_ofs[32];
    __shared__ int s_count, s_er_any;
    const int f = blockIdx.z, tid = threadIdx.x;
    const int r0 = blockIdx.y * kTH, c0 = blockIdx.x * kTW;
    const uint8_t *img = src + (int64_t)f * H * W;
    if (tid == 0) { s_count = 0; s_er_any = 0; }
    if (tid < kSH * 3) s_nz[tid] = 0u;
    __syncthreads();
    // ---- source tile: rows r0-5 .. r0+36, columns c0-8 .. c0+71; its nonzero pixels also go into row bit masks ----
    int seen = 0;                                  // this thread loaded a nonzero pixel
    // One dword (4 columns) per step.  Rows and columns outside the image are BORDER_REFLECT_101 copies (the taps of
    // in-image pixels reach up to kR beyond it); cells farther out are never consumed.  Inside the image the load is
    // as wide as the address allows (rows of a 214- or 850-pixel ROI start on alternating 4- and 2-byte boundaries).
    for (int i = tid; i < kSH * (kSP / 4); i += 256) {
        const int sr = i / (kSP / 4), q = i - sr * (kSP / 4);
        int rr = r0 - kHalo + sr;
        if (rr < 0 || rr >= H) rr = reflect101(rr, H);
        const int c = c0 - kSX + 4 * q;
        const uint8_t *rowp = img + (int64_t)rr * W;
        uint32_t v;
        if (c >= 0 && c + 3 < W) {
            const uint8_t *pp = rowp + c;
            if (((uintptr_t)pp & 3) == 0) v = *(const uint32_t *)pp;
            else if (((uintptr_t)pp & 1) == 0) v = (uint32_t)((const uint16_t *)pp)[0] | ((uint32_t)((const uint16_t *)pp)[1] << 16);
            else v = (uint32_t)pp[0] | ((uint32_t)pp[1] << 8) | ((uint32_t)pp[2] << 16) | ((uint32_t)pp[3] << 24);
        } else {
            v = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) v |= (uint32_t)rowp[reflect101(c + k, W)] << (8 * k);
        }
        ((uint32_t *)s_src)[i] = v;
        if (v) {
            seen = 1;
            const uint32_t nib = ((v & 0xffu) ? 1u : 0u) | ((v & 0xff00u) ? 2u : 0u) | ((v & 0xff0000u) ? 4u : 0u) |
                                 ((v & 0xff000000u) ? 8u : 0u);
            atomicOr(&s_nz[sr * 3 + (q >> 3)], nib << (4 * (q & 7)));
        }
    }
    // empty source window: every stage outputs zero, which the (cleared) buffers already hold
    if (!__syncthreads_or(seen)) return;
    for (int i = tid; i < 256; i += 256) s_cw[i] = color_w[i];
    if (tid < maxk) {
        s_sw[tid] = space_w[tid];
        s_ofs[tid] = (int)tdr[tid] * kSP + (int)tdc[tid];
    }
    for (int i = tid; i < kBH * kBW / 4; i += 256) ((uint32_t *)s_thr)[i] = 0u;
    for (int i = tid; i < kEH * kEW / 4; i += 256) ((uint32_t *)s_er)[i] = 0u;
    // ---- horizontal 7-wide OR, one source row per thread, as shifts of the row mask ----
    if (tid < kSH) {
        const unsigned long long lo = (unsigned long long)s_nz[tid * 3] | ((unsigned long long)s_nz[tid * 3 + 1] << 32);
        const unsigned long long hi = s_nz[tid * 3 + 2];
        unsigned long long alo = 0, ahi = 0;
#pragma unroll
        for (int d = kSX - kHalo; d <= kSX - kHalo + 2 * kR; ++d) {       // ring column lc = source columns lc+3 .. lc+9
            alo |= (lo >> d) | (hi << (64 - d));
            ahi |= hi >> d;
        }
        s_hlo[tid] = alo;
        s_hhi[tid] = (uint32_t)ahi;
    }
    __syncthreads();
    // ---- live ring cells: in the image and with a nonzero pixel somewhere in their 7x7 neighbourhood.
    //      One ring row per thread: vertical OR of seven row masks, then the set bits go to the list. ----
    if (tid < kBH) {
        const int lr = tid;
        const int r = r0 - 2 + lr;
        unsigned long long vlo = 0;
        uint32_t vhi = 0;
        if (r >= 0 && r < H) {
#pragma unroll
            for (int j = 0; j <= 2 * kR; ++j) { vlo |= s_hlo[lr + j]; vhi |= s_hhi[lr + j]; }
            // columns inside the image: lc in [max(0, 2 - c0), min(kBW - 1, W + 1 - c0)]
            const int lc_lo = 2 - c0 > 0 ? 2 - c0 : 0;
            const int lc_hi = W + 1 - c0 < kBW - 1 ? W + 1 - c0 : kBW - 1;
            unsigned long long mlo = ~0ull;
            uint32_t mhi = (1u << (kBW - 64)) - 1u;
            if (lc_lo > 0) mlo &= ~((1ull << lc_lo) - 1ull);
            if (lc_hi < 63) { mlo &= (1ull << (lc_hi + 1)) - 1ull; mhi = 0u; }
            else if (lc_hi < kBW - 1) mhi &= (1u << (lc_hi - 63)) - 1u;
            vlo &= mlo; vhi &= mhi;
        }
        const int cnt = __popcll(vlo) + __popc(vhi);
        if (cnt) {
            int pos = atomicAdd(&s_count, cnt);
            while (vlo) { const int bit = __ffsll((long long)vlo) - 1; vlo &= vlo - 1; s_list[pos++] = (uint16_t)(lr * kBW + bit); }
            while (vhi) { const int bit = __ffs(vhi) - 1; vhi &= vhi - 1; s_list[pos++] = (uint16_t)(lr * kBW + 64 + bit); }
        }
    }
    __syncthreads();
    // ---- bilateral + threshold on the compacted list ----
    const int nlive = s_count;
    for (int j = tid; j < nlive; j += 256) {
        const int i = s_list[j];
        const int lr = i / kBW, lc = i - lr * kBW;
        const int ctr = (lr + kR) * kSP + lc + (kSX - 2);             // ring (lr, lc) = source (lr+3, lc+6)
        const int v0 = s_src[ctr];
        float sum = 0.f, wsum = 0.f;
        for (int k = 0; k < maxk; ++k) {
            const int v = s_src[ctr + s_ofs[k]];
            const int dv = v - v0;
            const float w = s_sw[k] * s_cw[dv < 0 ? -dv : dv];
            if (use_fma) sum = __fmaf_rn((float)v, w, sum);
            else sum = sum + (float)v * w;
            wsum = wsum + w;
        }
        uint8_t outv = (uint8_t)__float2int_rn(sum / wsum);             // cvRound: half to even
        const int r = r0 - 2 + lr, c = c0 - 2 + lc;
        const bool inner = lr >= 2 && lr < kBH - 2 && lc >= 2 && lc < kBW - 2;
        if (inner && bil_out && outv) bil_out[((int64_t)f * H + r) * W + c] = outv;
        outv = outv > thresh ? outv : (uint8_t)0;
        if (inner && thr_out && outv) thr_out[((int64_t)f * H + r) * W + c] = outv;
        s_thr[i] = outv;
    }
    __syncthreads();
    // ---- erosion (window coordinates clamp to the IMAGE: scipy 'reflect' == edge replicate at radius 1).
    //      Only a live cell can have survived the threshold, and the minimum over a window is 0 as soon as its
    //      centre is: the pass runs over the list (cells of the eroded ring outside the image are never read). ----
    int er_any = 0;
    for (int j = tid; j < nlive; j += 256) {
        const int i = s_list[j];
        int acc = s_thr[i];
        if (!acc) continue;
        const int lr = i / kBW, lc = i - lr * kBW;
        if (lr < 1 || lr > kEH || lc < 1 || lc > kEW) continue;          // outermost ring layer: not an erosion cell
        const int r = r0 - 2 + lr, c = c0 - 2 + lc;
        for (int dr = -1; dr <= 1; ++dr)
            for (int dc = -1; dc <= 1; ++dc) {
                const int v = s_thr[(clampi(r + dr, H) - r0 + 2) * kBW + (clampi(c + dc, W) - c0 + 2)];
                acc = v < acc ? v : acc;
            }
        s_er[(lr - 1) * kEW + (lc - 1)] = (uint8_t)acc;
        er_any |= acc;
    }
    if (er_any) s_er_any = 1;
    __syncthreads();
    if (!s_er_any) return;                           // opened tile is all zero: already the buffer's content
    // ---- dilation ----
    for (int i = tid; i < kTH * kTW; i += 256) {
        const int lr = i / kTW, lc = i - lr * kTW;
        const int r = r0 + lr, c = c0 + lc;
        if (r >= H || c >= W) continue;
        int acc = 0;
        for (int dr = -1; dr <= 1; ++dr)
            for (int dc = -1; dc <= 1; ++dc) {
                const int v = s_er[(clampi(r + dr, H) - r0 + 1) * kEW + (clampi(c + dc, W) - c0 + 1)];
                acc = v > acc ? v : acc;
            }
        if (acc) open_out[((int64_t)f * H + r) * W + c] = (uint8_t)acc;
    }
}

void launch_filter_fused(hipStream_t s, const uint8_t *src, int F, int H, int W, const BilateralTables &t,
                         int use_fma, int thresh, uint8_t *bil_out, uint8_t *thr_out, uint8_t *open_out)
{
    const int ntx = (W + kTW - 1) / kTW, nty = (H + kTH - 1) / kTH;
    const size_t plane = (size_t)F * H * W;
    // the kernel stores nonzero results only: outputs start cleared
    hipError_t me = hipMemsetAsync(open_out, 0, plane, s);
    if (me == hipSuccess && bil_out) me = hipMemsetAsync(bil_out, 0, plane, s);
    if (me == hipSuccess && thr_out) me = hipMemsetAsync(thr_out, 0, plane, s);
    if (me != hipSuccess) { g_launch_error = (int)me; return; }
    for (int f0 = 0; f0 < F; f0 += 32768) {
        const int fc = F - f0 < 32768 ? F - f0 : 32768;
        const int64_t o = (int64_t)f0 * H * W;
        hipLaunchKernelGGL(k_filter_fused, dim3(ntx, nty, fc), dim3(256), 0, s,
                           src + o, H, W, t.color_w, t.space_w, t.tap_dr, t.tap_dc, t.maxk, use_fma, thresh,
                           bil_out ? bil_out + o : nullptr, thr_out ? thr_out + o : nullptr, open_out + o);
    }
}

}  // namespace swk

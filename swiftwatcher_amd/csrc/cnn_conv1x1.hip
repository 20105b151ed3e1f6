// 1x1 convolutions of the receptive-field cropped classifier (the squeeze and expand1x1 convolutions of every Fire
// module, segment_classification.py of the reference :14-67 via torchvision's SqueezeNet-1.0) as ONE kernel each:
//
//     dst[n][off_y + y][off_x + x][c_off + co] = max(sum_ci src[n][crop_y + y][crop_x + x][ci] * W[co][ci] + bias[co], 0)
//
// i.e. convolution + bias + ReLU + placement into the next layer's tile (and, for expand1x1, the channel
// concatenation).  Through MIOpen the same step was four launches and three passes over the activation: a zero fill of
// the convolution's output, the implicit-GEMM kernel, and k_bias_relu_place reading that output again to write the tile --
// and the expand convolution ran over the whole squeeze tile although only its centre is used.
//
// A 1x1 convolution in channels-last layout is a plain GEMM, rows = pixels (n * h * w of them), K = input channels,
// N = output channels.  v_mfma_f32_32x32x2_f32 (exact float32: a k-ordered fmaf chain, 64 cycles per instruction per SIMD,
// the f32 matrix peak of 157 TFLOP/s): lane l supplies A[row l & 31][k = l >> 5] and B[k = l >> 5][col l & 31].
//   * a wave owns 32 pixels.  A lane (row r, half h) loads 8 consecutive input channels kb + 8h .. kb + 8h + 7 of its
//     pixel (two 16-byte loads of one 32-byte piece of the pixel's channel vector) and feeds them to 8 MFMA steps; step i
//     multiplies channels {kb + i, kb + 8 + i} -- a permuted k order, matched by reading weight row kb + 8h + i for half h.
//   * the weights (at most 512 x 64 or 64 x 256 floats) sit transposed in LDS, [ci][co] with pitch N + 1: the staging
//     reads W coalesced along ci and writes conflict-free, and a B-operand read is 32 consecutive floats per half.
//   * accumulators: N / 32 tiles of 16 registers; bias, ReLU and the strided store happen from the accumulator layout
//     (col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5)): every store instruction writes two 128-byte pieces.
//   * the next 16 input channels are loaded while the current ones are multiplied (8 x N / 32 MFMAs = 512 .. 4096 cycles
//     per 32 bytes per lane: the loads are far from the critical path).
// Launched on the CALLER's stream (PyTorch's current stream), like the other glue kernels (cnn_aux.hip).
#include "swk_internal.h"

namespace swk {

typedef float f16v __attribute__((ext_vector_type(16)));

template <int NBLK>
__global__ __launch_bounds__(512) void k_conv1x1_relu_place(const float *__restrict__ src, int64_t rows, int sh, int sw, int cin, int crop_y,
                                                            int crop_x, int h, int w, const float *__restrict__ wgt, const float *__restrict__ bias,
                                                            int cout, float *__restrict__ dst, int dH, int dW, int dC, int off_y, int off_x, int c_off)
{
    constexpr int NP = 32 * NBLK, PITCH = NP + 1;
    extern __shared__ float lds[];                 // weights [cin][PITCH], then 16 x 32 destination row offsets
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    int64_t *rowoff = (int64_t *)(lds + ((cin * PITCH + 1) & ~1)) + wave * 32;
    // ---- weights, transposed: coalesced along ci in W[co][ci], conflict-free in LDS ----
    for (int i = tid; i < cin * NP; i += 512) {
        const int co = i / cin, ci = i - co * cin;
        lds[ci * PITCH + co] = co < cout ? wgt[(int64_t)co * cin + ci] : 0.0f;
    }
    __syncthreads();
    float bv[NBLK];
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb) bv[nb] = nb * 32 + r < cout ? bias[nb * 32 + r] : 0.0f;

    const int hw = h * w;
    const int64_t ntiles = (rows + 31) >> 5;
    for (int64_t tile = (int64_t)blockIdx.x * 8 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 8) {
        const int64_t m = tile * 32 + r;
        const bool valid = m < rows;
        const int64_t mm = valid ? m : rows - 1;
        const int64_t b = mm / hw;
        const int rem = (int)(mm - b * hw);
        const int y = rem / w, x = rem - y * w;
        const float *p = src + ((b * sh + crop_y + y) * sw + crop_x + x) * (int64_t)cin + 8 * hh;
        if (hh == 0) rowoff[r] = valid ? ((b * dH + off_y + y) * dW + off_x + x) * (int64_t)dC + c_off : -1;
        f16v acc[NBLK];
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[nb][e] = 0.0f;
        float4 a0 = *(const float4 *)p, a1 = *(const float4 *)(p + 4);
        for (int kb = 0; kb < cin; kb += 16) {
            const float4 c0 = a0, c1 = a1;
            if (kb + 16 < cin) { a0 = *(const float4 *)(p + kb + 16); a1 = *(const float4 *)(p + kb + 20); }
            const float av[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            // B operands of step i + 1 are read from LDS while step i multiplies (two register sets, scheduling fences:
            // left alone the compiler reads each operand right before its MFMA and the wave waits out the LDS latency)
            const float *wrow = lds + (kb + 8 * hh) * PITCH + r;
            float bw[2][NBLK];
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb) bw[0][nb] = wrow[32 * nb];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i < 7) {
#pragma unroll
                    for (int nb = 0; nb < NBLK; ++nb) bw[(i + 1) & 1][nb] = wrow[(i + 1) * PITCH + 32 * nb];
                }
#pragma unroll
                for (int nb = 0; nb < NBLK; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bw[i & 1][nb], acc[nb], 0, 0, 0);
                if (i < 7) __builtin_amdgcn_sched_group_barrier(0x100 /* DS read */, NBLK, 0);
                __builtin_amdgcn_sched_group_barrier(0x008 /* MFMA */, NBLK, 0);
            }
        }
        // ---- bias + ReLU + placement straight from the accumulator layout ----
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int64_t ro = rowoff[(e & 3) + 8 * (e >> 2) + 4 * hh];          // written by this wave's own lanes: in order
            if (ro < 0) continue;
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb)
                if (nb * 32 + r < cout) dst[ro + nb * 32 + r] = fmaxf(acc[nb][e] + bv[nb], 0.0f);
        }
    }
}

template <int NBLK>
static int launch_conv1x1(hipStream_t s, const float *src, int64_t rows, int sh, int sw, int cin, int crop_y, int crop_x, int h, int w,
                          const float *wgt, const float *bias, int cout, float *dst, int dH, int dW, int dC, int off_y, int off_x, int c_off)
{
    const size_t lds = (size_t)((cin * (32 * NBLK + 1) + 1) & ~1) * sizeof(float) + 16 * 32 * sizeof(int64_t);
    static unsigned long long attr_mask = 0;
    if (!ensure_dyn_lds((const void *)k_conv1x1_relu_place<NBLK>, 160 * 1024 - 256, attr_mask)) return SWK_ERR_HIP;
    if (lds > 160 * 1024 - 256) return SWK_ERR_CAPACITY;
    const int64_t ntiles = (rows + 31) / 32;
    int64_t blocks = (ntiles + 7) / 8;
    const int64_t cap = lds > 80 * 1024 ? 256 : 512;          // one or two 512-thread workgroups per CU, persistent over the row tiles
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((k_conv1x1_relu_place<NBLK>), dim3((unsigned)blocks), dim3(512), lds, s, src, rows, sh, sw, cin, crop_y, crop_x, h, w,
                       wgt, bias, cout, dst, dH, dW, dC, off_y, off_x, c_off);
    return hipGetLastError() == hipSuccess ? SWK_OK : SWK_ERR_HIP;
}

}  // namespace swk

#pragma GCC visibility push(default)
extern "C" {

int32_t swk_nhwc_conv1x1_bias_relu_place(void *stream, const float *src, int32_t n, int32_t sh, int32_t sw, int32_t cin, int32_t crop_y,
                                         int32_t crop_x, int32_t h, int32_t w, const float *weight, const float *bias, int32_t cout,
                                         float *dst, int32_t dH, int32_t dW, int32_t dC, int32_t off_y, int32_t off_x, int32_t c_off)
{
    if (!src || !weight || !bias || !dst || n < 1 || h < 1 || w < 1 || cin < 16 || (cin & 15) || cin > 1024 || cout < 1 || cout > 256 ||
        crop_y < 0 || crop_x < 0 || crop_y + h > sh || crop_x + w > sw || off_y < 0 || off_x < 0 || off_y + h > dH || off_x + w > dW ||
        c_off < 0 || c_off + cout > dC || (((uintptr_t)src) & 15))
        return SWK_ERR_ARG;
    using namespace swk;
    const int64_t rows = (int64_t)n * h * w;
    hipStream_t s = (hipStream_t)stream;
    switch ((cout + 31) / 32) {
    case 1: return launch_conv1x1<1>(s, src, rows, sh, sw, cin, crop_y, crop_x, h, w, weight, bias, cout, dst, dH, dW, dC, off_y, off_x, c_off);
    case 2: return launch_conv1x1<2>(s, src, rows, sh, sw, cin, crop_y, crop_x, h, w, weight, bias, cout, dst, dH, dW, dC, off_y, off_x, c_off);
    case 3: return launch_conv1x1<3>(s, src, rows, sh, sw, cin, crop_y, crop_x, h, w, weight, bias, cout, dst, dH, dW, dC, off_y, off_x, c_off);
    case 4: return launch_conv1x1<4>(s, src, rows, sh, sw, cin, crop_y, crop_x, h, w, weight, bias, cout, dst, dH, dW, dC, off_y, off_x, c_off);
    case 6: return launch_conv1x1<6>(s, src, rows, sh, sw, cin, crop_y, crop_x, h, w, weight, bias, cout, dst, dH, dW, dC, off_y, off_x, c_off);
    case 8: return launch_conv1x1<8>(s, src, rows, sh, sw, cin, crop_y, crop_x, h, w, weight, bias, cout, dst, dH, dW, dC, off_y, off_x, c_off);
    default: return SWK_ERR_ARG;          // 5 and 7 blocks (129..160, 193..224 channels) do not occur in SqueezeNet-1.0
    }
}

}  // extern "C"
#pragma GCC visibility pop

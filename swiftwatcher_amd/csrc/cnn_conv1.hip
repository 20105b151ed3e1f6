// conv1 of the classifier (segment_classification.py of the reference :47-67 via torchvision's SqueezeNet-1.0: Conv2d(3, 96, 7,
// stride 2)) on the window the receptive-field cropped network reads, fused with bias and ReLU:
//
//     dst[n][y][x][co] = max(sum_{dy,dx,c} src[n][2 (lo + y) + dy][2 (lo + x) + dx][c] * W[co][c][dy][dx] + bias[co], 0)      y, x < m
//
// Through MIOpen this was an implicit-GEMM kernel at a third of the f32 matrix peak plus a pass that added the bias, applied the
// ReLU and cropped (k_bias_relu_place): the convolution's raw output made one round trip through HBM for nothing.
//
// Implicit GEMM on v_mfma_f32_32x32x2_f32 (exact float32), rows = output pixels of the whole batch, N = 96 output channels, K = the
// 7 x 7 x 3 = 147 patch values.  In channels-last layout a patch ROW is 21 consecutive floats, so K is ordered (dy, j = 3 dx + c)
// and split between the two k-halves of the MFMA by rows: half 0 multiplies patch rows 0..3, half 1 rows 4..6 and a zero row 7
// (84 k-steps instead of 74: the price of both halves reading the same 21 columns at a fixed distance of four image rows).
//   * B operand (pixels): a lane (pixel r, half h) loads the 21 floats of patch row dy + 4 h as float2s (the row starts at an
//     8-byte boundary: 24 x bytes), one patch row ahead of its use; the image is read out of L1 / L2 (19 KB per segment, every
//     value used by ~12 output pixels).
//   * A operand (weights): all of W, re-laid [dy][j][half][co] with pitch 97, sits in LDS (65 KB); one ds_read_b32 per MFMA.
//   * three accumulators (96 channels) per 32 pixels; a register quad = four consecutive channels of the lane's pixel: float4 stores.
// Launched on the CALLER's stream (PyTorch's current stream).
#include "swk_internal.h"

namespace swk {

typedef float f16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512, 4) void k_conv7x7s2_relu(const float *__restrict__ src, int64_t rows, int side, int lo, int m,
                                                        const float *__restrict__ wgt, const float *__restrict__ bias, float *__restrict__ dst, FastDiv fmm, FastDiv fm)
{
    constexpr int NB = 3, NP = 96, PITCH = NP + 1, KR = 21;          // column blocks, channels, LDS pitch, floats per patch row
    extern __shared__ float lds[];                                   // weights [4 dy][21 j][2 halves][PITCH], then the bias
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    float *lbias = lds + ((4 * KR * 2 * PITCH + 3) & ~3);
    // ---- weights: W[co][c][dy][dx] -> lds[((dyl * 21 + 3 dx + c) * 2 + half) * PITCH + co], patch row dy = dyl + 4 half (row 7: zeros) ----
    //      (walked in W's own order: consecutive lanes read consecutive floats; walked in LDS order the reads were a gather with a
    //      stride of 147 floats, 64 lines per instruction -- a third of a window-sized forward's conv1 time)
    for (int i = tid; i < NP * 147; i += 512) {
        const int co = i / 147, rem = i - co * 147, c = rem / 49, r2 = rem - c * 49, dy = r2 / 7, dx = r2 - dy * 7;
        const int half = dy >> 2, dyl = dy & 3;
        lds[((dyl * KR + 3 * dx + c) * 2 + half) * PITCH + co] = wgt[i];
    }
    for (int i = tid; i < KR * NP; i += 512) {          // patch row 7 (half 1, dyl 3): zeros
        const int j = i / NP, co = i - j * NP;
        lds[((3 * KR + j) * 2 + 1) * PITCH + co] = 0.0f;
    }
    if (tid < NP) lbias[tid] = bias[tid];
    __syncthreads();

    const int mm2 = m * m, rowf = side * 3;                          // floats per image row
    const int64_t ntiles = (rows + 31) >> 5;
    for (int64_t tile = (int64_t)blockIdx.x * 8 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 8) {
        const int64_t q = tile * 32 + r;
        const bool valid = q < rows;
        const unsigned qq = valid ? (unsigned)q : (unsigned)rows - 1u;          // rows < 2^31 (checked by the launcher)
        const unsigned b = fmm.div(qq), rem = qq - b * (unsigned)mm2, y = fm.div(rem), x = rem - y * (unsigned)m;
        // patch row 0 (half 0) or 4 (half 1) of this pixel
        const float *p = src + (((int64_t)b * side + 2 * (lo + y) + 4 * hh) * (int64_t)side + 2 * (lo + x)) * 3;
        f16v acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[nb][e] = 0.0f;
        float2 cur[11], nxt[11];
#pragma unroll
        for (int v = 0; v < 11; ++v) cur[v] = *(const float2 *)(p + 2 * v);          // 22 floats: the 21 of the row and one more
#pragma unroll 1
        for (int dyl = 0; dyl < 4; ++dyl) {
            // the next patch row travels while this one multiplies (after the last one: the same row again, unused -- a load under a
            // branch would be waited for where the branches join)
            const float *pn = p + (dyl < 3 ? dyl + 1 : 3) * rowf;
#pragma unroll
            for (int v = 0; v < 11; ++v) nxt[v] = *(const float2 *)(pn + 2 * v);
            const float *wrow = lds + ((dyl * KR) * 2 + hh) * PITCH + r;
            float aw[2][NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) aw[0][nb] = wrow[32 * nb];
#pragma unroll
            for (int j = 0; j < KR; ++j) {
                if (j < KR - 1) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) aw[(j + 1) & 1][nb] = wrow[(j + 1) * 2 * PITCH + 32 * nb];
                }
                const float bval = (j & 1) ? cur[j >> 1].y : cur[j >> 1].x;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[j & 1][nb], bval, acc[nb], 0, 0, 0);
                if (j < KR - 1) __builtin_amdgcn_sched_group_barrier(0x100 /* DS read */, NB, 0);
                __builtin_amdgcn_sched_group_barrier(0x008 /* MFMA */, NB, 0);
            }
#pragma unroll
            for (int v = 0; v < 11; ++v) cur[v] = nxt[v];
        }
        if (valid) {
            float *o = dst + q * NP + 4 * hh;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = nb * 32 + 8 * g;
                    const float4 b4 = *(const float4 *)(lbias + c + 4 * hh);
                    float4 v;
                    v.x = fmaxf(acc[nb][4 * g] + b4.x, 0.0f);
                    v.y = fmaxf(acc[nb][4 * g + 1] + b4.y, 0.0f);
                    v.z = fmaxf(acc[nb][4 * g + 2] + b4.z, 0.0f);
                    v.w = fmaxf(acc[nb][4 * g + 3] + b4.w, 0.0f);
                    *(float4 *)(o + c) = v;
                }
        }
    }
}

}  // namespace swk

#pragma GCC visibility push(default)
extern "C" {

int32_t swk_nhwc_conv7x7s2_bias_relu(void *stream, const float *src, int32_t n, int32_t side, int32_t lo, int32_t m, const float *weight,
                                     const float *bias, int32_t cout, float *dst)
{
    // every patch of the m x m outputs starting at output (lo, lo) must lie in the image -- including the zero-weighted eighth row and
    // the 22nd float of a patch row, which are read
    if (!src || !weight || !bias || !dst || n < 1 || side < 8 || (side & 1) || lo < 0 || m < 1 || cout != 96 || 2 * (lo + m - 1) + 8 > side ||
        (((uintptr_t)src) & 7) || (((uintptr_t)dst) & 15))
        return SWK_ERR_ARG;
    using namespace swk;
    const size_t lds = (size_t)(((4 * 21 * 2 * 97 + 3) & ~3) + 96) * sizeof(float);
    static unsigned long long attr_mask = 0;
    if (!ensure_dyn_lds((const void *)k_conv7x7s2_relu, 160 * 1024 - 256, attr_mask)) return SWK_ERR_HIP;
    const int64_t rows = (int64_t)n * m * m, ntiles = (rows + 31) / 32;
    if (rows > (int64_t)0x7fffff00) return SWK_ERR_CAPACITY;
    int64_t blocks = (ntiles + 7) / 8;
    if (blocks > 512) blocks = 512;          // two 8-wave workgroups per CU (65 KB of LDS each), persistent over the row tiles
    hipLaunchKernelGGL(k_conv7x7s2_relu, dim3((unsigned)blocks), dim3(512), lds, (hipStream_t)stream, src, rows, side, lo, m, weight, bias, dst,
                       FastDiv((unsigned)(m * m)), FastDiv((unsigned)m));
    return hipGetLastError() == hipSuccess ? SWK_OK : SWK_ERR_HIP;
}

}  // extern "C"
#pragma GCC visibility pop

// The 3x3 expand convolutions of the receptive-field cropped classifier -- three quarters of its multiply-accumulates --
// as ONE kernel each, on the f32 matrix cores:
//
//     dst[n][off_y + y][off_x + x][c_off + co] =
//         max(sum_{dy,dx,ci} src[n][y + dy][x + dx][ci] * W[co][dy][dx][ci] + bias[co], 0)        0 <= y, x < t - 2
//
// a VALID 3x3 convolution over the t x t squeeze tile (the tile carries the halo: the background ring of the persistent
// buffers, segment_classification.py), fused with bias, ReLU, placement into the next layer's tile and the channel
// concatenation behind the expand1x1 output.  Through MIOpen this was a zero fill, an implicit-GEMM kernel at about 60 % of
// the f32 matrix peak on these shapes, and k_bias_relu_place reading the result back to write the tile.
//
// Implicit GEMM, rows = output pixels of the whole batch (n (t-2)^2 of them, a 32-row tile may span segments), K = 9 taps x C
// squeeze channels, N = expand channels; v_mfma_f32_32x32x2_f32 (exact float32, 64 cycles per instruction per SIMD).
//   * A operand straight from global memory (the squeeze tile is read nine times, out of L1 / L2: 32 bytes per lane per
//     8 x NBLK x RT MFMAs): lane (row r, half h) loads channels c0 + 8h .. c0 + 8h + 7 of its pixel at the current tap and
//     feeds 8 MFMA steps, step i multiplying channels {c0 + i, c0 + 8 + i} -- the permuted k order of cnn_conv1x1.hip.
//     The next chunk's 32 bytes are loaded while the current chunk is multiplied.
//   * B operand: the weights, re-laid once on the host side as [tap][ci][co], stream through LDS in chunks of 16 k-rows
//     (one tap, 16 channels: 16 x N floats), double buffered, one workgroup barrier per chunk; the eight waves of a
//     workgroup work on different row tiles of the same chunk.
//   * a wave holds RT row tiles x NBLK column blocks = 8 accumulators (128 registers; 6 for 192 channels) whatever the layer,
//     so every chunk is 64 (48) back-to-back MFMAs between two barriers.
//   * epilogue from the accumulator layout, as in the 1x1 kernel.
// Launched on the CALLER's stream (PyTorch's current stream).
#include "swk_internal.h"

namespace swk {

typedef float f16v __attribute__((ext_vector_type(16)));

template <int NBLK, int RT>
__global__ __launch_bounds__(512) void k_conv3x3_relu_place(const float *__restrict__ src, int64_t rows, int t, int cin, const float *__restrict__ wt,
                                                            const float *__restrict__ bias, int cout, float *__restrict__ dst, int dH, int dW, int dC,
                                                            int off_y, int off_x, int c_off)
{
    constexpr int NP = 32 * NBLK, PITCH = NP + 1;
    extern __shared__ float lds[];                 // two weight chunks [16][PITCH], then destination row offsets [8 waves][RT][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    float *const wb0 = lds, *const wb1 = lds + 16 * PITCH;
    int64_t *rowoff = (int64_t *)(lds + ((2 * 16 * PITCH + 1) & ~1)) + wave * (RT * 32);
    const int o = t - 2, oo = o * o;               // output pixels per side / per segment
    const int chunks_per_tap = cin >> 4, nchunks = 9 * chunks_per_tap;
    float bv[NBLK];
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb) bv[nb] = nb * 32 + r < cout ? bias[nb * 32 + r] : 0.0f;

    // cooperative load of one weight chunk: 16 k-rows x cout floats, contiguous in wt ([tap][ci][co] = [9 cin][cout]).
    // Split in two: the global loads are issued before the chunk's MFMAs, the LDS stores come after them.
    const int wcount = 16 * cout;                  // floats per chunk
    float wreg[NBLK];
    auto chunk_load = [&](int c) {
        const float *g = wt + (int64_t)c * wcount;
#pragma unroll
        for (int j = 0; j < NBLK; ++j) {
            const int i = tid + j * 512, k = i / NP, n = i - k * NP;
            wreg[j] = n < cout ? g[k * cout + n] : 0.0f;
        }
    };
    auto chunk_store = [&](float *buf) {
#pragma unroll
        for (int j = 0; j < NBLK; ++j) {
            const int i = tid + j * 512, k = i / NP, n = i - k * NP;
            buf[k * PITCH + n] = wreg[j];
        }
    };

    const int64_t ntiles = (rows + 31) >> 5;
    const int64_t nsuper = (ntiles + 8 * RT - 1) / (8 * RT);
    for (int64_t super = blockIdx.x; super < nsuper; super += gridDim.x) {
        // ---- this wave's RT row tiles: source pointers (tap 0, 0) and destination offsets ----
        const float *p[RT];
#pragma unroll
        for (int q = 0; q < RT; ++q) {
            const int64_t m = ((super * 8 + wave) * RT + q) * 32 + r;
            const bool valid = m < rows;
            const int64_t mm = valid ? m : rows - 1;
            const int64_t b = mm / oo;
            const int rem = (int)(mm - b * oo);
            const int y = rem / o, x = rem - y * o;
            p[q] = src + ((b * t + y) * t + x) * (int64_t)cin + 8 * hh;
            if (hh == 0) rowoff[q * 32 + r] = valid ? ((b * dH + off_y + y) * dW + off_x + x) * (int64_t)dC + c_off : -1;
        }
        f16v acc[RT][NBLK];
#pragma unroll
        for (int q = 0; q < RT; ++q)
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[q][nb][e] = 0.0f;
        __syncthreads();                           // every wave is done with the previous super tile's weight buffers
        chunk_load(0);
        chunk_store(wb0);
        float4 a0[RT], a1[RT];
#pragma unroll
        for (int q = 0; q < RT; ++q) { a0[q] = *(const float4 *)p[q]; a1[q] = *(const float4 *)(p[q] + 4); }
        __syncthreads();
        for (int c = 0; c < nchunks; ++c) {
            // ---- operands of the NEXT chunk: A into registers, weights towards the other LDS buffer ----
            float4 c0[RT], c1[RT];
#pragma unroll
            for (int q = 0; q < RT; ++q) { c0[q] = a0[q]; c1[q] = a1[q]; }
            if (c + 1 < nchunks) {
                const int cn = c + 1, tap = cn / chunks_per_tap, ch = (cn - tap * chunks_per_tap) << 4;
                const int dy = tap / 3, dx = tap - 3 * dy;
                const int64_t aoff = (int64_t)(dy * t + dx) * cin + ch;
#pragma unroll
                for (int q = 0; q < RT; ++q) { a0[q] = *(const float4 *)(p[q] + aoff); a1[q] = *(const float4 *)(p[q] + aoff + 4); }
                chunk_load(cn);
            }
            // ---- 8 MFMA steps x NBLK column blocks x RT row tiles on the current chunk.  The B operands of step i + 1 are
            //      read from LDS while step i multiplies (two register sets); left to itself the compiler reads each
            //      operand right before the MFMA that needs it and the wave waits out the LDS latency every two MFMAs ----
            const float *wrow = ((c & 1) ? wb1 : wb0) + 8 * hh * PITCH + r;
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
                for (int q = 0; q < RT; ++q) {
                    const float av = i < 4 ? (i == 0 ? c0[q].x : i == 1 ? c0[q].y : i == 2 ? c0[q].z : c0[q].w)
                                           : (i == 4 ? c1[q].x : i == 5 ? c1[q].y : i == 6 ? c1[q].z : c1[q].w);
#pragma unroll
                    for (int nb = 0; nb < NBLK; ++nb) acc[q][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bw[i & 1][nb], acc[q][nb], 0, 0, 0);
                }
                if (i < 7) __builtin_amdgcn_sched_group_barrier(0x100 /* DS read */, NBLK, 0);
                __builtin_amdgcn_sched_group_barrier(0x008 /* MFMA */, NBLK * RT, 0);
            }
            // the other buffer is the one chunk c - 1 was read from: every wave left it at the last barrier
            if (c + 1 < nchunks) chunk_store((c & 1) ? wb0 : wb1);
            __syncthreads();
        }
        // ---- bias + ReLU + placement straight from the accumulator layout ----
#pragma unroll
        for (int q = 0; q < RT; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t ro = rowoff[q * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh];
                if (ro < 0) continue;
#pragma unroll
                for (int nb = 0; nb < NBLK; ++nb)
                    if (nb * 32 + r < cout) dst[ro + nb * 32 + r] = fmaxf(acc[q][nb][e] + bv[nb], 0.0f);
            }
    }
}

template <int NBLK, int RT>
static int launch_conv3x3(hipStream_t s, const float *src, int64_t rows, int t, int cin, const float *wt, const float *bias, int cout,
                          float *dst, int dH, int dW, int dC, int off_y, int off_x, int c_off)
{
    const size_t lds = (size_t)((2 * 16 * (32 * NBLK + 1) + 1) & ~1) * sizeof(float) + (size_t)8 * RT * 32 * sizeof(int64_t);
    const int64_t ntiles = (rows + 31) / 32;
    int64_t blocks = (ntiles + 8 * RT - 1) / (8 * RT);
    if (blocks > 512) blocks = 512;                // persistent over the super tiles; two workgroups fit a CU at 128 registers
    hipLaunchKernelGGL((k_conv3x3_relu_place<NBLK, RT>), dim3((unsigned)blocks), dim3(512), lds, s, src, rows, t, cin, wt, bias, cout, dst,
                       dH, dW, dC, off_y, off_x, c_off);
    return hipGetLastError() == hipSuccess ? SWK_OK : SWK_ERR_HIP;
}

}  // namespace swk

#pragma GCC visibility push(default)
extern "C" {

int32_t swk_nhwc_conv3x3_bias_relu_place(void *stream, const float *src, int32_t n, int32_t t, int32_t cin, const float *weight_t,
                                         const float *bias, int32_t cout, float *dst, int32_t dH, int32_t dW, int32_t dC, int32_t off_y,
                                         int32_t off_x, int32_t c_off)
{
    if (!src || !weight_t || !bias || !dst || n < 1 || t < 3 || cin < 16 || (cin & 15) || cin > 1024 || cout < 1 || cout > 256 ||
        off_y < 0 || off_x < 0 || off_y + t - 2 > dH || off_x + t - 2 > dW || c_off < 0 || c_off + cout > dC || (((uintptr_t)src) & 15))
        return SWK_ERR_ARG;
    using namespace swk;
    const int64_t rows = (int64_t)n * (t - 2) * (t - 2);
    hipStream_t s = (hipStream_t)stream;
    switch ((cout + 31) / 32) {
    case 1: return launch_conv3x3<1, 4>(s, src, rows, t, cin, weight_t, bias, cout, dst, dH, dW, dC, off_y, off_x, c_off);
    case 2: return launch_conv3x3<2, 4>(s, src, rows, t, cin, weight_t, bias, cout, dst, dH, dW, dC, off_y, off_x, c_off);
    case 3: return launch_conv3x3<3, 2>(s, src, rows, t, cin, weight_t, bias, cout, dst, dH, dW, dC, off_y, off_x, c_off);
    case 4: return launch_conv3x3<4, 2>(s, src, rows, t, cin, weight_t, bias, cout, dst, dH, dW, dC, off_y, off_x, c_off);
    case 6: return launch_conv3x3<6, 1>(s, src, rows, t, cin, weight_t, bias, cout, dst, dH, dW, dC, off_y, off_x, c_off);
    case 8: return launch_conv3x3<8, 1>(s, src, rows, t, cin, weight_t, bias, cout, dst, dH, dW, dC, off_y, off_x, c_off);
    default: return SWK_ERR_ARG;          // 5 and 7 blocks do not occur in SqueezeNet-1.0
    }
}

}  // extern "C"
#pragma GCC visibility pop

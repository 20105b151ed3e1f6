// Max-pool + the squeeze convolution behind it as ONE kernel (the three MaxPool2d(3, 2) of SqueezeNet-1.0 are each followed by a Fire
// module's 1x1 squeeze: features 2 -> 3, 6 -> 7, 11 -> 12 of torchvision's network, segment_classification.py of the reference :47-67):
//
//     dst[n][off_y + y][off_x + x][co] = max(sum_ci W[co][ci] * max_{dy,dx < 3} src[n][2 y + dy][2 x + dx][ci] + bias[co], 0)
//
// Separately the pooled tensor made one round trip through HBM (written by k_maxpool3s2, read back by k_conv1x1_relu_place): 332 KB of
// the 1.07 MB the pair moves per segment at pool3 + fire9.  Here a workgroup owns a segment: its P x P pooled pixels (64 or 81) are the
// MFMA's pixel operand, formed 32 channels at a time in LDS while the previous 32 multiply.
//   * pooling stage (all threads): item = (pooled pixel, channel quad of the chunk): nine float4 loads (a pixel's 32 channels are one
//     128-byte line; neighbouring windows share lines through L1), eight float4 maxima, four LDS stores into B[k][pixel].
//   * the chunk's weights W[co][kb .. kb + 31] are read along ci (128 bytes per output channel) and transposed into LDS A[k][co].
//   * product stage: wave (pt, nb) owns 32 pixels x 32 output channels, 16 v_mfma_f32_32x32x2_f32 per chunk (exact float32: the same
//     k-ordered chain as k_conv1x1_relu_place: channels {kb + i, kb + 16 + i} in step i); accumulator register quads = four consecutive
//     output channels of the lane's pixel: float4 stores after bias and ReLU.
// Double-buffered LDS (chunk c + 1 is pooled and staged while chunk c multiplies), one barrier per chunk.  Launched on the CALLER's
// stream (PyTorch's current stream), like the other classifier kernels.
#include "swk_internal.h"

namespace swk {

typedef float f16v __attribute__((ext_vector_type(16)));

template <int PT, int NB>
__global__ __launch_bounds__(64 * PT * NB) void k_pool_squeeze(const float *__restrict__ src, int n, int T, int C, int P, const float *__restrict__ wgt,
                                                               const float *__restrict__ bias, int S, float *__restrict__ dst, int dH, int dW, int dC,
                                                               int off_y, int off_x, const float *__restrict__ ring, int live_lo, int live_n)
{
    constexpr int NW = PT * NB, NT = 64 * NW, PP = 32 * PT + 1, SP = 32 * NB + 1, KC = 32;
    __shared__ float sB[2][KC * PP];          // pooled pixels of a chunk: [channel k][pixel]
    __shared__ float sA[2][KC * SP];          // the chunk's weights: [channel k][output channel]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int pt = wave % PT, nb = wave / PT;
    const int npix = P * P, nchunk = C / KC;
    const int64_t rs = (int64_t)T * C;          // floats per source row

    auto stage = [&](int seg, int c, int buf) {
        const int kb = c * KC;
        // ---- weights: (co, quad of k) items, float4 along ci ----
        for (int i = tid; i < 32 * NB * 8; i += NT) {
            const int co = i >> 3, q = i & 7;
            float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
            if (co < S) w = *(const float4 *)(wgt + (int64_t)co * C + kb + 4 * q);
            float *a = &sA[buf][(4 * q) * SP + co];
            a[0] = w.x; a[SP] = w.y; a[2 * SP] = w.z; a[3 * SP] = w.w;
        }
        // ---- pooled pixels: (pixel, quad of k) items ----
        // pixels outside the live square [live_lo, live_lo + live_n)^2 hold the same values in every segment's tile (the ring the
        // tiles were created with): they are read from ONE tile (`ring`, cache-resident) instead of the segment's own -- at pool2 /
        // pool3 half of a tile's bytes.  ring == src's first tile and live = the whole tile when the caller has no ring.
        const float *base = src + (int64_t)seg * T * rs + kb, *rbase = ring + kb;
        for (int i = tid; i < 32 * PT * 8; i += NT) {
            const int p = i >> 3, q = i & 7;
            float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p < npix) {
                const int py = p / P, px = p - py * P;
                const int64_t o0 = (int64_t)(2 * py) * rs + (int64_t)(2 * px) * C + 4 * q;
                bool iny[3], inx[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    iny[d] = (unsigned)(2 * py + d - live_lo) < (unsigned)live_n;
                    inx[d] = (unsigned)(2 * px + d - live_lo) < (unsigned)live_n;
                }
                m = *(const float4 *)((iny[0] && inx[0] ? base : rbase) + o0);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        if (dy == 0 && dx == 0) continue;
                        const float4 v = *(const float4 *)((iny[dy] && inx[dx] ? base : rbase) + o0 + dy * rs + dx * C);
                        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
                    }
            }
            float *b = &sB[buf][(4 * q) * PP + p];
            b[0] = m.x; b[PP] = m.y; b[2 * PP] = m.z; b[3 * PP] = m.w;
        }
    };

    for (int seg = blockIdx.x; seg < n; seg += gridDim.x) {
        f16v acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
        stage(seg, 0, 0);
        __syncthreads();
        for (int c = 0; c < nchunk; ++c) {
            const int buf = c & 1;
            if (c + 1 < nchunk) stage(seg, c + 1, buf ^ 1);
            const float *a = &sA[buf][(16 * hh) * SP + 32 * nb + r];
            const float *b = &sB[buf][(16 * hh) * PP + 32 * pt + r];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i * SP], b[i * PP], acc, 0, 0, 0);
            __syncthreads();
        }
        // ---- bias + ReLU + placement: register quad g = output channels 32 nb + 8 g + 4 hh .. + 3 of pixel 32 pt + r ----
        const int p = 32 * pt + r;
        if (p < npix) {
            const int py = p / P, px = p - py * P;
            float *o = dst + (((int64_t)seg * dH + off_y + py) * dW + off_x + px) * dC + 32 * nb + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = 32 * nb + 8 * g + 4 * hh;
                if (co < S) {
                    const float4 b4 = *(const float4 *)(bias + co);
                    float4 v;
                    v.x = fmaxf(acc[4 * g] + b4.x, 0.0f);
                    v.y = fmaxf(acc[4 * g + 1] + b4.y, 0.0f);
                    v.z = fmaxf(acc[4 * g + 2] + b4.z, 0.0f);
                    v.w = fmaxf(acc[4 * g + 3] + b4.w, 0.0f);
                    *(float4 *)(o + 8 * g) = v;
                }
            }
        }
    }
}

template <int PT, int NB>
static int launch_pool_squeeze(hipStream_t s, const float *src, int n, int T, int C, int P, const float *wgt, const float *bias, int S, float *dst,
                               int dH, int dW, int dC, int off_y, int off_x, const float *ring, int live_lo, int live_n)
{
    int blocks = n < 256 * 8 ? n : 256 * 8;          // persistent over the segments beyond a few workgroups per CU
    hipLaunchKernelGGL((k_pool_squeeze<PT, NB>), dim3((unsigned)blocks), dim3(64 * PT * NB), 0, s, src, n, T, C, P, wgt, bias, S, dst, dH, dW, dC,
                       off_y, off_x, ring, live_lo, live_n);
    return hipGetLastError() == hipSuccess ? SWK_OK : SWK_ERR_HIP;
}

}  // namespace swk

#pragma GCC visibility push(default)
extern "C" {

int32_t swk_nhwc_maxpool3s2_conv1x1_bias_relu_place(void *stream, const float *src, int32_t n, int32_t t, int32_t cin, const float *weight,
                                                    const float *bias, int32_t cout, float *dst, int32_t dH, int32_t dW, int32_t dC,
                                                    int32_t off_y, int32_t off_x, const float *ring, int32_t live_lo, int32_t live_n)
{
    if (!src || !weight || !bias || !dst || n < 1 || t < 3 || cin < 32 || (cin & 31) || cout < 4 || (cout & 3) || cout > 64 || (dC & 3) ||
        off_y < 0 || off_x < 0 || (((uintptr_t)src | (uintptr_t)dst | (uintptr_t)weight | (uintptr_t)bias) & 15))
        return SWK_ERR_ARG;
    if (!ring) { ring = src; live_lo = 0; live_n = t; }
    if (live_lo < 0 || live_n < 0 || live_lo + live_n > t || ((uintptr_t)ring & 15)) return SWK_ERR_ARG;
    const int P = (t - 3) / 2 + 1;
    if (off_y + P > dH || off_x + P > dW || cout > dC || P * P > 96) return SWK_ERR_ARG;
    using namespace swk;
    hipStream_t s = (hipStream_t)stream;
    const int PT = (P * P + 31) / 32, NB = (cout + 31) / 32;
#define SWK_PS_ARGS s, src, n, t, cin, P, weight, bias, cout, dst, dH, dW, dC, off_y, off_x, ring, live_lo, live_n
    if (PT == 1 && NB == 1) return launch_pool_squeeze<1, 1>(SWK_PS_ARGS);
    if (PT == 2 && NB == 1) return launch_pool_squeeze<2, 1>(SWK_PS_ARGS);
    if (PT == 3 && NB == 1) return launch_pool_squeeze<3, 1>(SWK_PS_ARGS);
    if (PT == 1 && NB == 2) return launch_pool_squeeze<1, 2>(SWK_PS_ARGS);
    if (PT == 2 && NB == 2) return launch_pool_squeeze<2, 2>(SWK_PS_ARGS);
    if (PT == 3 && NB == 2) return launch_pool_squeeze<3, 2>(SWK_PS_ARGS);
#undef SWK_PS_ARGS
    return SWK_ERR_ARG;
}

}  // extern "C"
#pragma GCC visibility pop

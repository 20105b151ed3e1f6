// Fire -> Fire hand-over without the expand1x1 tensor (segment_classification.py of the reference :14-67: torchvision's SqueezeNet-1.0).
//
// The output of a Fire module's two expand convolutions is read by exactly one consumer, the squeeze convolution of the next Fire
// (fire2 -> 3, 3 -> 4, 5 -> 6, 6 -> 7, 7 -> 8): E = [E1 | E3] is written (64 .. 192 + 64 .. 192 channels per pixel) and read back to
// be contracted to 16 .. 64 channels.  The squeeze is linear in E before its bias and ReLU, so its sum splits by producer,
//
//     S' = relu(Wq1 E1 + Wq3 E3 + bq),        E1 = relu(W1 s + b1)   (a 1 x 1 convolution of the squeeze tile s),
//
// and the first term needs nothing but the pixel's own squeeze vector s: this kernel computes  P1 = Wq1 relu(W1 s + b1)  per pixel --
// two chained GEMMs on the f32 matrix cores -- and writes the 16 .. 64 partial sums instead of the 64 .. 192 expand1x1 channels.  The
// next squeeze (cnn_conv1x1.hip with an addend) then reads only E3 and P1.  E1 never reaches memory.
//
// Chaining costs no shuffle: v_mfma_f32_32x32x2_f32 computes D = A B with lane l supplying A[i = l & 31][k = l >> 5] and
// B[k = l >> 5][j = l & 31]; with the weights as A and the pixels as B (cnn_conv1x1.hip) accumulator register e = 4 g + q of lane l holds
// output channel 8 g + 4 (l >> 5) + q of pixel l & 31 -- which IS a B operand whose two k values are the channels 8 g + q and
// 8 g + 4 + q: the second GEMM runs one MFMA per accumulator register of the first, its weights read from LDS in that channel order.
// Only the n x n pixels whose squeeze vector depends on the segment are computed (on the ring of the next layer's square P1 keeps the
// blank image's value, written once when the buffers are made), like expand1x1 before.
// Float32 throughout (a k-ordered fmaf chain per MFMA); the squeeze's sum is now formed as (E1 part) + (E3 part) instead of one chain:
// a different rounding order of the same products (scores move by a few 1e-7; tests/test_classifier.py).
// Launched on the CALLER's stream (PyTorch's current stream).
#include "swk_internal.h"

namespace swk {

typedef float f16v __attribute__((ext_vector_type(16)));

// S = squeeze channels of this Fire (K of the first GEMM), NB1 = expand1x1 channels / 32, RB = ceil(next squeeze channels / 32)
template <int S, int NB1, int RB, int NWV>
__global__ __launch_bounds__(64 * NWV) void k_expand1x1_sq_partial(const float *__restrict__ src, int64_t rows, int sh, int sw, int crop_y, int crop_x,
                                                              int h, int w, const float *__restrict__ w1, const float *__restrict__ b1,
                                                              const float *__restrict__ wq, int wq_stride, int sq_out, float *__restrict__ dst,
                                                              int dH, int dW, int dC, int off_y, int off_x, FastDiv fhw, FastDiv fw)
{
    constexpr int C1 = 32 * NB1, P1 = C1 + 1, SQP = 32 * RB, P2 = SQP + 1, KH = S / 2, NV = KH / 4;
    extern __shared__ float lds[];                 // W1^T [S][P1], Wq^T [C1][P2], b1 [C1]
    float *lw1 = lds, *lwq = lds + S * P1, *lb1 = lwq + C1 * P2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5, NT = 64 * NWV;
    const int hw = h * w;
    const int64_t ntiles = (rows + 31) >> 5, stride = (int64_t)gridDim.x * NWV;
    int64_t tile = (int64_t)blockIdx.x * NWV + wave;

    auto locate = [&](int64_t t, int64_t &ro) -> const float * {
        const unsigned m = (unsigned)t * 32u + (unsigned)r;
        const bool valid = m < (unsigned)rows;
        const unsigned mm = valid ? m : (unsigned)rows - 1u;
        const unsigned b = fhw.div(mm), rem = mm - b * (unsigned)hw;
        const unsigned y = fw.div(rem), x = rem - y * (unsigned)w;
        ro = valid ? (((int64_t)b * dH + off_y + y) * dW + off_x + x) * (int64_t)dC : -1;
        return src + (((int64_t)b * sh + crop_y + y) * sw + crop_x + x) * (int64_t)S + KH * hh;
    };
    float4 a[NV], an[NV];
    int64_t ro = -1, ro_next = -1;
    const float *p = src;
    if (tile < ntiles) {
        p = locate(tile, ro);
#pragma unroll
        for (int v = 0; v < NV; ++v) a[v] = *(const float4 *)(p + 4 * v);
    }
    // ---- both weight matrices transposed into LDS: [input channel][output channel], pitch odd ----
    for (int co = tid >> 2; co < C1; co += NT / 4)
        for (int c4 = (tid & 3) * 4; c4 < S; c4 += 16) {
            const float4 wv = *(const float4 *)(w1 + (int64_t)co * S + c4);
            float *q = lw1 + c4 * P1 + co;
            q[0] = wv.x; q[P1] = wv.y; q[2 * P1] = wv.z; q[3 * P1] = wv.w;
        }
    for (int so = tid >> 4; so < SQP; so += NT / 16)
        for (int c4 = (tid & 15) * 4; c4 < C1; c4 += 64) {
            const float4 wv = so < sq_out ? *(const float4 *)(wq + (int64_t)so * wq_stride + c4) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            float *q = lwq + c4 * P2 + so;
            q[0] = wv.x; q[P2] = wv.y; q[2 * P2] = wv.z; q[3 * P2] = wv.w;
        }
    for (int i = tid; i < C1; i += NT) lb1[i] = b1[i];
    __syncthreads();

    for (; tile < ntiles; tile += stride) {
        const bool more = tile + stride < ntiles;
        if (more) {
            const float *pn = locate(tile + stride, ro_next);
#pragma unroll
            for (int v = 0; v < NV; ++v) an[v] = *(const float4 *)(pn + 4 * v);
        }
        float av[KH];
#pragma unroll
        for (int v = 0; v < NV; ++v) { av[4 * v] = a[v].x; av[4 * v + 1] = a[v].y; av[4 * v + 2] = a[v].z; av[4 * v + 3] = a[v].w; }
        // ---- E1 = relu(W1 s + b1): step i multiplies squeeze channels {i, KH + i} (half hh reads weight row KH hh + i) ----
        f16v e1[NB1];
#pragma unroll
        for (int nb = 0; nb < NB1; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) e1[nb][e] = 0.0f;
        {
            const float *wrow = lw1 + (KH * hh) * P1 + r;
#pragma unroll
            for (int i = 0; i < KH; ++i)
#pragma unroll
                for (int nb = 0; nb < NB1; ++nb) e1[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[i * P1 + 32 * nb], av[i], e1[nb], 0, 0, 0);
        }
#pragma unroll
        for (int nb = 0; nb < NB1; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b4 = *(const float4 *)(lb1 + 32 * nb + 8 * g + 4 * hh);
                e1[nb][4 * g] = fmaxf(e1[nb][4 * g] + b4.x, 0.0f);
                e1[nb][4 * g + 1] = fmaxf(e1[nb][4 * g + 1] + b4.y, 0.0f);
                e1[nb][4 * g + 2] = fmaxf(e1[nb][4 * g + 2] + b4.z, 0.0f);
                e1[nb][4 * g + 3] = fmaxf(e1[nb][4 * g + 3] + b4.w, 0.0f);
            }
        // ---- P1 = Wq1 E1: register (nb, 4 g + q) of the first product is the B operand for channels 32 nb + 8 g + q (+ 4) ----
        f16v p1[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int e = 0; e < 16; ++e) p1[rb][e] = 0.0f;
#pragma unroll
        for (int nb = 0; nb < NB1; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float *wrow = lwq + (32 * nb + 8 * g + 4 * hh + q) * P2 + r;
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) p1[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[32 * rb], e1[nb][4 * g + q], p1[rb], 0, 0, 0);
                }
        if (ro >= 0) {
            float *o = dst + ro + 4 * hh;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = 32 * rb + 8 * g;
                    if (c + 4 * hh < sq_out)
                        *(float4 *)(o + c) = make_float4(p1[rb][4 * g], p1[rb][4 * g + 1], p1[rb][4 * g + 2], p1[rb][4 * g + 3]);
                }
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) a[v] = an[v];
        ro = ro_next;
    }
}

template <int S, int NB1, int RB>
static int launch_expand_sq(hipStream_t s, const float *src, int64_t rows, int sh, int sw, int crop_y, int crop_x, int h, int w, const float *w1,
                            const float *b1, const float *wq, int wq_stride, int sq_out, float *dst, int dH, int dW, int dC, int off_y, int off_x)
{
    constexpr int C1 = 32 * NB1;
    const size_t lds = (size_t)(S * (C1 + 1) + C1 * (32 * RB + 1) + C1) * sizeof(float);
    const int64_t ntiles = (rows + 31) / 32;
    // few row tiles (a FrameQueue window's segments): 4-wave workgroups spread them over more CUs, like the 1 x 1 kernel
    const bool small = ntiles < 8 * 256;
    static unsigned long long mask8 = 0, mask4 = 0;
    if (lds > 160 * 1024 - 256 || rows > (int64_t)0x7fffff00) return SWK_ERR_CAPACITY;
    const FastDiv fhw((unsigned)(h * w)), fw((unsigned)w);
    if (small) {
        if (!ensure_dyn_lds((const void *)k_expand1x1_sq_partial<S, NB1, RB, 4>, 160 * 1024 - 256, mask4)) return SWK_ERR_HIP;
        int64_t blocks = (ntiles + 3) / 4;
        const int64_t cap = 256 * 2;
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL((k_expand1x1_sq_partial<S, NB1, RB, 4>), dim3((unsigned)blocks), dim3(256), lds, s, src, rows, sh, sw, crop_y, crop_x, h, w, w1,
                           b1, wq, wq_stride, sq_out, dst, dH, dW, dC, off_y, off_x, fhw, fw);
    } else {
        if (!ensure_dyn_lds((const void *)k_expand1x1_sq_partial<S, NB1, RB, 8>, 160 * 1024 - 256, mask8)) return SWK_ERR_HIP;
        int64_t blocks = (ntiles + 7) / 8;
        const int64_t per_cu = (160 * 1024 - 256) / (int64_t)lds >= 2 ? 2 : 1;
        const int64_t cap = 256 * per_cu;
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL((k_expand1x1_sq_partial<S, NB1, RB, 8>), dim3((unsigned)blocks), dim3(512), lds, s, src, rows, sh, sw, crop_y, crop_x, h, w, w1,
                           b1, wq, wq_stride, sq_out, dst, dH, dW, dC, off_y, off_x, fhw, fw);
    }
    return hipGetLastError() == hipSuccess ? SWK_OK : SWK_ERR_HIP;
}

}  // namespace swk

#pragma GCC visibility push(default)
extern "C" {

int32_t swk_nhwc_expand1x1_squeeze_partial(void *stream, const float *src, int32_t n, int32_t sh, int32_t sw, int32_t cin, int32_t crop_y,
                                           int32_t crop_x, int32_t h, int32_t w, const float *w1, const float *b1, int32_t c1,
                                           const float *wq, int32_t wq_stride, int32_t sq_out, float *dst, int32_t dH, int32_t dW, int32_t dC,
                                           int32_t off_y, int32_t off_x)
{
    if (!src || !w1 || !b1 || !wq || !dst || n < 1 || h < 1 || w < 1 || sq_out < 4 || sq_out > 64 || (sq_out & 3) || (dC & 3) || sq_out > dC ||
        wq_stride < c1 || (wq_stride & 3) || crop_y < 0 || crop_x < 0 || crop_y + h > sh || crop_x + w > sw || off_y < 0 || off_x < 0 ||
        off_y + h > dH || off_x + w > dW || ((((uintptr_t)src) | ((uintptr_t)w1) | ((uintptr_t)wq) | ((uintptr_t)dst)) & 15))
        return SWK_ERR_ARG;
    using namespace swk;
    const int64_t rows = (int64_t)n * h * w;
    hipStream_t s = (hipStream_t)stream;
#define SWK_ES_ARGS s, src, rows, sh, sw, crop_y, crop_x, h, w, w1, b1, wq, wq_stride, sq_out, dst, dH, dW, dC, off_y, off_x
    // the Fire modules of SqueezeNet-1.0 whose output feeds another Fire: (squeeze, expand1x1) = (16, 64), (32, 128), (48, 192)
    if (cin == 16 && c1 == 64) return sq_out <= 32 ? launch_expand_sq<16, 2, 1>(SWK_ES_ARGS) : launch_expand_sq<16, 2, 2>(SWK_ES_ARGS);
    if (cin == 32 && c1 == 128) return sq_out <= 32 ? launch_expand_sq<32, 4, 1>(SWK_ES_ARGS) : launch_expand_sq<32, 4, 2>(SWK_ES_ARGS);
    if (cin == 48 && c1 == 192) return sq_out <= 32 ? launch_expand_sq<48, 6, 1>(SWK_ES_ARGS) : launch_expand_sq<48, 6, 2>(SWK_ES_ARGS);
#undef SWK_ES_ARGS
    return SWK_ERR_ARG;
}

}  // extern "C"
#pragma GCC visibility pop

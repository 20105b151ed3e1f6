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
// A 1x1 convolution in channels-last layout is a plain GEMM, K = input channels.  v_mfma_f32_32x32x2_f32 (exact float32:
// a k-ordered fmaf chain, 64 cycles per instruction per SIMD, the f32 matrix peak of 157 TFLOP/s) computes D = A B with lane l
// supplying A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31].  Here the WEIGHTS are the A operand (i = output channel)
// and the PIXELS the B operand (j = pixel), so that an accumulator register quad holds four consecutive output channels of
// the lane's own pixel (i = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5), j = l & 31): the epilogue is 4 float4 stores per 32
// channels, addressed by the lane's own pixel, no transposition.
//   * a wave owns 32 pixels.  A lane (pixel r, half h) loads KC / 2 consecutive input channels kb + (KC/2) h .. of its
//     pixel as float4s -- with KC = 32 the two halves fetch one whole 128-byte line of the pixel's channel vector at once, so a
//     line is requested once (with 16-channel chunks it was requested two to four times, spaced by thousands of cycles, and the
//     eight waves' lines do not survive that long in L1).  Step i multiplies channels {kb + i, kb + KC/2 + i}: a permuted k
//     order, matched by reading weight row kb + (KC/2) h + i for half h.
//   * D such chunks are in flight per lane (a register ring), and the ring runs on into the wave's NEXT row tile: neither a
//     tile's first multiply nor its epilogue waits for memory.
//   * the weights (at most 512 x 64 or 64 x 256 floats) sit transposed in LDS, [ci][co] with pitch N + 1: the staging
//     reads W coalesced along ci and writes conflict-free, and an operand read is 32 consecutive floats per half.
// Launched on the CALLER's stream (PyTorch's current stream), like the other glue kernels (cnn_aux.hip).
#include "swk_internal.h"

namespace swk {

typedef float f16v __attribute__((ext_vector_type(16)));

// CS = column splits: CS waves share a row tile, each computing NBLK / CS of its 32-channel column blocks (they load the same
// activations; the accumulators of a wave shrink to 16 NBLK / CS registers).  NWV = waves per workgroup: 16 (one workgroup per
// CU whatever its weight matrix takes of the LDS, four waves per SIMD at <= 128 registers) or 8 (two per SIMD).
template <int NBLK, int KC, int D, int CS, int NWV>
__global__ __launch_bounds__(64 * NWV) void k_conv1x1_relu_place(const float *__restrict__ src, int64_t rows, int sh, int sw, int cin, int crop_y,
                                                            int crop_x, int h, int w, const float *__restrict__ wgt, const float *__restrict__ bias,
                                                            int cout, float *__restrict__ dst, int dH, int dW, int dC, int off_y, int off_x, int c_off,
                                                            FastDiv fhw, FastDiv fw)
{
    constexpr int NP = 32 * NBLK, PITCH = NP + 1, KH = KC / 2, NV = KH / 4, NB = NBLK / CS, NT = 64 * NWV, SLOTS = NWV / CS;
    static_assert(NBLK % CS == 0 && NWV % CS == 0, "column splits");
    extern __shared__ float lds[];                 // weights [cin][PITCH], then the bias padded to NP
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    float *lbias = lds + ((cin * PITCH + 3) & ~3);
    const int hw = h * w;
    const int cs = wave % CS;          // this wave's column blocks: cs * NB .. cs * NB + NB - 1
    const int64_t ntiles = (rows + 31) >> 5, stride = (int64_t)gridDim.x * SLOTS;
    int64_t tile = (int64_t)blockIdx.x * SLOTS + wave / CS;

    // source pointer and destination offset of this lane's pixel in a row tile (rows past the end repeat the last one)
    auto locate = [&](int64_t t, int64_t &ro) -> const float * {
        const unsigned m = (unsigned)t * 32u + (unsigned)r;          // rows < 2^31 (checked by the launcher): invariant-divisor division
        const bool valid = m < (unsigned)rows;
        const unsigned mm = valid ? m : (unsigned)rows - 1u;
        const unsigned b = fhw.div(mm), rem = mm - b * (unsigned)hw;
        const unsigned y = fw.div(rem), x = rem - y * (unsigned)w;
        ro = valid ? (((int64_t)b * dH + off_y + y) * dW + off_x + x) * (int64_t)dC + c_off + 32 * NB * cs : -1;
        return src + (((int64_t)b * sh + crop_y + y) * sw + crop_x + x) * (int64_t)cin + KH * hh;
    };
    float4 a[D][NV];
    int64_t ro = -1, ro_next = -1;
    const float *p = src, *pn = src;
    if (tile < ntiles) {           // the first pieces leave before the weights are staged
        p = locate(tile, ro);
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int v = 0; v < NV; ++v) a[d][v] = *(const float4 *)(p + KC * d + 4 * v);
    }
    // ---- weights, transposed: W[co][ci] read along ci as float4s (16 lanes = 64 channels of one output channel), [ci][co] in LDS.
    //      No index arithmetic beyond shifts: with `i / cin` per element the staging cost a window-sized forward a few microseconds
    //      per workgroup and kernel (cin is a runtime value: a 32-bit division is some forty instructions) ----
    for (int co = tid >> 4; co < NP; co += NT / 16)
        for (int c4 = (tid & 15) * 4; c4 < cin; c4 += 64) {
            const float4 wv = co < cout ? *(const float4 *)(wgt + (int64_t)co * cin + c4) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            float *q = lds + c4 * PITCH + co;
            q[0] = wv.x; q[PITCH] = wv.y; q[2 * PITCH] = wv.z; q[3 * PITCH] = wv.w;
        }
    for (int i = tid; i < NP; i += NT) lbias[i] = i < cout ? bias[i] : 0.0f;
    __syncthreads();

    for (; tile < ntiles; tile += stride) {
        const bool more = tile + stride < ntiles;
        if (more) pn = locate(tile + stride, ro_next);
        f16v acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[nb][e] = 0.0f;
        for (int kb = 0; kb < cin; kb += KC * D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int k0 = kb + KC * d;
                float av[KH];
#pragma unroll
                for (int v = 0; v < NV; ++v) { av[4 * v] = a[d][v].x; av[4 * v + 1] = a[d][v].y; av[4 * v + 2] = a[d][v].z; av[4 * v + 3] = a[d][v].w; }
                const int kn = k0 + KC * D;                      // the chunk that takes this ring slot next
                if (kn < cin) {
#pragma unroll
                    for (int v = 0; v < NV; ++v) a[d][v] = *(const float4 *)(p + kn + 4 * v);
                } else if (more) {
#pragma unroll
                    for (int v = 0; v < NV; ++v) a[d][v] = *(const float4 *)(pn + kn - cin + 4 * v);
                }
                // weight operands of step i + 1 are read from LDS while step i multiplies (two register sets, scheduling
                // fences: left alone the compiler reads each operand right before its MFMA and the wave waits out the LDS latency)
                const float *wrow = lds + (k0 + KH * hh) * PITCH + 32 * NB * cs + r;
                float bw[2][NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) bw[0][nb] = wrow[32 * nb];
#pragma unroll
                for (int i = 0; i < KH; ++i) {
                    if (i < KH - 1) {
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) bw[(i + 1) & 1][nb] = wrow[(i + 1) * PITCH + 32 * nb];
                    }
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(bw[i & 1][nb], av[i], acc[nb], 0, 0, 0);
                    if (i < KH - 1) __builtin_amdgcn_sched_group_barrier(0x100 /* DS read */, NB, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008 /* MFMA */, NB, 0);
                }
            }
        }
        // ---- bias + ReLU + placement: register quads = four consecutive output channels of the lane's pixel ----
        if (ro >= 0) {
            float *o = dst + ro + 4 * hh;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = nb * 32 + 8 * g;          // relative to the wave's first column block
                    if (32 * NB * cs + c + 4 * hh < cout) {
                        const float4 b4 = *(const float4 *)(lbias + 32 * NB * cs + c + 4 * hh);
                        float4 v;
                        v.x = fmaxf(acc[nb][4 * g] + b4.x, 0.0f);
                        v.y = fmaxf(acc[nb][4 * g + 1] + b4.y, 0.0f);
                        v.z = fmaxf(acc[nb][4 * g + 2] + b4.z, 0.0f);
                        v.w = fmaxf(acc[nb][4 * g + 3] + b4.w, 0.0f);
                        *(float4 *)(o + c) = v;
                    }
                }
        }
        p = pn;
        ro = ro_next;
    }
}

template <int NBLK, int KC, int D, int CS, int NWV>
static int launch_conv1x1_d(hipStream_t s, const float *src, int64_t rows, int sh, int sw, int cin, int crop_y, int crop_x, int h, int w,
                            const float *wgt, const float *bias, int cout, float *dst, int dH, int dW, int dC, int off_y, int off_x, int c_off)
{
    const size_t lds = (size_t)(((cin * (32 * NBLK + 1) + 3) & ~3) + 32 * NBLK) * sizeof(float);
    static unsigned long long attr_mask = 0;
    if (!ensure_dyn_lds((const void *)k_conv1x1_relu_place<NBLK, KC, D, CS, NWV>, 160 * 1024 - 256, attr_mask)) return SWK_ERR_HIP;
    if (lds > 160 * 1024 - 256 || rows > (int64_t)0x7fffff00) return SWK_ERR_CAPACITY;
    constexpr int SLOTS = NWV / CS;
    const int64_t ntiles = (rows + 31) / 32;
    int64_t blocks = (ntiles + SLOTS - 1) / SLOTS;
    // persistent over the row tiles: one 16-wave workgroup per CU, or one / two 8-wave ones
    // (as many workgroups per CU as their waves and their LDS allow)
    const int64_t by_lds = (int64_t)((160 * 1024 - 256) / lds), by_waves = 16 / NWV;
    const int64_t per_cu = by_lds < 1 ? 1 : (by_lds < by_waves ? by_lds : by_waves);
    const int64_t cap = 256 * per_cu;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((k_conv1x1_relu_place<NBLK, KC, D, CS, NWV>), dim3((unsigned)blocks), dim3(64 * NWV), lds, s, src, rows, sh, sw, cin, crop_y,
                       crop_x, h, w, wgt, bias, cout, dst, dH, dW, dC, off_y, off_x, c_off, FastDiv((unsigned)(h * w)), FastDiv((unsigned)w));
    return hipGetLastError() == hipSuccess ? SWK_OK : SWK_ERR_HIP;
}

int g_expand_split_bf16 = getenv("SWK_EXPAND_SPLIT_BF16") ? atoi(getenv("SWK_EXPAND_SPLIT_BF16")) : 0;
int g_conv1x1_ring = 0;          // A/B knob: 0 = 16-wave workgroups, column blocks of the wide expands split over two waves; 1 = the
                                 // first layout (8 waves, every wave all column blocks, activation ring as deep as fits)

#define SWK_C1_ARGS s, src, rows, sh, sw, cin, crop_y, crop_x, h, w, wgt, bias, cout, dst, dH, dW, dC, off_y, off_x, c_off
template <int NBLK>
static int launch_conv1x1(hipStream_t s, const float *src, int64_t rows, int sh, int sw, int cin, int crop_y, int crop_x, int h, int w,
                          const float *wgt, const float *bias, int cout, float *dst, int dH, int dW, int dC, int off_y, int off_x, int c_off)
{
    constexpr int CS = NBLK >= 6 ? 2 : 1;          // 64 accumulator registers per wave at most
    if (g_conv1x1_ring == 0) {
        // A workgroup takes NWV / CS row tiles (32 pixels each) per round and there is one workgroup per CU (the weight matrix fills
        // the LDS): a FrameQueue window's worth of segments (a few hundred rows of the batch, 10-40 k pixels) would keep 20-80 of the 256
        // CUs busy with 16-wave workgroups (82 us for 512 -> 64 on 9 x 9 at batch 256, 14 us per 256 rows at batch 4096).  Fewer waves
        // per workgroup spread the same tiles over more CUs.
        const int64_t ntiles = (rows + 31) / 32;
        const int nwv = ntiles >= 16 * 160 / CS ? 16 : ntiles >= 8 * 160 / CS ? 8 : 4;
#define SWK_C1_BY_WAVES(KC)                                                              \
        do {                                                                             \
            if (nwv == 16) return launch_conv1x1_d<NBLK, KC, 1, CS, 16>(SWK_C1_ARGS);    \
            if (nwv == 8) return launch_conv1x1_d<NBLK, KC, 1, CS, 8>(SWK_C1_ARGS);      \
            return launch_conv1x1_d<NBLK, KC, 1, CS, 4>(SWK_C1_ARGS);                    \
        } while (0)
        if (cin % 32 == 0) SWK_C1_BY_WAVES(32);          // whole 128-byte lines per pixel and chunk
        if (cin == 48) SWK_C1_BY_WAVES(48);              // one pixel = one chunk (192 bytes)
        SWK_C1_BY_WAVES(16);
#undef SWK_C1_BY_WAVES
    }
    if (cin % 32 == 0) {
        const int n = cin / 32, dmax = NBLK >= 6 ? 2 : 4;          // ring registers: 16 D; accumulators: 16 NBLK
        const int d = dmax >= 4 && n % 4 == 0 ? 4 : dmax >= 3 && n % 3 == 0 ? 3 : dmax >= 2 && n % 2 == 0 ? 2 : 1;
        switch (d) {
        case 4: if constexpr (NBLK < 6) return launch_conv1x1_d<NBLK, 32, 4, 1, 8>(SWK_C1_ARGS);
        case 3: if constexpr (NBLK < 6) return launch_conv1x1_d<NBLK, 32, 3, 1, 8>(SWK_C1_ARGS);
        case 2: return launch_conv1x1_d<NBLK, 32, 2, 1, 8>(SWK_C1_ARGS);
        default: return launch_conv1x1_d<NBLK, 32, 1, 1, 8>(SWK_C1_ARGS);
        }
    }
    if (cin == 48) return launch_conv1x1_d<NBLK, 48, 1, 1, 8>(SWK_C1_ARGS);
    return launch_conv1x1_d<NBLK, 16, 1, 1, 8>(SWK_C1_ARGS);
}
#undef SWK_C1_ARGS

}  // namespace swk

#pragma GCC visibility push(default)
extern "C" {

int32_t swk_nhwc_conv1x1_bias_relu_place(void *stream, const float *src, int32_t n, int32_t sh, int32_t sw, int32_t cin, int32_t crop_y,
                                         int32_t crop_x, int32_t h, int32_t w, const float *weight, const float *bias, int32_t cout,
                                         float *dst, int32_t dH, int32_t dW, int32_t dC, int32_t off_y, int32_t off_x, int32_t c_off)
{
    if (!src || !weight || !bias || !dst || n < 1 || h < 1 || w < 1 || cin < 16 || (cin & 15) || cin > 1024 || cout < 4 || cout > 256 ||
        (cout & 3) || (dC & 3) || (c_off & 3) || (((uintptr_t)dst) & 15) ||
        crop_y < 0 || crop_x < 0 || crop_y + h > sh || crop_x + w > sw || off_y < 0 || off_x < 0 || off_y + h > dH || off_x + w > dW ||
        c_off < 0 || c_off + cout > dC || ((((uintptr_t)src) | ((uintptr_t)weight)) & 15))
        return SWK_ERR_ARG;
    using namespace swk;
    const int64_t rows = (int64_t)n * h * w;
    hipStream_t s = (hipStream_t)stream;
    if (g_expand_split_bf16 && cout == 4 * cin) {          // the Fire modules' expand1x1 shapes: float32 products from split bf16 operands
        const int rc = launch_expand1x1_split_bf16(s, src, rows, sh, sw, cin, crop_y, crop_x, h, w, weight, bias, cout, dst, dH, dW, dC, off_y, off_x,
                                                   c_off);
        if (rc != SWK_ERR_ARG) return rc;
    }
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

// Measurement knobs of the classifier kernels (A/B runs).  knob 0: activation ring of the 1x1 kernel (0 = deepest that fits, 1 = one chunk
// in flight; results do not depend on it).  knob 1: 1 = expand1x1 shapes on the split-bf16 kernel (cnn_expand_bf16.hip: float32-accurate,
// another summation order than the float32 kernel's, so scores move in the last bits); 0, the default: every 1x1 on the float32 kernel.
int32_t swk_set_cnn_tuning(int32_t knob, int32_t value)
{
    if (knob == 0 && (value == 0 || value == 1)) { swk::g_conv1x1_ring = value; return SWK_OK; }
    if (knob == 1 && (value == 0 || value == 1)) { swk::g_expand_split_bf16 = value; return SWK_OK; }
    return SWK_ERR_ARG;
}

}  // extern "C"
#pragma GCC visibility pop

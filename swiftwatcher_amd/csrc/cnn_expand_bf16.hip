// expand1x1 of the Fire modules (16 -> 64, 32 -> 128, 48 -> 192, 64 -> 256; segment_classification.py of the reference :14-67 via torchvision's
// SqueezeNet-1.0) with every float32 product formed as a SUM OF bfloat16 PRODUCTS OF SPLIT OPERANDS:
//
//     x = x1 + x2 + x3,  x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)          (the remainders are exact in float32: 3 x 8 = 24 bits)
//     w x  ~  w1 x3 + w2 x2 + w3 x1 + w1 x2 + w2 x1 + w1 x1                                  (the three terms below 2^-24 |w x| are dropped)
//
// A bf16 x bf16 product is exact in float32 and v_mfma_f32_32x32x16_bf16 accumulates in float32: the sum carries the error of a float32
// multiply-add chain (measured against float64: 2-4e-7 of the output scale, the chain of k_conv1x1_relu_place 6e-7; tools/r4/
// bf16_split_accuracy.py, test_split_bf16_expand_kernel_is_float32_accurate) -- at six MFMAs of 32 cycles per 16 channels and 32 x 32
// outputs instead of eight of 64.  Why here: in exact float32 the matrix pipe gives 157 TFLOP/s against 6 TB/s of memory, a ridge at 26
// flop per byte, and these layers write four bytes per 8-32 flop of their own: the float32 kernel holds them at 2.8-3.7 TB/s with the pipe
// half busy; the split products take the pipe out of the way.  The split itself is vector work (47 instructions per
// eight activations); the weights are split once per workgroup while they are staged.
//
// STATUS (round 4): built, float32-accurate on the hardware (the test above), and OFF by default (swk_set_cnn_tuning knob 1, or
// SWK_EXPAND_SPLIT_BF16=1).  Measured per 4,096 rows (profiles/r4_cnn_split_bf16_expand.txt): 16 -> 64 33 us against 38, 32 -> 128 80 against
// 98, 48 -> 192 102 / 125 against 119 / 147 -- and 64 -> 256 430 / 187 against 349 / 143: its split weights take 96 KB of LDS, one 8-wave
// workgroup per CU at 168 registers, too few waves to cover the stores.  Net 0.8 % of a forward with 64 -> 256 left on the float32 kernel:
// not worth a second arithmetic on the default path; kept as the first, measured kernel of the route DESIGN section 10 describes.
//
// Layout as in k_conv1x1_relu_place: weights = A operand (i = output channel), pixels = B operand (j = pixel), so that an accumulator
// register quad holds four consecutive output channels of the lane's own pixel.  v_mfma_f32_32x32x16_bf16: lane l supplies
// A[i = l & 31][k = 8 (l >> 5) .. + 7] and B[k = 8 (l >> 5) .. + 7][j = l & 31]: a lane (pixel r, half h) loads the eight consecutive
// channels 16 s + 8 h .. + 7 of its pixel for k-step s (two float4s), all steps of a tile at once (K <= 64: at most eight float4s), and
// the next tile's while this one multiplies.  LDS: the split weights [step][column block][part][half][output channel][8] -- an operand
// read is one ds_read_b128 per lane, conflict-free.  Launched on the CALLER's stream.
#include "swk_internal.h"

namespace swk {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(const float4 lo, const float4 hi, bf16x8 &p1, bf16x8 &p2, bf16x8 &p3)
{
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 a = (__bf16)v[j];
        const float r = v[j] - (float)a;
        const __bf16 b = (__bf16)r;
        const float r2 = r - (float)b;
        p1[j] = a; p2[j] = b; p3[j] = (__bf16)r2;
    }
}

// NBLK column blocks of 32 output channels; KS = cin / 16 k-steps; CS waves share a row tile (each NBLK / CS column blocks); NWV waves.
template <int NBLK, int KS, int CS, int NWV>
__global__ __launch_bounds__(64 * NWV) void k_expand1x1_bf16s(const float *__restrict__ src, int64_t rows, int sh, int sw, int crop_y, int crop_x,
                                                         int h, int w, const float *__restrict__ wgt, const float *__restrict__ bias, int cout,
                                                         float *__restrict__ dst, int dH, int dW, int dC, int off_y, int off_x, int c_off,
                                                         FastDiv fhw, FastDiv fw)
{
    constexpr int CIN = 16 * KS, NP = 32 * NBLK, NB = NBLK / CS, NT = 64 * NWV, SLOTS = NWV / CS;
    static_assert(NBLK % CS == 0 && NWV % CS == 0, "column splits");
    extern __shared__ uint4 lds_e[];               // split weights: [KS][NBLK][3][2][32] x 16 bytes, then the bias padded to NP
    float *lbias = (float *)(lds_e + KS * NBLK * 3 * 64);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int hw = h * w, cs = wave % CS;
    const int64_t ntiles = (rows + 31) >> 5, stride = (int64_t)gridDim.x * SLOTS;
    int64_t tile = (int64_t)blockIdx.x * SLOTS + wave / CS;

    auto locate = [&](int64_t t, int64_t &ro) -> const float * {
        const unsigned m = (unsigned)t * 32u + (unsigned)r;
        const bool valid = m < (unsigned)rows;
        const unsigned mm = valid ? m : (unsigned)rows - 1u;
        const unsigned b = fhw.div(mm), rem = mm - b * (unsigned)hw;
        const unsigned y = fw.div(rem), x = rem - y * (unsigned)w;
        ro = valid ? (((int64_t)b * dH + off_y + y) * dW + off_x + x) * (int64_t)dC + c_off + 32 * NB * cs : -1;
        return src + (((int64_t)b * sh + crop_y + y) * sw + crop_x + x) * (int64_t)CIN + 8 * hh;
    };
    float4 raw[KS][2];
    int64_t ro = -1, ro_next = -1;
    const float *p = src;
    if (tile < ntiles) {           // the first tile's pixels leave before the weights are staged
        p = locate(tile, ro);
#pragma unroll
        for (int s = 0; s < KS; ++s) { raw[s][0] = *(const float4 *)(p + 16 * s); raw[s][1] = *(const float4 *)(p + 16 * s + 4); }
    }
    // ---- weights: item = (k-step, column block, half, output channel): eight floats of W[co][16 s + 8 half ..], split, three 16-byte stores ----
    for (int i = tid; i < KS * NBLK * 64; i += NT) {
        const int m = i & 31, kg = (i >> 5) & 1, nb = (i >> 6) % NBLK, s = (i >> 6) / NBLK, co = 32 * nb + m;
        float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
        if (co < cout) {
            const float *q = wgt + (int64_t)co * CIN + 16 * s + 8 * kg;
            lo = *(const float4 *)q; hi = *(const float4 *)(q + 4);
        }
        bf16x8 w1, w2, w3;
        split3(lo, hi, w1, w2, w3);
        uint4 *d = lds_e + ((s * NBLK + nb) * 3) * 64 + kg * 32 + m;
        d[0] = __builtin_bit_cast(uint4, w1); d[64] = __builtin_bit_cast(uint4, w2); d[128] = __builtin_bit_cast(uint4, w3);
    }
    for (int i = tid; i < NP; i += NT) lbias[i] = i < cout ? bias[i] : 0.0f;
    __syncthreads();

    for (; tile < ntiles; tile += stride) {
        const bool more = tile + stride < ntiles;
        if (more) p = locate(tile + stride, ro_next);
        f16v acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[nb][e] = 0.0f;
        // operand reads run one (k-step, column block) ahead of the products (two register sets, scheduling fences: left alone the
        // compiler reads every operand of the tile first -- 190 registers -- or each right before its MFMAs)
        bf16x8 wq[2][3];
        auto read_w = [&](int s, int nb, int set) {
            const uint4 *a = lds_e + ((s * NBLK + NB * cs + nb) * 3) * 64 + lane;
            wq[set][0] = __builtin_bit_cast(bf16x8, a[0]); wq[set][1] = __builtin_bit_cast(bf16x8, a[64]); wq[set][2] = __builtin_bit_cast(bf16x8, a[128]);
        };
        read_w(0, 0, 0);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            // this k-step's activations, split; the registers they came in take the next tile's
            bf16x8 x1, x2, x3;
            split3(raw[s][0], raw[s][1], x1, x2, x3);
            if (more) { raw[s][0] = *(const float4 *)(p + 16 * s); raw[s][1] = *(const float4 *)(p + 16 * s + 4); }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int i = s * NB + nb, set = i & 1;
                if (i + 1 < KS * NB) read_w((i + 1) / NB, (i + 1) % NB, set ^ 1);
                // smallest terms first
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[set][0], x3, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[set][1], x2, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[set][2], x1, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[set][0], x2, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[set][1], x1, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[set][0], x1, acc[nb], 0, 0, 0);
                if (i + 1 < KS * NB) __builtin_amdgcn_sched_group_barrier(0x100 /* DS read */, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008 /* MFMA */, 6, 0);
            }
        }
        // ---- bias + ReLU + placement: register quads = four consecutive output channels of the lane's pixel ----
        if (ro >= 0) {
            float *o = dst + ro + 4 * hh;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = nb * 32 + 8 * g;
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
        ro = ro_next;
    }
}

template <int NBLK, int KS, int CS, int NWV>
static int launch_expand_bf16s(hipStream_t s, const float *src, int64_t rows, int sh, int sw, int crop_y, int crop_x, int h, int w, const float *wgt,
                               const float *bias, int cout, float *dst, int dH, int dW, int dC, int off_y, int off_x, int c_off)
{
    const size_t lds = (size_t)KS * NBLK * 3 * 64 * 16 + 32 * NBLK * sizeof(float);
    static unsigned long long attr_mask = 0;
    if (!ensure_dyn_lds((const void *)k_expand1x1_bf16s<NBLK, KS, CS, NWV>, 160 * 1024 - 256, attr_mask)) return SWK_ERR_HIP;
    if (lds > 160 * 1024 - 256 || rows > (int64_t)0x7fffff00) return SWK_ERR_CAPACITY;
    constexpr int SLOTS = NWV / CS;
    const int64_t ntiles = (rows + 31) / 32;
    int64_t blocks = (ntiles + SLOTS - 1) / SLOTS;
    const int64_t by_lds = (int64_t)((160 * 1024 - 256) / lds), by_waves = 16 / NWV;
    const int64_t per_cu = by_lds < 1 ? 1 : (by_lds < by_waves ? by_lds : by_waves);
    if (blocks > 256 * per_cu) blocks = 256 * per_cu;
    hipLaunchKernelGGL((k_expand1x1_bf16s<NBLK, KS, CS, NWV>), dim3((unsigned)blocks), dim3(64 * NWV), lds, s, src, rows, sh, sw, crop_y, crop_x, h, w,
                       wgt, bias, cout, dst, dH, dW, dC, off_y, off_x, c_off, FastDiv((unsigned)(h * w)), FastDiv((unsigned)w));
    return hipGetLastError() == hipSuccess ? SWK_OK : SWK_ERR_HIP;
}

// the Fire shapes (cout = 4 cin): SWK_ERR_ARG for anything else (the caller then takes the float32 kernel)
int launch_expand1x1_split_bf16(hipStream_t s, const float *src, int64_t rows, int sh, int sw, int cin, int crop_y, int crop_x, int h, int w,
                                const float *wgt, const float *bias, int cout, float *dst, int dH, int dW, int dC, int off_y, int off_x, int c_off)
{
#define SWK_EB_ARGS s, src, rows, sh, sw, crop_y, crop_x, h, w, wgt, bias, cout, dst, dH, dW, dC, off_y, off_x, c_off
    if (cin == 16 && cout == 64) return launch_expand_bf16s<2, 1, 1, 8>(SWK_EB_ARGS);
    if (cin == 32 && cout == 128) return launch_expand_bf16s<4, 2, 1, 8>(SWK_EB_ARGS);
    if (cin == 48 && cout == 192) return launch_expand_bf16s<6, 3, 2, 8>(SWK_EB_ARGS);
    if (cin == 64 && cout == 256) return launch_expand_bf16s<8, 4, 2, 8>(SWK_EB_ARGS);
#undef SWK_EB_ARGS
    return SWK_ERR_ARG;
}

}  // namespace swk

// IALM streaming pass on the f64 matrix cores (v_mfma_f64_16x16x4_f64), A/Y-state formulation.
//
// One wave owns a tile of 16 pixels x all n frames and keeps ONE register layout for every
// element-wise step and for both matrix products:
//
//   layout L:  lane l, register t  <->  pixel p0 + (l & 15),  frame 4t + (l >> 4)
//
// * A-update, transposed:  A_new^T (frames x pixels) = B^T (frames x frames) * M^T (frames x pixels).
//   For k-step t the MFMA B operand B_op[k = l>>4][j = l&15] = m(pixel j, frame 4t+k) IS register t
//   of layout L, and the result tile for out-frame block b, D[row = (l>>4)+4r][col = l&15], is
//   register t = 4b + r of layout L again.  So X/A/Y are loaded once, in L, and A/Y/S are stored
//   from L: per frame 16 consecutive pixels = one 128-byte line of the f64 planes.
//   The A operand, B[4t + (l>>4)][16b + (l&15)], comes from an LDS copy of the n x n matrix.
// * Gram matrix of the NEXT iteration's M: G += M'^T M' sums over pixels, so it needs
//   lane <-> (pixel 4g + (l>>4), frame 16f + (l&15)).  That is one transpose through a private
//   per-wave LDS tile ([frame][pixel], pitch 17 doubles: conflict-free both ways); the transposed
//   register serves as BOTH MFMA operands.  Only frame-block pairs ib <= jb are accumulated
//   (G is symmetric); accumulators stay in registers over the wave's whole tile loop.
//
// No barrier inside the loop: waves only share the read-only B matrix.
// This file: the A/Y-state pass (k_ialm_pass_v2: 33 B/element/iteration, +1 for the sparse image) -- what runs when the caller asks
// for A or E, and what reruns a window the M-state pass (ialm_mstate.hip, 21 B) could not decide; and k_select_sparse of the latter.
// 4n^2 flop/pixel -> 7.8 flop/B at n = 64 on 21 B, under the f64 ridge of ~9.8 flop/B (78.6 TF / 8 TB/s); at n = 21 it is 2.5 flop/B.
#include "swk_internal.h"

namespace swk {

typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double shrink2(double raw, double thr)
{
    return fmax(raw - thr, 0.0) + fmin(raw + thr, 0.0);          // image_filtering.py:283
}

__device__ __forceinline__ uint8_t sparse_u8b(double e)
{
    double v = -e;                                                // :244
    v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);                 // :245
    return (uint8_t)v;
}

template <int NB>
struct V2Cfg {
    static constexpr int NPAD = 16 * NB;                          // frames padded to MFMA blocks
    static constexpr int NK = 4 * NB;                             // k-steps of 4 frames = registers per lane
    static constexpr int BP = (NPAD % 32 == 0) ? NPAD + 16 : NPAD;   // LDS pitch of B: rows 32 banks apart
    static constexpr int TP = 17;                                 // LDS pitch of the transpose tile
    static constexpr int NPAIR = NB * (NB + 1) / 2;
    // + 256 doubles: table of x / dual_norm for the 256 possible u8 values (first two passes)
    static constexpr size_t lds_bytes = (size_t)(NPAD * BP + 4 * NPAD * TP + 256) * sizeof(double);
};

typedef int v2i __attribute__((ext_vector_type(2)));
constexpr unsigned kOob = 0x80000000u;        // a byte offset past every buffer: loads give 0, stores are dropped

__device__ __forceinline__ double buf_ld64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_st64(double v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, v), r, voff, soff, 0);
}
__device__ __forceinline__ int buf_ld8(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    return (int)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(r, voff, soff, 0);
}
__device__ __forceinline__ void buf_st8(int v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)v, r, voff, soff, 0);
}

// Two waves per SIMD (<= 256 registers): while one wave is in a matrix phase the other runs its
// element-wise phase or waits for memory.  Every global access is a buffer instruction: the per-lane
// byte offset is computed once per tile, the per-frame-row step is a wave-uniform SGPR offset, and the
// hardware range check replaces all predication -- lanes past the last pixel (and X/S rows past the last
// frame) get an out-of-range offset, so their loads read 0 and their stores vanish; zeros flow through
// the arithmetic as zeros (A = Y = E = M = 0), which is exactly what padding must contribute.
// FULL: n == 16 NB, no padded frame rows (saves the per-row offset selects and their registers)
template <int NB, int MODE, bool WRITE_E, bool FULL>
__global__ __launch_bounds__(256, 2) void k_ialm_pass_v2(IalmBuffers b)
{
    using C = V2Cfg<NB>;
    constexpr int NPAD = C::NPAD, NK = C::NK, BP = C::BP, TP = C::TP;
    extern __shared__ double lds[];
    double *sB = lds;                                             // [NPAD][BP]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double *sT = lds + NPAD * BP + wave * (NPAD * TP);           // this wave's [NPAD][TP]
    const int w = blockIdx.y;
    const IalmWin &st = b.win[w];
    if (st.done) return;
    if (MODE == 0 && st.int_gram) return;        // the start pass's only product already came from k_gram_u8 (exact X^T X)
    const int n = b.n, P = b.P;
    const unsigned ps32 = (unsigned)b.pstride, P32 = (unsigned)P;
    const double inv_mu = st.cur.inv_mu, thr = st.cur.thr, mu = st.cur.mu;
    const double inv_mu2 = st.nxt.inv_mu, thr2 = st.nxt.thr;
    const double dual = st.dual_norm;
    // A/Y/E hold b.fpad (n rounded up to 16) zero-initialised planes per window, X/S exactly n
    const int fbytes = b.fpad * (int)ps32 * 8;
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void *)(b.X + (int64_t)w * n * P), 0, n * P, 0x00020000);
    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void *)(b.S + (int64_t)w * n * P), 0, n * P, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void *)(b.A + (int64_t)w * b.fpad * b.pstride), 0, fbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc((void *)(b.Y + (int64_t)w * b.fpad * b.pstride), 0, fbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rE = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(WRITE_E ? b.E + (int64_t)w * b.fpad * b.pstride : b.A), 0, WRITE_E ? fbytes : 0, 0x00020000);

    // Y0 = X / dual_norm (:272) takes one of 256 values: divide once per value, exactly, and look it up
    double *sY0 = lds + NPAD * BP + 4 * NPAD * TP;
    if (MODE != 2) sY0[tid] = (double)tid / dual;
    if (MODE != 0) {
        const double *Bm = b.Bm + (int64_t)w * n * n;
        for (int i = tid; i < NPAD * NPAD; i += 256) {
            const int k = i / NPAD, c = i % NPAD;
            sB[k * BP + c] = (k < n && c < n) ? Bm[k * n + c] : 0.0;
        }
    }
    __syncthreads();

    const int pl = lane & 15, fr0 = lane >> 4;
    const int flim = n - fr0;                    // frame 4t + fr0 exists  <=>  4t < flim
    d4 G[C::NPAIR];
#pragma unroll
    for (int i = 0; i < C::NPAIR; ++i) G[i] = d4{0.0, 0.0, 0.0, 0.0};
    double zz = 0.0;

    const int ntiles = (P + 15) >> 4;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const unsigned p = (unsigned)(tile * 16 + pl);
        const bool pvalid = p < P32;
        const unsigned vo8 = pvalid ? ((unsigned)fr0 * ps32 + p) * 8u : kOob;     // f64 planes
        const unsigned vo1 = pvalid ? (unsigned)fr0 * P32 + p : kOob;             // u8 planes
        int xi[NK];
        double yv[NK], mv[NK];
        // ---- loads (all in flight together), then the finish-iteration element-wise part ----
        double av[NK];
#pragma unroll
        for (int t = 0; t < NK; ++t) {
            // frame rows past n (n not a multiple of 16) get the out-of-range offset too: they read 0 and
            // are never stored, so padding costs no HBM traffic
            const bool fvalid = FULL || 4 * t < flim;
            xi[t] = buf_ld8(rX, fvalid ? vo1 : kOob, (unsigned)(4 * t) * P32);
            if (MODE == 2) {
                av[t] = buf_ld64(rA, fvalid ? vo8 : kOob, (unsigned)(4 * t) * ps32 * 8u);
                yv[t] = buf_ld64(rY, fvalid ? vo8 : kOob, (unsigned)(4 * t) * ps32 * 8u);
            }
        }
        if (MODE != 0) {
#pragma unroll
            for (int t = 0; t < NK; ++t) {
                const double x = (double)xi[t];
                double raw;
                if (MODE == 2) {
                    raw = (x - av[t]) + inv_mu * yv[t];                        // :282
                } else {
                    yv[t] = sY0[xi[t]];                                        // x / dual, :272 (A = 0, :273)
                    raw = x + inv_mu * yv[t];
                }
                const double e = shrink2(raw, thr);                            // :283
                mv[t] = (x - e) + inv_mu * yv[t];                              // :284 (SVD input)
                sT[(4 * t + fr0) * TP + pl] = e;        // parked in the cell M' of the same element will take
            }
        }
        // ---- A_new^T = B^T M^T on the matrix cores, two out-frame blocks at a time (two independent
        //      accumulator chains), then Z, Y, stores and the start of the next iteration ----
#pragma unroll
        for (int bq0 = 0; bq0 < NB; bq0 += 2) {
            d4 acc[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
            if (MODE != 0) {
#pragma unroll
                for (int t = 0; t < NK; ++t) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        if (bq0 + h < NB) {
                            const double bop = sB[(4 * t + fr0) * BP + 16 * (bq0 + h) + pl];
                            acc[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(bop, mv[t], acc[h], 0, 0, 0);
                        }
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (bq0 + h >= NB) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = 4 * (bq0 + h) + r;
                    const double x = (double)xi[t];
                    double a_new, y;
                    if (MODE == 0) {
                        a_new = 0.0;
                        y = sY0[xi[t]];
                    } else {
                        const double e = sT[(4 * t + fr0) * TP + pl];
                        a_new = acc[h][r];                                             // :290
                        const double z = (x - a_new) - e;                              // :293
                        y = yv[t] + mu * z;                                            // :294
                        zz += z * z;
                        const unsigned so8 = (unsigned)(4 * t) * ps32 * 8u;
                        const bool fvalid = FULL || 4 * t < flim;
                        const unsigned vo8t = fvalid ? vo8 : kOob;
                        buf_st64(a_new, rA, vo8t, so8);
                        buf_st64(y, rY, vo8t, so8);
                        buf_st8((int)sparse_u8b(e), rS, fvalid ? vo1 : kOob, (unsigned)(4 * t) * P32);
                        if (WRITE_E) buf_st64(e, rE, vo8t, so8);
                    }
                    const double raw2 = (x - a_new) + inv_mu2 * y;
                    const double e2 = shrink2(raw2, thr2);
                    sT[(4 * t + fr0) * TP + pl] = (x - e2) + inv_mu2 * y;
                }
            }
        }
        // ---- Gram of M': transposed registers are both MFMA operands ----
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            double tr[NB];
#pragma unroll
            for (int fb = 0; fb < NB; ++fb) tr[fb] = sT[(16 * fb + pl) * TP + 4 * g + fr0];
            int pair = 0;
#pragma unroll
            for (int ib = 0; ib < NB; ++ib)
#pragma unroll
                for (int jb = ib; jb < NB; ++jb) {
                    G[pair] = __builtin_amdgcn_mfma_f64_16x16x4f64(tr[ib], tr[jb], G[pair], 0, 0, 0);
                    ++pair;
                }
        }
    }

    // ---- block-level, fixed-order combination of the four waves' Gram accumulators ----
    __syncthreads();
    double *sG = lds;                           // reuse: [NPAD][NPAD] needs NPAD*NPAD <= NPAD*BP + 4*NPAD*TP
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
            int pair = 0;
#pragma unroll
            for (int ib = 0; ib < NB; ++ib)
#pragma unroll
                for (int jb = ib; jb < NB; ++jb) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * ib + fr0 + 4 * r, j = 16 * jb + pl;
                        if (wv == 0) sG[i * NPAD + j] = G[pair][r];
                        else sG[i * NPAD + j] += G[pair][r];
                    }
                    ++pair;
                }
        }
        __syncthreads();
    }
    double *gp = b.gpart + ((int64_t)w * b.nblk + blockIdx.x) * n * n;
    for (int idx = tid; idx < n * n; idx += 256) {
        const int i = idx / n, j = idx % n;
        if ((i >> 4) <= (j >> 4)) gp[idx] = sG[i * NPAD + j];
    }
    if (MODE != 0) {
        for (int off = 32; off; off >>= 1) zz += __shfl_down(zz, off);
        __syncthreads();
        if (lane == 0) lds[NPAD * NPAD + wave] = zz;
        __syncthreads();
        if (tid == 0)
            b.zzpart[(int64_t)w * b.nblk + blockIdx.x] =
                ((lds[NPAD * NPAD] + lds[NPAD * NPAD + 1]) + lds[NPAD * NPAD + 2]) + lds[NPAD * NPAD + 3];
    }
}

// M-state pass: the last iteration's sparse image sits in S[(iter-1) & 1]: bring the odd ones to S[0]
__global__ __launch_bounds__(256) void k_select_sparse(IalmBuffers b)
{
    const int w = blockIdx.y;
    const int it = b.win[w].iter;
    if (it < 1 || (((it - 1) & 1) == 0)) return;
    const int64_t total = (int64_t)b.n * b.P;
    const uint8_t *src = b.Salt + (int64_t)w * total;
    uint8_t *dst = b.S + (int64_t)w * total;
    const int64_t stride = (int64_t)gridDim.x * 256;
    if ((total & 15) == 0 && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0) {
        const int64_t nv = total >> 4;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += stride)
            ((uint4 *)dst)[i] = ((const uint4 *)src)[i];
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) dst[i] = src[i];
    }
}

void launch_select_sparse(hipStream_t s, const IalmBuffers &b)
{
    const int64_t total = (int64_t)b.n * b.P;
    int bx = (int)((total / 16 + 255) / 256);
    if (bx > 64) bx = 64;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(k_select_sparse, dim3(bx, b.nwin), dim3(256), 0, s, b);
}

bool ialm_v2_supported(int n) { return n >= 1 && n <= kMaxN; }

template <int NB, int MODE, bool WE, bool FULL>
static void launch_full(hipStream_t s, const IalmBuffers &b)
{
    static unsigned long long attr_mask = 0;
    if (!ensure_dyn_lds((const void *)k_ialm_pass_v2<NB, MODE, WE, FULL>, V2Cfg<NB>::lds_bytes, attr_mask)) return;
    hipLaunchKernelGGL((k_ialm_pass_v2<NB, MODE, WE, FULL>), dim3(b.nblk, b.nwin), dim3(256), V2Cfg<NB>::lds_bytes, s, b);
    note_launch();
}

template <int NB, int MODE, bool WE>
static void launch_one(hipStream_t s, const IalmBuffers &b)
{
    if (b.n == 16 * NB) launch_full<NB, MODE, WE, true>(s, b);
    else launch_full<NB, MODE, WE, false>(s, b);
}

template <int NB>
static void launch_nb(hipStream_t s, const IalmBuffers &b, int mode)
{
    const bool we = b.E != nullptr;
    if (mode == 0) launch_one<NB, 0, false>(s, b);
    else if (mode == 1) { if (we) launch_one<NB, 1, true>(s, b); else launch_one<NB, 1, false>(s, b); }
    else { if (we) launch_one<NB, 2, true>(s, b); else launch_one<NB, 2, false>(s, b); }
}

void launch_ialm_pass_v2(hipStream_t s, const IalmBuffers &b, int mode)
{
    const int nb = (b.n + 15) / 16;
    switch (nb) {
    case 1: launch_nb<1>(s, b, mode); break;
    case 2: launch_nb<2>(s, b, mode); break;
    case 3: launch_nb<3>(s, b, mode); break;
    default: launch_nb<4>(s, b, mode); break;
    }
}

}  // namespace swk

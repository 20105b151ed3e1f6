// IALM streaming pass on the f64 matrix cores (v_mfma_f64_16x16x4_f64) -- the hot kernel.
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
// Roofline: HBM.  33 B/element/iteration against 4n^2 flop/pixel -> 7.8 flop/B at n = 64, under the
// f64 ridge of ~9.8 flop/B (78.6 TF / 8 TB/s); at n = 21 it is 2.5 flop/B.
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
    static constexpr size_t lds_bytes = (size_t)(NPAD * BP + 4 * NPAD * TP) * sizeof(double);
};

template <int NB, int MODE, bool WRITE_E>
__global__ __launch_bounds__(256, 1) void k_ialm_pass_v2(IalmBuffers b)
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
    const int n = b.n, P = b.P;
    const int64_t ps = b.pstride;
    const double inv_mu = st.cur.inv_mu, thr = st.cur.thr, mu = st.cur.mu;
    const double inv_mu2 = st.nxt.inv_mu, thr2 = st.nxt.thr;
    const double dual = st.dual_norm;
    // A/Y/E hold NPAD-aligned frame counts per window (b.fpad planes): rows f >= n are never stored
    // and whatever is loaded from them is discarded, so the f64 streams need no per-lane clamping and
    // every address is  window base (SGPR) + 32-bit lane offset + wave-uniform row step.
    const uint8_t *X = b.X + (int64_t)w * n * P;
    uint8_t *S = b.S + (int64_t)w * n * P;
    double *A = b.A + (int64_t)w * b.fpad * ps, *Y = b.Y + (int64_t)w * b.fpad * ps;
    double *Eo = WRITE_E ? b.E + (int64_t)w * b.fpad * ps : nullptr;
    const unsigned ps32 = (unsigned)ps, P32 = (unsigned)P;

    if (MODE != 0) {
        const double *Bm = b.Bm + (int64_t)w * n * n;
        for (int i = tid; i < NPAD * NPAD; i += 256) {
            const int k = i / NPAD, c = i % NPAD;
            sB[k * BP + c] = (k < n && c < n) ? Bm[k * n + c] : 0.0;
        }
    }
    __syncthreads();

    const int pl = lane & 15, fr0 = lane >> 4;
    d4 G[C::NPAIR];
#pragma unroll
    for (int i = 0; i < C::NPAIR; ++i) G[i] = d4{0.0, 0.0, 0.0, 0.0};
    double zz = 0.0;

    // One wave per SIMD (the kernel needs > 256 registers at n = 64), so HBM latency is hidden
    // inside the wave: the raw X/A/Y values of the wave's NEXT tile are loaded into a second
    // register set before the matrix work of the current tile starts.
    const int ntiles = (P + 15) >> 4;
    const int tstride = gridDim.x * 4;
    int tile = blockIdx.x * 4 + wave;
    int xn[NK];
    double an[NK], yn[NK];
    auto load_raw = [&](int tl) {
        const int p_ = tl * 16 + pl;
        const unsigned pc_ = (unsigned)(p_ < P ? p_ : P - 1);
        const unsigned o64 = (unsigned)fr0 * ps32 + pc_;
#pragma unroll
        for (int t = 0; t < NK; ++t) {
            const int f = 4 * t + fr0;
            const unsigned fc = (unsigned)(f < n ? f : n - 1);
            xn[t] = X[fc * P32 + pc_];
            if (MODE == 2) {
                an[t] = A[o64 + (unsigned)(4 * t) * ps32];
                yn[t] = Y[o64 + (unsigned)(4 * t) * ps32];
            }
        }
    };
    if (tile < ntiles) load_raw(tile);
    for (; tile < ntiles; tile += tstride) {
        const int p = tile * 16 + pl;
        const bool pvalid = p < P;
        int xi[NK];
        double yv[NK], ev[NK], mv[NK];
        // ---- finish-iteration element-wise part on the tile loaded one trip ago ----
#pragma unroll
        for (int t = 0; t < NK; ++t) {
            const int f = 4 * t + fr0;
            xi[t] = xn[t];
            const double x = (double)xi[t];
            if (MODE == 2) {
                yv[t] = yn[t];
                const double raw = (x - an[t]) + inv_mu * yv[t];              // :282
                ev[t] = shrink2(raw, thr);
                mv[t] = (f < n) ? (x - ev[t]) + inv_mu * yv[t] : 0.0;          // :284
            } else if (MODE == 1) {
                yv[t] = x / dual;                                              // :272 (A = 0, :273)
                const double raw = x + inv_mu * yv[t];
                ev[t] = shrink2(raw, thr);
                mv[t] = (f < n) ? (x - ev[t]) + inv_mu * yv[t] : 0.0;
            }
        }
        // ---- prefetch the next tile (clamped: the last trip re-reads a valid tile) ----
        {
            const int nt = tile + tstride;
            load_raw(nt < ntiles ? nt : tile);
        }
        // ---- A_new^T = B^T M^T on the matrix cores, two out-frame blocks at a time (two independent
        //      accumulator chains), then Z, Y, stores and the start of the next iteration ----
#pragma unroll
        for (int bq0 = 0; bq0 < NB; bq0 += 2) {
            constexpr int kZero = 0;
            d4 acc[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
            (void)kZero;
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
                    const int f = 4 * t + fr0;
                    const bool ok = pvalid && f < n;
                    const double x = (double)xi[t];
                    double a_new, y;
                    if (MODE == 0) {
                        a_new = 0.0;
                        y = x / dual;
                    } else {
                        a_new = acc[h][r];                                             // :290
                        const double z = (x - a_new) - ev[t];                          // :293
                        y = yv[t] + mu * z;                                            // :294
                        if (ok) {
                            zz += z * z;
                            const unsigned o = (unsigned)fr0 * ps32 + (unsigned)p + (unsigned)(4 * t) * ps32;
                            A[o] = a_new;
                            Y[o] = y;
                            S[(unsigned)f * P32 + (unsigned)p] = sparse_u8b(ev[t]);
                            if (WRITE_E) Eo[o] = ev[t];
                        }
                    }
                    const double raw2 = (x - a_new) + inv_mu2 * y;
                    const double e2 = shrink2(raw2, thr2);
                    const double m2 = ok ? (x - e2) + inv_mu2 * y : 0.0;
                    sT[f * TP + pl] = m2;
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

bool ialm_v2_supported(int n) { return n >= 1 && n <= kMaxN; }

template <int NB, int MODE, bool WE>
static void launch_one(hipStream_t s, const IalmBuffers &b)
{
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)k_ialm_pass_v2<NB, MODE, WE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)V2Cfg<NB>::lds_bytes);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_ialm_pass_v2<NB, MODE, WE>), dim3(b.nblk, b.nwin), dim3(256), V2Cfg<NB>::lds_bytes, s, b);
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

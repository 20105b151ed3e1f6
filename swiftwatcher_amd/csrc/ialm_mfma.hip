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
// Roofline: HBM.  v2: 33 B/element/iteration (+1 for the sparse image), v3 below: 21 B (+1 when it stores the sparse image); against 4n^2 flop/pixel -> 7.8 flop/B at n = 64, under the
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

// ---------------------------------------------------------------------------------
// v3: the same pass on HALF the f64 state.  With A_k = M_k B_k (:290) the multiplier update (:294) collapses:
//   Y_k = Y_{k-1} + mu_{k-1} (X - A_k - E_k)  and  M_k = X - E_k + Y_{k-1}/mu_{k-1}   =>   Y_k = mu_{k-1} (M_k - A_k),
// so A_k and Y_k are both functions of M_k and the small matrix B_k, and M_k alone (8 B/element) is the state
// carried between passes instead of A and Y (16 B).  Per pass and element:
//   read  X u8, M_k f64, U_{k-1} f16      write  M_{k+1} f64, U_k f16, clip(-E_{k+1}) u8        = 22 B (v2: 34 B)
// U = Y/mu is kept, in binary16, ONLY for the stopping norm ||Z_k||_F, Z_k = X - A_k - E_k = (M_k - A_k) - U_{k-1}
// (:293, :297).  The test is ||Z||_F < 1e-3 ||X||_F (:297, tol = 0.001): at the decision |z| ~ 0.1 grey levels
// against |U| ~ 0.5, so U's rounding (2^-12 relative) adds ||delta||^2 ~ 2e-6 ||Z||^2 -- far inside the margin by
// which consecutive iterations differ (>= 20 % in ||Z||).  The state itself never sees the rounded value (U_k is
// recomputed in f64 from M_k).
// The sparse image has to come from an exact E: pass k computes E_{k+1} exactly (it builds M_{k+1} from it) and
// writes its u8 form to S[k & 1]; when iteration K turns out to be the last, E_K is what pass K-1 left in
// S[(K-1) & 1] (k_select_sparse moves it to S[0] for odd K-1).  Those stores are 16-byte row pieces and cost 2.5x
// their share of the bytes, so k_ialm_small switches them off while ||Z|| is still far above the threshold
// (IalmWin::ws) and flags the window for a rerun should the iteration stop anyway (IalmWin::redo).  A and E in f64 are not produced: callers that ask
// for them run v2.
// ---------------------------------------------------------------------------------
// U travels as binary16 of U / 128: |U| <= 1/mu_1 < ||X||_F / 1.8 <= 2.3e6 for every admissible window, so
// U / 128 never overflows binary16, and the format's subnormal step is 7.6e-6 in U's units
constexpr float kUScale = 1.0f / 128.0f, kUUnscale = 128.0f;
__device__ __forceinline__ float buf_ld16h(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const unsigned short bits = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0);
    return (float)__builtin_bit_cast(_Float16, bits) * kUUnscale;
}
__device__ __forceinline__ void buf_st16h(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const _Float16 h = (_Float16)(v * kUScale);
    __builtin_amdgcn_raw_buffer_store_b16((short)__builtin_bit_cast(unsigned short, h), r, voff, soff, 0);
}

template <int NB, int MODE, bool FULL>
__global__ __launch_bounds__(256, 2) void k_ialm_pass_v3(IalmBuffers b, int sel)
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
    if (MODE == 0 && st.int_gram) return;        // the start pass's only product already came from k_gram_u8
    const bool ws = st.ws != 0;                  // sparse-image stores on for this pass (k_ialm_small decides)
    const bool ru = st.ru != 0, wu = st.wu != 0; // all of U read (full ||Z||) / written in this pass; else frames 0..3 only
    const int n = b.n, P = b.P;
    const unsigned ps32 = (unsigned)b.pstride, P32 = (unsigned)P;
    constexpr unsigned ROWSTEP = 128u;           // (4 t) * ROWSTEP = t * 512 elements: one chunk of M / U per k-step
    const double inv_mu = st.cur.inv_mu, thr = st.cur.thr, mu = st.cur.mu;
    const double inv_mu2 = st.nxt.inv_mu, thr2 = st.nxt.thr;
    const double dual = st.dual_norm;
    const int felems = b.fpad * (int)ps32;
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void *)(b.X + (int64_t)w * n * P), 0, n * P, 0x00020000);
    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void *)((sel ? b.Salt : b.S) + (int64_t)w * n * P), 0, n * P, 0x00020000);
    const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc((void *)(b.A + (int64_t)w * b.fpad * b.pstride), 0, felems * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void *)(b.U + (int64_t)w * b.fpad * b.pstride), 0, felems * 2, 0x00020000);

    // Y0 = X / dual_norm (:272) for the first two passes: the quotient of a small integer, formed in registers
    // as one Newton step on x * (1/dual) -- the correctly rounded x / dual without a per-element division
    const double rdual = 1.0 / dual;
    auto y0_of = [&](double x) {
        const double q = x * rdual;
        return __builtin_fma(__builtin_fma(-q, dual, x), rdual, q);
    };
    if (MODE != 0) {
        const double *Bm = b.Bm + (int64_t)w * n * n;
        for (int i = tid; i < NPAD * NPAD; i += 256) {
            const int k = i / NPAD, c = i % NPAD;
            sB[k * BP + c] = (k < n && c < n) ? Bm[k * n + c] : 0.0;
        }
    }
    __syncthreads();

    const int pl = lane & 15, fr0 = lane >> 4;
    const int flim = n - fr0;
    d4 G[C::NPAIR];
#pragma unroll
    for (int i = 0; i < C::NPAIR; ++i) G[i] = d4{0.0, 0.0, 0.0, 0.0};
    double zz = 0.0, zz0 = 0.0;                  // sum of z^2 over frames >= 4 / frames 0..3

    // a block owns groups of 8 consecutive tiles = 128 pixels: every 128-byte line of the u8 planes (and every
    // pair of half lines of the f32 plane) is touched by ONE workgroup, two tiles per wave back to back
    const int ntiles = (P + 15) >> 4;
    const int nsteps = 2 * ((((ntiles + 7) >> 3) - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x);
    for (int it = 0; it < nsteps; ++it) {
        const int tile = ((int)blockIdx.x + (it >> 1) * (int)gridDim.x) * 8 + wave * 2 + (it & 1);
        if (tile >= ntiles) continue;
        const unsigned p = (unsigned)(tile * 16 + pl);
        const bool pvalid = p < P32;
        // M and U are private to this kernel.  Layout [group of 128 pixels][k-step t][tile 0..7][frame 4t + 0..3][16 px]:
        // what one wave instruction touches (4 frame rows x 16 pixels of one k-step) is ONE contiguous piece -- 512 B
        // of M, a full 128-B line of U -- and the 64 rows a workgroup streams together form one 64 KB (16 KB) block.
        // Against frame-major planes (four lines 700 KB apart per instruction, 32-byte pieces of U): -6 % per launch.
        const unsigned ge = ((unsigned)tile >> 3) * (unsigned)b.fpad * 128u + ((unsigned)tile & 7u) * 64u + (unsigned)fr0 * 16u + (unsigned)pl;
        const unsigned vo8 = pvalid ? ge * 8u : kOob;                             // f64 state
        const unsigned vo2 = pvalid ? ge * 2u : kOob;                             // binary16 copy of Y/mu
        const unsigned vo1 = pvalid ? (unsigned)fr0 * P32 + p : kOob;             // u8 planes
        const unsigned vo1s = ws ? vo1 : kOob;
        const unsigned vo2r = ru ? vo2 : kOob, vo2w = wu ? vo2 : kOob;
        int xi[NK];
        double mv[NK];
        float uf[NK];
#pragma unroll
        for (int t = 0; t < NK; ++t) {
            const bool fvalid = FULL || 4 * t < flim;
            xi[t] = buf_ld8(rX, fvalid ? vo1 : kOob, (unsigned)(4 * t) * P32);
            if (MODE == 2) {
                mv[t] = buf_ld64(rM, fvalid ? vo8 : kOob, (unsigned)(4 * t) * ROWSTEP * 8u);
                uf[t] = buf_ld16h(rU, fvalid ? (t == 0 ? vo2 : vo2r) : kOob, (unsigned)(4 * t) * ROWSTEP * 2u);
            }
        }
        if (MODE == 1) {
            // first iteration: A_0 = 0 (:273) and Y_0 = X / dual (:272), so M_1 is a function of X alone
#pragma unroll
            for (int t = 0; t < NK; ++t) {
                const double x = (double)xi[t];
                const double u0 = inv_mu * y0_of(x);
                const double e = shrink2(x + u0, thr);                             // :282-283
                mv[t] = (x - e) + u0;                                              // :284
            }
        }
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
                        y = y0_of(x);
                    } else {
                        a_new = acc[h][r];                                             // :290
                        const double pk = mv[t] - a_new;                               // M_k - A_k = Y_k / mu_{k-1}
                        const double uprev = MODE == 2 ? (double)uf[t] : inv_mu * y0_of(x);
                        const double z = pk - uprev;                                   // :293
                        if (t == 0) zz0 += z * z; else zz += z * z;
                        y = mu * pk;                                                   // :294
                    }
                    const double u = inv_mu2 * y;
                    const double e2 = shrink2((x - a_new) + u, thr2);
                    const double m2 = (x - e2) + u;
                    sT[(4 * t + fr0) * TP + pl] = m2;
                    const bool fvalid = FULL || 4 * t < flim;
                    if (MODE != 0) {         // the start pass leaves no state: pass 1 rebuilds M_1 from X
                        buf_st64(m2, rM, fvalid ? vo8 : kOob, (unsigned)(4 * t) * ROWSTEP * 8u);
                        buf_st16h((float)u, rU, fvalid ? (t == 0 ? vo2 : vo2w) : kOob, (unsigned)(4 * t) * ROWSTEP * 2u);
                    }
                    buf_st8((int)sparse_u8b(e2), rS, fvalid ? vo1s : kOob, (unsigned)(4 * t) * P32);
                }
            }
        }
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

    __syncthreads();
    double *sG = lds;
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
        zz = zz0 + ((MODE == 1 || ru) ? zz : 0.0);           // without all of U only the first four frames count
        for (int off = 32; off; off >>= 1) zz += __shfl_down(zz, off);
        __syncthreads();
        if (lane == 0) lds[NPAD * NPAD + wave] = zz;
        __syncthreads();
        if (tid == 0)
            b.zzpart[(int64_t)w * b.nblk + blockIdx.x] =
                ((lds[NPAD * NPAD] + lds[NPAD * NPAD + 1]) + lds[NPAD * NPAD + 2]) + lds[NPAD * NPAD + 3];
    }
}

// the last iteration's sparse image sits in S[(iter-1) & 1]: bring the odd ones to S[0]
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

template <int NB, int MODE, bool FULL>
static void launch3_full(hipStream_t s, const IalmBuffers &b, int sel)
{
    static unsigned long long attr_mask = 0;
    if (!ensure_dyn_lds((const void *)k_ialm_pass_v3<NB, MODE, FULL>, V2Cfg<NB>::lds_bytes, attr_mask)) return;
    hipLaunchKernelGGL((k_ialm_pass_v3<NB, MODE, FULL>), dim3(b.nblk, b.nwin), dim3(256), V2Cfg<NB>::lds_bytes, s, b, sel);
    note_launch();
}

template <int NB>
static void launch3_nb(hipStream_t s, const IalmBuffers &b, int mode, int sel)
{
    // The FULL instantiation is also correct for a partially filled last block: rows past the last frame read X = 0
    // (range check), stay exactly zero in M and U, and their sparse-image stores are dropped -- it merely moves the
    // padded rows of M and U as well.  With 3 or 4 blocks that is the better deal: the per-row offset selects of the
    // other instantiation make the compiler issue the 48 loads one by one (n = 49: 2.3 instead of 4.2 TB/s).
    const bool full = b.n == 16 * NB || NB >= 3;
    if (mode == 0) { if (full) launch3_full<NB, 0, true>(s, b, sel); else launch3_full<NB, 0, false>(s, b, sel); }
    else if (mode == 1) { if (full) launch3_full<NB, 1, true>(s, b, sel); else launch3_full<NB, 1, false>(s, b, sel); }
    else { if (full) launch3_full<NB, 2, true>(s, b, sel); else launch3_full<NB, 2, false>(s, b, sel); }
}

void launch_ialm_pass_v3(hipStream_t s, const IalmBuffers &b, int mode, int k)
{
    const int nb = (b.n + 15) / 16;
    const int sel = k & 1;
    switch (nb) {
    case 1: launch3_nb<1>(s, b, mode, sel); break;
    case 2: launch3_nb<2>(s, b, mode, sel); break;
    case 3: launch3_nb<3>(s, b, mode, sel); break;
    default: launch3_nb<4>(s, b, mode, sel); break;
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

// IALM streaming pass on the f64 matrix cores, M-state formulation: the same pass as k_ialm_pass_v2 (ialm_mfma.hip) on HALF the
// f64 state.  With A_k = M_k B_k (image_filtering.py:290) the multiplier update (:294) collapses:
//   Y_k = Y_{k-1} + mu_{k-1} (X - A_k - E_k)  and  M_k = X - E_k + Y_{k-1}/mu_{k-1}   =>   Y_k = mu_{k-1} (M_k - A_k),
// so A_k and Y_k are both functions of M_k and the small matrix B_k, and M_k alone (8 B/element) is the state
// carried between passes instead of A and Y (16 B).  Per pass and element:
//   read  X u8, M_k f64, U_{k-1} f16      write  M_{k+1} f64, U_k f16, clip(-E_{k+1}) u8        = 22 B (A/Y state: 34 B)
// U = Y/mu is kept, in binary16, ONLY for the stopping norm ||Z_k||_F, Z_k = X - A_k - E_k = (M_k - A_k) - U_{k-1}
// (:293, :297).  The test is ||Z||_F < 1e-3 ||X||_F (:297, tol = 0.001): at the decision |z| ~ 0.1 grey levels
// against |U| ~ 0.5, so U's rounding (2^-12 relative) adds ||delta||^2 ~ 2e-6 ||Z||^2 -- far inside the margin by
// which consecutive iterations differ (>= 20 % in ||Z||); a window that lands within the guard band of the threshold
// all the same is rerun by the A/Y-state pass (swk_set_norm_guard).  The state itself never sees the rounded value
// (U_k is recomputed in f64 from M_k).  U travels as binary16 of U / 128: |U| <= 1/mu_1 < ||X||_F / 1.8 <= 2.3e6 for
// every admissible window, so U / 128 never overflows binary16, and the format's subnormal step is 7.6e-6 in U's units.
// The sparse image has to come from an exact E: pass k computes E_{k+1} exactly (it builds M_{k+1} from it) and
// writes its u8 form to S[k & 1]; when iteration K turns out to be the last, E_K is what pass K-1 left in
// S[(K-1) & 1] (k_select_sparse moves it to S[0] for odd K-1).  Those stores are 16-byte row pieces and cost 2.5x
// their share of the bytes, so k_ialm_small switches them off while ||Z|| is still far above the threshold
// (IalmWin::ws) and flags the window for a rerun should the iteration stop anyway (IalmWin::redo).  A and E in f64 are
// not produced: callers that ask for them run the A/Y-state pass.
//
// The kernel is instantiated per NUMBER OF K-STEPS NK = ceil(n / 4) (round 1's was per 16-frame block):
//
//   * a window of n frames moves 4 NK frame rows (n rounded up to 4, not to 16): the CLI's queue of 21 frames runs
//     6 k-steps, 12 + 12 MFMAs per 16-pixel tile and 24 rows of state instead of 8, 16 + 12 and 32;
//     n = 49 streams 52 rows instead of 64.  Rows n .. 4 NK - 1 read X = 0 through the buffer range check, stay
//     exactly zero in M and U, and their sparse-image stores fall outside the buffer and are dropped.
//   * what bounds the pass on gfx950 is the f64 execution unit, not HBM: f64 vector instructions and f64 MFMAs do
//     not overlap on a SIMD (tools/f64_pipe_probe.hip: a wave pair running 8 MFMAs + 64 FMAs each takes the SUM of the
//     two alone), so per 16-pixel tile the time is 104 MFMAs x 64 cycles + (f64 vector instructions) x ~5 cycles.
//     The element-wise part is therefore written with the fewest f64 instructions (9-13 per element instead of 22: see
//     pass_loop), and the per-window switches of a pass (sparse-image stores, U read / written) select one of eight
//     specialised copies of the tile loop instead of predicating stores that are computed anyway.
//   * a wave that has the matrix pipe to itself issues MFMAs that rotate over many accumulators more slowly than
//     MFMAs that chain on one (tools/f64_mfma_chain_probe.hip: 103 cycles each over eight accumulators, 72 on one; two
//     waves issuing together always reach 64): the A update runs one out-frame block at a time and the Gram phase one
//     block pair at a time.
//   * up to 32 frames the tile loop is software-pipelined (the next tile's loads are issued before the Gram phase);
//     with more frames that makes no difference (measured) and the registers go to the Gram operands.  The plain tile
//     loop stays as the cross-check (swk_set_ialm_variant(5)).
//   * tried and measured without effect: raised priority or a delayed start for one wave of each SIMD pair
//     (swk_set_pass_tuning keeps them as A/B knobs).
//
// Register layout (lane l, register t <-> pixel p0 + (l & 15), frame 4t + (l >> 4)), LDS tiles and buffer addressing are
// those of k_ialm_pass_v2 (ialm_mfma.hip, header).
#include "swk_internal.h"

namespace swk {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));

namespace {

constexpr unsigned kOob = 0x80000000u;        // a byte offset past every buffer: loads give 0, stores are dropped
constexpr float kUScale = 1.0f / 128.0f, kUUnscale = 128.0f;      // U travels as binary16 of U / 128 (header)

__device__ __forceinline__ double shrink2(double raw, double thr)
{
    return fmax(raw - thr, 0.0) + fmin(raw + thr, 0.0);          // image_filtering.py:283
}
__device__ __forceinline__ int sparse_u8b(double e)
{
    double v = -e;                                                // :244
    v = fmin(fmax(v, 0.0), 255.0);                               // :245
    return (int)v;                                               // astype(uint8) truncates
}
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
// returns U / 128 as stored: the stopping norm is accumulated in that scale and multiplied by 128^2 once at the end
__device__ __forceinline__ float buf_ld16h(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const unsigned short bits = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0);
    return (float)__builtin_bit_cast(_Float16, bits);
}
__device__ __forceinline__ void buf_st16h(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const _Float16 h = (_Float16)(v * kUScale);
    __builtin_amdgcn_raw_buffer_store_b16((short)__builtin_bit_cast(unsigned short, h), r, voff, soff, 0);
}

template <int NK>
struct MCfg {
    static constexpr int NB = (NK + 3) / 4;                        // 16-frame output blocks
    static constexpr int NPAD = 16 * NB;
    static constexpr int BP = (NPAD % 32 == 0) ? NPAD + 16 : NPAD;   // LDS pitch of B: rows 32 banks apart
    static constexpr int TP = 17;                                  // LDS pitch of the transpose tile
    static constexpr int NPAIR = NB * (NB + 1) / 2;
    static constexpr size_t lds_bytes = (size_t)(NPAD * BP + 4 * NPAD * TP) * sizeof(double);
    // the first-iteration pass (MODE 1) appends two 256-entry tables (M_1 and U_0 as functions of the pixel value): + 4 KB, which at
    // 64 frames makes a workgroup exactly half of the CU's 160 KB
    static constexpr size_t lut_bytes = 2 * 256 * sizeof(double);
};

constexpr unsigned ROWSTEP = 128u;           // (4 t) * ROWSTEP = t * 512 elements: one chunk of M / U per k-step

}  // namespace

struct PassCtx {
    __amdgpu_buffer_rsrc_t rX, rS, rM, rU;
    double *sB, *sT;
    const double *lut;          // MODE 1: [256] M_1(x), then [256] U_0(x) = Y_0(x) / mu_0
    double inv_mu, thr, inv_mu2, thr2, dual, rdual, ratio;
    float ratio_f;
    unsigned P32, fpad;
    int pl, fr0, wave, ntiles, nsteps, bx, gx;
};

// The tile loop of one wave.  WS: this pass stores the sparse image; RU / WU: it reads / writes all of U (else frames 0..3).
// umax: largest |U_{k-1}| / 128 this wave read in a pass near the stopping decision (WS: the sparse image is being stored) -- what the
// error bound of the float32 / binary16 stopping norm needs (small_prologue, ialm_small_dev.h)
template <int NK, int MODE, bool WS, bool RU, bool WU>
__device__ __forceinline__ void pass_loop(const PassCtx &cx, d4 (&G)[MCfg<NK>::NPAIR], float &zz, float &zz0, float &umax)
{
    using C = MCfg<NK>;
    constexpr int NB = C::NB, BP = C::BP, TP = C::TP;
    constexpr bool PIPE = NB <= 2;
    auto y0_of = [&](double x) {
        const double q = x * cx.rdual;
        return __builtin_fma(__builtin_fma(-q, cx.dual, x), cx.rdual, q);
    };
    auto tile_of = [&](int it) { return (cx.bx + (it >> 1) * cx.gx) * 8 + cx.wave * 2 + (it & 1); };
    int xi[NK];
    double mv[NK];
    float uf[NK];
    unsigned vo8, vo2, vo1;                      // per-lane byte offsets of the current tile: f64 state, f16 copy of Y/mu, u8 planes
    auto offsets = [&](int tile, unsigned &o8, unsigned &o2, unsigned &o1) {
        const unsigned p = (unsigned)(tile * 16 + cx.pl);
        const bool pvalid = tile < cx.ntiles && p < cx.P32;
        // M and U are private to this kernel: [group of 128 pixels][k-step t][tile 0..7][frame 4t + 0..3][16 px]
        const unsigned ge = ((unsigned)tile >> 3) * (unsigned)cx.fpad * 128u + ((unsigned)tile & 7u) * 64u + (unsigned)cx.fr0 * 16u + (unsigned)cx.pl;
        o8 = pvalid ? ge * 8u : kOob;
        o2 = pvalid ? ge * 2u : kOob;
        o1 = pvalid ? (unsigned)cx.fr0 * cx.P32 + p : kOob;
    };
    auto load_tile = [&](unsigned o8, unsigned o2, unsigned o1) {
        const unsigned o2r = RU ? o2 : kOob;
#pragma unroll
        for (int t = 0; t < NK; ++t) {
            xi[t] = buf_ld8(cx.rX, o1, (unsigned)(4 * t) * cx.P32);
            if (MODE == 2) {
                mv[t] = buf_ld64(cx.rM, o8, (unsigned)(4 * t) * ROWSTEP * 8u);
                uf[t] = buf_ld16h(cx.rU, t == 0 ? o2 : o2r, (unsigned)(4 * t) * ROWSTEP * 2u);
            }
        }
    };

    int tile = cx.nsteps > 0 ? tile_of(0) : cx.ntiles;
    offsets(tile, vo8, vo2, vo1);
    load_tile(vo8, vo2, vo1);
    for (int it = 0; it < cx.nsteps && tile < cx.ntiles; ++it) {
        if (MODE == 1) {
            // first iteration: A_0 = 0 (:273) and Y_0 = X / dual (:272), so M_1 is a function of X alone
#pragma unroll
            for (int t = 0; t < NK; ++t) mv[t] = cx.lut[xi[t]];          // a function of the 8-bit value alone: tabulated (kernel prologue)
        }
        // ---- A_k^T = B^T M_k^T on the matrix cores, ONE out-frame block at a time: a wave that has the matrix pipe to
        //      itself issues back-to-back MFMAs on one accumulator every 72 cycles, on two alternating ones every 76, on
        //      eight every 103 (tools/f64_mfma_chain_probe.hip; two waves issuing together always reach the pipe's 64) --
        //      then Z, Y, the start of the next iteration and the stores of that block's four frame rows ----
#pragma unroll
        for (int bq = 0; bq < NB; ++bq) {
            d4 acc1 = d4{0.0, 0.0, 0.0, 0.0};
            if (MODE != 0) {
#pragma unroll
                for (int t = 0; t < NK; ++t) {
                    const double bop = cx.sB[(4 * t + cx.fr0) * BP + 16 * bq + cx.pl];
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(bop, mv[t], acc1, 0, 0, 0);
                }
            }
            // The f64 vector unit and the f64 matrix pipe do not overlap on gfx950 (tools/f64_pipe_probe.hip), so every
            // f64 instruction here is time taken from the MFMAs: the element-wise part is written with the fewest of them.
            // With c = clamp(raw, -thr, +thr) the shrinkage (:283) is E = raw - c exactly (fl(raw - thr) / fl(raw + thr) / 0
            // in the three cases, the same roundings as max(.) + min(.)), and M_{k+1} = X - E + U (:284) = A_k + c; the
            // stopping norm (:297, compared at a relative 1e-3 against iterates that move by 20 % and more) is formed in
            // float32 from the float32 copy of M_k - A_k that the binary16 store of U needs anyway.
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = 4 * bq + r;
                if (t >= NK) continue;
                const double x = (double)xi[t];
                double a_new, raw;
                float pkf = 0.f;
                if (MODE == 0) {
                    a_new = 0.0;
                    raw = x + cx.inv_mu2 * y0_of(x);                               // :282 with A_0 = 0, Y_0 = X / dual
                } else {
                    a_new = acc1[r];                                               // :290
                    const double pk = mv[t] - a_new;                               // M_k - A_k = Y_k / mu_{k-1}  (:293-294)
                    raw = __builtin_fma(pk, cx.ratio, x - a_new);                  // :282, (X - A_k) + Y_k / mu_k
                    if (MODE == 1) {
                        const float zf = (float)(pk - cx.lut[256 + xi[t]]) * kUScale;   // :293 with U_0 = Y_0 / mu_0, in units of 128
                        if (t == 0) zz0 += zf * zf; else zz += zf * zf;
                        if (t == 0 || WU) pkf = (float)pk;
                    } else if (t == 0 || RU || WU) {
                        pkf = (float)pk;
                        if (t == 0 || RU) {
                            const float zf = __builtin_fmaf(pkf, kUScale, -uf[t]);  // :293, in units of 128
                            if (t == 0) zz0 += zf * zf; else zz += zf * zf;
                            if (WS) umax = fmaxf(umax, fabsf(uf[t]));
                        }
                    }
                }
                const double c = fmin(fmax(raw, -cx.thr2), cx.thr2);
                const double m2 = a_new + c;                                       // :284
                cx.sT[(4 * t + cx.fr0) * TP + cx.pl] = m2;
                if (MODE != 0) {         // the start pass leaves no state: pass 1 rebuilds M_1 from X
                    buf_st64(m2, cx.rM, vo8, (unsigned)(4 * t) * ROWSTEP * 8u);
                    if (t == 0 || WU) buf_st16h(pkf * cx.ratio_f, cx.rU, vo2, (unsigned)(4 * t) * ROWSTEP * 2u);
                }
                if (WS) buf_st8(sparse_u8b(raw - c), cx.rS, vo1, (unsigned)(4 * t) * cx.P32);   // clip(-E) of :244-245
            }
        }
        // ---- up to 32 frames (registers to spare, four waves per SIMD, fewer flops per byte) the next tile's loads go out
        //      here, where xi / mv / uf are dead, and fly under the Gram phase: 3.1 instead of 3.5 ms per launch at
        //      n = 21.  With more frames the registers hold the Gram operands instead and the loads follow it ----
        const int tile_n = it + 1 < cx.nsteps ? tile_of(it + 1) : cx.ntiles;
        unsigned n8, n2, n1;
        offsets(tile_n, n8, n2, n1);
        if (PIPE) load_tile(n8, n2, n1);
        // ---- Gram of M_{k+1}: the transposed registers are both MFMA operands; one block pair at a time, its four
        //      MFMAs (16 pixels = 4 k-steps) back to back on the pair's accumulator ----
        {
            double tr[4][NB];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int fb = 0; fb < NB; ++fb) tr[g][fb] = cx.sT[(16 * fb + cx.pl) * TP + 4 * g + cx.fr0];
            int pair = 0;
#pragma unroll
            for (int ib = 0; ib < NB; ++ib)
#pragma unroll
                for (int jb = ib; jb < NB; ++jb) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) G[pair] = __builtin_amdgcn_mfma_f64_16x16x4f64(tr[g][ib], tr[g][jb], G[pair], 0, 0, 0);
                    ++pair;
                }
        }
        if (!PIPE) load_tile(n8, n2, n1);
        tile = tile_n; vo8 = n8; vo2 = n2; vo1 = n1;
    }

}

// tune: bit 0 = raised priority for the wave in the odd hardware slot of its SIMD, bit 1 = that wave also starts
// half a tile late (A/B knobs; results do not depend on them)
template <int NK, int MODE>
__global__ __launch_bounds__(256, 2) void k_ialm_pass_m(IalmBuffers b, int sel, int tune)
{
    using C = MCfg<NK>;
    constexpr int NB = C::NB, NPAD = C::NPAD, BP = C::BP, TP = C::TP;
    extern __shared__ double lds[];
    double *sB = lds;                                             // [NPAD][BP]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double *sT = lds + NPAD * BP + wave * (NPAD * TP);           // this wave's [NPAD][TP]
    const int w = blockIdx.y;
    const IalmWin &st = b.win[w];
    if (st.done) return;
    if (MODE == 0 && st.int_gram) return;        // the start pass's only product already came from k_gram_u8
    if (tune & 3) {
        const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | 4);      // HW_ID.wave_id: this wave's slot on its SIMD
        if (slot & 1u) {
            if (tune & 1) __builtin_amdgcn_s_setprio(1);
            if (tune & 2) { __builtin_amdgcn_s_sleep(64); }
        }
    }
    const bool ws = st.ws != 0;                  // sparse-image stores on for this pass (k_ialm_small decides)
    const bool ru = st.ru != 0, wu = st.wu != 0; // all of U read (full ||Z||) / written in this pass; else frames 0..3 only
    const int n = b.n, P = b.P;
    const unsigned P32 = (unsigned)P;
    const double inv_mu = st.cur.inv_mu, thr = st.cur.thr, mu = st.cur.mu;
    const double inv_mu2 = st.nxt.inv_mu, thr2 = st.nxt.thr;
    const double dual = st.dual_norm;
    const double ratio = mu * inv_mu2;           // U_k = Y_k / mu_k = (M_k - A_k) mu_{k-1} / mu_k
    const float ratio_f = (float)ratio;
    const int felems = b.fpad * (int)b.pstride;                   // b.fpad == 4 NK
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void *)(b.X + (int64_t)w * n * P), 0, n * P, 0x00020000);
    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void *)((sel ? b.Salt : b.S) + (int64_t)w * n * P), 0, n * P, 0x00020000);
    const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc((void *)(b.A + (int64_t)w * b.fpad * b.pstride), 0, felems * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void *)(b.U + (int64_t)w * b.fpad * b.pstride), 0, felems * 2, 0x00020000);

    // Y0 = X / dual_norm (:272) for the first two passes: one Newton step on x * (1/dual), the correctly rounded quotient
    const double rdual = 1.0 / dual;
    if (MODE != 0) {
        const double *Bm = b.Bm + (int64_t)w * n * n;
        for (int i = tid; i < NPAD * NPAD; i += 256) {
            const int k = i / NPAD, c = i % NPAD;
            sB[k * BP + c] = (k < n && c < n) ? Bm[k * n + c] : 0.0;
        }
    }
    // first iteration: M_1 = (x - E_1) + U_0 with U_0 = Y_0 / mu_0, Y_0 = x / dual (:272, the correctly rounded quotient by one
    // Newton step) and E_1 = shrink(x + U_0) (:282-284) depend on the pixel's 8-bit value only: 256 threads tabulate them once
    // (the same operations in the same order as the element-wise code they replace: 11 f64 instructions per element less)
    double *lut = lds + NPAD * BP + 4 * NPAD * TP;
    if (MODE == 1) {
        const double x = (double)tid;
        const double q = x * rdual;
        const double y0 = __builtin_fma(__builtin_fma(-q, dual, x), rdual, q);
        const double u0 = inv_mu * y0;
        const double e = shrink2(x + u0, thr);
        lut[tid] = (x - e) + u0;
        lut[256 + tid] = u0;
    }
    // frame rows 4 NK .. NPAD - 1 of the transpose tile are never written: they must read as zeros in the Gram phase
    for (int i = lane; i < (NPAD - 4 * NK) * TP; i += 64) sT[4 * NK * TP + i] = 0.0;
    __syncthreads();

    const int pl = lane & 15, fr0 = lane >> 4;
    d4 G[C::NPAIR];
#pragma unroll
    for (int i = 0; i < C::NPAIR; ++i) G[i] = d4{0.0, 0.0, 0.0, 0.0};
    float zz = 0.f, zz0 = 0.f;                   // sum of z^2 over frames >= 4 / frames 0..3 (float32: see the element-wise part)
    float umax = 0.f;

    // a block owns groups of 8 consecutive tiles = 128 pixels (every 128-byte line of the u8 planes is touched by ONE
    // workgroup), two tiles per wave back to back; the valid tiles of a wave are a prefix of its sequence
    const int ntiles = (P + 15) >> 4;
    const int nsteps = 2 * ((((ntiles + 7) >> 3) - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x);

    PassCtx cx{rX, rS, rM, rU, sB, sT, lut, inv_mu, thr, inv_mu2, thr2, dual, rdual, ratio, ratio_f, P32, (unsigned)b.fpad,
               pl, fr0, wave, ntiles, nsteps, (int)blockIdx.x, (int)gridDim.x};
    // the per-window switches of this pass (sparse-image stores, all of U read / written) are wave-uniform but only
    // known on the device: one specialised copy of the tile loop per combination, chosen once
    const int flags = (ws ? 1 : 0) | (ru ? 2 : 0) | (wu ? 4 : 0);
    if (MODE == 0) pass_loop<NK, 0, true, true, true>(cx, G, zz, zz0, umax);
    else if (MODE == 1) { if (wu) pass_loop<NK, 1, true, true, true>(cx, G, zz, zz0, umax); else pass_loop<NK, 1, true, true, false>(cx, G, zz, zz0, umax); }
    else switch (flags) {
        case 0: pass_loop<NK, 2, false, false, false>(cx, G, zz, zz0, umax); break;
        case 1: pass_loop<NK, 2, true, false, false>(cx, G, zz, zz0, umax); break;
        case 2: pass_loop<NK, 2, false, true, false>(cx, G, zz, zz0, umax); break;
        case 3: pass_loop<NK, 2, true, true, false>(cx, G, zz, zz0, umax); break;
        case 4: pass_loop<NK, 2, false, false, true>(cx, G, zz, zz0, umax); break;
        case 5: pass_loop<NK, 2, true, false, true>(cx, G, zz, zz0, umax); break;
        case 6: pass_loop<NK, 2, false, true, true>(cx, G, zz, zz0, umax); break;
        default: pass_loop<NK, 2, true, true, true>(cx, G, zz, zz0, umax); break;
    }

    // ---- block-level, fixed-order combination of the four waves' Gram accumulators ----
    __syncthreads();
    double *sG = lds;                           // reuse: [NPAD][NPAD] <= NPAD * BP + 4 * NPAD * TP
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
        // without all of U only the first four frames count; the sums are in units of 128^2 (buf_ld16h)
        double zsum = ((double)zz0 + ((MODE == 1 || ru) ? (double)zz : 0.0)) * ((double)kUUnscale * (double)kUUnscale);
        for (int off = 32; off; off >>= 1) zsum += __shfl_down(zsum, off);
        __syncthreads();
        if (lane == 0) lds[NPAD * NPAD + wave] = zsum;
        __syncthreads();
        for (int off = 32; off; off >>= 1) umax = fmaxf(umax, __shfl_down(umax, off));
        if (lane == 0) lds[NPAD * NPAD + 4 + wave] = (double)umax;
        __syncthreads();
        if (tid == 0) {
            b.zzpart[(int64_t)w * b.nblk + blockIdx.x] =
                ((lds[NPAD * NPAD] + lds[NPAD * NPAD + 1]) + lds[NPAD * NPAD + 2]) + lds[NPAD * NPAD + 3];
            // second half of the array: the block's largest |U_{k-1}| (0 in the passes that do not track it)
            b.zzpart[(int64_t)(b.nwin + w) * b.nblk + blockIdx.x] =
                fmax(fmax(lds[NPAD * NPAD + 4], lds[NPAD * NPAD + 5]), fmax(lds[NPAD * NPAD + 6], lds[NPAD * NPAD + 7])) * (double)kUUnscale;
        }
    }
}

template <int NK, int MODE>
static void launch_m_one(hipStream_t s, const IalmBuffers &b, int sel, int tune)
{
    static unsigned long long attr_mask = 0;
    constexpr size_t lds = MCfg<NK>::lds_bytes + (MODE == 1 ? MCfg<NK>::lut_bytes : 0);
    if (!ensure_dyn_lds((const void *)k_ialm_pass_m<NK, MODE>, lds, attr_mask)) return;
    hipLaunchKernelGGL((k_ialm_pass_m<NK, MODE>), dim3(b.nblk, b.nwin), dim3(256), lds, s, b, sel, tune);
    note_launch();
}

template <int NK>
static void launch_m_nk(hipStream_t s, const IalmBuffers &b, int mode, int sel, int tune, bool pipe)
{
    (void)pipe;
    if (mode == 0) launch_m_one<NK, 0>(s, b, sel, tune);
    else if (mode == 1) launch_m_one<NK, 1>(s, b, sel, tune);
    else launch_m_one<NK, 2>(s, b, sel, tune);
}

// frames per window -> planes of M / U state per window for this kernel
int ialm_mstate_fpad(int n) { return (n + 3) & ~3; }

void launch_ialm_pass_m(hipStream_t s, const IalmBuffers &b, int mode, int k, int tune, bool pipe)
{
    const int sel = k & 1;
    switch ((b.n + 3) / 4) {
    case 1: launch_m_nk<1>(s, b, mode, sel, tune, pipe); break;
    case 2: launch_m_nk<2>(s, b, mode, sel, tune, pipe); break;
    case 3: launch_m_nk<3>(s, b, mode, sel, tune, pipe); break;
    case 4: launch_m_nk<4>(s, b, mode, sel, tune, pipe); break;
    case 5: launch_m_nk<5>(s, b, mode, sel, tune, pipe); break;
    case 6: launch_m_nk<6>(s, b, mode, sel, tune, pipe); break;
    case 7: launch_m_nk<7>(s, b, mode, sel, tune, pipe); break;
    case 8: launch_m_nk<8>(s, b, mode, sel, tune, pipe); break;
    case 9: launch_m_nk<9>(s, b, mode, sel, tune, pipe); break;
    case 10: launch_m_nk<10>(s, b, mode, sel, tune, pipe); break;
    case 11: launch_m_nk<11>(s, b, mode, sel, tune, pipe); break;
    case 12: launch_m_nk<12>(s, b, mode, sel, tune, pipe); break;
    case 13: launch_m_nk<13>(s, b, mode, sel, tune, pipe); break;
    case 14: launch_m_nk<14>(s, b, mode, sel, tune, pipe); break;
    case 15: launch_m_nk<15>(s, b, mode, sel, tune, pipe); break;
    default: launch_m_nk<16>(s, b, mode, sel, tune, pipe); break;
    }
}

}  // namespace swk

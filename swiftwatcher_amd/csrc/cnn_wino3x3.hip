// The 3x3 expand convolutions of the wide Fire modules (fire4 .. fire9: 32 -> 128, 48 -> 192, 64 -> 256 channels, 94 of the
// classifier's 125 M multiply-accumulates per segment) by Winograd's minimal filtering F(2x2, 3x3) on the f32 matrix cores:
//
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A        per 2 x 2 output tile, 4 x 4 input patch d, 3 x 3 filter g
//
// 16 multiplies per four outputs instead of 36: 2.25 x fewer MFMAs than the direct kernel (cnn_conv3x3.hip), which already
// runs at three quarters of the f32 matrix peak.  Same interface and the same fused epilogue (bias + ReLU + placement behind
// the expand1x1 channels); the filter transform U = G g G^T is made once per layer on the host
// (swk_winograd_f2x2_3x3_weights, float64 then rounded).  Float32 results differ from the direct kernel by the usual
// F(2x2, 3x3) rounding (a few 1e-7 relative to the output scale; tests/test_classifier.py).
//
// In the transformed domain the layer is 16 independent GEMMs, one per position p = (xi, nu) of the 4 x 4 patch:
// M_p[tile][co] = sum_ci V_p[tile][ci] U_p[ci][co], and every output needs all 16 of them -- kept as 16 accumulators they
// would be four times the outputs.  Instead the positions run one after the other, M_p is ONE accumulator, and after its
// last k-step it is added with its coefficient A^T[i][xi] A^T[j][nu] in {0, +1, -1} into the four output accumulators Y_ij.
//   * workgroup = CG x TGN waves: 32 TGN output tiles (2 x 2 pixels each, of any segments) x all output channels; wave
//     (cg, tg) owns tile group tg (32 tiles) and NBW column blocks of 32 output channels: Y = 4 x NBW x 16 registers, M = 16.
//     NBW = 1 is the default: 128 registers, four waves per SIMD (8-wave workgroups for 64 -> 256, 4-wave ones for 32 -> 128;
//     48 -> 192 has six column blocks and runs as ONE 12-wave workgroup per CU, three waves on every SIMD -- two 6-wave
//     workgroups do not become co-resident and leave two SIMDs half empty).  NBW = 2 (226 registers, two waves per SIMD)
//     measured 6 % slower on 64 -> 256 and 15 % on 32 -> 128 (round 2) and is no longer instantiated.
//   * V_p = B^T d B restricted to position p is a signed sum of FOUR patch pixels.  The workgroup forms V_p for its tiles
//     once (NBW (tile, 4-channel) items per thread: 4 float4 loads, 3 fmas per component), transposes it into LDS as
//     [channel][tile] -- the matrix cores' pixel operand -- double buffered: position p + 1 is staged while p multiplies
//     (loads issued in the first, transform and LDS store in the last phase of the position).
//   * U_p streams through LDS by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write pass) in phases of SPP k-chunks
//     (16 channels each) of one column block of every wave, double buffered: 8 to 18 KB per phase, laid out by the host
//     exactly as it sits in LDS ([p][column block][chunk][k half][quad][channel][4]: an operand read is one ds_read_b128 per
//     four k-steps, 512 contiguous bytes per half wave).  One barrier per phase = per 8 SPP MFMAs of a wave.
//     (A first version read U_p from L2 into registers one k-chunk ahead: the waves spent 43 % of their cycles waiting for
//     those loads, SQ_WAIT_ANY / SQ_WAVE_CYCLES, and the matrix pipe was 43 % busy.)
//   * MFMA roles as in cnn_conv1x1.hip: weights = A operand, tiles = B operand, so a register quad of an accumulator is four
//     consecutive output channels of the lane's own tile: float4 stores.
// Where it stands (MI355X, batch 4096, tools/bench_convs.py): 64 -> 256 on 16 x 16 outputs 1.30-1.34 ms against 2.63 ms for the
// direct kernel; the matrix pipe is 70 % busy (SQ_VALU_MFMA_BUSY_CYCLES), the f32 vector instructions that share it
// (profiles/r2_f32_pipe_probe.txt) another 10 %; s_memtime brackets (tools/wino_stamp.py) put a wave's remaining time into
// the barrier (19 %) and the phase prologue / epilogue; skipping the global loads altogether gains 9 %.
// Launched on the CALLER's stream (PyTorch's current stream).
#include "swk_internal.h"

#include <type_traits>

namespace swk {

typedef float f16v __attribute__((ext_vector_type(16)));

// Diagnostic build only (-DSWK_WINO_STAMP, tools/wino_stamp.py): s_memtime brackets around the parts of a phase, summed per wave
// and stored to g_wino_stamp[wave][5] by lane 0.  Its fences forbid overlaps the real kernel has: read the shares, not the length.
#ifdef SWK_WINO_STAMP
std::atomic<int> g_launch_error{0};          // the diagnostic library is this file alone
__device__ unsigned long long *g_wino_stamp;
#define SWK_STAMP(k)                                                                                 \
    do {                                                                                             \
        unsigned long long t_;                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                   \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        stamp_sum[k] += t_ - stamp_last;                                                             \
        stamp_last = t_;                                                                             \
    } while (0)
#else
#define SWK_STAMP(k) do { } while (0)
#endif

template <int NBLK, int NBW, int TGN, int SPP, int WPS, bool PRIV>
__global__ __launch_bounds__(64 * (NBLK / NBW) * TGN, WPS) void k_wino3x3_relu_place(const float *__restrict__ src, int nseg, int t, int T,
                                                                      const float *__restrict__ w2, const float *__restrict__ bias, int cout,
                                                                      float *__restrict__ dst, int dH, int dW, int dC, int off_y, int off_x,
                                                                      int c_off, FastDiv fTT, FastDiv fT)
{
    // S k-chunks of 16 channels per position; a phase = SPP of them for one column block: PHS phases per (position, column block),
    // PPOS per position
    // a wave owns NBW column blocks of 32 output channels (WPS waves per SIMD fit: NBW = 2 -> 2, NBW = 1 -> 4)
    constexpr int CG = NBLK / NBW, NW = CG * TGN, NT = 64 * NW, CIN = 8 * NBLK, S = CIN / 16, SLOTS = 32 * TGN, VP = SLOTS + 1, G4 = CIN / 4,
                  NP = 32 * NBLK, CGR = 32 * CG, PHS = S / SPP, PPOS = NBW * PHS;
    // PRIV (one column block per wave, 64 -> 256): every wave copies the 2 KB of filter operands of a phase that it reads itself
    // into a slice of its own -- its vmcnt tells it when they have landed, and the workgroup meets only once per position (for
    // V): 3 % faster there, 5-13 % slower on the narrower shapes.  Otherwise the waves share the copy of a phase (WPH floats,
    // PPW pieces each) and meet after every phase.
    static_assert(!PRIV || NBW == 1, "private filter slices: one column block per wave");
    constexpr int WCG = PRIV ? 32 : CGR;          // output channels side by side in a phase buffer
    constexpr int WPH = PRIV ? NW * 16 * SPP * 32 : 16 * SPP * CGR, PPW = WPH / (256 * NW);
    static_assert(NBW * NT == SLOTS * G4, "NBW staging items per thread");
    static_assert(S % SPP == 0 && PPW * NW * 256 == WPH && (!PRIV || SPP == 1), "whole 1 KB pieces per wave");
    extern __shared__ float lds[];                 // W[2][WPH] (filter operands of two phases), V[2][CIN][VP], the bias padded to NP
    float *const W0 = lds, *const W1 = lds + WPH;
    float *const V0 = lds + 2 * WPH, *const V1 = V0 + CIN * VP, *const lbias = V1 + CIN * VP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int cg = wave % CG, tg = wave / CG;
    const int o = t - 2, TT = T * T;
    const int64_t ntiles = (int64_t)nseg * TT, src_floats = (int64_t)nseg * t * t * CIN;
    const int64_t ntasks = (ntiles + SLOTS - 1) / SLOTS;
    for (int i = tid; i < NP; i += NT) lbias[i] = i < cout ? bias[i] : 0.0f;

    // Every vector instruction of this kernel is paid in matrix-pipe time: the exact-f32 MFMA and the f32 vector unit are
    // one execution unit on gfx950 (profiles/r2_f32_pipe_probe.txt).  Hence 32-bit byte offsets against scalar bases
    // (one v_min + one v_add per patch load), LDS-DMA pieces addressed by instruction offsets, zero coefficients skipped.
    // ---- staging items of this thread: (tile slot, 4-channel group), channel group fastest (a wave reads whole pixels) ----
    int sg[NBW], sslot[NBW];
#pragma unroll
    for (int k = 0; k < NBW; ++k) { sg[k] = (tid + k * NT) % G4; sslot[k] = (tid + k * NT) / G4; }
    unsigned sbase[NBW];          // byte offset of the item's patch origin in src
    int slim[NBW];              // largest byte offset a patch load of the item may add (the last float4 of src)
    const char *const srcb = (const char *)src;
    auto stage_setup = [&](int64_t task) {
#pragma unroll
        for (int k = 0; k < NBW; ++k) {
            int64_t m = task * SLOTS + sslot[k];
            if (m >= ntiles) m = ntiles - 1;
            const unsigned bu = fTT.div((unsigned)m), rem = (unsigned)m - bu * (unsigned)TT, ty = fT.div(rem), tx = rem - ty * (unsigned)T;
            const int64_t b = bu;          // ntiles < 2^31 (checked by the launcher): invariant-divisor division
            const int64_t base = (((b * t + 2 * ty) * t + 2 * tx) * (int64_t)CIN + 4 * sg[k]) * 4;
            const int64_t lim = src_floats * 4 - 16 - base;          // a patch may reach one row / column past an odd-sized tile
            sbase[k] = (unsigned)base;
            slim[k] = (int)(lim < (1 << 30) ? lim : (1 << 30));
        }
    };
    float4 st[4];
    // patch rows (columns) that position xi (nu) combines: B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
    //   xi = 0: d0 - d2,  1: d1 + d2,  2: d2 - d1,  3: d1 - d3        -> first row {0,1,2,1}, second {2,2,1,3}, sign {-,+,-,-}
    auto stage_issue = [&](int p, int k) {
        const int xi = p >> 2, nu = p & 3;
        const int ra0 = (0x1210 >> (4 * xi)) & 15, ra1 = (0x3122 >> (4 * xi)) & 15;
        const int rb0 = (0x1210 >> (4 * nu)) & 15, rb1 = (0x3122 >> (4 * nu)) & 15;
        const int lim = slim[k];
        const int o00 = (ra0 * t + rb0) * (CIN * 4), o01 = (ra0 * t + rb1) * (CIN * 4), o10 = (ra1 * t + rb0) * (CIN * 4),
                  o11 = (ra1 * t + rb1) * (CIN * 4);
        st[0] = *(const float4 *)(srcb + (sbase[k] + (unsigned)(o00 < lim ? o00 : lim)));
        st[1] = *(const float4 *)(srcb + (sbase[k] + (unsigned)(o01 < lim ? o01 : lim)));
        st[2] = *(const float4 *)(srcb + (sbase[k] + (unsigned)(o10 < lim ? o10 : lim)));
        st[3] = *(const float4 *)(srcb + (sbase[k] + (unsigned)(o11 < lim ? o11 : lim)));
    };
    auto stage_store = [&](int p, int k, float *Vn) {
        const float sx = (p >> 2) == 1 ? 1.0f : -1.0f, sn = (p & 3) == 1 ? 1.0f : -1.0f;
        float4 v;
        v.x = __builtin_fmaf(__builtin_fmaf(st[3].x, sn, st[2].x), sx, __builtin_fmaf(st[1].x, sn, st[0].x));
        v.y = __builtin_fmaf(__builtin_fmaf(st[3].y, sn, st[2].y), sx, __builtin_fmaf(st[1].y, sn, st[0].y));
        v.z = __builtin_fmaf(__builtin_fmaf(st[3].z, sn, st[2].z), sx, __builtin_fmaf(st[1].z, sn, st[0].z));
        v.w = __builtin_fmaf(__builtin_fmaf(st[3].w, sn, st[2].w), sx, __builtin_fmaf(st[1].w, sn, st[0].w));
        float *q = Vn + 4 * sg[k] * VP + sslot[k];
        q[0] = v.x; q[VP] = v.y; q[2 * VP] = v.z; q[3 * VP] = v.w;
    };

    // ---- filter operands of phase ph = 2 p + h: WPH floats, contiguous in w2, copied as they lie by LDS-DMA; PPW 1 KB pieces per
    //      wave, addressed by the instruction offset (it advances the global and the LDS address alike).
    // Written as an asm statement: through __builtin_amdgcn_global_load_lds the compiler treats the copy as an LDS store that
    // every later ds_read may alias and waits for it (s_waitcnt vmcnt(0)) before the very next operand read -- the copy is then
    // no longer asynchronous.  The waits are placed by hand instead (wait_copies(), before the barrier that ends a phase).
    const unsigned wvoff = PRIV ? (unsigned)(lane * 16) : (unsigned)((wave * PPW) * 1024 + lane * 16);
    const unsigned wpiece = __builtin_amdgcn_readfirstlane((unsigned)(wave * PPW) * 1024u);
    const int cg_u = __builtin_amdgcn_readfirstlane(cg);
    auto w_issue = [&](int ph, float *Wb) {
        // shared: the phase's block as it lies; private: this wave's column block of the phase ([phase][cg][512 floats])
        const float *g = PRIV ? w2 + ((int64_t)ph * CG + cg_u) * 512 : w2 + (int64_t)ph * WPH;          // uniform
        const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)Wb) + wpiece;
        unsigned keep;
        if constexpr (PPW == 1)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(wvoff), "s"(l), "s"(g) : "memory");
        else if constexpr (PPW == 2)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                         "global_load_lds_dwordx4 %1, %3 offset:1024\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(wvoff), "s"(l), "s"(g) : "memory");
        else if constexpr (PPW == 3)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                         "global_load_lds_dwordx4 %1, %3 offset:1024\n\tglobal_load_lds_dwordx4 %1, %3 offset:2048\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(wvoff), "s"(l), "s"(g) : "memory");
        else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                         "global_load_lds_dwordx4 %1, %3 offset:1024\n\tglobal_load_lds_dwordx4 %1, %3 offset:2048\n\t"
                         "global_load_lds_dwordx4 %1, %3 offset:3072\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(wvoff), "s"(l), "s"(g) : "memory");
    };
    static_assert(PPW >= 1 && PPW <= 4, "LDS-DMA pieces per wave and phase");
    // the copies of a phase are issued BEFORE its patch loads: vmcnt(4) retires them and leaves the four patch loads in flight
    auto wait_copies = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    auto wait_copies_keep4 = [&]() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); };
    // this lane's operand quads in a phase buffer: [chunk][k half][quad][CGR channels][4]
    const int wlane = PRIV ? wave * (PPW * 256) + ((hh * 2) * 32 + r) * 4 : ((hh * 2) * CGR + cg * 32 + r) * 4;
    constexpr int CB = 32 * NBW;          // output channels of a wave

#ifdef SWK_WINO_STAMP
    unsigned long long stamp_sum[6] = {0, 0, 0, 0, 0, 0}, stamp_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
#endif
    int64_t task = blockIdx.x;
    if (task < ntasks) {
        stage_setup(task);
#pragma unroll
        for (int k = 0; k < NBW; ++k) { stage_issue(0, k); stage_store(0, k, V0); }
        w_issue(0, W0);
    }
    wait_copies();
    __syncthreads();
    for (; task < ntasks; task += gridDim.x) {
        // ---- this lane's tile as the matrix cores see it: destination of its 2 x 2 outputs ----
        const int64_t m = task * SLOTS + tg * 32 + r;
        const bool valid = m < ntiles;
        const int64_t mm = valid ? m : ntiles - 1;
        const unsigned bu = fTT.div((unsigned)mm), rem = (unsigned)mm - bu * (unsigned)TT;
        const int ty = (int)fT.div(rem), tx = (int)(rem - (unsigned)ty * (unsigned)T);
        const int64_t b = bu;
        const int64_t ro = ((b * dH + off_y + 2 * ty) * dW + off_x + 2 * tx) * (int64_t)dC + c_off + CB * cg + 4 * hh;
        const bool vy1 = 2 * ty + 1 < o, vx1 = 2 * tx + 1 < o;
        const bool more = task + gridDim.x < ntasks;
        f16v Y[2][2][NBW];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) Y[i][j][nb][e] = 0.0f;
        // One phase = SPP k-chunks of one 32-channel column block of one position: 8 SPP dependent MFMAs into M (the waves
        // that share a SIMD alternate: a dependent chain costs nothing, profiles/r2_f32_pipe_probe.txt).  After the last phase of
        // a (position, column block): Y_ij[nb] += c_ij M for the pairs the position feeds.
        // The operands of k-chunk s + 1 are read from LDS while chunk s multiplies (two register sets).  (Pinning the reads one MFMA
        // pair ahead inside a chunk as well changed nothing: with two to four waves per SIMD the LDS latency is covered.)
        auto phase = [&](const float *Wc, const float *Vc, int sub0, f16v M) -> f16v {
            const float *wq = Wc + wlane;
            const float *vrow = Vc + (sub0 * 16 + 8 * hh) * VP + tg * 32 + r;
            float4 w0[2], w1[2];
            float bv[2][8];
            w0[0] = *(const float4 *)wq;
            w1[0] = *(const float4 *)(wq + 4 * WCG);
#pragma unroll
            for (int i = 0; i < 8; ++i) bv[0][i] = vrow[i * VP];
#pragma unroll
            for (int sub = 0; sub < SPP; ++sub) {
                const int c = sub & 1, n = c ^ 1;
                if (sub + 1 < SPP) {
                    w0[n] = *(const float4 *)(wq + (sub + 1) * (16 * WCG));
                    w1[n] = *(const float4 *)(wq + (sub + 1) * (16 * WCG) + 4 * WCG);
#pragma unroll
                    for (int i = 0; i < 8; ++i) bv[n][i] = vrow[((sub + 1) * 16 + i) * VP];
                }
                M = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[c].x, bv[c][0], M, 0, 0, 0);
                M = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[c].y, bv[c][1], M, 0, 0, 0);
                M = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[c].z, bv[c][2], M, 0, 0, 0);
                M = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[c].w, bv[c][3], M, 0, 0, 0);
                M = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[c].x, bv[c][4], M, 0, 0, 0);
                M = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[c].y, bv[c][5], M, 0, 0, 0);
                M = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[c].z, bv[c][6], M, 0, 0, 0);
                M = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[c].w, bv[c][7], M, 0, 0, 0);
                if (sub + 1 < SPP) __builtin_amdgcn_sched_group_barrier(0x100 /* DS read */, 6, 0);
                if (SPP > 1) __builtin_amdgcn_sched_group_barrier(0x008 /* MFMA */, 8, 0);
            }
            return M;
        };
        for (int p = 0; p < 16; ++p) {
            const float *Vc = (p & 1) ? V1 : V0;
            float *Vn = (p & 1) ? V0 : V1;
            const int pn = (p + 1) & 15;
            if (p == 15 && more) stage_setup(task + gridDim.x);
            // Y_ij += A^T[i][xi] A^T[j][nu] M_p,   A^T = [1 1 1 0; 0 1 -1 -1]
            const int xi = p >> 2, nu = p & 3;
            const float ax[2] = {xi < 3 ? 1.0f : 0.0f, xi == 0 ? 0.0f : xi == 1 ? 1.0f : -1.0f};
            const float an[2] = {nu < 3 ? 1.0f : 0.0f, nu == 0 ? 0.0f : nu == 1 ? 1.0f : -1.0f};
            // (after a workgroup's last position the loads below fetch position 0 of the same tiles again, unused: conditional
            //  loads would make the compiler wait for them where the branches join, i.e. at once)
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                const f16v zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                f16v M = zero;
#pragma unroll
                for (int h = 0; h < PHS; ++h) {
                    // ---- phase q of the position: the next phase's filter operands travel while this one multiplies; the two items
                    //      of the next position's patch pixels travel during its first and its second half ----
                    const int q = nb * PHS + h;
                    int gn = p * PPOS + q + 1;
                    if (gn == 16 * PPOS) gn = 0;
                    SWK_STAMP(5);
                    const int par = (p * PPOS + q) & 1;          // phase buffers alternate over the whole sequence (PPOS may be odd)
                    w_issue(gn, par ? W0 : W1);
                    if (h == 0) stage_issue(pn, nb);          // item nb: issued in the first, stored in the last phase of block nb
                    SWK_STAMP(0);
                    M = phase(par ? W1 : W0, Vc, h * SPP, M);
                    SWK_STAMP(1);
                    if (h == PHS - 1) stage_store(pn, nb, Vn);
                    if (h == PHS - 1) {
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const float c = ax[i] * an[j];
                                if (c != 0.0f) {          // uniform; 36 of the 64 (position, pair) combinations
#pragma unroll
                                    for (int e = 0; e < 16; ++e) Y[i][j][nb][e] = __builtin_fmaf(M[e], c, Y[i][j][nb][e]);
                                }
                            }
                    }
                    SWK_STAMP(2);
                    // patch loads issued in this phase and stored in a later one stay in flight across the barrier
                    if (h == 0 && PHS > 1) wait_copies_keep4();
                    else wait_copies();
                    SWK_STAMP(3);
                    // shared filter copies: every wave's pieces have landed, the other buffers are free.  Private ones: the
                    // workgroup meets only when the next position's V is complete
                    if (!PRIV || q == PPOS - 1) __syncthreads();
                    SWK_STAMP(4);
                }
            }
        }
        // ---- bias + ReLU + placement: register quads = four consecutive output channels of the lane's tile ----
        if (valid) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if ((i == 1 && !vy1) || (j == 1 && !vx1)) continue;
                    float *q = dst + ro + ((int64_t)i * dW + j) * dC;
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int c = CB * cg + 32 * nb + 8 * g + 4 * hh;
                            if (c < cout) {
                                const float4 b4 = *(const float4 *)(lbias + c);
                                float4 v;
                                v.x = fmaxf(Y[i][j][nb][4 * g] + b4.x, 0.0f);
                                v.y = fmaxf(Y[i][j][nb][4 * g + 1] + b4.y, 0.0f);
                                v.z = fmaxf(Y[i][j][nb][4 * g + 2] + b4.z, 0.0f);
                                v.w = fmaxf(Y[i][j][nb][4 * g + 3] + b4.w, 0.0f);
                                *(float4 *)(q + 32 * nb + 8 * g) = v;
                            }
                        }
                }
        }
    }
#ifdef SWK_WINO_STAMP
    SWK_STAMP(5);
    if (lane == 0 && g_wino_stamp) {
        unsigned long long *o = g_wino_stamp + ((int64_t)blockIdx.x * NW + wave) * 6;
        for (int i = 0; i < 6; ++i) o[i] = stamp_sum[i];
    }
#endif
}

// column blocks per wave of the configuration a shape runs on (the filter layout depends on it): one for every Fire shape
static int wino_nbw(int, int) { return 1; }
// private per-wave filter slices (their own layout): the 64 -> 256 configuration with one column block per wave
static bool wino_priv(int cin, int cout) { return cin == 64 && cout == 256 && wino_nbw(cin, cout) == 1; }

template <int NBLK, int NBW, int TGN, int SPP, int WPS, bool PRIV>
static int launch_wino3x3(hipStream_t s, const float *src, int n, int t, const float *w2, const float *bias, int cout, float *dst, int dH,
                          int dW, int dC, int off_y, int off_x, int c_off)
{
    constexpr int CIN = 8 * NBLK, SLOTS = 32 * TGN, NT = 64 * (NBLK / NBW) * TGN, CGR = 32 * (NBLK / NBW);
    constexpr int WPH = PRIV ? (NT / 64) * 16 * SPP * 32 : 16 * SPP * CGR;
    const size_t lds = (size_t)(2 * WPH + 2 * CIN * (SLOTS + 1) + 32 * NBLK) * sizeof(float);
    static unsigned long long attr_mask = 0;
    if (!ensure_dyn_lds((const void *)k_wino3x3_relu_place<NBLK, NBW, TGN, SPP, WPS, PRIV>, 160 * 1024 - 256, attr_mask)) return SWK_ERR_HIP;
    if ((int64_t)n * t * t * CIN * 4 >= ((int64_t)1 << 32)) return SWK_ERR_CAPACITY;          // 32-bit byte offsets into src (and tile indices)
    const int T = (t - 2 + 1) / 2;
    const int64_t ntiles = (int64_t)n * T * T;
    int64_t blocks = (ntiles + SLOTS - 1) / SLOTS;
    // persistent workgroups of three or four waves (~230 registers): two waves per SIMD = two workgroups per CU, whose barriers
    // are independent -- one's phase change (drain, update, barrier, first operand reads) is covered by the other's MFMAs
    const int64_t cap = 256 * ((4 * WPS) / (NT / 64));
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((k_wino3x3_relu_place<NBLK, NBW, TGN, SPP, WPS, PRIV>), dim3((unsigned)blocks), dim3(NT), lds, s, src, n, t, T, w2, bias, cout, dst, dH, dW,
                       dC, off_y, off_x, c_off, FastDiv((unsigned)(T * T)), FastDiv((unsigned)T));
    return hipGetLastError() == hipSuccess ? SWK_OK : SWK_ERR_HIP;
}

}  // namespace swk

#pragma GCC visibility push(default)
extern "C" {

#ifdef SWK_WINO_STAMP
int32_t swk_wino_stamp_buffer(unsigned long long *buf)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(swk::g_wino_stamp), &buf, sizeof(buf)) == hipSuccess ? SWK_OK : SWK_ERR_HIP;
}
#endif

int32_t swk_winograd_f2x2_3x3_weights(const float *weight, int32_t cout, int32_t cin, float *out)
{
    if (!weight || !out || cout < 1 || cin < 16 || (cin & 15)) return SWK_ERR_ARG;
    // operand layout of k_wino3x3_relu_place, input channel = 16 chunk + 8 (k half) + 4 quad + j, output channels padded to whole
    // column blocks; NBW = column blocks per wave of the kernel configuration this shape runs on (see the loop below)
    const bool priv = swk::wino_priv(cin, cout);
    const int NBW = swk::wino_nbw(cin, cout), CB = 32 * NBW, CG = (cout + CB - 1) / CB, CGR = 32 * CG, S = cin / 16;
    static const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    for (int64_t i = 0, e = (int64_t)16 * cin * NBW * CGR; i < e; ++i) out[i] = 0.0f;
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            const float *g = weight + ((int64_t)co * cin + ci) * 9;
            double tmp[4][3], U[4][4];
            for (int a = 0; a < 4; ++a)
                for (int c = 0; c < 3; ++c) tmp[a][c] = G[a][0] * g[c] + G[a][1] * g[3 + c] + G[a][2] * g[6 + c];
            for (int a = 0; a < 4; ++a)
                for (int c = 0; c < 4; ++c) U[a][c] = tmp[a][0] * G[c][0] + tmp[a][1] * G[c][1] + tmp[a][2] * G[c][2];
            const int sub = ci >> 4, hh = (ci >> 3) & 1, q = (ci >> 2) & 1, j = ci & 3;
            const int cg = co / CB, h = (co % CB) >> 5, r = co & 31;
            for (int p = 0; p < 16; ++p) {
                int64_t idx;
                if (priv) {              // [p][chunk][cg][k half][quad][r][4]: a wave's 2 KB of a phase are contiguous
                    idx = ((int64_t)p * S + sub) * CG + cg;
                    idx = ((idx * 2 + hh) * 2 + q) * 32 + r;
                } else {                 // [p][h][chunk][k half][quad][cg * 32 + r][4]: the phase's block as it sits in LDS
                    idx = (int64_t)p * NBW + h;
                    idx = idx * S + sub;
                    idx = (idx * 2 + hh) * 2 + q;
                    idx = idx * CGR + cg * 32 + r;
                }
                out[idx * 4 + j] = (float)U[p >> 2][p & 3];
            }
        }
    return SWK_OK;
}

int32_t swk_nhwc_conv3x3_winograd_bias_relu_place(void *stream, const float *src, int32_t n, int32_t t, int32_t cin, const float *weight_w,
                                                  const float *bias, int32_t cout, float *dst, int32_t dH, int32_t dW, int32_t dC,
                                                  int32_t off_y, int32_t off_x, int32_t c_off)
{
    if (!src || !weight_w || !bias || !dst || n < 1 || t < 3 || cout < 4 || (cout & 3) || (dC & 3) || (c_off & 3) || off_y < 0 || off_x < 0 ||
        off_y + t - 2 > dH || off_x + t - 2 > dW || c_off < 0 || c_off + cout > dC || (((uintptr_t)src | (uintptr_t)dst | (uintptr_t)weight_w) & 15))
        return SWK_ERR_ARG;
    using namespace swk;
    hipStream_t s = (hipStream_t)stream;
    // the squeeze ratio of SqueezeNet's Fire modules: 8 input channels per 32 output channels
#define SWK_W3_ARGS s, src, n, t, weight_w, bias, cout, dst, dH, dW, dC, off_y, off_x, c_off
    if (cin == 16 && cout == 64) return launch_wino3x3<2, 1, 2, 1, 4, false>(SWK_W3_ARGS);
    if (cin == 32 && cout == 128) return launch_wino3x3<4, 1, 1, 1, 4, false>(SWK_W3_ARGS);
    // (48 -> 192: one k-chunk per phase since round 4 -- three times the barriers of SPP = 3, a third of the filter buffers: 3 % faster;
    //  the other shapes re-checked against 4 / 8-wave workgroups and two chunks per phase: as they are)
    if (cin == 48 && cout == 192) return launch_wino3x3<6, 1, 2, 1, 3, false>(SWK_W3_ARGS);
    if (cin == 64 && cout == 256) return launch_wino3x3<8, 1, 1, 1, 4, true>(SWK_W3_ARGS);
#undef SWK_W3_ARGS
    return SWK_ERR_ARG;
}

}  // extern "C"
#pragma GCC visibility pop

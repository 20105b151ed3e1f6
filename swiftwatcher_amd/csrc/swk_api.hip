// C ABI of libswk.so (include/swk.h): context, workspaces, stage orchestration.
// Host side only; every kernel lives in ialm*.hip / filters.hip / ccl.hip.
#include "swk_internal.h"

#include <math.h>
#include <stdio.h>
#include <string.h>
#include <atomic>
#include <string>
#include <vector>

using namespace swk;

namespace swk { std::atomic<int> g_launch_error{0}; }

namespace {

std::string g_create_error;

enum Slot {
    SL_ROI = 0, SL_X, SL_S, SL_BIL, SL_THR, SL_OPEN, SL_LAB8, SL_LAB32, SL_A, SL_Y, SL_E, SL_PN,
    SL_BM, SL_VPREV, SL_GPART, SL_ZZPART, SL_WIN, SL_ACTIVE, SL_PARENT, SL_ROOTBITS, SL_WORDPREFIX,
    SL_NCOMP, SL_TABLE, SL_SUMS, SL_SEGS, SL_NSEG, SL_ITERS, SL_TMP_IN, SL_TMP_OUT, SL_COLORW, SL_SPACEW,
    SL_TAPDR, SL_TAPDC, SL_SALT, SL_WIDE, SL_REDO_X, SL_REDO_S, SL_REDO_X2, SL_REDO_S2, SL_SEGOFFS, SL_CL_CROPS, SL_CL_OFFS, SL_CL_HW, SL_CL_PATCH, SL_CL_NET, SL_COUNT
};

struct EventPair { hipEvent_t a, b; int fam; };

}  // namespace

struct swk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    void *slot[SL_COUNT] = {nullptr};
    size_t slot_bytes[SL_COUNT] = {0};
    int *h_active = nullptr;             // pinned
    // bilateral tables currently on the device
    int bil_d = -1; double bil_sc = -1, bil_ss = -1;
    BilateralTables bil{};
    // profiling
    bool prof_on = false;
    std::vector<EventPair> pending;
    std::vector<hipEvent_t> pool;
    double prof_ms[SWK_K_COUNT] = {0};
    int64_t prof_n[SWK_K_COUNT] = {0};
    int64_t window_iters = 0;
    int ialm_variant = 0;
    int pass_tune = 0;             // k-step-templated pass: bit 0 priority, bit 1 stagger for the odd hardware wave slot
    double sparse_spec = 16.0;     // M-state pass: sparse image stores start at 16 x tol (<= 0: every pass)
    double norm_guard = 1e-3;      // M-state pass: |ratio / tol - 1| below this does not decide (the window is rerun with the f64 norm)
    int64_t guard_windows = 0;     // windows rerun for that reason
    double start_refine = 1e-5;    // estimated first-iteration error above which a window gets the accurate start (ialm_refine.hip)
    int64_t refined_windows = 0, unrefined_windows = 0;
    int64_t redo_batches = 0;      // batches in which a guess of the M-state pass failed ...
    int64_t redo_windows = 0;      // ... and the windows that were run again for it
    int last_eig_sweeps = 0;       // largest IalmWin::sweeps of the last batch (Newton-Schulz iterations, or 100 + Jacobi sweeps)
    int last_int_start = 0;        // windows of the last batch whose first Gram matrix came from the integer matrix cores
    int use_gram8 = 1;             // M-state pass: first Gram matrix from k_gram_u8 (A/B knob)
    unsigned long long pass_b16 = 0;   // sum over windows of IalmWin::pass_b16 since the last swk_prof_reset
    double norm_spec = 256.0;      // M-state pass: ||Z|| every other iteration while above 256 x tol (<= 0: every iteration)
    int sparse_backoff = 0, norm_backoff = 0;   // batches for which a guess stays off after it failed (same video, same behaviour)
    int eig_method = 0;                  // 0 Newton-Schulz (MFMA), 1 Jacobi
    hipEvent_t ev_poll[2] = {nullptr, nullptr};   // the host polls convergence two iterations late (run_ialm)
    std::vector<IalmWin> last_hw;        // host copy of the last batch's per-window IALM state (gather_iters): diagnostics
    IalmWin *last_win = nullptr;         // per-window IALM state of the last run
    int last_nwin = 0;
    int64_t pstride = 0;                 // plane pitch of the A/Y/E workspaces of the last IALM run
    int fpad = 0;                        // planes per window in them
    // what the last swk_batch_run left on the device for swk_segment_inputs_last: its frames (SL_ROI copy of a host input,
    // or the caller's device frames) and region records; valid until a call reuses those buffers
    struct LastBatch {
        bool valid = false;
        const uint8_t *frames = nullptr;
        int64_t fs = 0, rs = 0;
        int nwin = 0, n = 0, Hc = 0, Wc = 0, x0 = 0, y0 = 0, frame_h = 0, frame_w = 0, cap = 0;
        int total = -1;          // segments of the batch (regions beyond cap not counted), when nseg was copied to the host
        const swk_segment *segs = nullptr;
        const int32_t *nseg = nullptr;
    } last;
};

namespace {

#define HIPCHK(ctx, call)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            char buf_[512];                                                                    \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            (ctx)->err = buf_;                                                                 \
            return SWK_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

int fail(swk_ctx *ctx, int code, const char *msg)
{
    if (ctx) ctx->err = msg;
    return code;
}

int need(swk_ctx *ctx, Slot s, size_t bytes, void **out)
{
    if (bytes == 0) bytes = 16;
    if (ctx->slot_bytes[s] < bytes) {
        if (ctx->slot[s]) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(ctx->slot[s])); }
        ctx->slot[s] = nullptr;
        ctx->slot_bytes[s] = 0;
        hipError_t e = hipMalloc(&ctx->slot[s], bytes);
        if (e != hipSuccess) {
            char buf[256];
            snprintf(buf, sizeof buf, "hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
            ctx->err = buf;
            return SWK_ERR_NOMEM;
        }
        ctx->slot_bytes[s] = bytes;
    }
    *out = ctx->slot[s];
    return SWK_OK;
}

#define NEED(ctx, slot, bytes, ptr)                                          \
    do { void *p_; int rc_ = need(ctx, slot, bytes, &p_); if (rc_) return rc_; ptr = (decltype(ptr))p_; } while (0)

// ---- profiling --------------------------------------------------------------------
hipEvent_t take_event(swk_ctx *ctx)
{
    if (!ctx->pool.empty()) { hipEvent_t e = ctx->pool.back(); ctx->pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

struct Timed {
    swk_ctx *ctx; int fam; hipStream_t st; hipEvent_t a{}, b{};
    Timed(swk_ctx *c, int f, hipStream_t stream = nullptr) : ctx(c), fam(f), st(stream ? stream : c->stream)
    {
        if (ctx->prof_on) { a = take_event(ctx); b = take_event(ctx); (void)hipEventRecord(a, st); }
    }
    ~Timed()
    {
        if (ctx->prof_on) { (void)hipEventRecord(b, st); ctx->pending.push_back({a, b, fam}); }
    }
};

int ensure_poll_events(swk_ctx *ctx)
{
    for (int i = 0; i < 2; ++i)
        if (!ctx->ev_poll[i]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_poll[i], hipEventDisableTiming));
    return SWK_OK;
}

void drain_prof(swk_ctx *ctx)
{
    for (auto &p : ctx->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { ctx->prof_ms[p.fam] += ms; ctx->prof_n[p.fam] += 1; }
        ctx->pool.push_back(p.a);
        ctx->pool.push_back(p.b);
    }
    ctx->pending.clear();
}

int sync(swk_ctx *ctx)
{
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipGetLastError());
    if (const int le = g_launch_error.exchange(0)) {          // a launcher could not set a kernel attribute or start a kernel
        HIPCHK(ctx, (hipError_t)le);
    }
    drain_prof(ctx);
    return SWK_OK;
}

// ---- bilateral weight tables (OpenCV 4.1.0 bilateralFilter_8u set-up, host side) ----
int ensure_bilateral(swk_ctx *ctx, int d, double sigma_color, double sigma_space)
{
    if (ctx->bil_d == d && ctx->bil_sc == sigma_color && ctx->bil_ss == sigma_space) return SWK_OK;
    double sc = sigma_color <= 0 ? 1 : sigma_color, ss = sigma_space <= 0 ? 1 : sigma_space;
    const double gc = -0.5 / (sc * sc), gs = -0.5 / (ss * ss);
    int radius = d <= 0 ? (int)lrint(ss * 1.5) : d / 2;
    if (radius < 1) radius = 1;
    if (radius > 4) return fail(ctx, SWK_ERR_ARG, "bilateral diameter > 9 is not supported");
    float cw[256], sw[81];
    int8_t dr[81], dc[81];
    for (int i = 0; i < 256; ++i) cw[i] = (float)exp((double)(i * i) * gc);
    int k = 0;
    for (int i = -radius; i <= radius; ++i)
        for (int j = -radius; j <= radius; ++j) {
            const double r = sqrt((double)i * i + (double)j * j);
            if (r > radius) continue;
            sw[k] = (float)exp(r * r * gs);
            dr[k] = (int8_t)i; dc[k] = (int8_t)j;
            ++k;
        }
    NEED(ctx, SL_COLORW, sizeof cw, ctx->bil.color_w);
    NEED(ctx, SL_SPACEW, sizeof sw, ctx->bil.space_w);
    NEED(ctx, SL_TAPDR, sizeof dr, ctx->bil.tap_dr);
    NEED(ctx, SL_TAPDC, sizeof dc, ctx->bil.tap_dc);
    HIPCHK(ctx, hipMemcpyAsync(ctx->bil.color_w, cw, sizeof cw, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->bil.space_w, sw, sizeof(float) * k, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->bil.tap_dr, dr, k, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->bil.tap_dc, dc, k, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));     // cw/sw live on this stack frame
    ctx->bil.maxk = k;
    ctx->bil.radius = radius;
    ctx->bil_d = d; ctx->bil_sc = sigma_color; ctx->bil_ss = sigma_space;
    return SWK_OK;
}

int ensure_ccl(swk_ctx *ctx, int F, int H, int W, CclBuffers *b)
{
    b->Pp = (int)ccl_padded(H, W);
    b->words = (int)ccl_words(H, W);
    NEED(ctx, SL_PARENT, (size_t)F * b->Pp * 4, b->parent);
    NEED(ctx, SL_ROOTBITS, (size_t)F * b->words * 4, b->rootbits);
    NEED(ctx, SL_WORDPREFIX, (size_t)F * b->words * 4, b->wordprefix);
    NEED(ctx, SL_NCOMP, (size_t)F * 4, b->ncomp);
    NEED(ctx, SL_TABLE, (size_t)F * 256 * 8 * 4, b->table);
    NEED(ctx, SL_SUMS, (size_t)F * 256 * 2 * 8, b->sums);
    return SWK_OK;
}

// ---- IALM driver ----------------------------------------------------------------------
int run_ialm(swk_ctx *ctx, const uint8_t *dX, int nwin, int n, int P, double lmbda, double tol, int maxiter,
             bool want_A, bool want_E, uint8_t *dS, int32_t *h_iters /*host, optional*/, int32_t *d_iters /*device, optional*/,
             bool speculate = true, int force_variant = 0)
{
    if (n < 1 || n > kMaxNWide) return fail(ctx, SWK_ERR_ARG, "frames per window must be in 1..128");
    const bool wide = n > kMaxN;          // 65 .. 128 frames: the plain f64 kernels (A/Y state, Jacobi in global memory)
    IalmBuffers b{};
    b.X = dX; b.S = dS; b.nwin = nwin; b.n = n; b.P = P;
    // auto: the M-state pass (k-step-templated, 21 B/element) unless the caller wants the f64 low-rank / sparse matrices,
    // which only the A/Y-state pass (v2, 34 B/element) materialises
    int variant = force_variant ? force_variant : ctx->ialm_variant;
    if (variant == 0) variant = 4;
    if (variant >= 4 && variant != 6 && (want_A || want_E)) variant = 2;
    if (wide) variant = 6;
    const bool mstate = variant == 4 || variant == 5;          // the M-state pass (ialm_mstate.hip), with / without the software pipeline
    b.nblk = ialm_pass_nblk(variant, n, P, nwin);   // blocks per window, sized per launch
    b.nred = b.nblk > 4 ? 1 : b.nblk;       // several slabs: reduce them chip-wide first (k_gram_reduce)
    b.pstride = ((int64_t)P + 127) & ~(int64_t)127;      // whole groups of 8 tiles
    b.fpad = mstate ? ialm_mstate_fpad(n) : (n + 15) & ~15;
    ctx->pstride = b.pstride;
    ctx->fpad = b.fpad;
    if ((int64_t)b.fpad * b.pstride >= (1ll << 28)) return fail(ctx, SWK_ERR_ARG, "window too large: frames x ROI pixels must stay below 2^28");
    const size_t elems = (size_t)nwin * n * P;
    const size_t felems = (size_t)nwin * b.fpad * b.pstride;
    NEED(ctx, SL_A, felems * 8, b.A);
    NEED(ctx, SL_Y, felems * 8, b.Y);
    if (mstate) {
        b.U = (uint16_t *)b.Y;               // binary16 planes in the Y slot
        b.spec = (speculate && ctx->sparse_backoff == 0) ? ctx->sparse_spec : 0.0;
        b.nspec = (speculate && ctx->norm_backoff == 0) ? ctx->norm_spec : 0.0;
        b.guard = ctx->norm_guard;
        if (speculate) {
            if (ctx->sparse_backoff > 0) ctx->sparse_backoff -= 1;
            if (ctx->norm_backoff > 0) ctx->norm_backoff -= 1;
        }
        NEED(ctx, SL_SALT, elems, b.Salt);
    }
    if (want_E) NEED(ctx, SL_E, felems * 8, b.E);
    NEED(ctx, SL_BM, (size_t)nwin * n * n * 8, b.Bm);
    NEED(ctx, SL_VPREV, (size_t)nwin * n * n * 8, b.Vprev);
    NEED(ctx, SL_GPART, (size_t)nwin * b.nblk * n * n * 8, b.gpart);
    NEED(ctx, SL_ZZPART, (size_t)2 * nwin * b.nblk * 8, b.zzpart);          // [nwin][nblk] sums of z^2, then [nwin][nblk] max |U| (M-state pass)
    NEED(ctx, SL_WIN, (size_t)nwin * sizeof(IalmWin), b.win);
    NEED(ctx, SL_ACTIVE, 16 * sizeof(int), b.active);
    double *wide_work = nullptr;
    if (variant == 6) NEED(ctx, SL_WIDE, (size_t)nwin * ialm_small_wide_doubles(n) * sizeof(double), wide_work);
    hipStream_t s = ctx->stream;
    HIPCHK(ctx, hipMemsetAsync(b.win, 0, (size_t)nwin * sizeof(IalmWin), s));
    HIPCHK(ctx, hipMemsetAsync(b.active, 0, 16 * sizeof(int), s));
    HIPCHK(ctx, hipMemsetAsync(dS, 0, elems, s));
    if (mstate) {
        HIPCHK(ctx, hipMemsetAsync(b.Salt, 0, elems, s));
    } else {
        // a window that stops before writing A (all-zero input) must still read back zeros
        HIPCHK(ctx, hipMemsetAsync(b.A, 0, felems * 8, s));
        // padded frame planes (n..fpad-1) of Y are read by the MFMA pass and must contribute zeros
        if (b.fpad != n) HIPCHK(ctx, hipMemsetAsync(b.Y, 0, felems * 8, s));
    }
    if (want_E) HIPCHK(ctx, hipMemsetAsync(b.E, 0, felems * 8, s));

    // One chain on the context's stream: start, then per iteration pass -> (slab sum) -> small-matrix step.  (Rounds 1 / 2 cut the
    // batch into window groups whose eigen-solves ran on CU-masked side streams beside the other groups' passes; with the
    // Newton-Schulz solver the small-matrix kernel is ~3 % of a step and the overlap stopped paying: removed in round 3.)
    int rc = ensure_poll_events(ctx);
    if (rc) return rc;
    const int check_from = 6;     // no window converges earlier (mu grows 1.5x per iteration)
    // window statistics (||X||_F, max) and, for the M-state pass, the first Gram matrix in the same read of X
    // on the integer matrix cores; windows it does not cover get the f64 start pass below
    b.use_gram8 = ((mstate || variant == 2 || variant == 1) && ctx->use_gram8 && gram_u8_supported(b)) ? 1 : 0;
    b.refine = wide ? 0.0 : ctx->start_refine;
    { Timed t(ctx, SWK_K_IALM_STATS);
      if (b.use_gram8) launch_gram_u8(s, b); else launch_ialm_stats(s, b);
      launch_ialm_init(s, b, lmbda); }
    // the Gram-only start pass reads X alone (1 B/element): booked with the statistics family so
    // SWK_K_IALM_PASS times only the full streaming passes
    { Timed t(ctx, SWK_K_IALM_STATS); launch_ialm_pass(s, b, 0, variant, 0, ctx->pass_tune); }
    auto small_step = [&](int k) {
        Timed t(ctx, SWK_K_IALM_SMALL);
        if (b.nblk > 4) launch_gram_reduce(s, b);
        if (variant == 6) launch_ialm_small_wide(s, b, k, lmbda, tol, maxiter, wide_work);
        else launch_ialm_small(s, b, k, lmbda, tol, maxiter, ctx->eig_method);
    };
    small_step(0);
    // windows that step found ill-conditioned get their first iteration's matrix B_1 again, from a double-double Cholesky factor
    // of the exact integer X^T X -- of a double-double M_1^T M_1 where the window had no integer start -- (one workgroup per flagged
    // window; the others leave at the first branch)
    if (b.refine > 0.0) { Timed t(ctx, SWK_K_IALM_SMALL); launch_ialm_refine_start(s, b); }
    for (int k = 1; k <= maxiter + 2; ++k) {
        // convergence is polled two iterations late so the host never stalls the queue; the launches made
        // meanwhile for an already finished batch return at their first branch
        const int kc = k - 2;
        if (kc >= check_from) {
            HIPCHK(ctx, hipEventSynchronize(ctx->ev_poll[kc & 1]));
            if (ctx->h_active[kc & 1] <= 0) break;
        }
        if (k > maxiter) break;
        { Timed t(ctx, SWK_K_IALM_PASS); launch_ialm_pass(s, b, k == 1 ? 1 : 2, variant, k, ctx->pass_tune); }
        small_step(k);
        if (k >= check_from) {
            HIPCHK(ctx, hipMemcpyAsync(&ctx->h_active[k & 1], b.active, sizeof(int), hipMemcpyDeviceToHost, s));
            HIPCHK(ctx, hipEventRecord(ctx->ev_poll[k & 1], s));
        }
    }
    if (mstate) { Timed t(ctx, SWK_K_IALM_STATS); launch_select_sparse(s, b); }
    ctx->last_win = b.win;
    ctx->last_nwin = nwin;
    if (mstate && (b.spec > 0.0 || b.nspec > 0.0 || b.guard > 0.0)) {
        // Windows the M-state pass could not finish on its own terms (IalmWin::redo): bits 0 / 1 = a guess failed (the window stopped
        // right after a pass that had its sparse-image stores switched off, or a partial norm could not rule out that an iteration
        // was the last); bit 2 = the float32 stopping norm fell inside the guard band around the tolerance.  Only THOSE windows run
        // again -- the first kind together, in one call with the guesses off; the second kind through the A/Y-state pass (norm in
        // float64, statement by statement the reference's :293-297) -- and their sparse images, iteration counts and diagnostics
        // replace the windows' entries.  (Until round 3 one failed guess reran the whole batch.)
        std::vector<IalmWin> hw(nwin);
        HIPCHK(ctx, hipMemcpyAsync(hw.data(), b.win, (size_t)nwin * sizeof(IalmWin), hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipStreamSynchronize(s));
        std::vector<int> guess, band;
        int redo = 0;
        for (int w = 0; w < nwin; ++w) {
            redo |= hw[w].redo;
            if (hw[w].redo & 3) guess.push_back(w);
            else if (hw[w].redo & 4) band.push_back(w);
        }
        if (redo & 3) {
            // windows of one video behave alike: a guess that failed stays off for the next batches
            ctx->redo_batches += 1;
            ctx->redo_windows += (int64_t)guess.size();
            if (redo & 1) ctx->sparse_backoff = 64;
            if (redo & 2) ctx->norm_backoff = 64;
        }
        ctx->guard_windows += (int64_t)band.size();
        const int64_t pstride = ctx->pstride;
        const int fpad = ctx->fpad;
        for (int kind = 0; kind < 2; ++kind) {
            const std::vector<int> &list = kind == 0 ? guess : band;
            if (list.empty()) continue;
            const int cnt = (int)list.size();
            const size_t wbytes = (size_t)n * P;
            // the listed windows side by side: their pixels gathered, their sparse images scattered back
            // (a slot pair per kind: the run of the first kind may itself send windows of ITS batch through the second)
            uint8_t *gx, *gs;
            NEED(ctx, kind == 0 ? SL_REDO_X : SL_REDO_X2, (size_t)cnt * wbytes + 4, gx);
            NEED(ctx, kind == 0 ? SL_REDO_S : SL_REDO_S2, (size_t)cnt * wbytes, gs);
            for (int i = 0; i < cnt; ++i)
                HIPCHK(ctx, hipMemcpyAsync(gx + (size_t)i * wbytes, dX + (size_t)list[i] * wbytes, wbytes, hipMemcpyDeviceToDevice, s));
            const int rc1 = run_ialm(ctx, gx, cnt, n, P, lmbda, tol, maxiter, false, false, gs, nullptr, nullptr, false, kind == 0 ? variant : 2);
            if (rc1) return rc1;
            std::vector<IalmWin> sub(cnt);
            HIPCHK(ctx, hipMemcpyAsync(sub.data(), ctx->last_win, (size_t)cnt * sizeof(IalmWin), hipMemcpyDeviceToHost, s));
            for (int i = 0; i < cnt; ++i)
                HIPCHK(ctx, hipMemcpyAsync(dS + (size_t)list[i] * wbytes, gs + (size_t)i * wbytes, wbytes, hipMemcpyDeviceToDevice, s));
            HIPCHK(ctx, hipStreamSynchronize(s));
            for (int i = 0; i < cnt; ++i) {
                IalmWin one = sub[i];
                one.pass_b16 += hw[list[i]].pass_b16;          // the roofline books every pass a window ran, the abandoned ones included
                hw[list[i]] = one;
            }
        }
        if (!guess.empty() || !band.empty()) {
            // the nested runs used the window-state slot and the A / Y workspaces for their own (smaller) batches: this batch's
            // entries go back, and the context describes THIS batch again
            NEED(ctx, SL_WIN, (size_t)nwin * sizeof(IalmWin), b.win);
            HIPCHK(ctx, hipMemcpyAsync(b.win, hw.data(), (size_t)nwin * sizeof(IalmWin), hipMemcpyHostToDevice, s));
            HIPCHK(ctx, hipStreamSynchronize(s));
            ctx->pstride = pstride;
            ctx->fpad = fpad;
            ctx->last_win = b.win;
            ctx->last_nwin = nwin;
        }
    }
    (void)h_iters; (void)d_iters;
    return SWK_OK;
}

// Iteration counts live in the per-window state structs.  Called once the stream has drained.
int gather_iters(swk_ctx *ctx, int32_t *h_iters, int32_t *d_iters)
{
    const int nwin = ctx->last_nwin;
    std::vector<IalmWin> hw(nwin);
    // (on the context's own, non-blocking stream: a copy on the null stream would wait for every blocking stream of the process, and
    //  fails outright while another thread captures a HIP graph on one -- the classifier does, segment_classification.py)
    HIPCHK(ctx, hipMemcpyAsync(hw.data(), ctx->last_win, (size_t)nwin * sizeof(IalmWin), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<int32_t> it(nwin);
    ctx->last_hw = hw;
    ctx->last_int_start = 0;
    ctx->last_eig_sweeps = 0;
    for (int w = 0; w < nwin; ++w) {
        it[w] = hw[w].iter; ctx->window_iters += hw[w].iter; ctx->pass_b16 += hw[w].pass_b16;
        ctx->last_int_start += hw[w].int_gram ? 1 : 0;
        if (hw[w].refine == 2) ctx->refined_windows += 1;
        else if (hw[w].refine != 0) ctx->unrefined_windows += 1;
        if (hw[w].sweeps > ctx->last_eig_sweeps) ctx->last_eig_sweeps = hw[w].sweeps;
    }
    if (h_iters) memcpy(h_iters, it.data(), (size_t)nwin * 4);
    if (d_iters) {
        HIPCHK(ctx, hipMemcpyAsync(d_iters, it.data(), (size_t)nwin * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));          // `it` is a local
    }
    return SWK_OK;
}

int segment_inputs_impl(swk_ctx *ctx, const uint8_t *frames, int64_t fs, int64_t rs, int F, int x0, int y0, int frame_h, int frame_w,
                        const swk_segment *segs, const int32_t *nseg, int seg_cap, int min_h, int min_w, const float *mean,
                        const float *std_, int pad, bool nhwc, int first, int net_cap, float *net, int32_t *seg_frame, int32_t *total,
                        int32_t *skipped, int known_total = -1)
{
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    int32_t *doffs;
    NEED(ctx, SL_SEGOFFS, ((size_t)F + 2) * 4, doffs);           // [F+1] prefix sums, then the skipped-box counter
    int32_t *dskip = doffs + F + 1;
    HIPCHK(ctx, hipMemsetAsync(dskip, 0, 4, s));
    launch_segment_prefix(s, nseg, F, seg_cap, doffs);
    int32_t tot = known_total;
    if (tot < 0) {          // the caller does not know how many segments the batch holds: one round trip
        HIPCHK(ctx, hipMemcpyAsync(&tot, doffs + F, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipStreamSynchronize(s));
    }
    *total = tot;
    if (skipped) *skipped = 0;
    int count = tot - first;
    if (count > net_cap) count = net_cap;
    if (count < 1) return SWK_OK;
    launch_segment_inputs(s, frames, fs, rs, frame_h, frame_w, x0, y0, segs, doffs, F, seg_cap,
                          min_h, min_w, first, count, net, seg_frame, pad, nhwc, mean, std_, dskip);
    int32_t sk = 0;
    HIPCHK(ctx, hipMemcpyAsync(&sk, dskip, 4, hipMemcpyDeviceToHost, s));
    int rc = sync(ctx);
    if (rc) return rc;
    if (skipped) *skipped = sk;
    return SWK_OK;
}

int copy_out(swk_ctx *ctx, void *dst, const void *src, size_t bytes, int mem)
{
    if (!dst || dst == src) return SWK_OK;
    HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, mem == SWK_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice,
                               ctx->stream));
    return SWK_OK;
}

}  // namespace

// =====================================================================================
#pragma GCC visibility push(default)
extern "C" {

int32_t swk_abi_version(void) { return SWK_ABI_VERSION; }

void swk_params_default(swk_params *p)
{
    memset(p, 0, sizeof *p);
    p->lmbda = 0.01; p->tol = 0.001; p->maxiter = 100;
    p->bil_d = 7; p->bil_sigma_color = 15.0; p->bil_sigma_space = 1.0; p->bil_fma = 0;
    p->thresh = 15; p->open_kh = 3; p->open_kw = 3;
    p->connectivity = 8; p->label_order = SWK_ORDER_BLOCK2X2; p->gray_mode = SWK_GRAY_Q14;
}

const char *swk_last_error(const swk_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int32_t swk_ctx_create(int32_t device, int32_t max_windows, int32_t max_n, int32_t max_Hc, int32_t max_Wc, swk_ctx **out)
{
    if (!out) return SWK_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        g_create_error = "no HIP device visible: libswk has no CPU fallback";
        return SWK_ERR_NOGPU;
    }
    if (device < 0 || device >= count) { g_create_error = "device index out of range"; return SWK_ERR_ARG; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { g_create_error = "hipGetDeviceProperties failed"; return SWK_ERR_HIP; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("device is ") + prop.gcnArchName + ", libswk is built for gfx950 (MI355X) only";
        return SWK_ERR_NOGPU;
    }
    if (hipSetDevice(device) != hipSuccess) { g_create_error = "hipSetDevice failed"; return SWK_ERR_HIP; }
    swk_ctx *ctx = new swk_ctx();
    ctx->device = device;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipHostMalloc((void **)&ctx->h_active, 256, hipHostMallocDefault) != hipSuccess) {
        g_create_error = "stream / pinned memory creation failed";
        delete ctx;
        return SWK_ERR_HIP;
    }
    // pre-size the big workspaces so the first batch does not pay for hipMalloc
    if (max_windows > 0 && max_n > 0 && max_Hc > 0 && max_Wc > 0) {
        const size_t elems = (size_t)max_windows * max_n * max_Hc * max_Wc;
        void *p;
        int rc = 0;
        rc = rc ? rc : need(ctx, SL_X, elems + 4, &p);
        rc = rc ? rc : need(ctx, SL_S, elems, &p);
        rc = rc ? rc : need(ctx, SL_OPEN, elems, &p);
        rc = rc ? rc : need(ctx, SL_LAB8, elems, &p);
        const size_t felems = (size_t)max_windows * ((max_n + 15) & ~15) * (((size_t)max_Hc * max_Wc + 127) & ~(size_t)127);
        rc = rc ? rc : need(ctx, SL_A, felems * 8, &p);
        rc = rc ? rc : need(ctx, SL_Y, felems * 8, &p);
        if (rc) { g_create_error = ctx->err; swk_ctx_destroy(ctx); return rc; }
    }
    *out = ctx;
    return SWK_OK;
}

void swk_ctx_destroy(swk_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    drain_prof(ctx);
    for (auto e : ctx->pool) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; ++i)
        if (ctx->ev_poll[i]) (void)hipEventDestroy(ctx->ev_poll[i]);
    for (int i = 0; i < SL_COUNT; ++i)
        if (ctx->slot[i]) (void)hipFree(ctx->slot[i]);
    if (ctx->h_active) (void)hipHostFree(ctx->h_active);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int64_t swk_ctx_device_bytes(const swk_ctx *ctx)
{
    int64_t t = 0;
    if (ctx) for (int i = 0; i < SL_COUNT; ++i) t += (int64_t)ctx->slot_bytes[i];
    return t;
}

int32_t swk_prof_enable(swk_ctx *ctx, int32_t on) { if (!ctx) return SWK_ERR_ARG; ctx->prof_on = on != 0; return SWK_OK; }
int32_t swk_prof_reset(swk_ctx *ctx)
{
    if (!ctx) return SWK_ERR_ARG;
    memset(ctx->prof_ms, 0, sizeof ctx->prof_ms);
    memset(ctx->prof_n, 0, sizeof ctx->prof_n);
    ctx->window_iters = 0;
    ctx->pass_b16 = 0;
    return SWK_OK;
}
int32_t swk_prof_get(swk_ctx *ctx, int32_t family, double *ms_total, int64_t *launches)
{
    if (!ctx || family < 0 || family >= SWK_K_COUNT) return SWK_ERR_ARG;
    if (ms_total) *ms_total = ctx->prof_ms[family];
    if (launches) *launches = ctx->prof_n[family];
    return SWK_OK;
}
int32_t swk_prof_window_iters(swk_ctx *ctx, int64_t *window_iters)
{
    if (!ctx || !window_iters) return SWK_ERR_ARG;
    *window_iters = ctx->window_iters;
    return SWK_OK;
}
int32_t swk_set_ialm_variant(swk_ctx *ctx, int32_t variant)
{
    if (!ctx || variant < 0 || variant > 6 || variant == 3) return SWK_ERR_ARG;          // 3 was round 1's M-state kernel (removed)
    ctx->ialm_variant = variant;
    return SWK_OK;
}

// Page-locked host memory for the caller's staging buffers: a host -> device copy out of it is one DMA instead of a
// driver-side staging copy plus a DMA (the FrameQueue drop-in stacks a window's crops into such a buffer).
int32_t swk_pinned_alloc(int32_t device, int64_t bytes, void **out)
{
    if (!out || bytes < 1 || device < 0) return SWK_ERR_ARG;
    *out = nullptr;
    // on the GPU the buffer feeds (a thread that has not chosen a device would otherwise open a context on GPU 0); portable: every
    // context of the process may copy out of it
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return SWK_ERR_ARG; }
    if (hipHostMalloc(out, (size_t)bytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return SWK_ERR_NOMEM; }
    return SWK_OK;
}

int32_t swk_pinned_free(void *p)
{
    if (!p) return SWK_OK;
    return hipHostFree(p) == hipSuccess ? SWK_OK : SWK_ERR_HIP;
}

int32_t swk_device_alloc(swk_ctx *ctx, int64_t bytes, void **out)
{
    if (!ctx || !out || bytes < 1) return fail(ctx, SWK_ERR_ARG, "bad argument");
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (hipMalloc(out, (size_t)bytes) != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return fail(ctx, SWK_ERR_NOMEM, "hipMalloc failed"); }
    return SWK_OK;
}

int32_t swk_device_free(swk_ctx *ctx, void *p)
{
    if (!ctx) return SWK_ERR_ARG;
    if (!p) return SWK_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipFree(p));
    return SWK_OK;
}

int32_t swk_device_read(swk_ctx *ctx, const void *src_device, void *dst_host, int64_t bytes)
{
    if (!ctx || !src_device || !dst_host || bytes < 1) return fail(ctx, SWK_ERR_ARG, "bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpyAsync(dst_host, src_device, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SWK_OK;
}

int32_t swk_set_pass_tuning(swk_ctx *ctx, int32_t flags)
{
    if (!ctx || flags < 0 || flags > 3) return SWK_ERR_ARG;
    ctx->pass_tune = flags;
    return SWK_OK;
}

int32_t swk_set_sparse_speculation(swk_ctx *ctx, double factor)
{
    if (!ctx) return SWK_ERR_ARG;
    ctx->sparse_spec = factor;
    return SWK_OK;
}
int32_t swk_set_integer_start(swk_ctx *ctx, int32_t on)
{
    if (!ctx) return SWK_ERR_ARG;
    ctx->use_gram8 = on ? 1 : 0;
    return SWK_OK;
}
int32_t swk_set_norm_guard(swk_ctx *ctx, double rel)
{
    if (!ctx || !(rel >= 0.0) || rel >= 1.0) return SWK_ERR_ARG;
    ctx->norm_guard = rel;
    return SWK_OK;
}
int32_t swk_prof_guard_windows(swk_ctx *ctx, int64_t *windows)
{
    if (!ctx || !windows) return SWK_ERR_ARG;
    *windows = ctx->guard_windows;
    return SWK_OK;
}
int32_t swk_set_norm_speculation(swk_ctx *ctx, double factor)
{
    if (!ctx) return SWK_ERR_ARG;
    ctx->norm_spec = factor;
    return SWK_OK;
}
int32_t swk_prof_pass_bytes_per_element(swk_ctx *ctx, double *bytes)
{
    if (!ctx || !bytes) return SWK_ERR_ARG;
    *bytes = (double)ctx->pass_b16 / 16.0;
    return SWK_OK;
}
int32_t swk_last_eig_sweeps(swk_ctx *ctx, int32_t *sweeps)
{
    if (!ctx || !sweeps) return SWK_ERR_ARG;
    *sweeps = ctx->last_eig_sweeps;
    return SWK_OK;
}
int32_t swk_last_integer_start_windows(swk_ctx *ctx, int32_t *windows)
{
    if (!ctx || !windows) return SWK_ERR_ARG;
    *windows = ctx->last_int_start;
    return SWK_OK;
}
int32_t swk_prof_redo_batches(swk_ctx *ctx, int64_t *batches)
{
    if (!ctx || !batches) return SWK_ERR_ARG;
    *batches = ctx->redo_batches;
    return SWK_OK;
}

int32_t swk_set_start_refine(swk_ctx *ctx, double tau)
{
    if (!ctx) return SWK_ERR_ARG;
    ctx->start_refine = tau;
    return SWK_OK;
}
int32_t swk_prof_refined_windows(swk_ctx *ctx, int64_t *refined, int64_t *unrefined)
{
    if (!ctx) return SWK_ERR_ARG;
    if (refined) *refined = ctx->refined_windows;
    if (unrefined) *unrefined = ctx->unrefined_windows;
    return SWK_OK;
}

int32_t swk_last_stopping_norms(swk_ctx *ctx, double *ratio, double *err_bound, int32_t cap)
{
    if (!ctx || cap < 0) return SWK_ERR_ARG;
    const int n = (int)ctx->last_hw.size();
    for (int w = 0; w < n && w < cap; ++w) {
        if (ratio) ratio[w] = ctx->last_hw[w].last_ratio;
        if (err_bound) err_bound[w] = ctx->last_hw[w].norm_err;
    }
    return n;
}

int32_t swk_prof_redo_windows(swk_ctx *ctx, int64_t *windows)
{
    if (!ctx || !windows) return SWK_ERR_ARG;
    *windows = ctx->redo_windows;
    return SWK_OK;
}

int32_t swk_set_eig_method(swk_ctx *ctx, int32_t method)
{
    if (!ctx || method < 0 || method > 1) return SWK_ERR_ARG;
    ctx->eig_method = method;
    return SWK_OK;
}

// -------------------------------------------------------------------------------------
int32_t swk_batch_run(swk_ctx *ctx, const swk_input *in, const swk_params *p, swk_output *out)
{
    if (!ctx) return SWK_ERR_ARG;
    if (!in || !p || !out || !in->frames) return fail(ctx, SWK_ERR_ARG, "null argument");
    if (in->nwin < 1 || in->n < 1 || in->Hc < 1 || in->Wc < 1) return fail(ctx, SWK_ERR_ARG, "empty batch");
    if (in->channels != 1 && in->channels != 3) return fail(ctx, SWK_ERR_ARG, "channels must be 1 or 3");
    if (in->n > kMaxNWide) return fail(ctx, SWK_ERR_ARG, "frames per window must be <= 128");
    if (p->open_kh != 3 || p->open_kw != 3) return fail(ctx, SWK_ERR_ARG, "only the (3,3) opening window is implemented");
    if (p->connectivity != 4 && p->connectivity != 8) return fail(ctx, SWK_ERR_ARG, "connectivity must be 4 or 8");
    if (p->bil_d / 2 != 3) return fail(ctx, SWK_ERR_ARG, "the fused filter kernel implements bilateral d=7 (radius 3) only");
    if (out->segs && (out->seg_cap < 1 || out->seg_cap > 255)) return fail(ctx, SWK_ERR_ARG, "seg_cap must be in 1..255");
    if (in->Hc < 4 || in->Wc < 4) return fail(ctx, SWK_ERR_ARG, "ROI must be at least 4x4");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int F = in->nwin * in->n, H = in->Hc, W = in->Wc, P = H * W;
    const size_t plane = (size_t)F * P;
    const bool dev_out = out->mem == SWK_MEM_DEVICE;
    const bool dev_planes = dev_out || out->planes_on_device != 0;       // the six u8 stage images stay on the device
    int rc;
    ctx->last.valid = false;

    // ---- input ----
    const uint8_t *dframes = in->frames;
    int64_t fs = in->frame_stride, rs = in->row_stride;
    int x0 = in->x0, y0 = in->y0;
    if (in->mem == SWK_MEM_HOST) {
        uint8_t *roi;
        NEED(ctx, SL_ROI, plane * in->channels, roi);
        Timed t(ctx, SWK_K_COPY);
        bool whole = false;
        const int64_t afs = fs < 0 ? -fs : fs;
        const size_t rowb = (size_t)W * in->channels;
        if (x0 == 0 && (size_t)rs == rowb && fs == (int64_t)H * rs) {
            // pre-cropped, densely packed ROI frames: one copy for the whole batch
            HIPCHK(ctx, hipMemcpyAsync(roi, in->frames + (int64_t)y0 * rs, plane * in->channels, hipMemcpyHostToDevice, s));
        } else if (rs >= (int64_t)(x0 + W) * in->channels && rs % in->channels == 0 && afs % rs == 0 && afs >= (int64_t)(y0 + H) * rs &&
                   (size_t)F * (size_t)afs <= 2 * plane * in->channels) {
            // ROI frames with a margin around them (FrameQueue stages the crop plus the half minimum segment size, so that
            // segment boxes can grow into it like they grow into the full frame, image_filtering.py:338-369): the whole
            // buffer in one copy; the kernels read the ROI at (x0, y0) of the device copy.  A NEGATIVE frame stride (queue
            // position 0 = the LAST frame in memory: a reader's block in file order) is copied as it lies and read backwards.
            NEED(ctx, SL_ROI, (size_t)F * (size_t)afs, roi);
            const uint8_t *lowest = fs < 0 ? in->frames + (int64_t)(F - 1) * fs : in->frames;
            HIPCHK(ctx, hipMemcpyAsync(roi, lowest, (size_t)F * (size_t)afs, hipMemcpyHostToDevice, s));
            whole = true;
        } else if (x0 == 0 && (size_t)rs == rowb) {
            for (int f = 0; f < F; ++f)                    // full-width rows: one contiguous block per frame
                HIPCHK(ctx, hipMemcpyAsync(roi + (size_t)f * P * in->channels, in->frames + (int64_t)f * fs + (int64_t)y0 * rs,
                                           (size_t)P * in->channels, hipMemcpyHostToDevice, s));
        } else {
            for (int f = 0; f < F; ++f)
                HIPCHK(ctx, hipMemcpy2DAsync(roi + (size_t)f * P * in->channels, rowb,
                                             in->frames + (int64_t)f * fs + (int64_t)y0 * rs + (int64_t)x0 * in->channels, (size_t)rs,
                                             rowb, H, hipMemcpyHostToDevice, s));
        }
        dframes = (whole && fs < 0) ? roi + (int64_t)(F - 1) * afs : roi;
        if (!whole) { fs = (int64_t)P * in->channels; rs = (int64_t)rowb; x0 = 0; y0 = 0; }
    }
    // ---- stage buffers (caller's device buffers are written in place) ----
    uint8_t *dX, *dS, *dBil = nullptr, *dThr = nullptr, *dOpen, *dLab;
    // (a gray plane whose size is not a whole number of dwords stays in the library's padded buffer: k_gram_u8 reads
    // it in dwords; the caller's copy is made at the end)
    const bool gray_in_place = dev_planes && out->gray && (plane & 3) == 0;
    if (gray_in_place) dX = out->gray; else NEED(ctx, SL_X, plane + 4, dX);
    if (dev_planes && out->rpca) dS = out->rpca; else NEED(ctx, SL_S, plane, dS);
    if (out->bilateral) { if (dev_planes) dBil = out->bilateral; else NEED(ctx, SL_BIL, plane, dBil); }
    if (out->thresh) { if (dev_planes) dThr = out->thresh; else NEED(ctx, SL_THR, plane, dThr); }
    if (dev_planes && out->opened) dOpen = out->opened; else NEED(ctx, SL_OPEN, plane, dOpen);
    if (dev_planes && out->labels) dLab = out->labels; else NEED(ctx, SL_LAB8, plane, dLab);

    { Timed t(ctx, SWK_K_GRAY); launch_gray(s, dframes, in->channels, fs, rs, x0, y0, F, H, W, p->gray_mode, dX); }

    rc = run_ialm(ctx, dX, in->nwin, in->n, P, p->lmbda, p->tol, p->maxiter, out->A != nullptr, out->E != nullptr, dS,
                  (!dev_out) ? out->iters : nullptr, dev_out ? out->iters : nullptr);
    if (rc) return rc;

    rc = ensure_bilateral(ctx, p->bil_d, p->bil_sigma_color, p->bil_sigma_space);
    if (rc) return rc;
    { Timed t(ctx, SWK_K_FILTER); launch_filter_fused(s, dS, F, H, W, ctx->bil, p->bil_fma, p->thresh, dBil, dThr, dOpen); }

    CclBuffers cb{};
    rc = ensure_ccl(ctx, F, H, W, &cb);
    if (rc) return rc;
    const bool want_props = out->segs || out->nseg;
    const int cap = out->segs ? out->seg_cap : 1;
    swk_segment *dsegs = nullptr; int32_t *dnseg = nullptr;
    if (want_props) {
        if (dev_out && out->segs) dsegs = out->segs; else NEED(ctx, SL_SEGS, (size_t)F * cap * sizeof(swk_segment), dsegs);
        if (dev_out && out->nseg) dnseg = out->nseg; else NEED(ctx, SL_NSEG, (size_t)F * 4, dnseg);
        HIPCHK(ctx, hipMemsetAsync(dsegs, 0, (size_t)F * cap * sizeof(swk_segment), s));
    }
    const bool fused_ccl = ccl_frame_supported(H, W);
    if (fused_ccl) {
        // one workgroup per frame: labels and region properties in a single kernel
        Timed t(ctx, SWK_K_CCL);
        launch_ccl_frame(s, dOpen, F, H, W, p->connectivity, p->label_order, cb, nullptr, dLab, want_props, cap, dsegs, dnseg);
    } else {
        Timed t(ctx, SWK_K_CCL);
        launch_ccl(s, dOpen, F, H, W, p->connectivity, p->label_order, cb, nullptr, dLab);
    }
    if (want_props) {
        if (!fused_ccl) { Timed t(ctx, SWK_K_PROPS); launch_regionprops(s, dLab, F, H, W, cb, cap, dsegs, dnseg); }
        if (!dev_out) {
            rc = copy_out(ctx, out->segs, dsegs, (size_t)F * cap * sizeof(swk_segment), out->mem); if (rc) return rc;
            rc = copy_out(ctx, out->nseg, dnseg, (size_t)F * 4, out->mem); if (rc) return rc;
        }
    }
    // ---- float outputs in the reference's (pixels, frames) layout ----
    if (out->A || out->E) {
        const size_t elems = (size_t)in->nwin * in->n * P;
        for (int which = 0; which < 2; ++which) {
            double *dst = which == 0 ? out->A : out->E;
            if (!dst) continue;
            const double *planes = (const double *)ctx->slot[which == 0 ? SL_A : SL_E];
            double *pn;
            if (dev_out) pn = dst; else NEED(ctx, SL_PN, elems * 8, pn);
            { Timed t(ctx, SWK_K_COPY); launch_planes_to_pn(s, planes, pn, in->nwin, in->n, P, ctx->pstride, ctx->fpad); }
            if (!dev_out) { rc = copy_out(ctx, dst, pn, elems * 8, out->mem); if (rc) return rc; HIPCHK(ctx, hipStreamSynchronize(s)); }
        }
    }
    if (dev_planes && out->gray && !gray_in_place) { rc = copy_out(ctx, out->gray, dX, plane, SWK_MEM_DEVICE); if (rc) return rc; }
    if (!dev_planes) {
        Timed t(ctx, SWK_K_COPY);
        rc = copy_out(ctx, out->gray, dX, plane, out->mem); if (rc) return rc;
        rc = copy_out(ctx, out->rpca, dS, plane, out->mem); if (rc) return rc;
        rc = copy_out(ctx, out->bilateral, dBil, plane, out->mem); if (rc) return rc;
        rc = copy_out(ctx, out->thresh, dThr, plane, out->mem); if (rc) return rc;
        rc = copy_out(ctx, out->opened, dOpen, plane, out->mem); if (rc) return rc;
        rc = copy_out(ctx, out->labels, dLab, plane, out->mem); if (rc) return rc;
    }
    rc = sync(ctx);
    if (rc) return rc;
    if (out->segs && in->channels == 3) {
        swk_ctx::LastBatch &lb = ctx->last;
        lb.frames = dframes; lb.fs = fs; lb.rs = rs;
        lb.nwin = in->nwin; lb.n = in->n; lb.Hc = H; lb.Wc = W; lb.x0 = x0; lb.y0 = y0;
        lb.frame_h = (int)((fs < 0 ? -fs : fs) / rs); lb.frame_w = (int)(rs / 3);
        lb.segs = dsegs; lb.nseg = dnseg; lb.cap = cap;
        lb.total = -1;
        if (!dev_out && out->nseg) {
            lb.total = 0;
            for (int f = 0; f < F; ++f) lb.total += out->nseg[f] < cap ? out->nseg[f] : cap;
        }
        lb.valid = true;
    }
    return gather_iters(ctx, dev_out ? nullptr : out->iters, dev_out ? out->iters : nullptr);
}

// ---- stage-level entry points (host buffers) ----------------------------------------
int32_t swk_bgr2gray(swk_ctx *ctx, const uint8_t *bgr, int32_t count, int32_t H, int32_t W, int32_t gray_mode, uint8_t *gray)
{
    if (!ctx || !bgr || !gray || count < 1 || H < 1 || W < 1) return fail(ctx, SWK_ERR_ARG, "bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t px = (size_t)count * H * W;
    uint8_t *din, *dout;
    NEED(ctx, SL_TMP_IN, px * 3, din);
    NEED(ctx, SL_TMP_OUT, px, dout);
    HIPCHK(ctx, hipMemcpyAsync(din, bgr, px * 3, hipMemcpyHostToDevice, ctx->stream));
    launch_gray(ctx->stream, din, 3, (int64_t)H * W * 3, (int64_t)W * 3, 0, 0, count, H, W, gray_mode, dout);
    HIPCHK(ctx, hipMemcpyAsync(gray, dout, px, hipMemcpyDeviceToHost, ctx->stream));
    return sync(ctx);
}

int32_t swk_ialm(swk_ctx *ctx, const uint8_t *planes, int32_t n, int32_t P, double lmbda, double tol, int32_t maxiter,
                 double *A, double *E, int32_t *iters)
{
    if (!ctx || !planes || n < 1 || P < 1) return fail(ctx, SWK_ERR_ARG, "bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t elems = (size_t)n * P;
    uint8_t *dX, *dS;
    NEED(ctx, SL_X, elems + 4, dX);
    NEED(ctx, SL_S, elems, dS);
    HIPCHK(ctx, hipMemcpyAsync(dX, planes, elems, hipMemcpyHostToDevice, ctx->stream));
    int rc = run_ialm(ctx, dX, 1, n, P, lmbda, tol, maxiter, A != nullptr, E != nullptr, dS, iters, nullptr);
    if (rc) return rc;
    for (int which = 0; which < 2; ++which) {
        double *dst = which == 0 ? A : E;
        if (!dst) continue;
        double *pn;
        NEED(ctx, SL_PN, elems * 8, pn);
        launch_planes_to_pn(ctx->stream, (const double *)ctx->slot[which == 0 ? SL_A : SL_E], pn, 1, n, P, ctx->pstride, ctx->fpad);
        HIPCHK(ctx, hipMemcpyAsync(dst, pn, elems * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    rc = sync(ctx);
    if (rc) return rc;
    return gather_iters(ctx, iters, nullptr);
}

int32_t swk_rpca_epilogue(swk_ctx *ctx, const double *E, int64_t count, uint8_t *S)
{
    if (!ctx || !E || !S || count < 1) return fail(ctx, SWK_ERR_ARG, "bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    double *din; uint8_t *dout;
    NEED(ctx, SL_PN, (size_t)count * 8, din);
    NEED(ctx, SL_TMP_OUT, (size_t)count, dout);
    HIPCHK(ctx, hipMemcpyAsync(din, E, (size_t)count * 8, hipMemcpyHostToDevice, ctx->stream));
    launch_rpca_epilogue(ctx->stream, din, count, dout);
    HIPCHK(ctx, hipMemcpyAsync(S, dout, (size_t)count, hipMemcpyDeviceToHost, ctx->stream));
    return sync(ctx);
}

int32_t swk_bilateral_u8(swk_ctx *ctx, const uint8_t *src, int32_t count, int32_t H, int32_t W, int32_t d,
                         double sigma_color, double sigma_space, int32_t use_fma, uint8_t *dst)
{
    if (!ctx || !src || !dst || count < 1 || H < 2 || W < 2) return fail(ctx, SWK_ERR_ARG, "bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = ensure_bilateral(ctx, d, sigma_color, sigma_space);
    if (rc) return rc;
    const size_t px = (size_t)count * H * W;
    uint8_t *din, *dout;
    NEED(ctx, SL_TMP_IN, px, din);
    NEED(ctx, SL_TMP_OUT, px, dout);
    HIPCHK(ctx, hipMemcpyAsync(din, src, px, hipMemcpyHostToDevice, ctx->stream));
    launch_bilateral(ctx->stream, din, count, H, W, ctx->bil, use_fma, dout);
    HIPCHK(ctx, hipMemcpyAsync(dst, dout, px, hipMemcpyDeviceToHost, ctx->stream));
    return sync(ctx);
}

int32_t swk_thresh_tozero_u8(swk_ctx *ctx, const uint8_t *src, int64_t count, int32_t thresh, uint8_t *dst)
{
    if (!ctx || !src || !dst || count < 1) return fail(ctx, SWK_ERR_ARG, "bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    uint8_t *din, *dout;
    NEED(ctx, SL_TMP_IN, (size_t)count, din);
    NEED(ctx, SL_TMP_OUT, (size_t)count, dout);
    HIPCHK(ctx, hipMemcpyAsync(din, src, (size_t)count, hipMemcpyHostToDevice, ctx->stream));
    launch_thresh(ctx->stream, din, count, thresh, dout);
    HIPCHK(ctx, hipMemcpyAsync(dst, dout, (size_t)count, hipMemcpyDeviceToHost, ctx->stream));
    return sync(ctx);
}

int32_t swk_grey_open3x3_u8(swk_ctx *ctx, const uint8_t *src, int32_t count, int32_t H, int32_t W, uint8_t *dst)
{
    if (!ctx || !src || !dst || count < 1 || H < 1 || W < 1) return fail(ctx, SWK_ERR_ARG, "bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t px = (size_t)count * H * W;
    uint8_t *din, *dout;
    NEED(ctx, SL_TMP_IN, px, din);
    NEED(ctx, SL_TMP_OUT, px, dout);
    HIPCHK(ctx, hipMemcpyAsync(din, src, px, hipMemcpyHostToDevice, ctx->stream));
    launch_open3x3(ctx->stream, din, count, H, W, dout);
    HIPCHK(ctx, hipMemcpyAsync(dst, dout, px, hipMemcpyDeviceToHost, ctx->stream));
    return sync(ctx);
}

int32_t swk_grey_open_u8(swk_ctx *ctx, const uint8_t *src, int32_t count, int32_t H, int32_t W, int32_t kh, int32_t kw, uint8_t *dst)
{
    if (!ctx || !src || !dst || count < 1 || H < 1 || W < 1 || kh < 1 || kw < 1 || kh > 255 || kw > 255) return fail(ctx, SWK_ERR_ARG, "bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t px = (size_t)count * H * W;
    uint8_t *din, *dout, *dtmp;
    NEED(ctx, SL_TMP_IN, px, din);
    NEED(ctx, SL_TMP_OUT, px, dout);
    NEED(ctx, SL_REDO_S2, px, dtmp);          // (a scratch plane: the rerun slots are idle outside run_ialm)
    HIPCHK(ctx, hipMemcpyAsync(din, src, px, hipMemcpyHostToDevice, ctx->stream));
    launch_grey_open(ctx->stream, din, count, H, W, kh, kw, dtmp, dout);
    HIPCHK(ctx, hipMemcpyAsync(dst, dout, px, hipMemcpyDeviceToHost, ctx->stream));
    return sync(ctx);
}

int32_t swk_resize_linear_u8(swk_ctx *ctx, const uint8_t *src, int32_t count, int32_t H, int32_t W, int32_t channels, int32_t dH, int32_t dW,
                             uint8_t *dst)
{
    if (!ctx || !src || !dst || count < 1 || H < 1 || W < 1 || dH < 1 || dW < 1 || channels < 1 || channels > 4)
        return fail(ctx, SWK_ERR_ARG, "bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // per-axis tables (OpenCV 4.1.0 resizeGeneric_ for INTER_LINEAR, 8u: float32 coordinates, 11-bit weights)
    std::vector<int> idx((size_t)dW + dH);
    std::vector<short> wts(2 * ((size_t)dW + dH));
    auto axis = [](int srcn, int dstn, int *ix, short *w) {
        const double scale = (double)srcn / dstn;
        for (int d = 0; d < dstn; ++d) {
            float f = (float)((d + 0.5) * scale - 0.5);
            int s0 = (int)floorf(f);
            f -= s0;
            if (s0 < 0) { s0 = 0; f = 0.f; }
            if (s0 >= srcn - 1) { s0 = srcn - 1; f = 0.f; }
            ix[d] = s0;
            w[2 * d] = (short)lrintf((1.f - f) * 2048.f);
            w[2 * d + 1] = (short)lrintf(f * 2048.f);
        }
    };
    axis(W, dW, idx.data(), wts.data());
    axis(H, dH, idx.data() + dW, wts.data() + 2 * dW);
    const size_t in_b = (size_t)count * H * W * channels, out_b = (size_t)count * dH * dW * channels;
    uint8_t *din, *dout;
    int *dix; short *dw;
    NEED(ctx, SL_TMP_IN, in_b, din);
    NEED(ctx, SL_TMP_OUT, out_b, dout);
    NEED(ctx, SL_REDO_S2, idx.size() * 4 + wts.size() * 2, dix);
    dw = (short *)(dix + idx.size());
    HIPCHK(ctx, hipMemcpyAsync(din, src, in_b, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dix, idx.data(), idx.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dw, wts.data(), wts.size() * 2, hipMemcpyHostToDevice, ctx->stream));
    launch_resize_linear(ctx->stream, din, count, H, W, channels, dH, dW, dix, dw, dix + dW, dw + 2 * dW, dout);
    HIPCHK(ctx, hipMemcpyAsync(dst, dout, out_b, hipMemcpyDeviceToHost, ctx->stream));
    return sync(ctx);          // (synchronous: idx / wts live on this stack frame)
}

int32_t swk_ccl_u8(swk_ctx *ctx, const uint8_t *src, int32_t count, int32_t H, int32_t W, int32_t connectivity,
                   int32_t label_order, int32_t *labels, int32_t *ncomp)
{
    if (!ctx || !src || count < 1 || H < 1 || W < 1) return fail(ctx, SWK_ERR_ARG, "bad argument");
    if (connectivity != 4 && connectivity != 8) return fail(ctx, SWK_ERR_ARG, "connectivity must be 4 or 8");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t px = (size_t)count * H * W;
    uint8_t *din; int32_t *dlab;
    NEED(ctx, SL_TMP_IN, px, din);
    NEED(ctx, SL_LAB32, px * 4, dlab);
    CclBuffers cb{};
    int rc = ensure_ccl(ctx, count, H, W, &cb);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(din, src, px, hipMemcpyHostToDevice, ctx->stream));
    if (ccl_frame_supported(H, W))
        launch_ccl_frame(ctx->stream, din, count, H, W, connectivity, label_order, cb, dlab, nullptr, false, 1, nullptr, nullptr);
    else
        launch_ccl(ctx->stream, din, count, H, W, connectivity, label_order, cb, dlab, nullptr);
    if (labels) HIPCHK(ctx, hipMemcpyAsync(labels, dlab, px * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (ncomp) HIPCHK(ctx, hipMemcpyAsync(ncomp, cb.ncomp, (size_t)count * 4, hipMemcpyDeviceToHost, ctx->stream));
    return sync(ctx);
}

int32_t swk_regionprops_u8(swk_ctx *ctx, const uint8_t *labels, int32_t count, int32_t H, int32_t W, int32_t seg_cap,
                           swk_segment *segs, int32_t *nseg)
{
    if (!ctx || !labels || !segs || !nseg || count < 1 || H < 1 || W < 1 || seg_cap < 1 || seg_cap > 255)
        return fail(ctx, SWK_ERR_ARG, "bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->last.valid = false;          // SL_SEGS / SL_NSEG are about to be reused
    const size_t px = (size_t)count * H * W;
    uint8_t *din; swk_segment *dsegs; int32_t *dnseg;
    NEED(ctx, SL_TMP_IN, px, din);
    NEED(ctx, SL_SEGS, (size_t)count * seg_cap * sizeof(swk_segment), dsegs);
    NEED(ctx, SL_NSEG, (size_t)count * 4, dnseg);
    CclBuffers cb{};
    int rc = ensure_ccl(ctx, count, H, W, &cb);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(din, labels, px, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(dsegs, 0, (size_t)count * seg_cap * sizeof(swk_segment), ctx->stream));
    launch_regionprops(ctx->stream, din, count, H, W, cb, seg_cap, dsegs, dnseg);
    HIPCHK(ctx, hipMemcpyAsync(segs, dsegs, (size_t)count * seg_cap * sizeof(swk_segment), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(nseg, dnseg, (size_t)count * 4, hipMemcpyDeviceToHost, ctx->stream));
    return sync(ctx);
}

int32_t swk_classifier_input_window(swk_ctx *ctx, const uint8_t *crops, int64_t crops_bytes, const int64_t *offsets,
                                    const int32_t *hw, int32_t nseg, const float mean[3], const float std_[3],
                                    int32_t pad, int32_t channels_last, uint8_t *patches, float *net, int32_t net_mem)
{
    if (!ctx || !crops || !offsets || !hw || !mean || !std_ || nseg < 1 || crops_bytes < 1 || (!patches && !net) ||
        pad < 0 || pad > 100 || (channels_last != 0 && channels_last != 1))
        return fail(ctx, SWK_ERR_ARG, "bad argument");
    for (int i = 0; i < nseg; ++i) {
        const int h = hw[2 * i], w = hw[2 * i + 1];
        if (h < 1 || w < 1 || h > 512 || w > 512) return fail(ctx, SWK_ERR_ARG, "segment crops must be 1..512 pixels on each side");
        if (offsets[i] < 0 || offsets[i] + (int64_t)h * w * 3 > crops_bytes) return fail(ctx, SWK_ERR_ARG, "crop outside the packed buffer");
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    uint8_t *dcrops, *dpatch = nullptr; int64_t *doffs; int32_t *dhw; float *dnet = nullptr;
    NEED(ctx, SL_CL_CROPS, (size_t)crops_bytes, dcrops);
    NEED(ctx, SL_CL_OFFS, (size_t)nseg * 8, doffs);
    NEED(ctx, SL_CL_HW, (size_t)nseg * 8, dhw);
    HIPCHK(ctx, hipMemcpyAsync(dcrops, crops, (size_t)crops_bytes, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(doffs, offsets, (size_t)nseg * 8, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(dhw, hw, (size_t)nseg * 8, hipMemcpyHostToDevice, s));
    if (patches) NEED(ctx, SL_CL_PATCH, (size_t)nseg * 24 * 24 * 3, dpatch);
    const size_t side = 24 + 2 * (size_t)pad;
    const size_t net_bytes = (size_t)nseg * 3 * side * side * sizeof(float);
    if (net) { if (net_mem == SWK_MEM_DEVICE) dnet = net; else NEED(ctx, SL_CL_NET, net_bytes, dnet); }
    launch_classifier_input(s, dcrops, doffs, dhw, nseg, dpatch, dnet, pad, channels_last != 0, mean, std_);
    if (patches) HIPCHK(ctx, hipMemcpyAsync(patches, dpatch, (size_t)nseg * 24 * 24 * 3, hipMemcpyDeviceToHost, s));
    if (net && net_mem != SWK_MEM_DEVICE) HIPCHK(ctx, hipMemcpyAsync(net, dnet, net_bytes, hipMemcpyDeviceToHost, s));
    return sync(ctx);
}

int32_t swk_segment_inputs(swk_ctx *ctx, const swk_input *in, int32_t frame_h, int32_t frame_w,
                           const swk_segment *segs, const int32_t *nseg, int32_t seg_cap, int32_t min_h, int32_t min_w,
                           const float mean[3], const float std_[3], int32_t pad, int32_t channels_last, int32_t first, int32_t net_cap,
                           float *net, int32_t *seg_frame, int32_t *total, int32_t *skipped)
{
    if (!ctx || !in || !in->frames || !segs || !nseg || !mean || !std_ || !net || !total)
        return fail(ctx, SWK_ERR_ARG, "bad argument");
    if (in->mem != SWK_MEM_DEVICE || in->channels != 3) return fail(ctx, SWK_ERR_ARG, "segment inputs are cut from device-resident BGR frames");
    const int64_t F64 = (int64_t)in->nwin * in->n;
    if (in->nwin < 1 || in->n < 1 || F64 > (1 << 24) || seg_cap < 1 || pad < 0 || pad > 100 || first < 0 || net_cap < 1 ||
        (channels_last != 0 && channels_last != 1) ||
        frame_h < 1 || frame_w < 1 || min_h < 1 || min_w < 1 || min_h > 512 || min_w > 512 ||
        in->x0 < 0 || in->y0 < 0 || in->x0 + in->Wc > frame_w || in->y0 + in->Hc > frame_h ||
        in->row_stride < (int64_t)frame_w * 3 || (in->frame_stride < 0 ? -in->frame_stride : in->frame_stride) < in->row_stride * frame_h)
        return fail(ctx, SWK_ERR_ARG, "bad geometry");
    return segment_inputs_impl(ctx, in->frames, in->frame_stride, in->row_stride, (int)F64, in->x0, in->y0, frame_h, frame_w, segs, nseg,
                               seg_cap, min_h, min_w, mean, std_, pad, channels_last != 0, first, net_cap, net, seg_frame, total, skipped);
}

int32_t swk_segment_inputs_last(swk_ctx *ctx, int32_t min_h, int32_t min_w, const float mean[3], const float std_[3], int32_t pad,
                                int32_t channels_last, int32_t first, int32_t net_cap, float *net, int32_t *seg_frame, int32_t *total,
                                int32_t *skipped)
{
    if (!ctx || !mean || !std_ || !net || !total || pad < 0 || pad > 100 || first < 0 || net_cap < 1 ||
        (channels_last != 0 && channels_last != 1) || min_h < 1 || min_w < 1 || min_h > 512 || min_w > 512)
        return fail(ctx, SWK_ERR_ARG, "bad argument");
    const swk_ctx::LastBatch &lb = ctx->last;
    if (!lb.valid) return fail(ctx, SWK_ERR_STALE, "no batch with BGR frames and region records is held by the context any more");
    const int known = *total >= 0 && lb.total >= 0 ? lb.total : -1;          // the batch's own count, when its nseg went to the host
    if (known >= 0 && *total != known) return fail(ctx, SWK_ERR_ARG, "*total does not match the batch");
    return segment_inputs_impl(ctx, lb.frames, lb.fs, lb.rs, lb.nwin * lb.n, lb.x0, lb.y0, lb.frame_h, lb.frame_w, lb.segs, lb.nseg, lb.cap,
                               min_h, min_w, mean, std_, pad, channels_last != 0, first, net_cap, net, seg_frame, total, skipped, known);
}

int32_t swk_classifier_input(swk_ctx *ctx, const uint8_t *crops, int64_t crops_bytes, const int64_t *offsets,
                             const int32_t *hw, int32_t nseg, const float mean[3], const float std_[3],
                             uint8_t *patches, float *net, int32_t net_mem)
{
    return swk_classifier_input_window(ctx, crops, crops_bytes, offsets, hw, nseg, mean, std_, 100, 0, patches, net, net_mem);
}

}  // extern "C"
#pragma GCC visibility pop

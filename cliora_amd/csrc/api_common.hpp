// Shared pieces of the C-ABI translation units (api_core / api_mlp / api_lstm / api_vl): the plan handle, error and
// profiling plumbing, and the launch helpers of the MFMA kernels.  Header-only templates: each unit instantiates what it uses,
// so the units compile in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/cliora_chart.h"
#include "chart_kernels.hpp"
#include "gemm_kernels.hpp"
#include "wgrad_tiles.hpp"
#include "plan.hpp"

using namespace cliora;

struct cliora_plan {
    Plan p;
    bool uploaded = false;
    int device = -1;            // HIP device the index tables live on (set at upload; every later call must run there)
    // side: the second chain of the inside / outside wavefront (forward: the outside pass, backward: the outside pass's backward);
    // side2: weight-gradient GEMMs beside the level chains.  Fork / join with the caller's stream by events.  The streams and events
    // belong to the DEVICE (api_core.hip: device_lanes), not to the plan: HIP deals streams round-robin onto a few hardware queues,
    // and a per-plan side stream that lands on the caller's queue serialises the two chains (measured: 6.45 -> 6.97 ms at L = 24
    // with five plans alive).  `lanes_mu` is held while a call enqueues (the events are shared by every plan of the device).
    hipStream_t side = nullptr, side2 = nullptr;
    hipEvent_t ev_fork[3] = {nullptr, nullptr, nullptr}, ev_join[3] = {nullptr, nullptr, nullptr};
    const hipEvent_t* ev_level = nullptr;   // one per chart level: "this chain has finished level k" for the other chain
    std::mutex* lanes_mu = nullptr;
    int ncu = 0;
    // forward workspaces whose weight images were NOT built (resident forward; the last eight): a launch-path backward on one of them
    // builds them first.  Only a mode switch between a forward and its backward gets there (tests do that).
    const void* imageless_ws[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int imageless_next = 0;
    void note_imageless(const void* ws) {
        for (const void* w : imageless_ws) if (w == ws) return;
        imageless_ws[imageless_next] = ws; imageless_next = (imageless_next + 1) % 8;
    }
    bool take_imageless(const void* ws) {          // true (and forgotten) if `ws` is one of them
        for (const void*& w : imageless_ws) if (w == ws && ws) { w = nullptr; return true; }
        return false;
    }
    unsigned long long* trace_words = nullptr;   // device words for diagnostic stamps (the device's, api_core.hip: DeviceLanes::trace)
    std::mutex upload_mu;                   // first-use upload of the tables (plans are shared between host threads)
};

extern thread_local std::string g_cliora_err;
static inline int fail(int code, const std::string& msg) { g_cliora_err = msg; return code; }

#define HIPOK(expr)                                                                            \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(CLIORA_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
    } while (0)
#define LAUNCHOK(name)                                                                         \
    do {                                                                                       \
        hipError_t e_ = hipGetLastError();                                                     \
        if (e_ != hipSuccess)                                                                  \
            return fail(CLIORA_EHIP, std::string("launch ") + name + ": " + hipGetErrorString(e_)); \
    } while (0)
#define OKR(expr) do { int rc_ = (expr); if (rc_ != CLIORA_OK) return rc_; } while (0)
// A launch that signals `ev` (or none: nullptr) itself -- hipExtLaunchKernelGGL's stop event is the dispatch packet's own completion
// signal; a hipEventRecord behind the launch is a barrier packet of its own on the queue, and the next kernel of that queue started
// ~1 us later for it at every level of the two-chain wavefronts (c2: 3.22 -> 3.17 ms with the forward's inside chain alone).
#define LAUNCH_SIGNALLING(ev, kernel, grid, block, lds, stream, ...)                                          \
    do {                                                                                                       \
        if (ev) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, nullptr, ev, 0, __VA_ARGS__);          \
        else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                \
    } while (0)
// CLIORA_STOP_EVENT=0: every event by hipEventRecord
static bool stop_events_on() {
    static const bool on = [] { const char* e = getenv("CLIORA_STOP_EVENT"); return !e || atoi(e) != 0; }();
    return on;
}

// kernels that need more than 64 KiB of dynamic LDS: the attribute is per device, so it is set once per (function, device)
int cliora_ensure_max_lds(const void* fn);
// uploads the plan's index tables on first use (current device) and checks that later calls run on that device
int cliora_plan_ready(cliora_plan* plan, hipStream_t st);

constexpr size_t TRACE_BYTES = (size_t)256 * 4 * (CLIORA_MAX_L + 1) * 10 * 8;   // diagnostic stamp words per device

// ------------------------------------------------------------------ profiling (HIP events)
struct ProfClass {
    bool on = false;
    std::vector<hipEvent_t> ev;   // pairs
    size_t used = 0;
    double total_ms = 0;
    long long launches = 0;
};
extern ProfClass g_cliora_prof[CLIORA_KCLASS_COUNT];
extern std::mutex g_cliora_prof_mu;      // the event lists are shared by every host thread that calls into the library
struct ProfScope {
    ProfClass* pc = nullptr;
    hipStream_t st;
    hipEvent_t stop{};
    ProfScope(int cls, hipStream_t s) : st(s) {
        ProfClass& c = g_cliora_prof[cls];
        if (!c.on) return;
        std::lock_guard<std::mutex> lk(g_cliora_prof_mu);
        if (c.used + 2 > c.ev.size()) {
            for (int k = 0; k < 256; ++k) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; c.ev.push_back(e); }
        }
        pc = &c;
        hipEventRecord(c.ev[c.used], st);
        stop = c.ev[c.used + 1];
        c.used += 2;
    }
    ~ProfScope() { if (pc) hipEventRecord(stop, st); }
};

// Arithmetic of the pair-level GEMMs: 1 = split-bf16 (three bf16 MFMAs per product, fp32 accumulate; see
// gemm_kernels.hpp), 0 = fp32-input MFMA (exact fp32 products).  CLIORA_MFMA=f32 selects the latter.
extern int g_cliora_wavefront;      // -1 auto, 0 off, 1 two streams, 2 merged: one queue (include/cliora_chart.h: cliora_set_wavefront)
extern int g_cliora_resident;       // -1 auto, 0 off, 1 on (include/cliora_chart.h: cliora_set_resident)
extern int g_cliora_resident_max_pairs;   // auto: span pairs per sentence (both passes) up to which the sentence-resident kernels are taken
extern int g_cliora_split_bf16;
static inline bool split_bf16() {
    if (g_cliora_split_bf16 < 0) {
        const char* e = getenv("CLIORA_MFMA");
        g_cliora_split_bf16 = (e && !strcmp(e, "f32")) ? 0 : 1;
    }
    return g_cliora_split_bf16 == 1;
}
// The per-cell projection GEMMs (leaf, PL/PR/QL and their backward) run on the exact fp32-input MFMA in both modes: their
// outputs feed the split scores, where a 2^-18 operand rounding shows up as ~1e-4 absolute on scores of magnitude ~15
// (measured in round 1), and they are latency-bound.  Their weights are read from fp32 fragment images (frag_weight_image).
static inline int image_stride(int K) { return (K + 31) / 32 * 32 + WS3_PAD; }
// image argument pair (pointer, kind) of a projection weight
#define PROJ_IMG(off) (ws + (off)), IMG_FRAG_F32

// A call forks work onto the device's side streams and joins it back before it returns.  If it returns EARLY (a failed launch or
// HIP call between fork and join), the caller drops its workspaces while side-stream kernels may still use them: this guard makes
// the caller's stream wait for whatever the side streams hold before the error propagates.  disarm() after the regular join.
struct ForkGuard {
    hipStream_t st;
    hipStream_t side[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    explicit ForkGuard(hipStream_t caller) : st(caller) {}
    void arm(int k, hipStream_t s, hipEvent_t e) { side[k] = s; ev[k] = e; }
    void disarm() { side[0] = side[1] = nullptr; }
    ~ForkGuard() {
        for (int k = 0; k < 2; ++k)
            if (side[k] && side[k] != st) {
                if (hipEventRecord(ev[k], side[k]) == hipSuccess) (void)hipStreamWaitEvent(st, ev[k], 0);
            }
    }
};

// ------------------------------------------------------------------ launch helpers
static int pick_tiles(int ntiles16) {
    for (int t : {5, 4, 2, 1}) if (ntiles16 % t == 0) return t;
    return 1;
}

template <int CT, int SC, int WAVES, class AP, class EP>
static int launch_rows_inst(hipStream_t st, const float* W, int Kseg, int nseg, int ncols, int nrows, AP ap, EP ep) {
    const size_t lds = (size_t)CT * 16 * (Kseg + WS_LDS_PAD) * sizeof(float);
    OKR(cliora_ensure_max_lds((const void*)rows_gemm_ws<CT, SC, WAVES, AP, EP>));
    const int ntiles = (nrows + 15) / 16;
    const int gy = ncols / (16 * CT);
    // one workgroup per CU (the weight block fills LDS): take the fewest passes over the row tiles the chip allows,
    // then the smallest grid that still does it in that many passes
    const int cap = std::max(1, 256 / gy);
    const int passes = (ntiles + WAVES * cap - 1) / (WAVES * cap);
    int gx = (ntiles + WAVES * passes - 1) / (WAVES * passes);
    // workgroups are dealt round-robin over the 8 XCDs by linear id (x + y*gx): with gx a multiple of 8
    // the gy column blocks that re-read the same A rows share one XCD's L2 (speed only, never correctness)
    if (gx >= 8 && (gx + 7) / 8 * 8 <= cap) gx = (gx + 7) / 8 * 8;
    hipLaunchKernelGGL((rows_gemm_ws<CT, SC, WAVES, AP, EP>), dim3(gx, gy), dim3(WAVES * 64), lds, st, W, Kseg * nseg, Kseg, nseg,
                       nrows, ap, ep);
    LAUNCHOK("rows_gemm_ws");
    return CLIORA_OK;
}

template <int CT, int SC, class AP, class EP>
static int launch_rows_waves(hipStream_t st, const float* W, int Kseg, int nseg, int ncols, int nrows, AP ap, EP ep) {
    // one wave per SIMD while the launch cannot fill the chip twice over; two per SIMD beyond that
    const long long tasks = (long long)((nrows + 15) / 16) * (ncols / (16 * CT));
    if (tasks > 1536) return launch_rows_inst<CT, SC, 8>(st, W, Kseg, nseg, ncols, nrows, ap, ep);
    return launch_rows_inst<CT, SC, 4>(st, W, Kseg, nseg, ncols, nrows, ap, ep);
}

template <int CT, class AP, class EP>
static int launch_rows_ct(hipStream_t st, const float* W, int Kseg, int nseg, int ncols, int nrows, AP ap, EP ep) {
    const int chunks = Kseg / 16;
    if (chunks % 5 == 0) return launch_rows_waves<CT, 5>(st, W, Kseg, nseg, ncols, nrows, ap, ep);
    if (chunks % 4 == 0) return launch_rows_waves<CT, 4>(st, W, Kseg, nseg, ncols, nrows, ap, ep);
    if (chunks % 2 == 0) return launch_rows_waves<CT, 2>(st, W, Kseg, nseg, ncols, nrows, ap, ep);
    return launch_rows_waves<CT, 1>(st, W, Kseg, nseg, ncols, nrows, ap, ep);
}

// small-row variant (per-level cell GEMMs): 32 x (CT*16) blocks, reduction split over the 4 waves, no weight staging
// One split-K launch: blocks of (RT*16 rows) x (CT*16 columns).  A block's MFMA work and operand bytes are fixed by
// its tile, so a launch with few blocks leaves most CUs idle while the busy ones work through a long reduction: the
// tile shrinks (2x5 -> 1x5 -> 1x1 sixteen-wide tiles) until the launch has enough blocks to cover the chip.
template <int RT, int CT, bool FRAG, class AP, class EP>
static int launch_ksplit_tile(hipStream_t st, const float* W, int K, int nt, int nrows, AP ap, EP ep) {
    const int nrg = ((nrows + 15) / 16 + RT - 1) / RT;
    // linear id = rg + cb * nrgp with nrgp a multiple of 8: the column blocks of one row group share an XCD (L2 reuse of A)
    const int nrgp = nrg >= 8 ? (nrg + 7) / 8 * 8 : nrg;
    hipLaunchKernelGGL((rows_gemm_ksplit<RT, CT, FRAG, AP, EP>), dim3(nrgp * (nt / CT)), dim3(256), 0, st, W, K, nrg, nrgp, nt / CT, nrows, ap, ep);
    LAUNCHOK("rows_gemm_ksplit");
    return CLIORA_OK;
}
static int g_ksplit_min_blocks = -1;   // per translation unit (read from the environment once each)
template <int CT, bool FRAG, class AP, class EP>
static int launch_ksplit_ct(hipStream_t st, const float* W, int K, int nt, int nrows, AP ap, EP ep) {
    if (g_ksplit_min_blocks < 0) { const char* e = getenv("CLIORA_KSPLIT_MIN_BLOCKS"); g_ksplit_min_blocks = e ? atoi(e) : 1000; }   // MI355X sweep 0..2000: 5.61 ms/step at 0, 5.44 at 160, 5.37 at 1000
    const int nrt = (nrows + 15) / 16;
    if (((nrt + 1) / 2) * (nt / CT) >= g_ksplit_min_blocks) return launch_ksplit_tile<2, CT, FRAG>(st, W, K, nt, nrows, ap, ep);
    if (CT == 1 || nrt * (nt / CT) >= g_ksplit_min_blocks) return launch_ksplit_tile<1, CT, FRAG>(st, W, K, nt, nrows, ap, ep);
    return launch_ksplit_tile<1, 1, FRAG>(st, W, K, nt, nrows, ap, ep);
}
template <bool FRAG, class AP, class EP>
static int launch_ksplit_f32(hipStream_t st, const float* W, int K, int nt, int nrows, AP ap, EP ep) {
    if (nt % 5 == 0) return launch_ksplit_ct<5, FRAG>(st, W, K, nt, nrows, ap, ep);
    if (nt % 4 == 0) return launch_ksplit_ct<4, FRAG>(st, W, K, nt, nrows, ap, ep);
    if (nt % 2 == 0) return launch_ksplit_ct<2, FRAG>(st, W, K, nt, nrows, ap, ep);
    return launch_ksplit_ct<1, FRAG>(st, W, K, nt, nrows, ap, ep);
}

// image kinds a split-K launch can take beside the plain weight
enum { IMG_NONE = 0, IMG_FRAG_F32 = 2 };

template <class AP, class EP>
static int launch_rows_direct(hipStream_t st, const float* W, const float* img, int kind, int K, int ncols, int nrows, AP ap, EP ep) {
    if (nrows <= 0) return CLIORA_OK;
    const int nt = ncols / 16;
    if (kind == IMG_FRAG_F32) return launch_ksplit_f32<true>(st, img, K, nt, nrows, ap, ep);
    return launch_ksplit_f32<false>(st, W, K, nt, nrows, ap, ep);
}

// out[r][j] = sum_k A(r,k) W[j][k] for j < ncols (multiple of 16); k runs over nseg segments of Kseg (multiple of 16)
template <class AP, class EP>
static int launch_rows(hipStream_t st, const float* W, int Kseg, int nseg, int ncols, int nrows, AP ap, EP ep) {
    if (nrows <= 0) return CLIORA_OK;
    const size_t budget = 150 * 1024;
    const int nt = ncols / 16;
    for (int ct : {5, 4, 2, 1}) {
        if (nt % ct) continue;
        if ((size_t)ct * 16 * (Kseg + WS_LDS_PAD) * sizeof(float) > budget) continue;
        switch (ct) {
            case 5: return launch_rows_ct<5>(st, W, Kseg, nseg, ncols, nrows, ap, ep);
            case 4: return launch_rows_ct<4>(st, W, Kseg, nseg, ncols, nrows, ap, ep);
            case 2: return launch_rows_ct<2>(st, W, Kseg, nseg, ncols, nrows, ap, ep);
            default: return launch_rows_ct<1>(st, W, Kseg, nseg, ncols, nrows, ap, ep);
        }
    }
    return fail(CLIORA_EINVAL, "weight block does not fit LDS");
}

// weight-gradient GEMMs of at most this many rows AND at most this many 16 x 16 output blocks run as one launch (tn_gemm_direct: a
// workgroup per block walks every row with element loads -- the shape of the configs[0]-sized plans; at d = 400 the 1 875 blocks of the
// low levels' projection gradient took 190 us that way, beside and slowing the pair rows' tail, against 35 us through the slab)
constexpr int TN_DIRECT_ROWS = 4096;
constexpr int TN_DIRECT_BLOCKS = 256;

template <int T, class AP, class BP>
static int launch_tn_t(hipStream_t st, int nrows, int Mi, int Nj, AP ap, BP bp, float* slab, size_t slab_floats,
                       float* out, float* colsum_out, int accumulate) {
    const int blocks = (Mi / (T * 16)) * (Nj / (T * 16));
    size_t per_slice = (size_t)Mi * Nj + (colsum_out ? Mi : 0);
    int nsl = (int)std::min<size_t>(slab_floats / per_slice, (size_t)std::max(1, 2048 / blocks));
    nsl = std::min(nsl, (nrows + 15) / 16);
    nsl = std::max(4, nsl / 4 * 4);
    if ((size_t)nsl * per_slice > slab_floats) return fail(CLIORA_ENOMEM, "slab too small for the weight-gradient GEMM");
    int rps = (nrows + nsl - 1) / nsl;
    rps = (rps + 3) / 4 * 4;
    float* csl = slab + (size_t)nsl * Mi * Nj;
    if (colsum_out)
        hipLaunchKernelGGL((tn_gemm<T, T, true, AP, BP>), dim3(blocks, nsl / 4), dim3(WS_THREADS), 0, st, nrows, rps, Mi, Nj, ap, bp, slab, csl);
    else
        hipLaunchKernelGGL((tn_gemm<T, T, false, AP, BP>), dim3(blocks, nsl / 4), dim3(WS_THREADS), 0, st, nrows, rps, Mi, Nj, ap, bp, slab, csl);
    LAUNCHOK("tn_gemm");
    const size_t n = (size_t)Mi * Nj;
    launch_slab_reduce(st, slab, nsl, n, out, accumulate, csl, (size_t)Mi, colsum_out);
    LAUNCHOK("slab_reduce");
    return CLIORA_OK;
}

// out[i][j] (+)= sum_r A(r,i) B(r,j); colsum_out[i] (+)= sum_r A(r,i) (optional)
template <class AP, class BP>
static int launch_tn(hipStream_t st, int nrows, int Mi, int Nj, int Dp, AP ap, BP bp, float* slab, size_t slab_floats,
                     float* out, float* colsum_out, int accumulate = 0) {
    if (nrows <= 0) {
        if (accumulate) return CLIORA_OK;
        HIPOK(hipMemsetAsync(out, 0, (size_t)Mi * Nj * sizeof(float), st));
        if (colsum_out) HIPOK(hipMemsetAsync(colsum_out, 0, (size_t)Mi * sizeof(float), st));
        return CLIORA_OK;
    }
    if (nrows <= TN_DIRECT_ROWS && (Mi / 16) * (Nj / 16) <= TN_DIRECT_BLOCKS) {            // small plans: one launch, no slab (tn_gemm_direct)
        const int blocks = (Mi / 16) * (Nj / 16);
        if (colsum_out)
            hipLaunchKernelGGL((tn_gemm_direct<true, AP, BP>), dim3(blocks), dim3(TND_WAVES * 64), 0, st, nrows, Mi, Nj, ap, bp, out, colsum_out, accumulate);
        else
            hipLaunchKernelGGL((tn_gemm_direct<false, AP, BP>), dim3(blocks), dim3(TND_WAVES * 64), 0, st, nrows, Mi, Nj, ap, bp, out, colsum_out, accumulate);
        LAUNCHOK("tn_gemm_direct");
        return CLIORA_OK;
    }
    switch (pick_tiles(Dp / 16)) {
        case 5: return launch_tn_t<5>(st, nrows, Mi, Nj, ap, bp, slab, slab_floats, out, colsum_out, accumulate);
        case 4: return launch_tn_t<4>(st, nrows, Mi, Nj, ap, bp, slab, slab_floats, out, colsum_out, accumulate);
        case 2: return launch_tn_t<2>(st, nrows, Mi, Nj, ap, bp, slab, slab_floats, out, colsum_out, accumulate);
        default: return launch_tn_t<1>(st, nrows, Mi, Nj, ap, bp, slab, slab_floats, out, colsum_out, accumulate);
    }
}

// the big weight-gradient GEMM over span-pair rows: C = DZ^T X (Mi = Nj = Dp), LDS-DMA fed, split over row slices
template <int NIT, int NJT>
static int launch_tn_pairs_inst(hipStream_t st, const float* DZ, const float* X, int nrows, int Dp, int nkb, float* slab,
                                size_t slab_floats, float* out, float* colsum_out, int accumulate, int slices_cap, int ldz, int ldx) {
    const size_t per_slice = (size_t)Dp * Dp + Dp;
    int nsl = (int)std::min<size_t>(slab_floats / per_slice, (size_t)std::max(1, 256 / nkb));
    nsl = std::max(1, std::min(nsl, (nrows + 63) / 64));
    int rps = (nrows + nsl - 1) / nsl;
    rps = (rps + TN_RS - 1) / TN_RS * TN_RS;
    nsl = (nrows + rps - 1) / rps;
    float* csl = slab + (size_t)nsl * Dp * Dp;
    const size_t lds3 = (size_t)2 * TN3_RS * (Dp + NJT * 16) * sizeof(float) + (size_t)NJT * 2048;   // two fp32 stages + the split column fragments
    const bool strided = ldz != Dp || ldx != Dp;           // operands that are column blocks of wider matrices: the eight-wave kernel only
    if (split_bf16() && lds3 <= 160 * 1024 && (Dp + NJT * 16) / 32 <= TN3_NP) {
        OKR(cliora_ensure_max_lds((const void*)tn_gemm_dma3<NIT, NJT, true>));
        // whole groups of 8 slices (one per XCD), all resident at once: one workgroup per CU, no second round -- and a tenth of the CUs
        // left to the small weight-gradient GEMMs of the side stream, which otherwise queue up behind this kernel (a workgroup of it fills
        // a CU's registers): measured at c2 with 80 / 72 / 64 / 56 slices: 3.566 / 3.538 / 3.539 / 3.576 ms per step
        static const int cap_env = [] { const char* e = getenv("CLIORA_WGRAD_SLICES"); return e ? atoi(e) : 0; }();
        const int nsl_cap = slices_cap > 0 ? slices_cap : cap_env > 0 ? cap_env : std::max(1, 230 / (8 * nkb)) * 8;
        if (nsl > nsl_cap) { nsl = nsl_cap; rps = (nrows + nsl - 1) / nsl; }
        rps = (rps + TN3_RS - 1) / TN3_RS * TN3_RS;
        nsl = (nrows + rps - 1) / rps;
        csl = slab + (size_t)nsl * Dp * Dp;
        static const bool eight = [] { const char* e = getenv("CLIORA_WGRAD_WAVES"); return !e || atoi(e) != 4; }();
        bool launched = false;
        if constexpr (NIT == 7 && NJT == 9) {              // d = 400: two waves per SIMD, each with half of the block's j-tiles
            if (eight) {                                   // (the second half has at most NJT - 5 = 4 of its 5 slots taken: the spare one holds the ones-tile)
                OKR(cliora_ensure_max_lds((const void*)tn_gemm_dma3x<NIT, NJT, 5, true>));
                hipLaunchKernelGGL((tn_gemm_dma3x<NIT, NJT, 5, true>), dim3(8 * nkb * ((nsl + 7) / 8)), dim3(512), lds3, st, DZ, ldz, X, ldx, nrows, rps, nsl, Dp, Dp, nkb, slab, csl);
                LAUNCHOK("tn_gemm_dma3x");
                launched = true;
            }
        }
        if (!launched) {
            if (strided) return fail(CLIORA_EINVAL, "strided operands need the eight-wave weight-gradient kernel (tn_pairs_strided_ok)");
            hipLaunchKernelGGL((tn_gemm_dma3<NIT, NJT, true>), dim3(8 * nkb * ((nsl + 7) / 8)), dim3(256), lds3, st, DZ, X, nrows, rps, nsl, Dp, Dp, nkb, slab, csl);
            LAUNCHOK("tn_gemm_dma3");
        }
    } else {
        if (strided) return fail(CLIORA_EINVAL, "strided operands need the eight-wave weight-gradient kernel (tn_pairs_strided_ok)");
        const size_t lds = (size_t)2 * TN_RS * (Dp + NJT * 16) * sizeof(float);
        OKR(cliora_ensure_max_lds((const void*)tn_gemm_dma<NIT, NJT, true>));
        hipLaunchKernelGGL((tn_gemm_dma<NIT, NJT, true>), dim3(nkb * nsl), dim3(256), lds, st, DZ, X, nrows, rps, Dp, Dp, nkb, slab, csl);
        LAUNCHOK("tn_gemm_dma");
    }
    const size_t n = (size_t)Dp * Dp;
    launch_slab_reduce(st, slab, nsl, n, out, accumulate, csl, (size_t)Dp, colsum_out);
    LAUNCHOK("slab_reduce");
    return CLIORA_OK;
}

// The eight-wave d = 400 kernel over TWO row ranges of the pair rows in one launch (rows [0, n1) and n2 rows from row `start2`): one
// slab and one reduction.  false: this shape has no such kernel (the caller launches the ranges one after the other).
static bool tn_pairs_two_ranges_ok(int Dp) {
    static const bool eight = [] { const char* e = getenv("CLIORA_WGRAD_WAVES"); return !e || atoi(e) != 4; }();
    static const bool off = [] { const char* e = getenv("CLIORA_WGRAD_TAIL_MERGED"); return e && atoi(e) == 0; }();
    const int NT = Dp / 16;
    return split_bf16() && eight && !off && NT > 20 && NT <= 27;
}
static int launch_tn_pairs_two_ranges(hipStream_t st, const float* DZ, const float* X, int n1, long long start2, int n2, int Dp, float* slab,
                                      size_t slab_floats, float* out, float* colsum_out, int slices_cap) {
    constexpr int NIT = 7, NJT = 9, nkb = 3;
    const size_t per_slice = (size_t)Dp * Dp + Dp;
    const size_t lds3 = (size_t)2 * TN3_RS * (Dp + NJT * 16) * sizeof(float) + (size_t)NJT * 2048;
    int cap = std::min<int>((int)(slab_floats / per_slice), slices_cap > 0 ? slices_cap : std::max(1, 230 / (8 * nkb)) * 8);
    cap = std::max(2, cap);
    int rps = (n1 + n2 + cap - 1) / cap;
    rps = (rps + TN3_RS - 1) / TN3_RS * TN3_RS;
    int s1 = (n1 + rps - 1) / rps, s2 = (n2 + rps - 1) / rps;
    while (s1 + s2 > cap) { rps += TN3_RS; s1 = (n1 + rps - 1) / rps; s2 = (n2 + rps - 1) / rps; }
    const int nsl = s1 + s2;
    float* csl = slab + (size_t)nsl * Dp * Dp;
    OKR(cliora_ensure_max_lds((const void*)tn_gemm_dma3x<NIT, NJT, 5, true>));
    hipLaunchKernelGGL((tn_gemm_dma3x<NIT, NJT, 5, true>), dim3(8 * nkb * ((nsl + 7) / 8)), dim3(512), lds3, st, DZ, Dp, X, Dp, n1, rps, nsl, Dp, Dp, nkb,
                       slab, csl, s1, n2, start2);
    LAUNCHOK("tn_gemm_dma3x(two ranges)");
    launch_slab_reduce(st, slab, nsl, (size_t)Dp * Dp, out, 0, csl, (size_t)Dp, colsum_out);
    LAUNCHOK("slab_reduce");
    return CLIORA_OK;
}

// out[i][j] (+)= sum over the cells [off, off + hi) of every sentence of A[cell][acol0 + i] * Bm[cell][j] (i, j < Dp; A of row stride lda,
// Bm of row stride Dp; colsum_out (+)= the column sums of that A block): one block of the projections' weight gradient on the
// eight-wave split-bf16 kernel, the rows remapped inside it.  Only where tn_pairs_strided_ok(Dp).
// nblk > 1 (round 4): the blocks acol0, acol0 + Dp, ... of A in ONE launch (gridDim.y) and one reduction -- out / colsum_out are then
// nblk consecutive Dp x Dp / Dp outputs.
static int launch_tn_level_block(hipStream_t st, const float* A, int lda, int acol0, const float* Bm, int nsent, int C, int off, int hi, int Dp,
                                 float* slab, size_t slab_floats, float* out, float* colsum_out, int accumulate, int slices_cap, int nblk = 1) {
    constexpr int NIT = 7, NJT = 9, nkb = 3;
    const int nrows = nsent * hi;
    if (nrows <= 0) return CLIORA_OK;
    const size_t per_slice = ((size_t)Dp * Dp + Dp) * nblk;
    const size_t lds3 = (size_t)2 * TN3_RS * (Dp + NJT * 16) * sizeof(float) + (size_t)NJT * 2048;
    int cap = std::min<int>((int)(slab_floats / per_slice), slices_cap > 0 ? slices_cap : std::max(1, 230 / (8 * nkb)) * 8);
    cap = std::max(1, std::min(cap, (nrows + 63) / 64));
    int rps = (nrows + cap - 1) / cap;
    rps = (rps + TN3_RS - 1) / TN3_RS * TN3_RS;
    const int nsl = (nrows + rps - 1) / rps;
    float* csl = slab + (size_t)nsl * nblk * Dp * Dp;
    OKR(cliora_ensure_max_lds((const void*)tn_gemm_dma3x<NIT, NJT, 5, true>));
    hipLaunchKernelGGL((tn_gemm_dma3x<NIT, NJT, 5, true>), dim3(8 * nkb * ((nsl + 7) / 8), nblk), dim3(512), lds3, st, A + acol0, lda, Bm, Dp, nrows, rps, nsl,
                       Dp, Dp, nkb, slab, csl, 0x7fffffff, 0, 0LL, hi, C, off, (long long)Dp);
    LAUNCHOK("tn_gemm_dma3x(level rows)");
    launch_slab_reduce(st, slab, nsl, (size_t)nblk * Dp * Dp, out, accumulate, csl, (size_t)nblk * Dp, colsum_out);
    LAUNCHOK("slab_reduce");
    return CLIORA_OK;
}

// The pair rows' weight gradient from the TILED split-bf16 operands (wgrad_tiles.hpp): tiles [tile0, tile0 + ntiles) and, in the same
// launch, [tile0b, tile0b + ntilesb).  Only where pair_tiles_ok(Dp).
static bool pair_tiles_ok(int Dp) {
    static const bool off = [] { const char* e = getenv("CLIORA_PAIR_TILES"); return e && atoi(e) == 0; }();
    return !off && split_bf16() && Dp == 400;      // the <7, 9, 5> instance and level_compose_bwd's <5, 25> instance
}
static int launch_tn_tiles(hipStream_t st, const float* DZt, const float* Xt, long long tile0, long long ntiles, long long tile0b, long long ntilesb,
                           int Dp, float* slab, size_t slab_floats, float* out, float* colsum_out, int accumulate, int slices_cap) {
    const int NT = Dp / 16;
    const long long nt = ntiles + ntilesb;
    if (nt <= 0) {
        if (accumulate) return CLIORA_OK;
        HIPOK(hipMemsetAsync(out, 0, (size_t)Dp * Dp * sizeof(float), st));
        if (colsum_out) HIPOK(hipMemsetAsync(colsum_out, 0, (size_t)Dp * sizeof(float), st));
        return CLIORA_OK;
    }
    constexpr int NTc = 25, NIT = 7, NJT = 9, NJW = 5, nkb = 3;      // d = 400 (pair_tiles_ok)
    if (NT != NTc) return fail(CLIORA_EINVAL, "tiled pair-row operands: d = 400 only");
    const size_t per_slice = (size_t)Dp * Dp + Dp;
    long long cap = std::min<long long>((long long)(slab_floats / per_slice), slices_cap > 0 ? slices_cap : std::max(1, 230 / (8 * nkb)) * 8);
    cap = std::max<long long>(2, std::min(cap, (nt + 3) / 4));
    long long tps = (nt + cap - 1) / cap;
    tps = (tps + 1) / 2 * 2;                                   // whole stages (two tiles)
    long long s1 = (ntiles + tps - 1) / tps, s2 = (ntilesb + tps - 1) / tps;
    while (s1 + s2 > cap) { tps += 2; s1 = (ntiles + tps - 1) / tps; s2 = (ntilesb + tps - 1) / tps; }
    const int nsl = (int)(s1 + s2);
    float* csl = slab + (size_t)nsl * Dp * Dp;
    const size_t lds = (size_t)2 * 2048 * (NT + NJT) + 1024;  // two stage buffers + the spare KiB the last X slot reads into
    // the next stage's LDS-DMA: 0 = two pieces beside each row tile's MFMAs, 1 = all of it right behind the barrier (the kernel alone at
    // L 40: 2.30 / 2.48 ms; c2 step 3.21 / 3.24 ms)
    static const int issue_mode = [] { const char* e = getenv("CLIORA_TILES_ISSUE"); return e ? atoi(e) : 0; }();
    OKR(cliora_ensure_max_lds((const void*)tn_gemm_tiles<NTc, NIT, NJT, NJW, true>));
    hipLaunchKernelGGL((tn_gemm_tiles<NTc, NIT, NJT, NJW, true>), dim3(8 * nkb * ((nsl + 7) / 8)), dim3(512), lds, st, reinterpret_cast<const uint32_t*>(DZt),
                       reinterpret_cast<const uint32_t*>(Xt), tile0, (int)ntiles, (int)tps, nsl, nkb, slab, csl, (int)s1, tile0b, (int)ntilesb, issue_mode);
    LAUNCHOK("tn_gemm_tiles");
    launch_slab_reduce(st, slab, nsl, (size_t)Dp * Dp, out, accumulate, csl, (size_t)Dp, colsum_out);
    LAUNCHOK("slab_reduce");
    return CLIORA_OK;
}

// accumulate: add to out / colsum_out instead of overwriting; slices_cap > 0: at most that many row slices (= workgroups / nkb)
// ldz / ldx: row strides of DZ / X when they are Dp-wide column blocks of wider matrices (0: Dp); only where tn_pairs_strided_ok()
static bool tn_pairs_strided_ok(int Dp) {
    static const bool eight = [] { const char* e = getenv("CLIORA_WGRAD_WAVES"); return !e || atoi(e) != 4; }();
    const int NT = Dp / 16;
    return split_bf16() && eight && NT > 20 && NT <= 27;       // the <7, 9> instance: d = 400
}
static int launch_tn_pairs(hipStream_t st, const float* DZ, const float* X, int nrows, int Dp, float* slab, size_t slab_floats,
                           float* out, float* colsum_out, int accumulate = 0, int slices_cap = 0, int ldz = 0, int ldx = 0) {
    if (ldz == 0) ldz = Dp;
    if (ldx == 0) ldx = Dp;
    if (nrows <= 0) {
        if (accumulate) return CLIORA_OK;
        HIPOK(hipMemsetAsync(out, 0, (size_t)Dp * Dp * sizeof(float), st));
        if (colsum_out) HIPOK(hipMemsetAsync(colsum_out, 0, (size_t)Dp * sizeof(float), st));
        return CLIORA_OK;
    }
    // ... and the pair rows only up to 1024 of them: at configs[0] (3 960 rows, 64 x 64 = 16 blocks) the direct form is 16 workgroups
    // walking 495 rows per wave, 42 us on the caller's stream behind resident_bwd, against 10 + 7 us through the LDS-DMA kernel and
    // the (round 4) parallel slab reduction
    if (nrows <= 1024 && (Dp / 16) * (Dp / 16) <= TN_DIRECT_BLOCKS && slices_cap == 0)
        return launch_tn(st, nrows, Dp, Dp, Dp, PlainRowsA{DZ, ldz}, PlainRowsA{X, ldx}, slab, slab_floats, out, colsum_out, accumulate);
    // tile counts are compile-time (straight-line MFMA code): NIT = ceil(NT/4) i-tiles per wave,
    // nkb column blocks of NJT = ceil(NT/nkb) j-tiles
    const int NT = Dp / 16;
#define TN_CASE(nit, njt, nkb) return launch_tn_pairs_inst<nit, njt>(st, DZ, X, nrows, Dp, nkb, slab, slab_floats, out, colsum_out, accumulate, slices_cap, ldz, ldx)
    if (NT <= 4) TN_CASE(1, 4, 1);
    if (NT <= 8) TN_CASE(2, 8, 1);
    if (NT <= 12) TN_CASE(3, 12, 1);
    if (NT <= 16) TN_CASE(4, 8, 2);
    if (NT <= 20) TN_CASE(5, 10, 2);
    if (NT <= 27) TN_CASE(7, 9, 3);     // 63 accumulator tiles = 252 registers: the most that stays spill-free
    TN_CASE(8, 8, 4);
#undef TN_CASE
}

// ---- split-bf16 weight-stationary rows GEMM (rows_gemm_ws3): out[r][j] = sum_k A(r,k) W[j][k], Wimg = split_weight_image of W
template <int CT, int WAVES, int K16, class AP, class EP>
static int launch_rows3_k(hipStream_t st, const uint32_t* Wimg, int S, int K, int ncols, int nrows, AP ap, EP ep) {
    constexpr int PD = 4;     // four k-steps of row operands in flight per wave; deeper rings (6, 7) measured the same on MI355X
    const size_t lds = (size_t)CT * 16 * S * sizeof(uint32_t);
    OKR(cliora_ensure_max_lds((const void*)rows_gemm_ws3<CT, WAVES, PD, K16, AP, EP>));
    const int ntiles = (nrows + 15) / 16;
    const int gy = ncols / (16 * CT);
    // same grid rule as the fp32 kernel: fewest passes over the row tiles, then the smallest grid that does it
    const int cap = std::max(1, 256 / gy);
    const int passes = (ntiles + WAVES * cap - 1) / (WAVES * cap);
    int gx = (ntiles + WAVES * passes - 1) / (WAVES * passes);
    if (gx >= 8 && (gx + 7) / 8 * 8 <= cap) gx = (gx + 7) / 8 * 8;
    hipLaunchKernelGGL((rows_gemm_ws3<CT, WAVES, PD, K16, AP, EP>), dim3(gx, gy), dim3(WAVES * 64), lds, st, Wimg, S, K, nrows, ap, ep);
    LAUNCHOK("rows_gemm_ws3");
    return CLIORA_OK;
}
template <int CT, int WAVES, class AP, class EP>
static int launch_rows3_inst(hipStream_t st, const uint32_t* Wimg, int S, int K, int ncols, int nrows, AP ap, EP ep) {
    if (K == 400) return launch_rows3_k<CT, WAVES, 25>(st, Wimg, S, K, ncols, nrows, ap, ep);       // d = 400: the fully unrolled instance
    return launch_rows3_k<CT, WAVES, 0>(st, Wimg, S, K, ncols, nrows, ap, ep);
}
template <class AP, class EP>
static int launch_rows3(hipStream_t st, const uint32_t* Wimg, int S, int K, int ncols, int nrows, AP ap, EP ep) {
    if (nrows <= 0) return CLIORA_OK;
    const size_t budget = 150 * 1024;
    const int nt = ncols / 16;
    const long long ntiles = (nrows + 15) / 16;
    for (int ct : {5, 4, 2, 1}) {
        if (nt % ct) continue;
        if ((size_t)ct * 16 * S * sizeof(uint32_t) > budget) continue;
        const bool two = ntiles * (nt / ct) > 1536;      // two waves per SIMD once the launch fills the chip
        switch (ct) {
#define WS3_CASE(c) case c: return two ? launch_rows3_inst<c, 8>(st, Wimg, S, K, ncols, nrows, ap, ep) : launch_rows3_inst<c, 4>(st, Wimg, S, K, ncols, nrows, ap, ep)
            WS3_CASE(5); WS3_CASE(4); WS3_CASE(2);
            default: return two ? launch_rows3_inst<1, 8>(st, Wimg, S, K, ncols, nrows, ap, ep) : launch_rows3_inst<1, 4>(st, Wimg, S, K, ncols, nrows, ap, ep);
#undef WS3_CASE
        }
    }
    return fail(CLIORA_EINVAL, "weight block does not fit LDS");
}

// split-bf16 images (see split_weight_image) of weight matrices already in the workspace
struct ImageList {
    SplitImageTab tab{};
    int n = 0, max_rows = 0, max_S = 0;
    void add(const float* src, float* dst, int nrows, int ldw, int K) {
        tab.src[n] = src; tab.dst[n] = reinterpret_cast<uint32_t*>(dst); tab.nrows[n] = nrows; tab.ldw[n] = ldw; tab.K[n] = K;
        max_rows = std::max(max_rows, nrows);
        max_S = std::max(max_S, (K + 31) / 32 * 32 + WS3_PAD);
        ++n;
    }
};
static int build_weight_images(hipStream_t st, const ImageList& l) {
    if (l.n == 0) return CLIORA_OK;
    hipLaunchKernelGGL(split_weight_image, dim3((l.max_S + 255) / 256, l.max_rows, l.n), dim3(256), 0, st, l.tab);
    LAUNCHOK("split_weight_image");
    return CLIORA_OK;
}
static int build_frag_images3(hipStream_t st, const ImageList& l) {       // split-bf16 fragment images (rows_gemm_ksplit3)
    if (l.n == 0) return CLIORA_OK;
    hipLaunchKernelGGL(frag_weight_image3, dim3(256, 1, l.n), dim3(256), 0, st, l.tab);
    LAUNCHOK("frag_weight_image3");
    return CLIORA_OK;
}
static int build_all_images(hipStream_t st, const ImageList& split, const ImageList& frag, const ImageList& frag3) {      // one launch for all three kinds
    const int n = split.n + frag.n + frag3.n;
    if (n == 0) return CLIORA_OK;
    hipLaunchKernelGGL(weight_images_all, dim3(256, 1, n), dim3(256), 0, st, split.tab, split.n, frag.tab, frag.n, frag3.tab);
    LAUNCHOK("weight_images_all");
    return CLIORA_OK;
}
// the RT x CT form (rows_gemm_ksplit3x): shape = 10 RT + CT
static constexpr int GEMM3_SHAPES[] = {11, 21, 22, 32, 42, 23, 33, 25, 13, 15, 24, 12, 14};
// a shape code from the environment, validated ONCE where it is read: an unknown code would otherwise fail every backward much later
// with a bare CLIORA_EINVAL.  0 (where allowed) = the kernel the variable replaces; anything else not in GEMM3_SHAPES falls back to 23.
static int gemm3_shape_env(const char* var, bool zero_ok) {
    const char* e = getenv(var);
    if (!e) return 23;
    const int v = atoi(e);
    if (v == 0 && zero_ok) return 0;
    for (int s : GEMM3_SHAPES) if (s == v) return v;
    fprintf(stderr, "cliora: %s=%s is not a tile shape of rows_gemm_ksplit3x (10 RT + CT in {11 21 22 32 42 23 33 25 13 15 24 12 14}%s); using 23\n", var, e,
            zero_ok ? ", or 0" : "");
    return 23;
}
template <class AP, class EP>
static int launch_rows_direct3x(hipStream_t st, const float* img3, int K, int ncols, int nrows, AP ap, EP ep, int shape) {
    if (nrows <= 0 || ncols <= 0) return CLIORA_OK;
    const int nt = ncols / 16, nrt = (nrows + 15) / 16;
    const uint32_t* I = reinterpret_cast<const uint32_t*>(img3);
#define R3X_CASE(rt, ct)                                                                                                             \
    case 10 * rt + ct: {                                                                                                             \
        const int nrg = (nrt + rt - 1) / rt, nrgp = nrg >= 8 ? (nrg + 7) / 8 * 8 : nrg;                                              \
        hipLaunchKernelGGL((rows_gemm_ksplit3x<rt, ct, AP, EP>), dim3(nrgp * ((nt + ct - 1) / ct)), dim3(256), 0, st, I, K, nrg, nrgp, nt, nrows, ap, ep); \
        break;                                                                                                                       \
    }
    switch (shape) {
        R3X_CASE(1, 1) R3X_CASE(2, 1) R3X_CASE(2, 2) R3X_CASE(3, 2) R3X_CASE(4, 2) R3X_CASE(2, 3) R3X_CASE(3, 3) R3X_CASE(2, 5)
        R3X_CASE(1, 3) R3X_CASE(1, 5) R3X_CASE(2, 4) R3X_CASE(1, 2) R3X_CASE(1, 4)
        default: return fail(CLIORA_EINVAL, "rows_gemm_ksplit3x: tile shape code " + std::to_string(shape) + " is not instantiated (CLIORA_BWD_GEMM3 / CLIORA_GEMM3_SHAPE)");
    }
#undef R3X_CASE
    LAUNCHOK("rows_gemm_ksplit3x");
    return CLIORA_OK;
}
// out = A W^T on split-bf16 products, W as its frag_weight_image3 (ncols a multiple of 16 output columns, the image's first `ncols`;
// K any multiple of 16): the TreeLSTM's gate projections and their backward in the default arithmetic mode
template <class AP, class EP>
static int launch_rows_direct3(hipStream_t st, const float* img3, int K, int ncols, int nrows, AP ap, EP ep) {
    if (nrows <= 0 || ncols <= 0) return CLIORA_OK;
    const int nt = ncols / 16, nrg = (nrows + 15) / 16;
    const uint32_t* I = reinterpret_cast<const uint32_t*>(img3);
    // RT x CT tiles (rows_gemm_ksplit3x) where the level has rows for them: CLIORA_GEMM3_SHAPE = 10 RT + CT, 0 = the 16 x 80 kernel below
    static const int shape3 = gemm3_shape_env("CLIORA_GEMM3_SHAPE", true);       // c5 L 20 / L 40: 5.54 / 30.56 ms (16 x 80) -> 5.38 / 29.21 (32 x 48); 2 x 2: 5.44 / 29.90, 2 x 5: 5.60 / 29.37
    if (shape3 > 0 && nrows >= 256) return launch_rows_direct3x(st, img3, K, ncols, nrows, ap, ep, shape3);
    const int nrgp = nrg >= 8 ? (nrg + 7) / 8 * 8 : nrg;
#define R3_CASE(ct) hipLaunchKernelGGL((rows_gemm_ksplit3<ct, AP, EP>), dim3(nrgp * (nt / ct)), dim3(256), 0, st, I, K, nrg, nrgp, nrows, ap, ep)
    if (nt % 5 == 0) R3_CASE(5);
    else if (nt % 4 == 0) R3_CASE(4);
    else if (nt % 2 == 0) R3_CASE(2);
    else R3_CASE(1);
#undef R3_CASE
    LAUNCHOK("rows_gemm_ksplit3");
    return CLIORA_OK;
}
static int build_frag_images(hipStream_t st, const ImageList& l) {
    if (l.n == 0) return CLIORA_OK;
    hipLaunchKernelGGL(frag_weight_image, dim3(256, 1, l.n), dim3(256), 0, st, l.tab);
    LAUNCHOK("frag_weight_image");
    return CLIORA_OK;
}

static int run_copies(hipStream_t st, const CopyTable& tab) {
    if (tab.n == 0) return CLIORA_OK;
    hipLaunchKernelGGL(copy2d_multi, dim3(64, tab.n), dim3(256), 0, st, tab);
    LAUNCHOK("copy2d_multi");
    return CLIORA_OK;
}

static void add_copy(CopyTable& t, float* dst, int ldd, int drows, int dcols, const float* s0, int ld0, int rows0, int cols0,
                     int r0, int c0, int T0, const float* s1 = nullptr, int ld1 = 0, int rows1 = 0, int cols1 = 0, int r1 = 0,
                     int c1 = 0, int T1 = 0) {
    CopyDesc& d = t.d[t.n++];
    d.dst = dst; d.ldd = ldd; d.drows = drows; d.dcols = dcols;
    d.s[0] = CopySrc{s0, ld0, rows0, cols0, r0, c0, T0};
    d.s[1] = CopySrc{s1, ld1, rows1, cols1, r1, c1, T1};
}

static inline unsigned cells_grid(int ncells) { return (unsigned)((ncells + 3) / 4); }

struct Dev {   // device views for one call
    const int32_t *arow, *brow, *trow;
    UseTab use[N_ROLES];
};
static Dev dev_views(const Plan& p) {
    Dev d;
    const int32_t* t = p.d_tables;
    d.arow = t + p.dev.arow; d.brow = t + p.dev.brow; d.trow = t + p.dev.trow;
    for (int r = 0; r < N_ROLES; ++r)
        d.use[r] = UseTab{t + p.dev.use_off[r], t + p.dev.use_row[r], t + p.dev.use_stride[r], t + p.dev.use_partner[r]};
    return d;
}

// sentence-affine block order of the one-workgroup-per-cell kernels (chart_kernels.hpp: cell_of_block); CLIORA_XCD_AFFINE=0: off
static int xcd_affine() { static const int v = [] { const char* e = getenv("CLIORA_XCD_AFFINE"); return e ? atoi(e) : 1; }(); return v; }
static LevelArgs level_args(const Plan& p, int level, bool outside_pass) {
    LevelArgs g;
    g.affine = xcd_affine();
    g.B = p.B; g.C = p.C; g.Dp = p.Dp; g.Lc = p.L - level;
    g.N = outside_pass ? p.Nout(level) : p.Nin(level);
    g.off = p.level_offset[level];
    g.rowbase = (int)(outside_pass ? p.row_base_out(level) : p.row_base_in(level));
    return g;
}

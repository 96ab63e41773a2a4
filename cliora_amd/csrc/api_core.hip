// C ABI, core unit: plan management, error / profiling plumbing, arithmetic mode, hook views and the CKY decode.
// See include/cliora_chart.h for the contract and the reference lines it replaces.
#include <set>
#include <utility>

#include "api_common.hpp"

thread_local std::string g_cliora_err;
ProfClass g_cliora_prof[CLIORA_KCLASS_COUNT];
int g_cliora_split_bf16 = -1;
std::mutex g_cliora_prof_mu;

int cliora_ensure_max_lds(const void* fn) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    HIPOK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({fn, dev})) return CLIORA_OK;
    HIPOK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    done.insert({fn, dev});
    return CLIORA_OK;
}

extern "C" int cliora_prof_enable(int cls, int on) {
    if (cls < 0 || cls >= CLIORA_KCLASS_COUNT) return fail(CLIORA_EINVAL, "bad kernel class");
    std::lock_guard<std::mutex> lk(g_cliora_prof_mu);
    g_cliora_prof[cls].on = on != 0;
    return CLIORA_OK;
}

extern "C" int cliora_prof_read(int cls, double* total_ms, long long* launches, void* stream) {
    if (cls < 0 || cls >= CLIORA_KCLASS_COUNT) return fail(CLIORA_EINVAL, "bad kernel class");
    std::lock_guard<std::mutex> lk(g_cliora_prof_mu);
    ProfClass& c = g_cliora_prof[cls];
    HIPOK(hipStreamSynchronize((hipStream_t)stream));
    for (size_t k = 0; k + 1 < c.used; k += 2) {
        float ms = 0;
        HIPOK(hipEventElapsedTime(&ms, c.ev[k], c.ev[k + 1]));
        c.total_ms += ms;
        c.launches += 1;
    }
    c.used = 0;
    if (total_ms) *total_ms = c.total_ms;
    if (launches) *launches = c.launches;
    c.total_ms = 0; c.launches = 0;
    return CLIORA_OK;
}

// ------------------------------------------------------------------ plan
extern "C" int cliora_plan_create(int B, int L, int D, int share, int normalize, int R, cliora_plan** out) {
    if (!out) return fail(CLIORA_EINVAL, "out is NULL");
    cliora_plan* pl = new (std::nothrow) cliora_plan();
    if (!pl) return fail(CLIORA_ENOMEM, "host allocation failed");
    const std::string e = build_plan(pl->p, B, L, D, share, normalize, R);
    if (!e.empty()) { delete pl; return fail(CLIORA_EINVAL, e); }
    *out = pl;
    return CLIORA_OK;
}

extern "C" int cliora_plan_create_ex(int B, int L, int D, int share, int normalize, int R, int arch, cliora_plan** out) {
    if (!out) return fail(CLIORA_EINVAL, "out is NULL");
    cliora_plan* pl = new (std::nothrow) cliora_plan();
    if (!pl) return fail(CLIORA_ENOMEM, "host allocation failed");
    const std::string e = build_plan(pl->p, B, L, D, share, normalize, R, arch);
    if (!e.empty()) { delete pl; return fail(CLIORA_EINVAL, e); }
    *out = pl;
    return CLIORA_OK;
}

extern "C" void cliora_plan_destroy(cliora_plan* plan) {
    if (!plan) return;
    if (plan->p.d_tables) (void)hipFree(plan->p.d_tables);
    delete plan;                       // streams and events belong to the device (device_lanes)
}

extern "C" size_t cliora_plan_fwd_workspace_bytes(const cliora_plan* plan) { return plan ? plan->p.fwd.total * sizeof(float) : 0; }
extern "C" size_t cliora_plan_pair_states_bytes(const cliora_plan* plan) { return plan ? plan->p.fwd.pair_h_floats * sizeof(float) : 0; }
extern "C" size_t cliora_plan_bwd_workspace_bytes(const cliora_plan* plan) { return plan ? plan->p.bwd.total * sizeof(float) : 0; }

extern "C" int cliora_plan_table(const cliora_plan* plan, const char* name, const int32_t** data, size_t* count) {
    if (!plan || !name || !data || !count) return fail(CLIORA_EINVAL, "NULL argument");
    cliora_plan* pl = const_cast<cliora_plan*>(plan);
    std::lock_guard<std::mutex> up(pl->upload_mu);                   // find_table may build the row maps; so may a concurrent upload
    const std::vector<int32_t>* v = find_table(pl->p, name);
    if (!v) return fail(CLIORA_EINVAL, std::string("unknown table ") + name);
    *data = v->data();
    *count = v->size();
    return CLIORA_OK;
}

// float offset of a named region of the forward workspace (tests and tooling: compare two runs region by region); (size_t)-1 = unknown
extern "C" size_t cliora_plan_fwd_offset(const cliora_plan* plan, const char* name) {
    if (!plan || !name) return (size_t)-1;
    const FwdLayout& f = plan->p.fwd;
    const std::string n = name;
    if (n == "t") return f.t;
    if (n == "pi") return f.pi;
    if (n == "po") return f.po;
    if (n == "sp") return f.sp;
    if (n == "pp") return f.pp;
    if (n == "hp") return f.hp;
    if (n == "hp_o") return f.hp_o;
    if (n == "ymask") return f.ymask;
    if (n == "nrmi") return f.nrmi;
    if (n == "nrmo") return f.nrmo;
    if (n == "qrleaf") return f.qrleaf;
    if (n == "total") return f.total;
    return (size_t)-1;
}

extern "C" size_t cliora_plan_device_bytes(const cliora_plan* plan) {
    if (!plan) return 0;
    const Plan& p = plan->p;
    size_t n = p.pair_a_in.size() + p.pair_b_in.size() + p.pair_a_out.size() + p.pair_b_out.size() + p.lvl_base_in.size() + p.level_geom.size() +
               (p.arch == 1 ? 3 * (size_t)(p.R_in + p.R_out) : 0);     // the batch-expanded row maps: TreeLSTM plans only
    for (int r = 0; r < N_ROLES; ++r) n += p.uses[r].off.size() + 3 * p.uses[r].row.size();
    return n * sizeof(int32_t);
}

// Side streams and fork / join / level events of one device, created at the first call there and kept for the life of the process.
struct DeviceLanes {
    hipStream_t side = nullptr, side2 = nullptr;
    hipStream_t side3 = nullptr;            // the callers' lane (cliora_device_side_stream), created at its first request
    hipEvent_t fork[3], join[3], level[CLIORA_MAX_L + 1];
    unsigned long long* trace = nullptr;    // TRACE_BYTES of device words for the diagnostic stamps of the resident kernels (CLIORA_RES_TRACE)
    int ncu = 0;
    std::mutex mu;
};
// Holds one workgroup for `ticks` of the 100 MHz wall clock: the probe of pick_concurrent_stream.
static __global__ void lane_probe(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
// HIP deals streams round-robin onto a few hardware queues; two streams on one queue run one after the other.  Create streams until
// one runs a probe kernel CONCURRENTLY with every stream of `busy` (the caller's stream, lanes picked before), measured once:
// 40 us probes on all of them take ~40 us together when they overlap and n x 40 us when any two share a queue.
static int pick_concurrent_stream(hipStream_t* busy, int nbusy, hipStream_t* out) {
    hipEvent_t e0 = nullptr, e1 = nullptr, ej = nullptr;
    HIPOK(hipEventCreate(&e0)); HIPOK(hipEventCreate(&e1)); HIPOK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    int clk_khz = 100000;
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    if (hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, cur_dev) != hipSuccess || clk_khz <= 0) clk_khz = 100000;
    const long long ticks = 40LL * clk_khz / 1000;
    std::vector<hipStream_t> rejected;
    hipStream_t pick = nullptr;
    // probes on the busy streams (+ the candidate c, if any) started together; best of two timed rounds after a warm-up round
    auto timed = [&](hipStream_t c, float* best) -> int {
        *best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            HIPOK(hipEventRecord(e0, busy[0]));
            for (int k = 1; k < nbusy; ++k) HIPOK(hipStreamWaitEvent(busy[k], e0, 0));
            if (c) HIPOK(hipStreamWaitEvent(c, e0, 0));
            for (int k = 0; k < nbusy; ++k) hipLaunchKernelGGL(lane_probe, dim3(1), dim3(64), 0, busy[k], ticks);
            if (c) hipLaunchKernelGGL(lane_probe, dim3(1), dim3(64), 0, c, ticks);
            for (int k = 1; k < nbusy; ++k) { HIPOK(hipEventRecord(ej, busy[k])); HIPOK(hipStreamWaitEvent(busy[0], ej, 0)); }
            if (c) { HIPOK(hipEventRecord(ej, c)); HIPOK(hipStreamWaitEvent(busy[0], ej, 0)); }
            HIPOK(hipEventRecord(e1, busy[0]));
            HIPOK(hipEventSynchronize(e1));
            float ms = 0.f;
            HIPOK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0) *best = std::min(*best, ms);
        }
        return CLIORA_OK;
    };
    float base = 0.f;                            // what the busy streams take without a candidate (they already overlap each other)
    OKR(timed(nullptr, &base));
    for (int attempt = 0; attempt < 8 && !pick; ++attempt) {
        hipStream_t c = nullptr;
        HIPOK(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
        float best = 0.f;
        OKR(timed(c, &best));
        if (best < base + 0.020f) pick = c;      // one more 40 us probe costs nothing when it overlaps, 40 us when it shares a queue
        else rejected.push_back(c);
    }
    if (!pick && !rejected.empty()) { pick = rejected.back(); rejected.pop_back(); }     // no concurrent queue found: still correct, only slower
    for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(ej);
    if (!pick) return fail(CLIORA_EHIP, "could not create a side stream");
    *out = pick;
    return CLIORA_OK;
}

static int device_lanes(int dev, hipStream_t st, DeviceLanes** out) {
    static std::mutex g_mu;
    static DeviceLanes* g_lanes[64] = {};
    if (dev < 0 || dev >= 64) return fail(CLIORA_EINVAL, "device index out of range");
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_lanes[dev]) {
        DeviceLanes* ln = new (std::nothrow) DeviceLanes();
        if (!ln) return fail(CLIORA_ENOMEM, "host allocation failed");
        hipStream_t busy[3] = {st, nullptr, nullptr};
        OKR(pick_concurrent_stream(busy, 1, &ln->side));
        busy[1] = ln->side;
        OKR(pick_concurrent_stream(busy, 2, &ln->side2));
        // The fork / join / level events order kernels of ONE device against each other, so a device-scope release at the record
        // would do (CLIORA_EVENT_SCOPE=device: hipEventReleaseToDevice instead of the default system-scope release).  Measured on
        // MI355X at c2 (round 4, tools/ab/env_ab.sh, three alternations): 3.512 / 3.547 / 3.528 ms (system) against 3.526 / 3.549 /
        // 3.567 (device) -- no gain, so the default stays.
        static const unsigned ev_flags = [] {
            const char* e = getenv("CLIORA_EVENT_SCOPE");
            return (unsigned)hipEventDisableTiming | ((e && !strcmp(e, "device")) ? (unsigned)hipEventReleaseToDevice : 0u);
        }();
        for (int k = 0; k < 3; ++k) {
            HIPOK(hipEventCreateWithFlags(&ln->fork[k], ev_flags));
            HIPOK(hipEventCreateWithFlags(&ln->join[k], ev_flags));
        }
        for (int k = 0; k <= CLIORA_MAX_L; ++k) HIPOK(hipEventCreateWithFlags(&ln->level[k], ev_flags));
        HIPOK(hipMalloc((void**)&ln->trace, TRACE_BYTES));
        HIPOK(hipMemsetAsync(ln->trace, 0, TRACE_BYTES, st));
        HIPOK(hipDeviceGetAttribute(&ln->ncu, hipDeviceAttributeMultiprocessorCount, dev));
        g_lanes[dev] = ln;
    }
    *out = g_lanes[dev];
    return CLIORA_OK;
}

// A fourth lane for the CALLERS of the chart path: work that is independent of the chart -- the word branch of a CLIORA step (word
// projections, word-region scorer, their backward; cliora.py:459-461) and the region-matrix half of the span-region scorer's backward --
// runs there beside the chart's two chains and its weight-gradient stream.  Picked like the library's own lanes (a stream that runs a
// probe kernel concurrently with the caller's stream and the two lanes), once per device, so that it does not land on one of their
// hardware queues.  The stream belongs to the library; callers only enqueue on it and order it against their own streams by events.
extern "C" int cliora_device_side_stream(void* caller_stream, void** out) {
    if (!out) return fail(CLIORA_EINVAL, "out is NULL");
    int dev = 0;
    HIPOK(hipGetDevice(&dev));
    hipStream_t st = (hipStream_t)caller_stream;
    DeviceLanes* ln = nullptr;
    OKR(device_lanes(dev, st, &ln));
    std::lock_guard<std::mutex> lk(ln->mu);
    if (!ln->side3) {
        hipStream_t busy[3] = {st, ln->side, ln->side2};
        OKR(pick_concurrent_stream(busy, 3, &ln->side3));
    }
    *out = (void*)ln->side3;
    return CLIORA_OK;
}

int cliora_plan_ready(cliora_plan* plan, hipStream_t st) {
    int dev = 0;
    HIPOK(hipGetDevice(&dev));
    auto check_device = [&]() -> int {
        if (dev != plan->device)
            return fail(CLIORA_EINVAL, "plan tables live on device " + std::to_string(plan->device) + " but the current device is " +
                                           std::to_string(dev) + " (call under the tensors' device)");
        return CLIORA_OK;
    };
    if (__atomic_load_n(&plan->uploaded, __ATOMIC_ACQUIRE)) return check_device();
    // plans are shared process-wide (cliora_amd/_lib.py) and ctypes drops the GIL: two host threads can make their first call on
    // one cold plan together.  One of them uploads, the other waits here and then sees `uploaded` (published last, release).
    std::lock_guard<std::mutex> up(plan->upload_mu);
    if (plan->uploaded) return check_device();
    std::vector<int32_t> flat = flatten_tables(plan->p);
    HIPOK(hipMalloc((void**)&plan->p.d_tables, flat.size() * sizeof(int32_t)));
    HIPOK(hipMemcpyAsync(plan->p.d_tables, flat.data(), flat.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIPOK(hipStreamSynchronize(st));   // `flat` dies at scope exit
    plan->p.d_tables_count = flat.size();
    DeviceLanes* ln = nullptr;
    OKR(device_lanes(dev, st, &ln));
    plan->side = ln->side; plan->side2 = ln->side2;
    for (int k = 0; k < 3; ++k) { plan->ev_fork[k] = ln->fork[k]; plan->ev_join[k] = ln->join[k]; }
    plan->ev_level = ln->level;
    plan->lanes_mu = &ln->mu;
    plan->ncu = ln->ncu; plan->trace_words = ln->trace;
    plan->device = dev;
    __atomic_store_n(&plan->uploaded, true, __ATOMIC_RELEASE);
    return CLIORA_OK;
}

// ------------------------------------------------------------------ hooks / CKY
extern "C" int cliora_inside_pair_states(const cliora_plan* plan, void* fwd_ws, int level, const float** scores, const float** h,
                                         size_t* rows, size_t* ldh) {
    if (!plan || !fwd_ws || !scores || !h || !rows || !ldh) return fail(CLIORA_EINVAL, "NULL argument");
    const Plan& p = plan->p;
    if (level < 1 || level >= p.L) return fail(CLIORA_EINVAL, "level out of range");
    const size_t r0 = (size_t)p.row_base_in(level);
    *scores = (const float*)fwd_ws + p.fwd.sp + r0;
    *h = (const float*)fwd_ws + p.fwd.pair_h + r0 * p.Dp;
    *rows = (size_t)p.B * (p.L - level) * level;
    *ldh = (size_t)p.Dp;
    return CLIORA_OK;
}

extern "C" int cliora_outside_pair_states(const cliora_plan* plan, void* fwd_ws, int level, const float** scores, const float** h,
                                          size_t* rows, size_t* ldh) {
    if (!plan || !fwd_ws || !scores || !h || !rows || !ldh) return fail(CLIORA_EINVAL, "NULL argument");
    const Plan& p = plan->p;
    if (level < 0 || level > p.L - 2) return fail(CLIORA_EINVAL, "level out of range");
    const size_t r0 = (size_t)p.row_base_out(level);
    *scores = (const float*)fwd_ws + p.fwd.sp + r0;
    *h = (const float*)fwd_ws + p.fwd.pair_h + r0 * p.Dp;
    *rows = (size_t)p.B * (p.L - level) * (p.L - level - 1);
    *ldh = (size_t)p.Dp;
    return CLIORA_OK;
}

// One wavefront per sentence.  val[] (chart of best scores) lives in LDS; leaves start at 1
// (analysis/cky.py:24-25, 39).  Candidate = (val_l + val_r) + (s_n - max_n s) in fp32, in that
// order (cky.py:83, utils.py:89-90); argmax keeps the first maximum (cky.py:86).
// One workgroup of CKY_WAVES wavefronts per sentence (round 5: was one wavefront walking every cell in turn -- 190 dependent steps at
// L 20; the cells of a level are independent, so wave w takes the positions w, w + CKY_WAVES, ... and the level ends with one barrier).
constexpr int CKY_WAVES = 4;
__global__ __launch_bounds__(CKY_WAVES * 64) void cky_kernel(int L, int C, const int32_t* __restrict__ level_off_tab_a, const int32_t* __restrict__ pair_a,
                                                 const int32_t* __restrict__ pair_b, const int32_t* __restrict__ lvl_base, int B,
                                                 const float* __restrict__ Sp, int32_t* __restrict__ split, int32_t* __restrict__ spans) {
    extern __shared__ float val[];                 // C best scores, then C chosen splits
    int32_t* bsp = reinterpret_cast<int32_t*>(val + C);
    (void)level_off_tab_a;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = threadIdx.x; c < C; c += CKY_WAVES * 64) val[c] = 1.f;
    for (int c = threadIdx.x; c < L; c += CKY_WAVES * 64) { bsp[c] = -1; if (split) split[(size_t)b * C + c] = -1; }
    __syncthreads();
    int off = L;   // cell id of (level 1, pos 0)
    for (int level = 1; level < L; ++level) {
        const int Lc = L - level, N = level;
        for (int pos = wave; pos < Lc; pos += CKY_WAVES) {          // operands live on lower levels: final since the last barrier
            const int loc = lvl_base[level] + pos * N;
            const size_t row0 = (size_t)B * lvl_base[level] + ((size_t)b * Lc + pos) * N;
            const bool an = lane < N;
            const float s = an ? Sp[row0 + lane] : -INFINITY;
            const float smax = wave_max(s);
            float cand = -INFINITY;
            if (an) cand = (val[pair_a[loc + lane]] + val[pair_b[loc + lane]]) + (s - smax);
            float best = cand; int bi = an ? lane : 0x7fffffff;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
                if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            if (lane == 0) { val[off + pos] = best; bsp[off + pos] = bi; if (split) split[(size_t)b * C + off + pos] = bi; }
        }
        __syncthreads();
        off += Lc;
    }
    // The tree's constituent spans (start, end), children before parents, left subtree first: the order of the REDUCE actions, i.e. what
    // the reference gets from get_spans(get_actions(tree)) (cliora/analysis/utils.py:3-49) -- L - 1 spans per sentence, the root last.
    // One lane walks the L - 1 internal nodes with an explicit stack (level, pos, phase) in LDS.
    if (spans && threadIdx.x == 0 && L > 1) {
        int32_t* stk = bsp + C;                     // 3 * L ints
        auto cell = [&](int level, int pos) { const int rem = L - level; return C - rem * (rem + 1) / 2 + pos; };
        int sp = 0, cnt = 0;
        stk[0] = L - 1; stk[1] = 0; stk[2] = 0;
        int32_t* out = spans + (size_t)b * (L - 1) * 2;
        while (sp >= 0) {
            const int level = stk[3 * sp], pos = stk[3 * sp + 1], phase = stk[3 * sp + 2];
            if (level == 0) { --sp; continue; }
            const int n = bsp[cell(level, pos)];
            if (phase == 0) { stk[3 * sp + 2] = 1; ++sp; stk[3 * sp] = n; stk[3 * sp + 1] = pos; stk[3 * sp + 2] = 0; }
            else if (phase == 1) { stk[3 * sp + 2] = 2; ++sp; stk[3 * sp] = level - n - 1; stk[3 * sp + 1] = pos + n + 1; stk[3 * sp + 2] = 0; }
            else { out[2 * cnt] = pos; out[2 * cnt + 1] = pos + level; ++cnt; --sp; }
        }
    }
}

static int launch_cky(cliora_plan* plan, void* fwd_ws, int32_t* split_out, int32_t* spans_out, void* stream, const char* who) {
    if (!plan || !fwd_ws) return fail(CLIORA_EINVAL, "NULL argument");
    if (!plan->uploaded) return fail(CLIORA_EINVAL, std::string(who) + " called before forward");
    OKR(cliora_plan_ready(plan, (hipStream_t)stream));
    Plan& p = plan->p;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)p.C * (sizeof(float) + sizeof(int32_t)) + (size_t)3 * (p.L + 1) * sizeof(int32_t);
    hipLaunchKernelGGL(cky_kernel, dim3(p.B), dim3(CKY_WAVES * 64), lds, st, p.L, p.C, (const int32_t*)nullptr,
                       p.d_tables + p.dev.pair_a_in, p.d_tables + p.dev.pair_b_in, p.d_tables + p.dev.lvl_base_in, p.B,
                       (const float*)fwd_ws + p.fwd.sp, split_out, spans_out);
    LAUNCHOK("cky_kernel");
    return CLIORA_OK;
}

extern "C" int cliora_cky_decode(cliora_plan* plan, void* fwd_ws, int32_t* split_out, void* stream) {
    if (!split_out) return fail(CLIORA_EINVAL, "NULL argument");
    return launch_cky(plan, fwd_ws, split_out, nullptr, stream, "cky");
}

extern "C" int cliora_cky_spans(cliora_plan* plan, void* fwd_ws, int32_t* split_out, int32_t* spans_out, void* stream) {
    if (!spans_out) return fail(CLIORA_EINVAL, "NULL argument");
    return launch_cky(plan, fwd_ws, split_out, spans_out, stream, "cky_spans");
}

extern "C" const char* cliora_last_error(void) { return g_cliora_err.c_str(); }
extern "C" int cliora_set_mfma_mode(int mode) {
    const int prev = split_bf16() ? CLIORA_MFMA_SPLIT_BF16 : CLIORA_MFMA_F32;
    g_cliora_split_bf16 = mode == CLIORA_MFMA_F32 ? 0 : 1;
    return prev;
}

int g_cliora_wavefront = [] { const char* e = getenv("CLIORA_WAVEFRONT"); return e ? std::max(0, std::min(atoi(e), 2)) : -1; }();
extern "C" int cliora_set_wavefront(int mode) {
    const int prev = g_cliora_wavefront;
    g_cliora_wavefront = mode < 0 ? -1 : std::min(mode, 2);
    return prev;
}

int g_cliora_resident = [] { const char* e = getenv("CLIORA_RESIDENT"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
int g_cliora_resident_max_pairs = [] { const char* e = getenv("CLIORA_RESIDENT_MAX_PAIRS"); return e ? atoi(e) : 1000; }();
extern "C" int cliora_set_resident(int mode) {
    const int prev = g_cliora_resident;
    g_cliora_resident = mode < 0 ? -1 : (mode != 0 ? 1 : 0);
    return prev;
}
// diagnostics: wall-clock stamps (100 MHz) of the last resident launch traced with CLIORA_RES_TRACE=1 (tools/resident_trace.py), `count` 64-bit words
extern "C" int cliora_resident_trace(cliora_plan* plan, unsigned long long* out, size_t count, void* stream) {
    if (!plan || !out) return fail(CLIORA_EINVAL, "NULL argument");
    if (!plan->uploaded) return fail(CLIORA_EINVAL, "no forward has run on this plan");
    if (count * 8 > TRACE_BYTES) return fail(CLIORA_EINVAL, "trace buffer holds fewer words");
    HIPOK(hipMemcpyAsync(out, plan->trace_words, count * 8, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPOK(hipStreamSynchronize((hipStream_t)stream));
    return CLIORA_OK;
}

extern "C" const char* cliora_version(void) { return "cliora_amd 0.3 (gfx950)"; }

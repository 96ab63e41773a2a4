// The level loop of the DioraMLP forward inside ONE launch (gfx950): cliora/net/diora.py:312-331 (inside_pass) and :378-398
// (outside_pass) as phases of a persistent kernel, one 512-thread workgroup per CU.
//
// Why: as separate launches every chart level is two dependent kernels that each pay a launch boundary (2.4 us), a re-staging of
// the SAME 133 KB weight block into LDS (2.5 us) and their ramp / drain, for a few microseconds of work (profiles/r02_*): the
// forward is a chain of ~76 latency floors.  Here the compose workgroups stage their column block of W2 ONCE and keep it in LDS
// across all levels, and a level boundary is a counter barrier.
//
// What makes the barrier cheap (tools/ubench/grid_barrier_bench.hip, MI355X): a flat agent-scope counter costs 3.9 us per phase
// when every workgroup waits right after it arrives, but 0.6 us when the wait comes one phase later -- the two passes are two
// independent chains (outside level L-k only needs inside levels <= k-1), so the phases interleave
//     C_I(k)  C_O(L-k)  P_I(k)  P_O(L-k)            C = compose + aggregate, P = norm + projection + the next level's scores
// and each chain's barrier latency hides under the other chain's phase (split-phase barrier: arrive after the phase, wait before
// the chain's next phase).
//
// Visibility between workgroups follows cdna_hip_programming.md Guideline 16, form R1: everything one phase hands to a later one
// (partial aggregates, projections, chart rows, scores) is stored write-through (sc1) and loaded with sc1 loads (L1 bypass); every
// storing wave drains its stores, the workgroup meets at a barrier, ONE lane adds to the chain's counter; ONE lane polls.
// No fence, no L2 write-back, no cache invalidate.  Weights and index tables are written before the launch and read normally.
//
// Arithmetic and summation order are those of level_kernels.hpp (level_compose_fwd, level_project, score_cell, level_finish),
// instruction for instruction where a rounding happens: the results are bitwise those of the launch-per-level path
// (tests/test_gpu_persistent.py).  The projection GEMM tile is computed by ONE wave here (the launch-per-level kernel splits the
// reduction over four waves): the four reduction quarters keep their own accumulators and meet in the same order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "level_kernels.hpp"

namespace cliora {

// one chart level of one pass, as the persistent kernel needs it (built on the host with the plan: plan.cpp persist_levels)
struct PLevel { int32_t Lc, N, off, rowbase, pbase, TG, SP, ntask; };

struct PersistFwd {
    const int32_t* tabs;                       // the plan's device tables
    const PLevel* lev;                         // [2][L]: inside levels, then outside levels
    uint32_t pa_in, pb_in, pa_out, pb_out;     // pair tables inside `tabs`
    float *PI, *PO, *HPi, *HPo, *Pp, *Sp, *IH, *OH, *IS, *OS, *nrmi, *nrmo;
    uint32_t* ymask;                           // ReLU bits for the backward (nullptr: no backward will follow)
    float* Y;                                  // per-pair compose outputs for the hooks (nullptr: not wanted)
    const float* QRleaf;
    const uint32_t* Wimg[2];                   // W2 of the inside / outside compose: split-bf16 image (exact mode: the fp32 matrix)
    const float* b2[2];
    const float *wcat_frag, *bcat, *w1ro_frag; // projection weights as fragment images
    unsigned* sync;                            // [0] inside chain counter, [64] outside chain counter (zeroed before the launch)
    unsigned* status;                          // device-wide: [0] barrier timeouts (never reset by the kernel)
    int B, L, C, Dp, ldpi, blk_plo, blk_qlo, normalize, share, S, K, ncb, run_outside;
    uint32_t hp_stride_bytes;
    uint32_t bytes_PI, bytes_PO, bytes_HP, bytes_R, bytes_H, bytes_S;
};

// ---------------------------------------------------------------------------------------------------------------------------
// coherent accessors: raw buffer loads / stores with sc1 (aux bit 4): write-through stores, L1-bypassing loads
// ---------------------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) unsigned pk_gu32;
#define PK_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr int PK_SC1 = 16;
using pk_rsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ pk_rsrc pk_make(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x27000);
}
__device__ __forceinline__ float4 cld4(pk_rsrc r, uint32_t off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, PK_SC1);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float cld1(pk_rsrc r, uint32_t off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, PK_SC1));
}
__device__ __forceinline__ void cst4(pk_rsrc r, uint32_t off, float4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, r, (int)off, 0,
                                           PK_SC1);
}
__device__ __forceinline__ void cst1(pk_rsrc r, uint32_t off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)off, 0, PK_SC1);
}

// ---------------------------------------------------------------------------------------------------------------------------
// P-phase tasks, one WAVE each
// ---------------------------------------------------------------------------------------------------------------------------
struct PCells {            // the cells of one level as rows r = b*Lc + p of a phase
    int ncell, Lc, C, off;
    __device__ __forceinline__ uint32_t crow(int r) const {
        const int rc = min(r, ncell - 1);
        const int b = rc / Lc;
        return (uint32_t)(b * C + off + (rc - b * Lc));
    }
};

// sum of the level's SP partial aggregates at one 16-byte position: part 0 first, then += part s (level_kernels.hpp sum_parts)
template <int SP>
__device__ __forceinline__ float4 pk_sum_parts(const float4 (&p)[SP]) {
    float4 a = p[0];
#pragma unroll
    for (int s = 1; s < SP; ++s) a = f4add(a, p[s]);
    return a;
}

// Streams the 16 rows of row group rg through the wave in the fetch-lane map (row lane >> 2, 16-byte piece lane & 3), reduction
// chunk by chunk, calling body(chunk, a) with the summed operand; returns the rows' squared norms in the order of level_project:
// per reduction quarter (the four waves of the launch-per-level kernel) a running sum over its chunks, folded over the row's four
// fetch lanes, the quarters then added in order.  Quarter boundaries are reported through flush(w).
template <int SP, int PD, class Body, class Flush>
__device__ __forceinline__ float pk_stream_rows(pk_rsrc rHP, uint32_t aoff, uint32_t stride, int nchunks, Body& body, Flush flush) {
    const int cbase = nchunks >> 2, crem = nchunks & 3;
    const int e0 = cbase + (crem > 0 ? 1 : 0), e1 = e0 + cbase + (crem > 1 ? 1 : 0), e2 = e1 + cbase + (crem > 2 ? 1 : 0);
    float4 rp[PD][SP];
    auto load = [&](int slot, int ch) {
#pragma unroll
        for (int s = 0; s < SP; ++s) rp[slot][s] = cld4(rHP, aoff + (uint32_t)s * stride + (uint32_t)ch * 64u);
    };
#pragma unroll
    for (int sl = 0; sl < PD; ++sl)
        if (sl < nchunks) { load(sl, sl); body.prefetch(sl, sl); }
    float ss = 0.f, S0 = 0.f, S1 = 0.f, S2 = 0.f;
    auto fold = [](float v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); return v; };
    for (int base = 0; base < nchunks; base += PD) {
#pragma unroll
        for (int sl = 0; sl < PD; ++sl) {
            const int c = base + sl;
            if (c < nchunks) {
                const float4 av = pk_sum_parts<SP>(rp[sl]);
                ss += f4dot(av, av);
                body.consume(sl, c, av);
                if (c + PD < nchunks) { load(sl, c + PD); body.prefetch(sl, c + PD); }
                __builtin_amdgcn_sched_barrier(0);
                if (c + 1 == e0) { S0 = fold(ss); ss = 0.f; flush(0); }
                else if (c + 1 == e1) { S1 = fold(ss); ss = 0.f; flush(1); }
                else if (c + 1 == e2) { S2 = fold(ss); ss = 0.f; flush(2); }
            }
        }
    }
    flush(3);
    // a quarter that owns no chunk (K < 64) keeps its exact zero, as the idle waves of the launch-per-level kernel do
    const float S3 = fold(ss);
    return ((S0 + S1) + S2) + S3;
}

// one 16 x 16 tile of  P = (sum_parts g) Wcat^T / max(||g||, eps) + bias  for row group rg, column tile ct
struct PkGemmBody {
    const float4* Wf;          // this column tile's fragments, lane-resolved: chunk ch at Wf[ch * 64]
    int psrc;
    float4 rw[8];
    f32x4 acc;
    __device__ __forceinline__ void prefetch(int slot, int ch) { rw[slot] = Wf[(size_t)ch * 64]; }
    __device__ __forceinline__ void consume(int slot, int, const float4& av) {
        const float4 am = to_mfma_lanes(psrc, av);
        acc = mfma16(rw[slot].x, am.x, acc);
        acc = mfma16(rw[slot].y, am.y, acc);
        acc = mfma16(rw[slot].z, am.z, acc);
        acc = mfma16(rw[slot].w, am.w, acc);
    }
};

template <int SP, int PD>
__device__ __forceinline__ void pk_gemm_tile(const PersistFwd& a, int role, const PCells& cl, int rg, int ct, int lane) {
    static_assert(PD <= 8, "ring depth");
    const int K = a.Dp, nchunks = K >> 4;
    const int i = lane & 15, q = lane >> 4;
    const int li = fetch_row_of(lane), lq = fetch_piece_of(lane);
    const pk_rsrc rHP = pk_make(role ? a.HPo : a.HPi, a.bytes_HP);
    const uint32_t aoff = cl.crow(rg * 16 + li) * (uint32_t)K * 4u + 16u * lq;
    PkGemmBody body;
    body.Wf = reinterpret_cast<const float4*>(role ? a.w1ro_frag : a.wcat_frag) + (size_t)ct * nchunks * 64 + lane;
    body.psrc = mfma_src_addr(lane);
    body.acc = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 P0 = f32x4{0.f, 0.f, 0.f, 0.f}, P1 = P0, P2 = P0, P3 = P0;
    auto flush = [&](int w) {
        if (w == 0) P0 = body.acc; else if (w == 1) P1 = body.acc; else if (w == 2) P2 = body.acc; else P3 = body.acc;
        body.acc = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // the last quarter's accumulator is parked by the final flush(3); quarters without chunks stay zero
    const float ssum = pk_stream_rows<SP, PD>(rHP, aoff, a.hp_stride_bytes, nchunks, body, flush);
    const float nr = sqrtf(ssum);                                   // fetch lanes 4r .. 4r+3 hold row r
    const float den_f = a.normalize ? fmaxf(nr, UNIT_EPS) : 1.f;
    const float den = __shfl(den_f, 4 * i);
    float4 v = make_float4(((P0[0] + P1[0]) + P2[0]) + P3[0], ((P0[1] + P1[1]) + P2[1]) + P3[1], ((P0[2] + P1[2]) + P2[2]) + P3[2],
                           ((P0[3] + P1[3]) + P2[3]) + P3[3]);
    v = make_float4(v.x / den, v.y / den, v.z / den, v.w / den);
    const int col = ct * 16 + 4 * q;
    const float* bias = role ? nullptr : a.bcat;
    if (bias) v = f4add(v, ld4(bias + col));
    const int row = rg * 16 + i;
    if (row < cl.ncell) {
        const int ldp = role ? a.Dp : a.ldpi;
        const pk_rsrc rP = role ? pk_make(a.PO, a.bytes_PO) : pk_make(a.PI, a.bytes_PI);
        cst4(rP, (cl.crow(row) * (uint32_t)ldp + (uint32_t)col) * 4u, v);
    }
}

struct PkNormBody {
    __device__ __forceinline__ void prefetch(int, int) {}
    __device__ __forceinline__ void consume(int, int, const float4&) {}
};

// chart rows of row group rg: H = g / max(||g||, eps) with the norm of the projection launch (level_project, cb == 0 blocks)
template <int SP, int PD>
__device__ __forceinline__ void pk_hwrite(const PersistFwd& a, int role, const PCells& cl, int rg, int lane) {
    const int K = a.Dp, nchunks = K >> 4, nv = K >> 2;
    const int li = fetch_row_of(lane), lq = fetch_piece_of(lane);
    const pk_rsrc rHP = pk_make(role ? a.HPo : a.HPi, a.bytes_HP);
    const pk_rsrc rH = pk_make(role ? a.OH : a.IH, a.bytes_H);
    float* nrm = role ? a.nrmo : a.nrmi;
    const uint32_t aoff = cl.crow(rg * 16 + li) * (uint32_t)K * 4u + 16u * lq;
    PkNormBody body;
    const float ssum = pk_stream_rows<SP, PD>(rHP, aoff, a.hp_stride_bytes, nchunks, body, [](int) {});
    const float nr_f = sqrtf(ssum);
    for (int rr = 0; rr < 16; ++rr) {
        const int r = rg * 16 + rr;
        if (r >= cl.ncell) break;
        const float nr = __shfl(nr_f, 4 * rr);
        const float d = a.normalize ? fmaxf(nr, UNIT_EPS) : 1.f;
        const uint32_t crow = cl.crow(r);
        const uint32_t src = crow * (uint32_t)K * 4u;
        for (int v4 = lane; v4 < nv; v4 += 64) {
            float4 p[SP];
#pragma unroll
            for (int s = 0; s < SP; ++s) p[s] = cld4(rHP, src + (uint32_t)s * a.hp_stride_bytes + 16u * v4);
            const float4 g = pk_sum_parts<SP>(p);
            cst4(rH, src + 16u * v4, make_float4(g.x / d, g.y / d, g.z / d, g.w / d));
        }
        if (lane == 0) nrm[crow] = nr;
    }
}

// a level whose cells need no projection (inside root, outside leaves): level_finish for one row
__device__ __forceinline__ void pk_finish_row(const PersistFwd& a, int role, const PCells& cl, int SP, int r, int lane) {
    const int Dp = a.Dp, nv = Dp >> 2;
    const pk_rsrc rHP = pk_make(role ? a.HPo : a.HPi, a.bytes_HP);
    float* H = role ? a.OH : a.IH;
    float* nrm = role ? a.nrmo : a.nrmi;
    const uint32_t crow = cl.crow(r);
    float4 v0 = f4zero(), v1 = f4zero();
    for (int s = 0; s < SP; ++s) {
        const uint32_t src = (uint32_t)s * a.hp_stride_bytes + crow * (uint32_t)Dp * 4u;
        if (lane < nv) v0 = f4add(v0, cld4(rHP, src + 16u * lane));
        if (lane + 64 < nv) v1 = f4add(v1, cld4(rHP, src + 16u * (lane + 64)));
    }
    const float nr = sqrtf(wave_sum(f4dot(v0, v0) + f4dot(v1, v1)));
    const float den = a.normalize ? fmaxf(nr, UNIT_EPS) : 1.f;
    float* h = H + (size_t)crow * Dp;
    if (lane < nv) st4(h + 4 * lane, make_float4(v0.x / den, v0.y / den, v0.z / den, v0.w / den));
    if (lane + 64 < nv) st4(h + 4 * (lane + 64), make_float4(v1.x / den, v1.y / den, v1.z / den, v1.w / den));
    if (lane == 0) nrm[crow] = nr;
}

// split scores, softmax and cell score of target cell t of level T (score_cell of level_kernels.hpp, one wave):
//   s_n = QL(a_n) . h(b_n) + S(a_n) + S(b_n),   p = softmax_n s,   S(t) = sum_n p_n s_n          (diora.py:125-149)
// newest >= 0: the cells of that level of the same pass exist only as SPn partial aggregates (their partner is then a leaf).
__device__ __forceinline__ void pk_score_cell(const PersistFwd& a, int role, int T, int newest, int SPn, int t, int lane) {
    const PLevel g = a.lev[role * a.L + T];
    const int Dp = a.Dp, nv = Dp >> 2, C = a.C;
    const int b = t / g.Lc, p = t - b * g.Lc;
    const int row0 = g.rowbase + t * g.N;
    const bool a0 = lane < nv, a1 = lane + 64 < nv;
    const uint32_t c0 = 16u * lane, c1 = 16u * (lane + 64);
    const int bC = b * C;
    const int32_t* pa = a.tabs + (role ? a.pa_out : a.pa_in) + g.pbase + p * g.N;
    const int32_t* pb = a.tabs + (role ? a.pb_out : a.pb_in) + g.pbase + p * g.N;
    int new_lo = 0, new_hi = 0;
    if (newest >= 0) { new_lo = a.lev[role * a.L + newest].off; new_hi = new_lo + (a.L - newest); }
    const pk_rsrc rPI = pk_make(a.PI, a.bytes_PI);
    const pk_rsrc rHB = pk_make(role ? a.OH : a.IH, a.bytes_H);
    const pk_rsrc rHP = pk_make(role ? a.HPo : a.HPi, a.bytes_HP);
    const pk_rsrc rSA = pk_make(a.IS, a.bytes_S);
    const pk_rsrc rSB = pk_make(role ? a.OS : a.IS, a.bytes_S);
    const uint32_t qa_base = (uint32_t)(role ? a.blk_qlo : 2) * Dp * 4u, ldA = (uint32_t)a.ldpi * 4u;
    const bool a_can_be_new = role == 0;
    float my_s = -INFINITY;
    constexpr int NB = 4;
    for (int n0 = 0; n0 < g.N; n0 += NB) {
        float4 u0[NB], u1[NB], v0[NB], v1[NB];
        float sa[NB], sb[NB];
        bool isnew[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int n = min(n0 + j, g.N - 1);
            const int ca = pa[n], cb = pb[n];
            const int ar = bC + ca, br = bC + cb;
            const bool a_new = a_can_be_new && ca >= new_lo && ca < new_hi;
            const bool b_new = cb >= new_lo && cb < new_hi;
            isnew[j] = a_new || b_new;
            u0[j] = f4zero(); u1[j] = f4zero(); v0[j] = f4zero(); v1[j] = f4zero();
            auto newest_row = [&](int crow, float4& x0, float4& x1) {       // un-normalised h of a newest-level cell
                for (int sp = 0; sp < SPn; ++sp) {
                    const uint32_t src = (uint32_t)sp * a.hp_stride_bytes + (uint32_t)crow * Dp * 4u;
                    if (a0) x0 = f4add(x0, cld4(rHP, src + c0));
                    if (a1) x1 = f4add(x1, cld4(rHP, src + c1));
                }
            };
            if (a_new) {                               // partner is a leaf: QR(leaf) = M h_b
                newest_row(ar, u0[j], u1[j]);
                const float* qr = a.QRleaf + ((size_t)b * a.L + cb) * Dp;
                if (a0) v0[j] = ld4(qr + 4 * lane);
                if (a1) v1[j] = ld4(qr + 4 * (lane + 64));
            } else {
                const uint32_t qa = qa_base + (uint32_t)ar * ldA;
                if (a0) u0[j] = cld4(rPI, qa + c0);
                if (a1) u1[j] = cld4(rPI, qa + c1);
                if (b_new) newest_row(br, v0[j], v1[j]);
                else {
                    const uint32_t hb = (uint32_t)br * Dp * 4u;
                    if (a0) v0[j] = cld4(rHB, hb + c0);
                    if (a1) v1[j] = cld4(rHB, hb + c1);
                }
            }
            sa[j] = cld1(rSA, (uint32_t)ar * 4u);
            sb[j] = cld1(rSB, (uint32_t)br * 4u);
            // which operand is the newest one decides whose norm divides the dot: keep it in u (a_new) or v (b_new)
            if (b_new && !a_new) { const float4 t0 = u0[j], t1 = u1[j]; u0[j] = v0[j]; u1[j] = v1[j]; v0[j] = t0; v1[j] = t1; }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (n0 + j < g.N) {                        // wave-uniform
                float den = 1.f;
                if (isnew[j]) {                        // u holds the newest operand
                    const float nr = sqrtf(wave_sum(f4dot(u0[j], u0[j]) + f4dot(u1[j], u1[j])));
                    den = a.normalize ? fmaxf(nr, UNIT_EPS) : 1.f;
                }
                const float s = wave_sum(f4dot(u0[j], v0[j]) + f4dot(u1[j], v1[j])) / den + sa[j] + sb[j];
                if (lane == n0 + j) my_s = s;
            }
        }
    }
    const float m = wave_max(my_s);
    const float e = lane < g.N ? expf(my_s - m) : 0.f;
    const float pn = e / wave_sum(e);
    const pk_rsrc rSp = pk_make(a.Sp, a.bytes_R), rPp = pk_make(a.Pp, a.bytes_R);
    if (lane < g.N) { cst1(rSp, (uint32_t)(row0 + lane) * 4u, my_s); cst1(rPp, (uint32_t)(row0 + lane) * 4u, pn); }
    const float st = wave_sum(lane < g.N ? pn * my_s : 0.f);
    if (lane == 0) cst1(rSB, (uint32_t)(bC + g.off + p) * 4u, st);       // Sout = the pass's own score chart (= rSB)
}

// ---------------------------------------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------------------------------------
template <int CT, int K16, bool F32>
__global__ __launch_bounds__(512) void chart_fwd_persist(PersistFwd a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_img[];
    constexpr int WAVES = 8, T = WAVES * 64, PD = 4;
    constexpr bool KS = K16 > 0;
    constexpr int UNROLL_STEPS = KS ? 64 : 1;
    const int K = KS ? K16 * 16 : a.K;
    const int S = F32 ? K : (KS ? (K16 + 1) / 2 * 32 + WS3_PAD : a.S);     // row stride of the LDS image in dwords
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int li = fetch_row_of(lane), lg = fetch_piece_of(lane), psrc = mfma_src_addr(lane);
    const int Kp = F32 ? (K + 31) / 32 * 32 : S - WS3_PAD, half = Kp >> 1;
    const int NW = gridDim.x, wg = blockIdx.x;
    const int B = a.B, L = a.L, C = a.C, Dp = a.Dp;

    // ---- which column block of which compose weight this workgroup keeps in LDS.  Workgroups are dealt round-robin over the 8
    // XCDs by id: the ncb column blocks that gather the same operand rows (same bx) sit on one XCD (speed only).
    const int nslots = a.ncb * (a.share ? 1 : 2);
    int bx, by, gx;
    if ((NW & 7) == 0 && (NW >> 3) >= nslots) {
        const int x = wg & 7, q = wg >> 3, per = (NW >> 3) / nslots;
        by = q % nslots; gx = per * 8;
        bx = q / nslots < per ? (q / nslots) * 8 + x : -1;
    } else {
        gx = NW / nslots; by = wg % nslots;
        bx = wg / nslots < gx ? wg / nslots : -1;
    }
    const bool composer = bx >= 0;
    const int img = by / a.ncb, cbk = by - img * a.ncb;
    const int col0 = cbk * (CT * 16);
    if (composer) stage_weight_image(a.Wimg[img] + (size_t)col0 * S, lds_img, CT * 16 * S, wave, lane, T);
    float4* red = reinterpret_cast<float4*>(lds_img + CT * 16 * S);      // [LC_SLOTS][CT][64]
    volatile int* sh_dead = reinterpret_cast<volatile int*>(red + LC_SLOTS * CT * 64);   // behind the reduction slots (launcher: + 16 bytes)
    if (threadIdx.x == 0) *sh_dead = 0;
    float4 bv[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) bv[c] = ld4(a.b2[img] + col0 + c * 16 + 4 * g);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- split-phase barrier on the chain's counter ----
    unsigned ep0 = 0, ep1 = 0;                        // arrivals this workgroup has made on each chain
    auto arrive = [&](int chain) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // every wave: its write-through stores have left
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add((pk_gu32*)(a.sync + 64 * chain), 1u, PK_RLX);
        if (chain) ++ep1; else ++ep0;
    };
    auto wait = [&](int chain) -> bool {              // false: gave up (another kernel holds CUs this grid needs); leave
        if (threadIdx.x == 0) {
            const unsigned target = (chain ? ep1 : ep0) * (unsigned)NW;
            unsigned n = 0;
            while (__hip_atomic_load((pk_gu32*)(a.sync + 64 * chain), PK_RLX) < target) {
                if (++n > (1u << 22)) { atomicAdd(a.status, 1u); *sh_dead = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        return *sh_dead == 0;
    };

    const int nsteps = Kp >> 5;
    const int nsteps_p = (nsteps + PD - 1) / PD * PD;
    int wimg_off = 0;

    // ---- C phase: level_compose_fwd of one level of one pass, this workgroup's share of the tasks ----
    auto compose = [&](int role, int level) {
        if (!composer || (!a.share && img != role)) return;
        const PLevel lv = a.lev[role * L + level];
        const int32_t* pa = a.tabs + (role ? a.pa_out : a.pa_in) + lv.pbase;
        const int32_t* pb = a.tabs + (role ? a.pb_out : a.pb_in) + lv.pbase;
        const int ncell = B * lv.Lc;
        const pk_rsrc rA = pk_make(a.PI, a.bytes_PI);
        const pk_rsrc rB = role ? pk_make(a.PO, a.bytes_PO) : rA;
        const uint32_t baseA = (uint32_t)(role ? a.blk_plo * Dp : 0) * 4u + 16u * lg, ldA = (uint32_t)a.ldpi * 4u;
        const uint32_t baseB = (role ? 0u : (uint32_t)Dp * 4u) + 16u * lg, ldB = (uint32_t)(role ? Dp : a.ldpi) * 4u;
        const pk_rsrc rHP = pk_make(role ? a.HPo : a.HPi, a.bytes_HP);
        const pk_rsrc rPp = pk_make(a.Pp, a.bytes_R);
        const int TG = lv.TG, SP = lv.SP, ntask = lv.ntask;
        const int WPG = WAVES / TG;                      // waves per cell tile (1, 2, 4 or 8)
        const int j = wave / WPG, r = wave - j * WPG;
        const int G = (ncell + 15) >> 4;
        const int Ns = (lv.N + SP - 1) / SP;

        struct Ctx { uint32_t oa, ob; };
        auto rowctx = [&](int gt, int n) {               // fetch-lane view of tile (gt, n)
            const int t = min(gt * 16 + li, ncell - 1);                // clamp: computed, masked out by p = 0 and never stored
            const int b = t / lv.Lc, p = t - b * lv.Lc;
            const int idx = p * lv.N + n;
            return Ctx{baseA + (uint32_t)(b * C + pa[idx]) * ldA, baseB + (uint32_t)(b * C + pb[idx]) * ldB};
        };
        Raw2 ra[PD][2];
        auto issue = [&](int slot, const Ctx& c, int s) {
            const uint32_t k = 128u * s;
            const uint32_t k2 = k + (32 * s + 16 < K ? 64u : 0u);
            ra[slot][0] = Raw2{cld4(rA, c.oa + k), cld4(rB, c.ob + k)};
            ra[slot][1] = Raw2{cld4(rA, c.oa + k2), cld4(rB, c.ob + k2)};
        };
        auto relu_add = [](const Raw2& q) {
            return make_float4(fmaxf(q.u.x + q.v.x, 0.f), fmaxf(q.u.y + q.v.y, 0.f), fmaxf(q.u.z + q.v.z, 0.f), fmaxf(q.u.w + q.v.w, 0.f));
        };
        for (int task = bx; task < ntask; task += gx) {
            const int gg = task / SP, s = task - gg * SP;
            const int gt = gg * TG + j;                  // this wave's cell tile
            const bool have = gt < G;
            const int n0 = s * Ns, n1 = min(lv.N, n0 + Ns);
            f32x4 hacc[CT];
#pragma unroll
            for (int c = 0; c < CT; ++c) hacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            const bool work = have && n0 + r < n1;
            Ctx ctx = rowctx(min(gt, G - 1), work ? n0 + r : n0);
            if (work) {
                int n = n0 + r;
#pragma unroll
                for (int sl = 0; sl < PD; ++sl) issue(sl, ctx, sl < nsteps ? sl : 0);
                while (true) {
                    const int nn = n + WPG;
                    const bool has_next = nn < n1;
                    const Ctx ctxn = rowctx(gt, has_next ? nn : n);
                    const int ti = gt * 16 + i;
                    const bool ok = ti < ncell;
                    const size_t prow = (size_t)lv.rowbase + (size_t)min(ti, ncell - 1) * lv.N + n;
                    const float pn = ok ? cld1(rPp, (uint32_t)prow * 4u) : 0.f;
                    f32x4 acc[CT];
#pragma unroll
                    for (int c = 0; c < CT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
                    asm volatile("" : "+v"(wimg_off));           // keep the weight-fragment LDS reads inside the tile loop
                    const uint32_t* wimg = lds_img + wimg_off;
                    StepOperand cur = make_operand<F32>(psrc, relu_add(ra[0][0]), relu_add(ra[0][1]));
#pragma unroll UNROLL_STEPS
                    for (int base = 0; base < nsteps_p; base += PD) {
#pragma unroll
                        for (int sl = 0; sl < PD; ++sl) {
                            const int st = base + sl;
                            if (st < nsteps) {
                                StepOperand nxt = cur;
                                if (st + 1 < nsteps) nxt = make_operand<F32>(psrc, relu_add(ra[(sl + 1) % PD][0]), relu_add(ra[(sl + 1) % PD][1]));
                                kstep_mfma<CT, F32>(wimg, i, g, S, half, st, 32 * st + 16 < K, cur, acc);
                                const int nst = st + PD;
                                const bool in_cur = nst < nsteps;
                                issue(sl, pick_pod(in_cur, ctx, ctxn), in_cur ? nst : (sl < nsteps ? sl : 0));
                                __builtin_amdgcn_sched_barrier(0);
                                cur = nxt;
                            }
                        }
                    }
                    uint32_t bits = 0;
#pragma unroll
                    for (int c = 0; c < CT; ++c) {
                        const float y0 = fmaxf(acc[c][0] + bv[c].x, 0.f), y1 = fmaxf(acc[c][1] + bv[c].y, 0.f);
                        const float y2 = fmaxf(acc[c][2] + bv[c].z, 0.f), y3 = fmaxf(acc[c][3] + bv[c].w, 0.f);
                        hacc[c][0] = fmaf(pn, y0, hacc[c][0]); hacc[c][1] = fmaf(pn, y1, hacc[c][1]);
                        hacc[c][2] = fmaf(pn, y2, hacc[c][2]); hacc[c][3] = fmaf(pn, y3, hacc[c][3]);
                        bits |= ((y0 > 0.f ? 1u : 0u) | (y1 > 0.f ? 2u : 0u) | (y2 > 0.f ? 4u : 0u) | (y3 > 0.f ? 8u : 0u)) << (4 * c);
                        if (a.Y && ok) st4(a.Y + prow * Dp + col0 + c * 16 + 4 * g, make_float4(y0, y1, y2, y3));
                    }
                    if (a.ymask && ok) a.ymask[(prow * a.ncb + cbk) * 4 + g] = bits;
                    if (!has_next) break;
                    ctx = ctxn;
                    n = nn;
                }
            }
            // sum over the WPG waves of a cell tile: the fixed tree of level_compose_fwd
#pragma unroll
            for (int stride = 4; stride >= 1; stride >>= 1) {
                if (WPG >= 2 * stride) {
                    const bool holding = r < 2 * stride;
                    const bool writer = holding && r >= stride;
                    if (writer) {
                        const int slot = j * stride + (r - stride);
#pragma unroll
                        for (int c = 0; c < CT; ++c) red[(slot * CT + c) * 64 + lane] = make_float4(hacc[c][0], hacc[c][1], hacc[c][2], hacc[c][3]);
                    }
                    __syncthreads();
                    if (holding && !writer) {
                        const int slot = j * stride + r;
#pragma unroll
                        for (int c = 0; c < CT; ++c) {
                            const float4 v = red[(slot * CT + c) * 64 + lane];
                            hacc[c][0] += v.x; hacc[c][1] += v.y; hacc[c][2] += v.z; hacc[c][3] += v.w;
                        }
                    }
                    __syncthreads();
                }
            }
            if (have && r == 0) {
                const int ti = gt * 16 + i;
                if (ti < ncell) {
                    const int b = ti / lv.Lc, p = ti - b * lv.Lc;
                    const uint32_t o = (uint32_t)s * a.hp_stride_bytes + ((uint32_t)(b * C + lv.off + p) * Dp + col0 + 4 * g) * 4u;
#pragma unroll
                    for (int c = 0; c < CT; ++c) cst4(rHP, o + 64u * c, make_float4(hacc[c][0], hacc[c][1], hacc[c][2], hacc[c][3]));
                }
            }
        }
    };

    // ---- P phase: norm + projection of the level's cells, the next level's scores, as one-wave tasks over the whole grid ----
    const int gw = wave * NW + wg, nwv = WAVES * NW;
    auto project = [&](int role, int level) {
        const PLevel lv = a.lev[role * L + level];
        PCells cl{B * lv.Lc, lv.Lc, C, lv.off};
        const int nrg = (cl.ncell + 15) >> 4;
        const bool proj = role ? level >= 1 : level < L - 1;
        const int Tn = role ? level - 1 : level + 1;
        const int nct = proj ? (role ? Dp : a.ldpi) >> 4 : 0;
        const int ngemm = nrg * nct, nscore = proj ? B * (L - Tn) : 0, nh = proj ? nrg : 0, nfin = proj ? 0 : cl.ncell;
        const int total = ngemm + nscore + nh + nfin;
        for (int t = gw; t < total; t += nwv) {
            if (t < ngemm) {
                const int ct = t / nrg, rg = t - ct * nrg;
                if (lv.SP == 1) pk_gemm_tile<1, 8>(a, role, cl, rg, ct, lane);
                else if (lv.SP == 2) pk_gemm_tile<2, 6>(a, role, cl, rg, ct, lane);
                else pk_gemm_tile<4, 4>(a, role, cl, rg, ct, lane);
            } else if (t < ngemm + nscore) {
                pk_score_cell(a, role, Tn, level, lv.SP, t - ngemm, lane);
            } else if (t < ngemm + nscore + nh) {
                const int rg = t - ngemm - nscore;
                if (lv.SP == 1) pk_hwrite<1, 8>(a, role, cl, rg, lane);
                else if (lv.SP == 2) pk_hwrite<2, 6>(a, role, cl, rg, lane);
                else pk_hwrite<4, 4>(a, role, cl, rg, lane);
            } else {
                pk_finish_row(a, role, cl, lv.SP, t - ngemm - nscore - nh, lane);
            }
        }
    };

    // ---- first scores of both chains: every operand is final (leaves / the outside root, written before the launch) ----
    {
        const int n_in = L > 1 ? B * (L - 1) : 0;
        const int n_out = (a.run_outside && L > 1) ? B * 2 : 0;          // outside level L-2 has two cells per sentence
        for (int t = gw; t < n_in + n_out; t += nwv) {
            if (t < n_in) pk_score_cell(a, 0, 1, -1, 0, t, lane);
            else pk_score_cell(a, 1, L - 2, -1, 0, t - n_in, lane);
        }
        arrive(0);
        if (a.run_outside) arrive(1);
    }
    for (int k = 1; k <= L; ++k) {
        for (int sub = 0; sub < 4; ++sub) {
            const int role = sub & 1;
            const int level = role ? L - k : k;
            const bool exists = role ? (a.run_outside && k >= 2) : (k <= L - 1);
            if (!exists) continue;
            if (!wait(role)) return;
            if (sub < 2) compose(role, level); else project(role, level);
            arrive(role);
        }
    }
}

}  // namespace cliora

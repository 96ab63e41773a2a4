// The level loop of the DioraMLP forward inside ONE launch (gfx950): cliora/net/diora.py:312-331 (inside_pass) and :378-398
// (outside_pass) as phases of a persistent kernel, one 512-thread workgroup per CU.
//
// Why: as separate launches every chart level is two dependent kernels that each pay a launch boundary (2.4 us), a re-staging of
// the SAME 133 KB weight block into LDS (2.5 us) and their ramp / drain, for a few microseconds of work (profiles/r02_*): the
// forward is a chain of ~76 latency floors.  Here the compose workgroups stage their column block of W2 ONCE and keep it in LDS
// across all levels, and a level boundary is a grid barrier on counters.
//
// Schedule.  Outside level L-k only needs inside levels <= k-1, so step k = 1 .. L runs the two passes
// side by side, as two slots with a barrier after each:
//     C(k)   compose + softmax-weighted aggregate of inside level k AND outside level L-k: ONE task list over the compose
//            workgroups (with shared weights any of them serves either pass), so the two levels fill the chip together;
//     P(k)   unit norm + projection of both levels' new cells and the split scores of the levels they unlock, as one list
//            of workgroup units: projection units (16 rows x 2 groups of column tiles, the reduction split over four waves as in
//            level_project) and units of eight one-wave tasks (a cell's scores, a row group's chart rows).
// (A first version ran C_I, C_O, P_I, P_O as four phases with split-phase barriers -- 0.6 us each, tools/ubench/
// grid_barrier_bench.hip -- but serialised the two passes' work: 1.85 ms against 1.12 ms for the launches; profiles/r03_*.)
//
// Barrier: two-level (per-XCD counter, last arriver -> top counter -> the XCD's generation word), 2.2 us against 3.9 us for one
// flat counter (same bench).  Which XCD a workgroup runs on is read from the hardware (XCC_ID) and counted at kernel start:
// no assumption on placement.
//
// Visibility between workgroups follows cdna_hip_programming.md Guideline 16, form R1: everything one phase hands to a later one
// (partial aggregates, projections, chart rows, scores) is stored write-through (sc1) and loaded with sc1 loads (L1 bypass); every
// storing wave drains its stores, the workgroup meets at a barrier, ONE lane signals; ONE lane polls.  No fence, no L2 write-back,
// no cache invalidate (tools/ubench/handoff_latency_bench.hip: an sc1 load of a line another XCD published costs 570 cycles at
// first touch, 308 from L2 afterwards; plain 560 / 252).  Weights and index tables are written before the launch and read normally.
//
// Arithmetic and summation order are those of level_kernels.hpp (level_compose_fwd, level_project, score_cell, level_finish),
// instruction for instruction where a rounding happens (the library is built with -ffp-contract=off so that no expression is
// contracted differently in two kernels): the results are bitwise those of the launch-per-level path
// (tests/test_gpu_persistent.py).
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
    unsigned* sync;                            // PK_SYNC_WORDS barrier words (zeroed before the launch), see pk_barrier
    unsigned* status;                          // device-wide: [0] barrier timeouts (never reset by the kernel)
    unsigned long long* trace;                 // diagnostics (CLIORA_PERSIST_TRACE=1): [workgroup][phase][2] wall-clock stamps, else nullptr
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
// LDS behind the weight image: the compose reduction slots, the level table, the give-up flag
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int PK_TW = 4;                                     // column tiles per wave of a wide projection task
#ifndef PK_WIDE_TASKS
#define PK_WIDE_TASKS 0
#endif
constexpr bool PK_WIDE = PK_WIDE_TASKS != 0;                 // 0: one column tile per task everywhere (measured faster: finer tasks balance better)
constexpr int PK_PART_BYTES = LC_SLOTS * 5 * 64 * 16;         // the compose reduction slots [LC_SLOTS][CT <= 5][64] float4
constexpr int PK_SS_BYTES = 0;
constexpr int PK_MAX_L = 64;
constexpr int PK_TAB_BYTES = 2 * PK_MAX_L * 16;              // [pass][level] {rowbase, pbase, TG | SP << 8, ntask}
constexpr int PK_LDS_EXTRA = PK_PART_BYTES + PK_SS_BYTES + PK_TAB_BYTES + 16;
// barrier words (each on a 256-byte line of its own): [0] top, [64 * (1 + x)] counter of XCD x, [64 * (9 + x)] its generation,
// [64 * 17 + x] census
constexpr int PK_SYNC_WORDS = 1280;

struct PkLevel { int Lc, N, off, rowbase, pbase, TG, SP, ntask; };

struct PCells {            // the cells of one level as rows r = b*Lc + p
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

// ---------------------------------------------------------------------------------------------------------------------------
// P slot, one-wave tasks
// ---------------------------------------------------------------------------------------------------------------------------
// TW column tiles of row group rg for the whole reduction, by ONE wave:  P = (sum_parts g) W^T / max(||g||, eps) + bias.
// level_project splits the reduction over four waves and adds their partial tiles ((q0 + q1) + q2) + q3: here the four reduction
// quarters run one after the other in the same accumulators, each quarter's result parked in registers at its end, and meet in
// that order -- bitwise the same tile.  TW independent accumulators keep the matrix pipe busy (a single 16 x 16 tile is a chain of
// 100 dependent MFMAs), the operand ring runs PD chunks ahead with every load issued unconditionally (counted vmcnt).
// NCHT > 0: the number of 16-deep chunks at compile time (straight-line code); 0: K from the plan.
template <int NCHT, int TW, int SP, int PD>
__device__ __forceinline__ void pk_gemm_wave(const PersistFwd& a, int role, const PCells& cl, int rg, int ct0, int nct, int lane) {
    const int K = a.Dp;
    const int nchunks = NCHT > 0 ? NCHT : K >> 4;
    const int cbase = nchunks >> 2, crem = nchunks & 3;
    const int e0 = cbase + (crem > 0 ? 1 : 0), e1 = e0 + cbase + (crem > 1 ? 1 : 0), e2 = e1 + cbase + (crem > 2 ? 1 : 0);
    const int i = lane & 15, q = lane >> 4;
    const int li = fetch_row_of(lane), lq = fetch_piece_of(lane), psrc = mfma_src_addr(lane);
    const pk_rsrc rHP = pk_make(role ? a.HPo : a.HPi, a.bytes_HP);
    const uint32_t aoff = cl.crow(rg * 16 + li) * (uint32_t)K * 4u + 16u * lq;
    const float4* Wf = reinterpret_cast<const float4*>(role ? a.w1ro_frag : a.wcat_frag) + lane;
    size_t wofs[TW];
#pragma unroll
    for (int c = 0; c < TW; ++c) wofs[c] = (size_t)min(ct0 + c, nct - 1) * nchunks * 64;      // ragged last group: computed, not stored
    f32x4 acc[TW], P0[TW], P1[TW], P2[TW];
#pragma unroll
    for (int c = 0; c < TW; ++c) { acc[c] = f32x4{0.f, 0.f, 0.f, 0.f}; P0[c] = acc[c]; P1[c] = acc[c]; P2[c] = acc[c]; }
    float ss = 0.f, S0 = 0.f, S1 = 0.f, S2 = 0.f;
    float4 ra[PD][SP];
    float4 rw[PD][TW];
    auto load = [&](int slot, int ch) {
        const int cc = min(ch, nchunks - 1);
#pragma unroll
        for (int s = 0; s < SP; ++s) ra[slot][s] = cld4(rHP, aoff + (uint32_t)s * a.hp_stride_bytes + (uint32_t)cc * 64u);
#pragma unroll
        for (int c = 0; c < TW; ++c) rw[slot][c] = Wf[wofs[c] + (size_t)cc * 64];
    };
    auto fold = [](float v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); return v; };
#pragma unroll
    for (int sl = 0; sl < PD; ++sl) load(sl, sl);
    constexpr int UNROLL_ALL = NCHT > 0 ? 64 : 1;
#pragma unroll UNROLL_ALL
    for (int base = 0; base < nchunks; base += PD) {
#pragma unroll
        for (int sl = 0; sl < PD; ++sl) {
            const int c = base + sl;
            if (c < nchunks) {
                const float4 av = pk_sum_parts<SP>(ra[sl]);
                ss += f4dot(av, av);
                const float4 am = to_mfma_lanes(psrc, av);
#pragma unroll
                for (int t = 0; t < TW; ++t) acc[t] = mfma16(rw[sl][t].x, am.x, acc[t]);
#pragma unroll
                for (int t = 0; t < TW; ++t) acc[t] = mfma16(rw[sl][t].y, am.y, acc[t]);
#pragma unroll
                for (int t = 0; t < TW; ++t) acc[t] = mfma16(rw[sl][t].z, am.z, acc[t]);
#pragma unroll
                for (int t = 0; t < TW; ++t) acc[t] = mfma16(rw[sl][t].w, am.w, acc[t]);
                load(sl, c + PD);
                __builtin_amdgcn_sched_barrier(0);
                // end of a reduction quarter: park its tile and its share of the squared norms, restart from zero
                const bool f0 = c + 1 == e0, f1 = !f0 && c + 1 == e1, f2 = !f0 && !f1 && c + 1 == e2;
                if (f0 || f1 || f2) {
                    const float sf = fold(ss);
#pragma unroll
                    for (int t = 0; t < TW; ++t) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            P0[t][j] = f0 ? acc[t][j] : P0[t][j];
                            P1[t][j] = f1 ? acc[t][j] : P1[t][j];
                            P2[t][j] = f2 ? acc[t][j] : P2[t][j];
                        }
                        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    S0 = f0 ? sf : S0; S1 = f1 ? sf : S1; S2 = f2 ? sf : S2;
                    ss = 0.f;
                }
            }
        }
    }
    // a quarter that owns no chunk (K < 64) keeps its exact zero, as the idle waves of the launch-per-level kernel do; whatever is
    // still in the accumulators is the LAST non-empty quarter, which is quarter 3 exactly when e2 < nchunks
    const float S3 = fold(ss);
    const float nr = sqrtf(((S0 + S1) + S2) + S3);                  // fetch lanes 4r .. 4r+3 hold row r
    const float den_f = a.normalize ? fmaxf(nr, UNIT_EPS) : 1.f;
    const float den = __shfl(den_f, 4 * i);
    const int row = rg * 16 + i;
    const int ldp = role ? a.Dp : a.ldpi;
    const pk_rsrc rP = role ? pk_make(a.PO, a.bytes_PO) : pk_make(a.PI, a.bytes_PI);
    const float* bias = role ? nullptr : a.bcat;
    const uint32_t prow = cl.crow(row) * (uint32_t)ldp;
#pragma unroll
    for (int t = 0; t < TW; ++t) {
        if (ct0 + t < nct) {
            float4 v = make_float4(((P0[t][0] + P1[t][0]) + P2[t][0]) + acc[t][0], ((P0[t][1] + P1[t][1]) + P2[t][1]) + acc[t][1],
                                   ((P0[t][2] + P1[t][2]) + P2[t][2]) + acc[t][2], ((P0[t][3] + P1[t][3]) + P2[t][3]) + acc[t][3]);
            v = make_float4(v.x / den, v.y / den, v.z / den, v.w / den);
            const int col = (ct0 + t) * 16 + 4 * q;
            if (bias) v = f4add(v, ld4(bias + col));
            if (row < cl.ncell) cst4(rP, (prow + (uint32_t)col) * 4u, v);
        }
    }
}

// Chart rows r0 .. r0+3 of a projected level: H = g / max(||g||, eps), with the norm summed as level_project sums it (per row
// 16 partial sums: reduction quarter x 16-byte piece of the chunk, each over its chunks in order; the four pieces folded by
// xor 1, xor 2; the quarters added in order).  16 lanes per row: lane = 16 * row + 4 * quarter + piece.
template <int SP>
__device__ __forceinline__ void pk_hrows(const PersistFwd& a, int role, const PCells& cl, int r0, int lane) {
    const int K = a.Dp, nchunks = K >> 4, nv = K >> 2;
    const int rl = lane >> 4, sub = lane & 15, qd = sub >> 2, lq = sub & 3;
    const int r = r0 + rl;
    const bool valid = r < cl.ncell;
    const pk_rsrc rHP = pk_make(role ? a.HPo : a.HPi, a.bytes_HP);
    const pk_rsrc rH = pk_make(role ? a.OH : a.IH, a.bytes_H);
    float* nrm = role ? a.nrmo : a.nrmi;
    const uint32_t crow = cl.crow(r);
    const uint32_t src = crow * (uint32_t)K * 4u;
    const int cbase = nchunks >> 2, crem = nchunks & 3;
    const int ch0 = qd * cbase + min(qd, crem);
    const int nch = cbase + (qd < crem ? 1 : 0);
    constexpr int MAXCH = 8;                       // K <= 512
    float4 rp[MAXCH][SP];
#pragma unroll
    for (int e = 0; e < MAXCH; ++e)
        if (e < nch) {
#pragma unroll
            for (int s = 0; s < SP; ++s) rp[e][s] = cld4(rHP, src + (uint32_t)s * a.hp_stride_bytes + (uint32_t)(ch0 + e) * 64u + 16u * lq);
        }
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < MAXCH; ++e)
        if (e < nch) { const float4 av = pk_sum_parts<SP>(rp[e]); ss += f4dot(av, av); }
    ss += __shfl_xor(ss, 1);
    ss += __shfl_xor(ss, 2);
    const int l0 = lane & 48;
    const float S0 = __shfl(ss, l0), S1 = __shfl(ss, l0 + 4), S2 = __shfl(ss, l0 + 8), S3 = __shfl(ss, l0 + 12);
    const float nr = sqrtf(((S0 + S1) + S2) + S3);
    const float d = a.normalize ? fmaxf(nr, UNIT_EPS) : 1.f;
    constexpr int MAXV = 8;                        // K <= 512: 128 vectors per row over 16 lanes
    float4 wp[MAXV][SP];
#pragma unroll
    for (int e = 0; e < MAXV; ++e)
        if (sub + 16 * e < nv) {
#pragma unroll
            for (int s = 0; s < SP; ++s) wp[e][s] = cld4(rHP, src + (uint32_t)s * a.hp_stride_bytes + 16u * (sub + 16 * e));
        }
#pragma unroll
    for (int e = 0; e < MAXV; ++e)
        if (sub + 16 * e < nv) {
            const float4 g = pk_sum_parts<SP>(wp[e]);
            if (valid) cst4(rH, src + 16u * (sub + 16 * e), make_float4(g.x / d, g.y / d, g.z / d, g.w / d));
        }
    if (sub == 0 && valid) nrm[crow] = nr;
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
__device__ __forceinline__ void pk_score_cell(const PersistFwd& a, int role, const PkLevel& g, int new_lo, int new_hi, int SPn, int t, int lane) {
    const int Dp = a.Dp, nv = Dp >> 2, C = a.C;
    const int b = t / g.Lc, p = t - b * g.Lc;
    const int row0 = g.rowbase + t * g.N;
    const bool a0 = lane < nv, a1 = lane + 64 < nv;
    const uint32_t c0 = 16u * lane, c1 = 16u * (lane + 64);
    const int bC = b * C;
    const int32_t* pa = a.tabs + (role ? a.pa_out : a.pa_in) + g.pbase + p * g.N;
    const int32_t* pb = a.tabs + (role ? a.pb_out : a.pb_in) + g.pbase + p * g.N;
    const int my_ca = lane < g.N ? pa[lane] : 0, my_cb = lane < g.N ? pb[lane] : 0;       // one split per lane (N <= 64)
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
            u0[j] = f4zero(); u1[j] = f4zero(); v0[j] = f4zero(); v1[j] = f4zero();
            sa[j] = 0.f; sb[j] = 0.f; isnew[j] = false;
            if (n0 + j < g.N) {                        // wave-uniform
                const int ca = __shfl(my_ca, n0 + j), cb = __shfl(my_cb, n0 + j);
                const int ar = bC + ca, br = bC + cb;
                const bool a_new = a_can_be_new && ca >= new_lo && ca < new_hi;
                const bool b_new = cb >= new_lo && cb < new_hi;
                isnew[j] = a_new || b_new;
                auto newest_row = [&](int crow, float4& x0, float4& x1) {       // un-normalised h of a newest-level cell
                    for (int sp = 0; sp < SPn; ++sp) {
                        const uint32_t src = (uint32_t)sp * a.hp_stride_bytes + (uint32_t)crow * Dp * 4u;
                        if (a0) x0 = f4add(x0, cld4(rHP, src + c0));
                        if (a1) x1 = f4add(x1, cld4(rHP, src + c1));
                    }
                };
                // u: the operand whose norm divides the dot when it is a newest-level cell; the products commute exactly
                if (a_new) {                               // partner is a leaf: QR(leaf) = M h_b
                    newest_row(ar, u0[j], u1[j]);
                    const float* qr = a.QRleaf + ((size_t)b * a.L + cb) * Dp;
                    if (a0) v0[j] = ld4(qr + 4 * lane);
                    if (a1) v1[j] = ld4(qr + 4 * (lane + 64));
                } else {
                    const uint32_t qa = qa_base + (uint32_t)ar * ldA;
                    if (a0) v0[j] = cld4(rPI, qa + c0);
                    if (a1) v1[j] = cld4(rPI, qa + c1);
                    if (b_new) newest_row(br, u0[j], u1[j]);
                    else {
                        const uint32_t hb = (uint32_t)br * Dp * 4u;
                        if (a0) u0[j] = cld4(rHB, hb + c0);
                        if (a1) u1[j] = cld4(rHB, hb + c1);
                    }
                }
                sa[j] = cld1(rSA, (uint32_t)ar * 4u);
                sb[j] = cld1(rSB, (uint32_t)br * 4u);
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (n0 + j < g.N) {
                float den = 1.f;
                if (isnew[j]) {
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
    unsigned char* ext = reinterpret_cast<unsigned char*>(lds_img + CT * 16 * S);
    float4* red = reinterpret_cast<float4*>(ext);                        // [LC_SLOTS][CT][64]
    int4* sh_tab = reinterpret_cast<int4*>(ext + PK_PART_BYTES + PK_SS_BYTES);
    volatile int* sh_flag = reinterpret_cast<volatile int*>(ext + PK_PART_BYTES + PK_SS_BYTES + PK_TAB_BYTES);   // [0] dead, [1] nx, [2] nlive
    for (int e = threadIdx.x; e < 2 * L; e += T) {
        const PLevel q = a.lev[e];
        sh_tab[e] = make_int4(q.rowbase, q.pbase, q.TG | (q.SP << 8), q.ntask);
    }
    float4 bv[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) bv[c] = ld4(a.b2[img] + col0 + c * 16 + 4 * g);

    // ---- the grid barrier.  XCD of this workgroup from the hardware; census of workgroups per XCD through the top counter ----
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    auto spin_ge = [&](unsigned* p, unsigned target) -> bool {
        for (unsigned n = 0;; ++n) {
            if (__hip_atomic_load((pk_gu32*)p, PK_RLX) >= target) return true;
            if (n > (1u << 22)) return false;        // seconds: another process holds CUs this grid needs; give up
            __builtin_amdgcn_s_sleep(1);
        }
    };
    if (threadIdx.x == 0) {
        sh_flag[0] = 0;
        // census word first, then the top counter as a RELEASE; the census reads follow the spin as ACQUIRES: a workgroup that has
        // seen all NW arrivals sees every census add (two relaxed adds to different words could be reordered by the compiler or land
        // out of order, and an under-counted nx releases every later barrier of this workgroup's XCD early)
        __hip_atomic_fetch_add((pk_gu32*)(a.sync + 64 * 17 + xcc), 1u, PK_RLX);
        __hip_atomic_fetch_add((pk_gu32*)a.sync, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (!spin_ge(a.sync, (unsigned)NW)) { atomicAdd(a.status, 1u); sh_flag[0] = 1; }
        unsigned nx = 0, nlive = 0;
        for (unsigned x = 0; x < 8; ++x) {
            const unsigned c = __hip_atomic_load((pk_gu32*)(a.sync + 64 * 17 + x), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            if (x == xcc) nx = c;
            nlive += c ? 1u : 0u;
        }
        sh_flag[1] = (int)nx; sh_flag[2] = (int)nlive;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // weight image staged
    __syncthreads();
    if (sh_flag[0]) return;
    const unsigned nx = (unsigned)sh_flag[1], nlive = (unsigned)sh_flag[2];
    unsigned bar_ph = 0;
    auto barrier = [&]() -> bool {                     // false: gave up; every workgroup leaves
        ++bar_ph;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // every wave: its write-through stores have left
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned* gen = a.sync + 64 * (9 + xcc);
            const unsigned old = __hip_atomic_fetch_add((pk_gu32*)(a.sync + 64 * (1 + xcc)), 1u, PK_RLX);
            bool ok;
            if (old + 1 == bar_ph * nx) {            // last of this XCD: tell the top counter, wait for every XCD, release this one
                __hip_atomic_fetch_add((pk_gu32*)a.sync, 1u, PK_RLX);
                ok = spin_ge(a.sync, (unsigned)NW + bar_ph * nlive);
                if (ok) __hip_atomic_store((pk_gu32*)gen, bar_ph, PK_RLX);
            } else {
                ok = spin_ge(gen, bar_ph);
            }
            if (!ok) { atomicAdd(a.status, 1u); sh_flag[0] = 1; }
        }
        __syncthreads();
        return sh_flag[0] == 0;
    };
    auto level_of = [&](int role, int level) {         // uniform: the table entry + what follows from (L, level)
        const int4 e = sh_tab[role * L + level];
        PkLevel v;
        v.Lc = L - level; v.N = role ? L - level - 1 : level; v.off = C - (L - level) * (L - level + 1) / 2;
        v.rowbase = __builtin_amdgcn_readfirstlane(e.x); v.pbase = __builtin_amdgcn_readfirstlane(e.y);
        const int ge = __builtin_amdgcn_readfirstlane(e.z);
        v.TG = ge & 255; v.SP = ge >> 8; v.ntask = __builtin_amdgcn_readfirstlane(e.w);
        return v;
    };

    const int nsteps = Kp >> 5;
    const int nsteps_p = (nsteps + PD - 1) / PD * PD;
    int wimg_off = 0;
    int tph = 0;                                      // trace slot of the running phase (2 * k + slot)
    auto stamp = [&](int which) {
        if (a.trace && threadIdx.x == 0) a.trace[((size_t)wg * (2 * (L + 1)) + tph) * 2 + which] = __builtin_amdgcn_s_memrealtime();
    };
#ifdef CLIORA_PERSIST_STAMPS      // diagnostic build: eight more stamps per (workgroup, slot), behind the slot stamps
#define PK_STAMP(id) do { if (a.trace && threadIdx.x == 0) a.trace[(size_t)NW * (2 * (L + 1)) * 2 + ((size_t)wg * (2 * (L + 1)) + tph) * 8 + (id)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PK_STAMP(id) do {} while (0)
#endif

    // ---- one task of level_compose_fwd: (group of TG cell tiles) x (part of the split range), the whole workgroup ----
    auto compose_task = [&](int role, const PkLevel& lv, int task) {
        const int32_t* pa = a.tabs + (role ? a.pa_out : a.pa_in) + lv.pbase;
        const int32_t* pb = a.tabs + (role ? a.pb_out : a.pb_in) + lv.pbase;
        const int ncell = B * lv.Lc;
        const pk_rsrc rA = pk_make(a.PI, a.bytes_PI);
        const pk_rsrc rB = role ? pk_make(a.PO, a.bytes_PO) : rA;
        const uint32_t baseA = (uint32_t)(role ? a.blk_plo * Dp : 0) * 4u + 16u * lg, ldA = (uint32_t)a.ldpi * 4u;
        const uint32_t baseB = (role ? 0u : (uint32_t)Dp * 4u) + 16u * lg, ldB = (uint32_t)(role ? Dp : a.ldpi) * 4u;
        const pk_rsrc rHP = pk_make(role ? a.HPo : a.HPi, a.bytes_HP);
        const pk_rsrc rPp = pk_make(a.Pp, a.bytes_R);
        const int TG = lv.TG, SP = lv.SP;
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
    };

    // ---- C slot of step k: inside level k and outside level L-k as ONE task list over the compose workgroups ----
    auto c_slot = [&](int k) {
        const bool exI = k <= L - 1, exO = a.run_outside && k >= 2;
        const PkLevel lvI = level_of(0, exI ? k : 1), lvO = level_of(1, exO ? L - k : 0);
        const int ntI = exI ? lvI.ntask : 0, ntO = exO ? lvO.ntask : 0;
        if (!composer) return;
        // shared weights: every compose workgroup serves both passes; else the inside image's workgroups take the inside tasks
        const int t0 = a.share ? bx : (img == 0 ? bx : ntI + bx);
        const int t1 = a.share ? ntI + ntO : (img == 0 ? ntI : ntI + ntO);
        for (int t = t0; t < t1; t += gx) {
            const int role = t >= ntI ? 1 : 0;
            compose_task(role, role ? lvO : lvI, role ? t - ntI : t);
        }
    };

    // ---- P slot of step k: both levels' projections, chart rows, and the scores of the levels they unlock, as one list of
    //      one-wave tasks over every wave of the grid (longest first) ----
    const int gw = wave * NW + wg, nwv = WAVES * NW;
    auto gemm_task = [&](int role, const PCells& cl, int SP, bool wide, int rg, int ct0, int nct) {
        constexpr int NCHT = KS ? K16 : 0;
        if (PK_WIDE && wide) pk_gemm_wave<NCHT, PK_TW, 1, 4>(a, role, cl, rg, ct0, nct, lane);
        else if (SP == 1) pk_gemm_wave<NCHT, 1, 1, 6>(a, role, cl, rg, ct0, nct, lane);
        else if (SP == 2) pk_gemm_wave<NCHT, 1, 2, 4>(a, role, cl, rg, ct0, nct, lane);
        else pk_gemm_wave<NCHT, 1, 4, 2>(a, role, cl, rg, ct0, nct, lane);
    };
    auto p_slot = [&](int k) {
        const bool exI = k <= L - 1, exO = a.run_outside && k >= 2;
        const int lI = exI ? k : 1, lO = exO ? L - k : 0;
        const PkLevel lvI = level_of(0, lI), lvO = level_of(1, lO);
        const bool projI = exI && lI < L - 1, finI = exI && lI == L - 1, projO = exO && lO >= 1, finO = exO && lO == 0;
        const PCells clI{B * lvI.Lc, lvI.Lc, C, lvI.off}, clO{B * lvO.Lc, lvO.Lc, C, lvO.off};
        const int nrgI = (clI.ncell + 15) >> 4, nrgO = (clO.ncell + 15) >> 4;
        const int nctI = a.ldpi >> 4, nctO = Dp >> 4;
        // projection tasks: PK_TW column tiles per wave while such tasks still cover half the waves, else one (more, shorter tasks)
        const int gwI = (nctI + PK_TW - 1) / PK_TW, gwO = (nctO + PK_TW - 1) / PK_TW;
        const bool wideI = PK_WIDE && lvI.SP == 1 && 2 * nrgI * gwI >= nwv, wideO = PK_WIDE && lvO.SP == 1 && 2 * nrgO * gwO >= nwv;
        const int gI = wideI ? gwI : nctI, gO = wideO ? gwO : nctO;
        const int ngI = projI ? nrgI * gI : 0, ngO = projO ? nrgO * gO : 0;
        // scores of inside level lI+1 / outside level lO-1, chart rows (4 per task), rows of a final level
        const PkLevel svI = level_of(0, projI ? lI + 1 : 1), svO = level_of(1, projO ? lO - 1 : 0);
        const int nsI = projI ? B * svI.Lc : 0, nsO = projO ? B * svO.Lc : 0;
        const int nhI = projI ? (clI.ncell + 3) >> 2 : 0, nhO = projO ? (clO.ncell + 3) >> 2 : 0;
        const int nfI = finI ? clI.ncell : 0, nfO = finO ? clO.ncell : 0;
        const int g1 = ngI + ngO, e0 = g1 + nsI, e1 = e0 + nsO, e2 = e1 + nhI, e3 = e2 + nhO, e4 = e3 + nfI, total = e4 + nfO;
#ifdef CLIORA_PERSIST_STAMPS
        unsigned long long tacc[4] = {0, 0, 0, 0}; unsigned tcnt[4] = {0, 0, 0, 0};
#endif
        for (int t = gw; t < total; t += nwv) {
#ifdef CLIORA_PERSIST_STAMPS
            const unsigned long long ts = __builtin_amdgcn_s_memrealtime();
            const int kind = t < g1 ? 0 : (t < e1 ? 1 : (t < e3 ? 2 : 3));
#endif
            if (t < ngI) {
                const int cg = t / nrgI, rg = t - cg * nrgI;
                gemm_task(0, clI, lvI.SP, wideI, rg, cg * (wideI ? PK_TW : 1), nctI);
            } else if (t < g1) {
                const int v = t - ngI;
                const int cg = v / nrgO, rg = v - cg * nrgO;
                gemm_task(1, clO, lvO.SP, wideO, rg, cg * (wideO ? PK_TW : 1), nctO);
            } else if (t < e0) pk_score_cell(a, 0, svI, lvI.off, lvI.off + lvI.Lc, lvI.SP, t - g1, lane);
            else if (t < e1) pk_score_cell(a, 1, svO, lvO.off, lvO.off + lvO.Lc, lvO.SP, t - e0, lane);
            else if (t < e3) {
                const int role = t >= e2 ? 1 : 0;
                const int r0 = 4 * (role ? t - e2 : t - e1);
                const int SP = role ? lvO.SP : lvI.SP;
                const PCells& cl = role ? clO : clI;
                if (SP == 1) pk_hrows<1>(a, role, cl, r0, lane);
                else if (SP == 2) pk_hrows<2>(a, role, cl, r0, lane);
                else pk_hrows<4>(a, role, cl, r0, lane);
            } else if (t < e4) pk_finish_row(a, 0, clI, lvI.SP, t - e3, lane);
            else pk_finish_row(a, 1, clO, lvO.SP, t - e4, lane);
#ifdef CLIORA_PERSIST_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            tacc[kind] += __builtin_amdgcn_s_memrealtime() - ts; tcnt[kind] += 1;
#endif
        }
#ifdef CLIORA_PERSIST_STAMPS
        if (a.trace && threadIdx.x == 0) {
            unsigned long long* f = a.trace + (size_t)NW * (2 * (L + 1)) * 2 + ((size_t)wg * (2 * (L + 1)) + tph) * 8;
            for (int q = 0; q < 4; ++q) { f[q] = tacc[q]; f[4 + q] = tcnt[q]; }
        }
#endif
    };

    // ---- first scores of both chains: every operand is final (leaves / the outside root, written before the launch) ----
    {
        const PkLevel sI = level_of(0, 1), sO = level_of(1, L - 2);
        const int n_in = B * (L - 1);
        const int n_out = a.run_outside ? B * 2 : 0;                     // outside level L-2 has two cells per sentence
        for (int t = gw; t < n_in + n_out; t += nwv) {
            if (t < n_in) pk_score_cell(a, 0, sI, 0, 0, 0, t, lane);
            else pk_score_cell(a, 1, sO, 0, 0, 0, t - n_in, lane);
        }
        stamp(0); stamp(1);
        if (!barrier()) return;
    }
    for (int k = 1; k <= L; ++k) {
        tph = 2 * k;
        stamp(0);
        c_slot(k);
        stamp(1);
        if (!barrier()) return;
        tph = 2 * k + 1;
        stamp(0);
        p_slot(k);
        stamp(1);
        if (k < L && !barrier()) return;
    }
}

}  // namespace cliora

// The pair rows' weight gradient dW2 = DZ^T X from operands that level_compose_bwd leaves ALREADY split and tiled (round 4).
//
// tn_gemm_dma3x (gemm_kernels.hpp) spends as many cycles forming operands as on its MFMAs: the contraction index is the ROW index of
// both matrices, so an operand register holds eight rows of one column -- every fp32 element comes out of LDS with a strided 4-byte
// read and is split into bf16 hi / lo inside the loop (8 LDS reads + ~40 VALU per operand, 12 operands per wave and 32-row stage
// against 105 MFMAs).  Here the producer stores, per 16-row tile T (one wave tile of level_compose_bwd: 16 target cells x one split)
// and 16-column tile c, the hi and the lo bf16 plane as 512 contiguous bytes each:
//     dword offset ((T * NT + c) * 2 + plane) * 128 + row * 8 + col / 2            (NT = Dp / 16; the same bytes per row as fp32)
// For the producer that is one 512-byte coalesced store per (tile, column tile, plane) instead of 16-byte pieces of 16 rows.  Here a
// stage (two tiles = 32 rows) is a straight LDS-DMA copy -- 2 x NT KiB of DZ, 2 x njt KiB of X, linear on both sides -- and
// ds_read_b64_tr_b16 (gfx950's transposing LDS read: a 16-lane group reads 4 rows x 16 columns of 16-bit elements and lane i receives
// column i of the four rows) delivers operand registers directly: lane (i, g) gets rows 4g .. 4g+3 of the stage's first tile in
// registers 0-1 and of its second tile in registers 2-3, for column i, of BOTH matrices -- which of the 32 k of the MFMA a row lands
// in is free as long as the two operands agree (the sum over rows is what is wanted).  A half-wave's two 4 x 16 blocks are 256
// contiguous bytes: conflict-free (SQ_LDS_BANK_CONFLICT = 0).  No VALU work in the loop besides addresses: 48 LDS reads and 105 MFMAs
// per wave and stage.  Work split (four i-groups x two j-halves of a j-block, row slices one per workgroup and XCD), accumulators,
// slab, ones-tile (the column sums of DZ = the bias gradient) and epilogue are tn_gemm_dma3x's.  Rows are counted in TILES here.
//
// The transposing reads are INLINE ASSEMBLY, not the builtin: with __builtin_amdgcn_ds_read_tr16_b64 the compiler (ROCm 7.2) puts an
// s_waitcnt vmcnt(0) in front of every group of reads that follows an LDS-DMA instruction -- it cannot tell that the DMA fills the
// OTHER stage buffer -- so the next stage's loads and this stage's MFMAs ran one after the other (measured at L 40, the kernel alone:
// 3.0 ms, against 1.7 ms for the loads alone and 1.9 ms for the reads + MFMAs alone).  The asm reads carry their own lgkmcnt waits
// (lgkm_wait ties the registers to the wait, so that no consumer is scheduled above it).
//
// Which pair row sits where inside the tiles is the producer's business (cliora::Plan::tile_base_in / _out): the sum runs over all
// of them.  Rows of a tile past the level's last cell are stored as zeros.
#pragma once
#include "gemm_kernels.hpp"

namespace cliora {

typedef uint32_t tr_u2 __attribute__((ext_vector_type(2)));

// a float4 as four bf16 hi + four bf16 lo: 8 bytes into each plane of a tile (`dst`: the lane's dword slot in the hi plane)
__device__ __forceinline__ void store_split_tile(uint32_t* dst, const float4 v) {
    const uint32_t h0 = pack_bf16(v.x, v.y), h1 = pack_bf16(v.z, v.w);
    const uint32_t l0 = pack_bf16(v.x - __uint_as_float(h0 << 16), v.y - __uint_as_float(h0 & 0xffff0000u));
    const uint32_t l1 = pack_bf16(v.z - __uint_as_float(h1 << 16), v.w - __uint_as_float(h1 & 0xffff0000u));
    // Non-temporal: the tiles (0.82 GB per c2 step, 13 GB at L 40) are read once, by the weight-gradient GEMM, after most of the backward --
    // written with the default policy they pass through the 256 MB Infinity Cache and push out the rows the next launches gather (DA, the
    // projections, dG).  c2 3.03-3.08 -> 2.97-3.01 ms, L 40 17.0 -> 16.6 (profiles/r05_notes.md section 13; the same hint on DA, HP, P and
    // the gathers' outputs -- all read by one of the next launches -- loses 0.5-2 %).
    typedef uint32_t v2u_ __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store(v2u_{h0, h1}, reinterpret_cast<v2u_*>(dst));
    __builtin_nontemporal_store(v2u_{l0, l1}, reinterpret_cast<v2u_*>(dst + 128));
}

// ds_read_b64_tr_b16 at LDS byte address `addr` (every lane of the wave must be active: the read gathers across lanes)
__device__ __forceinline__ tr_u2 lds_tr16(uint32_t addr) {
    tr_u2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr) : "memory");
    return r;
}
// the four reads of one operand pair: (hi plane, lo plane) x (first, second tile of the stage, `hs` bytes apart), raw
struct TrOperand { tr_u2 h0, l0, h1, l1; };
__device__ __forceinline__ TrOperand lds_tr_operand(uint32_t addr, uint32_t hs) {
    TrOperand o;
    o.h0 = lds_tr16(addr); o.l0 = lds_tr16(addr + 512u);
    o.h1 = lds_tr16(addr + hs); o.l1 = lds_tr16(addr + hs + 512u);
    return o;
}
__device__ __forceinline__ void lgkm_wait(TrOperand& o) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o.h0), "+v"(o.l0), "+v"(o.h1), "+v"(o.l1));
}
// (after the wait) second: the stage has a second tile -- else its half of the operand is zero
__device__ __forceinline__ void tr_finish(const TrOperand& o, bool second, u32x4& hi, u32x4& lo) {
    hi = u32x4{o.h0.x, o.h0.y, second ? o.h1.x : 0u, second ? o.h1.y : 0u};
    lo = u32x4{o.l0.x, o.l0.y, second ? o.l1.x : 0u, second ? o.l1.y : 0u};
}

template <int NT, int NIT, int NJT, int NJW, bool COLSUM>
static __global__ __launch_bounds__(512) void tn_gemm_tiles(const uint32_t* __restrict__ A, const uint32_t* __restrict__ B, long long tile0,
                                                            int ntiles, int tiles_per_slice, int nslices, int nkb,
                                                            float* __restrict__ slab, float* __restrict__ colsum,
                                                            int slice2, long long tile0b, int ntilesb, int issue_mode) {
    // slices >= slice2 walk a SECOND tile range (ntilesb tiles from tile0b): the two ends of the pair rows around the part whose
    // weight gradient started early -- one launch, one slab, one reduction for both
    static_assert(2 * NJW > NJT, "the second half of the j-tiles needs a spare slot for the ones-tile");
    static_assert(4 * NIT >= NT && NT <= 27, "four i-groups of at most NIT tiles");
    extern __shared__ __attribute__((aligned(1024))) uint32_t lds_q[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int iw = wave & 3, jw = wave >> 2;
    const int i = lane & 15, g = lane >> 4;
    const int xcd = blockIdx.x & 7, wq = blockIdx.x >> 3;
    const int kb = wq % nkb, slice = (wq / nkb) * 8 + xcd;
    if (slice >= nslices) return;
    constexpr int Mi = NT * 16, Nj = Mi;
    const int jbase = NT / nkb, jrem = NT % nkb;
    const int jt0 = kb * jbase + min(kb, jrem);
    const int njt = jbase + (kb < jrem ? 1 : 0);            // j-tiles of this block (<= NJT)
    const int ju0 = jw * NJW;                               // first j-tile (block-relative) of this wave
    const int njw = max(0, min(njt - ju0, NJW));            // j-tiles of this wave
    constexpr int ibase = NT / 4, irem = NT % 4;
    const int it0 = iw * ibase + min(iw, irem);
    const int nit = ibase + (iw < irem ? 1 : 0);            // i-tiles of this wave (<= NIT)
    // a stage buffer, in dwords: DZ part [h][c < NT][256], then X part [h][u < NJT][256]   (h: the stage's first / second tile)
    constexpr int xpart = 512 * NT, bufdw = 512 * (NT + NJT);

    const bool second = slice >= slice2;
    const long long tfirst = second ? tile0b : tile0;
    const int tcount = second ? ntilesb : ntiles;
    const int tbeg = (second ? slice - slice2 : slice) * tiles_per_slice;
    const int tend = min(tcount, tbeg + tiles_per_slice);
    const int nstages = tend > tbeg ? (tend - tbeg + 1) / 2 : 0;
    const uint32_t* Ab = A + (size_t)tfirst * NT * 256;     // 256 dwords per (tile, column tile): hi plane, lo plane
    const uint32_t* Bb = B + (size_t)tfirst * NT * 256;

    f32x4 acc[NIT][NJW];
#pragma unroll
    for (int a = 0; a < NIT; ++a)
#pragma unroll
        for (int b = 0; b < NJW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool ones_here = COLSUM && kb == 0 && jw == 1;    // the spare slot NJW - 1 of the second half

    // LDS-DMA pieces of a stage, 1 KiB each (64 lanes x 16 bytes): 2 * NT of DZ, then 2 * njt of X; piece = wave + 8 k
    const int npa = 2 * NT, npieces = npa + 2 * njt;
    constexpr int NPW = (2 * NT + 2 * NJT + 7) / 8;
    auto issue = [&](int stage, int k0, int k1) {
        const int t0 = tbeg + 2 * stage;
        const bool h1 = t0 + 1 < tend;
        uint32_t* buf = lds_q + (stage & 1) * bufdw;
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            if (k < k0 || k >= k1) continue;
            const int piece = wave + 8 * k;
            if (piece >= npieces) continue;
            const bool isA = piece < npa;
            const int e = isA ? piece : piece - npa;
            const int per = isA ? NT : njt;
            const int h = e >= per ? 1 : 0;
            if (h && !h1) continue;                  // an odd tile count: the last stage has one tile (its operand halves are zeroed)
            const int c = e - h * per;
            const uint32_t* src = (isA ? Ab : Bb) + ((size_t)(t0 + h) * NT + (isA ? c : jt0 + c)) * 256 + lane * 4;
            uint32_t* dst = buf + (isA ? (h * NT + c) * 256 : xpart + (h * NJT + c) * 256);
            // an X piece is read by ONE column block of ONE slice: non-temporal (aux 2), so that the stream does not take L2 / Infinity Cache
            // lines from the chains this GEMM runs beside; a DZ piece is read by all the column blocks of its slice: default policy
            // (c2 2.93-2.96 -> 2.92-2.93 ms, L 40 16.59 -> 16.54; nt on both: L 40 16.87 -- profiles/r05_notes.md section 13)
            if (isA) __builtin_amdgcn_global_load_lds((const void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            else __builtin_amdgcn_global_load_lds((const void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 2);
        }
    };

    // transposing read: lane 4q + p of a 16-lane group addresses row 4g + q, columns 4p .. 4p+3 of the plane.  Byte addresses of this
    // lane's slot in the wave's first DZ tile and first X tile of stage buffer 0.  A wave with fewer tiles than slots reads on into
    // the neighbouring tiles of the buffer (never stored; the launcher allocates a spare KiB behind the last buffer for the last X slot).
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t*)lds_q;
    const uint32_t lane_b = (uint32_t)((4 * g + (i >> 2)) * 32 + 8 * (i & 3));
    const uint32_t a_addr0 = lds0 + lane_b + (uint32_t)it0 * 1024u;
    const uint32_t b_addr0 = lds0 + lane_b + (uint32_t)(xpart * 4) + (uint32_t)ju0 * 1024u;

    if (nstages > 0) issue(0, 0, NPW);
    constexpr int PPT = (NPW + NIT - 1) / NIT;           // pieces of the NEXT stage issued beside each row tile's MFMAs (issue_mode 0)
    for (int st = 0; st < nstages; ++st) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                 // stage st has landed for every wave; buffer (st+1)&1 is free again
        const bool more = st + 1 < nstages;
        const bool h1 = tbeg + 2 * st + 1 < tend;
        const uint32_t a_addr = a_addr0 + (uint32_t)((st & 1) * bufdw * 4), b_addr = b_addr0 + (uint32_t)((st & 1) * bufdw * 4);
        if (more && issue_mode == 1) issue(st + 1, 0, NPW);
        TrOperand rb[NJW], ra[2];
#pragma unroll
        for (int u = 0; u < NJW; ++u) rb[u] = lds_tr_operand(b_addr + (uint32_t)u * 1024u, NJT * 1024u);
        ra[0] = lds_tr_operand(a_addr, NT * 1024u);
        u32x4 bh[NJW], bl[NJW];
#pragma unroll
        for (int u = 0; u < NJW; ++u) { lgkm_wait(rb[u]); tr_finish(rb[u], h1, bh[u], bl[u]); }
        if (ones_here) {                 // bf16 1.0 = 0x3F80 in all eight k of the spare fragment
            bh[NJW - 1] = u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
            bl[NJW - 1] = u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int t = 0; t < NIT; ++t) {
            u32x4 ah, al;
            lgkm_wait(ra[t & 1]);
            tr_finish(ra[t & 1], h1, ah, al);
            if (t + 1 < NIT) ra[(t + 1) & 1] = lds_tr_operand(a_addr + (uint32_t)(t + 1) * 1024u, NT * 1024u);
            if (more && issue_mode == 0) issue(st + 1, PPT * t, PPT * t + PPT);
#pragma unroll
            for (int u = 0; u < NJW; ++u) acc[t][u] = mfma32bf(al, bh[u], acc[t][u]);
#pragma unroll
            for (int u = 0; u < NJW; ++u) acc[t][u] = mfma32bf(ah, bl[u], acc[t][u]);
#pragma unroll
            for (int u = 0; u < NJW; ++u) acc[t][u] = mfma32bf(ah, bh[u], acc[t][u]);
        }
    }
    float* out = slab + (size_t)slice * Mi * Nj;
#pragma unroll
    for (int t = 0; t < NIT; ++t)
        if (t < nit)
#pragma unroll
            for (int u = 0; u < NJW; ++u)
                if (u < njw)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg)
                        out[(size_t)((it0 + t) * 16 + g * 4 + reg) * Nj + (jt0 + ju0 + u) * 16 + i] = acc[t][u][reg];
    if (ones_here && i == 0) {      // every column of the ones tile holds the same sums: lane column 0 writes
#pragma unroll
        for (int t = 0; t < NIT; ++t)
            if (t < nit)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) colsum[(size_t)slice * Mi + (it0 + t) * 16 + g * 4 + reg] = acc[t][NJW - 1][reg];
    }
}

}  // namespace cliora

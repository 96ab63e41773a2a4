// Micro-benchmark: what a grid-wide dependency costs INSIDE one persistent launch on MI355X, in the geometry of the level
// kernels (one 512-thread workgroup per CU holding ~150 KB of LDS), against the dependent-launch boundary of launch_bench.hip.
//
//   flat      one monotonic counter: every workgroup adds once per phase (one lane, after the workgroup's stores have drained),
//             one lane per workgroup polls it with sc1 loads
//   xcd       two-level: a counter per XCD (s_getreg XCC_ID; census in the first phase), the last arriver of an XCD adds to the top
//             counter, polls it and publishes the XCD's generation word, which the XCD's other workgroups poll
//   split     two independent chains, arrive(A) .. work(B) .. wait(A): the barrier latency of one chain under the other's phase
//
// Every phase each workgroup publishes `kb` KB with write-through (sc1) stores and reads the block another workgroup (another XCD:
// id + 3) published in the previous phase with sc1 loads, checking every word (hand-off recipe of cdna_hip_programming.md G16, R1).
// build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier_bench grid_barrier_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((address_space(1))) unsigned gu32;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct Sync {
    unsigned* flat;       // [2] chains, each on its own 256-B line (stride 64 words)
    unsigned* xcd_ctr;    // [2][8] lines
    unsigned* xcd_gen;    // [2][8] lines
    unsigned* top;        // [2] lines
    unsigned* census;     // [8] words
    unsigned* err;        // mismatches, timeouts
};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15u;
}
__device__ __forceinline__ bool spin_until_ge(unsigned* p, unsigned target, unsigned* err) {
    for (unsigned n = 0;; ++n) {
        if (__hip_atomic_load((gu32*)p, RLX) >= target) return true;
        if (n > (1u << 22)) { atomicAdd(err + 1, 1u); return false; }
        __builtin_amdgcn_s_sleep(1);
    }
}
// every wave has drained its stores; one lane signals
__device__ __forceinline__ void arrive_flat(const Sync& s, int chain) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add((gu32*)(s.flat + 64 * chain), 1u, RLX);
}
__device__ __forceinline__ void wait_flat(const Sync& s, int chain, unsigned phase) {
    if (threadIdx.x == 0) spin_until_ge(s.flat + 64 * chain, phase * gridDim.x, s.err);
    __syncthreads();
}
// returns (to thread 0) whether this workgroup was its XCD's last arriver: it then relays the top counter to the XCD in wait_xcd
__device__ __forceinline__ bool arrive_xcd(const Sync& s, int chain, unsigned phase, unsigned xcc, unsigned nx, unsigned nxcd_live) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bool last = false;
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add((gu32*)(s.xcd_ctr + 64 * (8 * chain + xcc)), 1u, RLX);
        last = old + 1 == phase * nx;
        if (last) __hip_atomic_fetch_add((gu32*)(s.top + 64 * chain), 1u, RLX);
    }
    return last;
}
__device__ __forceinline__ void wait_xcd(const Sync& s, int chain, unsigned phase, unsigned xcc, bool last, unsigned nxcd_live) {
    if (threadIdx.x == 0) {
        if (last) {
            spin_until_ge(s.top + 64 * chain, phase * nxcd_live, s.err);
            __hip_atomic_store((gu32*)(s.xcd_gen + 64 * (8 * chain + xcc)), phase, RLX);
        } else {
            spin_until_ge(s.xcd_gen + 64 * (8 * chain + xcc), phase, s.err);
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void busy_us(float us) {        // s_memrealtime ticks at 100 MHz
    if (us <= 0.f) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long dt = (unsigned long long)(us * 100.f);
    while (__builtin_amdgcn_s_memrealtime() - t0 < dt) __builtin_amdgcn_s_sleep(2);
}

// publish this workgroup's block for `tag`, then (after the barrier) check the partner's block
__device__ __forceinline__ void publish(u32x4* buf, int words4, unsigned tag) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(buf + (size_t)blockIdx.x * words4), 0, words4 * 16, 0x27000);
    for (int i = threadIdx.x; i < words4; i += blockDim.x) {
        const unsigned v = tag * 0x9E3779B1u + blockIdx.x * 7919u + i;
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{v, v + 1, v + 2, v + 3}, r, i * 16, 0, 16);   // aux 16 = sc1
    }
}
__device__ __forceinline__ void check(const u32x4* buf, int words4, unsigned tag, unsigned* err) {
    const unsigned src = (blockIdx.x + 3) % gridDim.x;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(buf + (size_t)src * words4), 0, words4 * 16, 0x27000);
    unsigned bad = 0;
    for (int i = threadIdx.x; i < words4; i += blockDim.x) {
        const u32x4 g = __builtin_amdgcn_raw_buffer_load_b128(r, i * 16, 0, 16);
        const unsigned v = tag * 0x9E3779B1u + src * 7919u + i;
        bad += (g.x != v) + (g.y != v + 1) + (g.z != v + 2) + (g.w != v + 3);
    }
    if (bad) atomicAdd(err, bad);
}

// mode 0 flat, 1 xcd, 2 split (flat counters, two chains), 3 split on xcd barriers
__global__ __launch_bounds__(512) void persistent(Sync s, int mode, int nphase, u32x4* buf0, u32x4* buf1, int words4, float work_us) {
    extern __shared__ unsigned lds[];
    if (threadIdx.x == 0) lds[0] = 1;
    const unsigned xcc = xcc_id();
    unsigned nx = 0, nlive = 0;
    // census: workgroups per XCD (flat barrier, once)
    if (threadIdx.x == 0) __hip_atomic_fetch_add((gu32*)(s.census + xcc), 1u, RLX);
    arrive_flat(s, 0);
    wait_flat(s, 0, 1);
    for (int x = 0; x < 8; ++x) { const unsigned c = __hip_atomic_load((gu32*)(s.census + x), RLX); if (x == (int)xcc) nx = c; nlive += c ? 1 : 0; }
    if (mode == 0 || mode == 1) {
        for (int ph = 1; ph <= nphase; ++ph) {
            u32x4* cur = (ph & 1) ? buf0 : buf1;
            publish(cur, words4, ph);
            busy_us(work_us);
            if (mode == 0) { arrive_flat(s, 0); wait_flat(s, 0, ph + 1); }
            else { const bool l = arrive_xcd(s, 0, ph, xcc, nx, nlive); wait_xcd(s, 0, ph, xcc, l, nlive); }
            check(cur, words4, ph, s.err);
        }
    } else {
        // chain A on buf0 / counter 0, chain B on buf1 / counter 1; A's barrier latency hides under B's phase and vice versa
        const bool hx = mode == 3;
        auto A = [&](unsigned tag) { return buf0 + (size_t)(tag & 1) * gridDim.x * words4; };     // double-buffered by tag parity
        auto Bf = [&](unsigned tag) { return buf1 + (size_t)(tag & 1) * gridDim.x * words4; };
        bool la = false, lb = false;
        publish(A(1), words4, 1); busy_us(work_us);
        if (hx) la = arrive_xcd(s, 0, 1, xcc, nx, nlive); else arrive_flat(s, 0);
        for (int ph = 1; ph <= nphase; ++ph) {
            publish(Bf(ph), words4, ph); busy_us(work_us);
            if (hx) lb = arrive_xcd(s, 1, ph, xcc, nx, nlive); else arrive_flat(s, 1);
            if (hx) wait_xcd(s, 0, ph, xcc, la, nlive); else wait_flat(s, 0, ph + 1);
            check(A(ph), words4, ph, s.err);
            publish(A(ph + 1), words4, ph + 1); busy_us(work_us);
            if (hx) la = arrive_xcd(s, 0, ph + 1, xcc, nx, nlive); else arrive_flat(s, 0);
            if (hx) wait_xcd(s, 1, ph, xcc, lb, nlive); else wait_flat(s, 1, ph);
            check(Bf(ph), words4, ph, s.err);
        }
    }
    if (lds[0] == 12345u) s.err[2] = 1;
}

int main(int argc, char** argv) {
    const int nphase = argc > 1 ? atoi(argv[1]) : 200;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device: %s, %d CUs\n", prop.name, ncu);
    unsigned* words; const size_t nwords = 64 * 64;
    CK(hipMalloc(&words, nwords * 4));
    Sync s; s.flat = words; s.xcd_ctr = words + 64 * 2; s.xcd_gen = words + 64 * 18; s.top = words + 64 * 34; s.census = words + 64 * 36; s.err = words + 64 * 37;
    CK(hipFuncSetAttribute((const void*)persistent, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[4] = {"flat", "xcd", "split/flat", "split/xcd"};
    for (int kb : {0, 4, 16, 64}) {
        const int words4 = kb * 1024 / 16;
        u32x4 *b0, *b1;
        CK(hipMalloc(&b0, (size_t)2 * ncu * (words4 + 1) * 16)); CK(hipMalloc(&b1, (size_t)2 * ncu * (words4 + 1) * 16));
        for (float work : {0.f, 5.f, 10.f}) {
            for (int mode = 0; mode < 4; ++mode) {
                float best = 1e9f; unsigned err[4] = {0, 0, 0, 0};
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemsetAsync(words, 0, nwords * 4, 0));
                    CK(hipEventRecord(e0, 0));
                    hipLaunchKernelGGL(persistent, dim3(ncu), dim3(512), 150 * 1024, 0, s, mode, nphase, b0, b1, words4, work);
                    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    best = ms < best ? ms : best;
                    unsigned e[4]; CK(hipMemcpy(e, s.err, 16, hipMemcpyDeviceToHost));
                    for (int i = 0; i < 3; ++i) err[i] += e[i];
                }
                const int syncs = mode >= 2 ? 2 * nphase : nphase;
                printf("%-10s publish %2d KB/WG, phase work %4.1f us: %7.2f us per phase (%d phases)  mismatches %u timeouts %u\n", names[mode], kb, work,
                       best * 1000.f / syncs, syncs, err[0], err[1]);
            }
        }
        CK(hipFree(b0)); CK(hipFree(b1));
    }
    return 0;
}

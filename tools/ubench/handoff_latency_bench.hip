// Micro-benchmark: latency of a DEPENDENT 16-byte load of bytes another workgroup published earlier in the same launch, by
// publish form (plain stores + agent release fence | write-through sc1 stores) and read form (agent acquire + plain loads | sc1
// loads), for a producer on the same XCD (id + 8) and on another one (id + 3); first touch and second touch of the same lines.
// This is the latency the operand ring of the compose phase has to cover inside the persistent kernel.
// build: hipcc --offload-arch=gfx950 -O3 -o handoff_latency_bench handoff_latency_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((address_space(1))) unsigned gu32;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int CHAIN = 32;          // dependent loads per measurement
constexpr int BLOCK_BYTES = 64 * 1024;

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add((gu32*)ctr, 1u, RLX);
        while (__hip_atomic_load((gu32*)ctr, RLX) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

// store_sc1: publish with write-through stores (else plain + release fence); load_sc1: read with sc1 loads (else acquire + plain)
__global__ __launch_bounds__(64) void latency(unsigned* ctr, u32x4* buf, int store_sc1, int load_sc1, int partner_delta, unsigned long long* out, int rounds) {
    const int nw = gridDim.x, wg = blockIdx.x, lane = threadIdx.x;
    const int n16 = BLOCK_BYTES / 16;
    for (int rd = 0; rd < rounds; ++rd) {
        u32x4* mine = buf + (size_t)wg * n16;
        __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)mine, 0, BLOCK_BYTES, 0x27000);
        // element e holds the index of the next element of a stride-67 walk (a different 128-B line every hop), salted by the round
        for (int e = lane; e < n16; e += 64) {
            const unsigned nxt = (unsigned)((e + 67 * 8 + rd) % n16);
            const u32x4 v = u32x4{nxt, nxt ^ 0x5a5a5a5au, (unsigned)rd, (unsigned)wg};
            if (store_sc1) __builtin_amdgcn_raw_buffer_store_b128(v, rm, e * 16, 0, 16);
            else mine[e] = v;
        }
        if (!store_sc1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
        grid_barrier(ctr, (unsigned)(nw * (2 * rd + 1)));
        if (!load_sc1) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        const int src = (wg + partner_delta) % nw;
        const u32x4* theirs = buf + (size_t)src * n16;
        __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)theirs, 0, BLOCK_BYTES, 0x27000);
        for (int pass = 0; pass < 2; ++pass) {
            unsigned idx = (unsigned)(lane * 4) % n16;          // 64 lanes walk 64 chains: one wave-instruction per hop
            unsigned bad = 0;
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            for (int h = 0; h < CHAIN; ++h) {
                u32x4 v;
                if (load_sc1) v = __builtin_amdgcn_raw_buffer_load_b128(rt, idx * 16, 0, 16);
                else v = theirs[idx];
                bad |= (v.y != (v.x ^ 0x5a5a5a5au)) | (v.z != (unsigned)rd) | (v.w != (unsigned)src);
                idx = v.x;
            }
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
            if (lane == 0 && rd == rounds - 1) { out[(size_t)wg * 4 + pass] = t1 - t0; }
            if (bad) atomicAdd((unsigned*)(out + (size_t)nw * 4), 1u);
        }
        grid_barrier(ctr, (unsigned)(nw * (2 * rd + 2)));
    }
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    unsigned* ctr; CK(hipMalloc(&ctr, 256));
    u32x4* buf; CK(hipMalloc(&buf, (size_t)ncu * BLOCK_BYTES));
    unsigned long long* out; CK(hipMalloc(&out, ((size_t)ncu * 4 + 1) * 8));
    unsigned long long* h = (unsigned long long*)malloc(((size_t)ncu * 4 + 1) * 8);
    printf("%d workgroups of one wave, %d dependent 16-B loads per measurement, shader clock cycles per load (median over workgroups)\n", ncu, CHAIN);
    for (int delta : {8, 3})
        for (int ssc : {0, 1})
            for (int lsc : {0, 1}) {
                CK(hipMemset(ctr, 0, 256)); CK(hipMemset(out, 0, ((size_t)ncu * 4 + 1) * 8));
                hipLaunchKernelGGL(latency, dim3(ncu), dim3(64), 0, 0, ctr, buf, ssc, lsc, delta, out, 3);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(h, out, ((size_t)ncu * 4 + 1) * 8, hipMemcpyDeviceToHost));
                double med[2];
                for (int pass = 0; pass < 2; ++pass) {
                    double v[1024]; int n = 0;
                    for (int w = 0; w < ncu; ++w) v[n++] = (double)h[(size_t)w * 4 + pass] / CHAIN;
                    for (int a = 0; a < n; ++a) for (int b = a + 1; b < n; ++b) if (v[b] < v[a]) { double t = v[a]; v[a] = v[b]; v[b] = t; }
                    med[pass] = v[n / 2];
                }
                printf("producer %s, publish %-22s read %-20s first touch %7.0f  second touch %7.0f cycles   mismatching chains %llu\n",
                       delta == 8 ? "same XCD (id+8) " : "other XCD (id+3)", ssc ? "sc1 stores" : "plain + release fence", lsc ? "sc1 loads" : "acquire + plain", med[0],
                       med[1], h[(size_t)ncu * 4]);
            }
    return 0;
}

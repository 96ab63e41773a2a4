// Micro-benchmark: what does a cross-stream dependency per step cost, by mechanism?  Two streams run chains of short kernels; chain B's
// step k may start only when chain A's step k is done (the backward wavefront's shape: api_mlp.hip).  Per step each chain runs NK kernels
// of ~T us.  Reported: us per step of the pair of chains.
//   none      no dependency at all (lower bound: the two chains side by side)
//   event     hipEventRecord behind A's last kernel + hipStreamWaitEvent on B
//   stopev    A's last kernel launched with hipExtLaunchKernelGGL(stop event) + hipStreamWaitEvent on B       (what the library does)
//   value     hipStreamWriteValue32 on A + hipStreamWaitValue32 on B (command-processor memory operations), if the device supports them
//   flag      A's last kernel bumps a device word when its last block retires; B starts the step with a ONE-WAVE gate kernel that spins on it
//   oneq      both chains' kernels interleaved on ONE stream (no concurrency, no sync): the serial reference
// build: hipcc --offload-arch=gfx950 -O3 -o xstream_sync_bench xstream_sync_bench.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void work(long long ticks, float* p) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    if (ticks < 0) p[0] = 1.f;
}
// the last kernel of A's step in the flag variant: the last block to retire publishes `step`
__global__ void work_signal(long long ticks, unsigned* count, unsigned* flag, unsigned step, unsigned nblocks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned prev = atomicAdd(count, 1u);
        if (prev == step * nblocks + nblocks - 1) { __threadfence(); __hip_atomic_store(flag, step + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
    }
}
__global__ void gate(const unsigned* flag, unsigned want) {
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
}
int main(int argc, char** argv) {
    const int NK = 3, STEPS = 200, NB = 64;
    const float T_us = argc > 1 ? atof(argv[1]) : 8.f;
    int clk_khz = 100000; CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0));
    const long long ticks = (long long)(T_us * clk_khz / 1000);
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    float* p; CK(hipMalloc(&p, 1 << 20));
    unsigned* words; CK(hipMalloc(&words, 4096)); CK(hipMemset(words, 0, 4096));
    hipEvent_t ev[STEPS]; for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t t0, t1, fork, join; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1)); CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    int can_wait = 0; (void)hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, 0);
    unsigned* sig = nullptr;
    if (can_wait && hipExtMallocWithFlags((void**)&sig, 4096, hipMallocSignalMemory) != hipSuccess) { sig = nullptr; }
    if (sig) CK(hipMemset(sig, 0, 4096));
    auto run = [&](const char* name, int mode) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipMemset(words, 0, 4096));
            if (sig) CK(hipMemset(sig, 0, 4096));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(t0, sa));
            CK(hipEventRecord(fork, sa)); CK(hipStreamWaitEvent(sb, fork, 0));
            for (int k = 0; k < STEPS; ++k) {
                hipStream_t qa = sa, qb = mode == 5 ? sa : sb;
                for (int i = 0; i < NK - 1; ++i) hipLaunchKernelGGL(work, dim3(NB), dim3(256), 0, qa, ticks, p);
                if (mode == 2) hipExtLaunchKernelGGL(work, dim3(NB), dim3(256), 0, qa, nullptr, ev[k], 0, ticks, p);
                else if (mode == 4) hipLaunchKernelGGL(work_signal, dim3(NB), dim3(256), 0, qa, ticks, words, words + 16, (unsigned)k, (unsigned)NB);
                else hipLaunchKernelGGL(work, dim3(NB), dim3(256), 0, qa, ticks, p);
                if (mode == 1) { CK(hipEventRecord(ev[k], qa)); CK(hipStreamWaitEvent(qb, ev[k], 0)); }
                if (mode == 2) CK(hipStreamWaitEvent(qb, ev[k], 0));
                if (mode == 3) { CK(hipStreamWriteValue32(qa, sig, (uint32_t)(k + 1), 0)); CK(hipStreamWaitValue32(qb, sig, (uint32_t)(k + 1), hipStreamWaitValueGte, 0xffffffffu)); }
                if (mode == 4) hipLaunchKernelGGL(gate, dim3(1), dim3(64), 0, qb, words + 16, (unsigned)(k + 1));
                for (int i = 0; i < NK; ++i) hipLaunchKernelGGL(work, dim3(NB), dim3(256), 0, qb, ticks, p);
            }
            CK(hipEventRecord(join, sb)); CK(hipStreamWaitEvent(sa, join, 0));
            CK(hipEventRecord(t1, sa)); CK(hipEventSynchronize(t1));
            float ms; CK(hipEventElapsedTime(&ms, t0, t1));
            if (rep > 0 && ms < best) best = ms;
        }
        printf("%-8s %7.2f us per step (%d kernels of %.0f us per chain and step; ideal side by side %.1f, serial %.1f)\n", name, best * 1000.f / STEPS, NK, T_us,
               NK * (T_us + 2.5f), 2 * NK * (T_us + 2.5f));
    };
    run("none", 0); run("event", 1); run("stopev", 2);
    if (sig) run("value", 3); else printf("value    not supported here (CanUseStreamWaitValue %d)\n", can_wait);
    run("flag", 4); run("oneq", 5);
    return 0;
}

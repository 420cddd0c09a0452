// Cost of handing over from a kernel on one stream to a kernel on another: (a) hipEventRecord + hipStreamWaitEvent,
// (b) the producer kernel bumps a counter itself and the consumer's stream waits for it (hipStreamWaitValue32 on signal memory),
// (c) both kernels on one stream.  Gap = s_memrealtime at the consumer's first wave minus at the producer's last (100 MHz ticks).
//   hipcc -O2 --offload-arch=gfx950 tools/stream_handover_bench.hip -o /tmp/handover && /tmp/handover
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void producer(unsigned long long *t_end, unsigned *flag, int spin)
{
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) { }
    if (threadIdx.x == 0) {
        atomicMax(t_end, __builtin_amdgcn_s_memrealtime());
        if (flag) { __threadfence(); atomicAdd(flag, 1u); }
    }
}
__global__ void consumer(unsigned long long *t_start)
{
    if (threadIdx.x == 0) atomicMin(t_start, __builtin_amdgcn_s_memrealtime());
}

int main()
{
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipEvent_t evd; CK(hipEventCreateWithFlags(&evd, hipEventDisableTiming | hipEventDisableSystemFence));
    unsigned long long *t; CK(hipMalloc(&t, 16));
    unsigned *flag = nullptr;
    int can = 0; CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    hipError_t fe = hipExtMallocWithFlags((void **)&flag, 8, hipMallocSignalMemory);
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d, signal memory: %s\n", can, hipGetErrorString(fe));
    if (fe != hipSuccess) { flag = nullptr; (void)hipGetLastError(); }
    else CK(hipMemset(flag, 0, 8));
    const int nblk = 128, spin = 4000;       /* 40 us */
    unsigned count = 0;
    hipEvent_t evk; CK(hipEventCreateWithFlags(&evk, hipEventDisableTiming | hipEventDisableSystemFence));
    for (int mode = 0; mode < 5; mode++) {
        if (mode == 1 && !flag) continue;
        std::vector<double> gaps;
        for (int it = 0; it < 60; it++) {
            unsigned long long init[2] = {0, ~0ull};
            CK(hipMemcpy(t, init, 16, hipMemcpyHostToDevice));
            CK(hipDeviceSynchronize());
            if (mode == 0) {
                hipLaunchKernelGGL(producer, dim3(nblk), dim3(256), 0, s1, t, (unsigned *)nullptr, spin);
                CK(hipEventRecord(ev, s1)); CK(hipStreamWaitEvent(s2, ev, 0));
                hipLaunchKernelGGL(consumer, dim3(nblk), dim3(256), 0, s2, t + 1);
            } else if (mode == 1) {
                count += nblk;
                hipLaunchKernelGGL(producer, dim3(nblk), dim3(256), 0, s1, t, flag, spin);
                CK(hipStreamWaitValue32(s2, flag, count, hipStreamWaitValueGte, 0xFFFFFFFFu));
                hipLaunchKernelGGL(consumer, dim3(nblk), dim3(256), 0, s2, t + 1);
            } else if (mode == 2) {
                hipLaunchKernelGGL(producer, dim3(nblk), dim3(256), 0, s1, t, (unsigned *)nullptr, spin);
                hipLaunchKernelGGL(consumer, dim3(nblk), dim3(256), 0, s1, t + 1);
            } else if (mode == 3) {
                hipLaunchKernelGGL(producer, dim3(nblk), dim3(256), 0, s1, t, (unsigned *)nullptr, spin);
                CK(hipEventRecord(evd, s1)); CK(hipStreamWaitEvent(s2, evd, 0));
                hipLaunchKernelGGL(consumer, dim3(nblk), dim3(256), 0, s2, t + 1);
            } else {
                unsigned long long *ta = t; unsigned *fnull = nullptr; int sp = spin;
                void *kargs[] = {&ta, &fnull, &sp};
                CK(hipExtLaunchKernel((const void *)producer, dim3(nblk), dim3(256), kargs, 0, s1, nullptr, evk, 0));
                CK(hipStreamWaitEvent(s2, evk, 0));
                hipLaunchKernelGGL(consumer, dim3(nblk), dim3(256), 0, s2, t + 1);
            }
            CK(hipDeviceSynchronize());
            unsigned long long out[2]; CK(hipMemcpy(out, t, 16, hipMemcpyDeviceToHost));
            gaps.push_back(((double)out[1] - (double)out[0]) / 100.0);
        }
        std::sort(gaps.begin(), gaps.end());
        printf("%-48s: gap median %.1f us (p10 %.1f, p90 %.1f)\n", mode == 0 ? "event record + stream wait event" : mode == 1 ? "kernel bumps a counter + hipStreamWaitValue32" : mode == 2 ? "same stream" : mode == 3 ? "event with hipEventDisableSystemFence" : "the kernel's own stop event (hipExtLaunchKernel)",
               gaps[gaps.size() / 2], gaps[gaps.size() / 10], gaps[gaps.size() * 9 / 10]);
    }
    return 0;
}

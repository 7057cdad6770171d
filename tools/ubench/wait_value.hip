// wait_value.hip -- does hipStreamWaitValue32 gate a stream on a word a running kernel of another stream writes, and how late?
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_signal_then_spin(unsigned *flag, unsigned long long *t, int spin_ticks) {
    if (threadIdx.x == 0) {
        t[0] = __builtin_amdgcn_s_memrealtime();
        __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        while ((long long)(__builtin_amdgcn_s_memrealtime() - t[0]) < spin_ticks) __builtin_amdgcn_s_sleep(8);
        t[1] = __builtin_amdgcn_s_memrealtime();
    }
}
__global__ void k_stamp(unsigned long long *t) {
    if (threadIdx.x == 0) t[2] = __builtin_amdgcn_s_memrealtime();
}
int main() {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    unsigned *flag;
    unsigned long long *t, ht[3];
    for (int kind = 0; kind < 2; kind++) {
        if (kind == 0) CK(hipMalloc(&flag, 64));
        else { hipError_t e = hipExtMallocWithFlags((void **)&flag, 64, hipMallocSignalMemory); if (e != hipSuccess) { printf("signal memory: %s\n", hipGetErrorString(e)); continue; } }
        CK(hipMalloc(&t, 64));
        hipStream_t a, b;
        CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
        CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
        for (int r = 0; r < 4; r++) {
            CK(hipMemset(flag, 0, 64));
            CK(hipMemset(t, 0, 64));
            CK(hipDeviceSynchronize());
            hipError_t e = hipStreamWaitValue32(b, flag, 1, hipStreamWaitValueGte, 0xffffffffu);
            if (e != hipSuccess) { printf("kind %d: hipStreamWaitValue32 -> %s\n", kind, hipGetErrorString(e)); break; }
            hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, b, t);
            // give the waiting stream time to be seen waiting, then start the signaller (which runs for 200 us after signalling)
            hipLaunchKernelGGL(k_signal_then_spin, dim3(1), dim3(64), 0, a, flag, t, 20000);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(ht, t, 24, hipMemcpyDeviceToHost));
            printf("kind %d (%s) run %d: signal -> gated kernel starts after %.1f us; signaller ran %.1f us%s\n", kind, kind ? "signal memory" : "hipMalloc", r,
                   (double)(long long)(ht[2] - ht[0]) / 100.0, (double)(ht[1] - ht[0]) / 100.0, ht[2] < ht[1] ? "  (overlapped: gate opened mid-kernel)" : "  (after it)");
        }
    }
    return 0;
}

// Probe: does hipExtAnyOrderLaunch let kernel B start while kernel A (same stream, launched first) still runs?
// A: 256 workgroups spin for ~40 us and stamp start/end; B: stamps its start.  Prints (B.start - A.start) and
// (A.end - A.start) in us of the 100 MHz constant clock, for a plain launch and for an any-order launch of B.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ void spin_kernel(unsigned long long* stamps, int us) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t0;
        stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
}
__global__ void stamp_kernel(unsigned long long* stamps) {
    if (threadIdx.x == 0) stamps[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
}

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    hipStream_t s;
    CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned long long *a, *b;
    CHK(hipMalloc(&a, 512 * 8));
    CHK(hipMalloc(&b, 256 * 8));
    std::vector<unsigned long long> ha(512), hb(256);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            int us = 40;
            void* args_a[] = {&a, &us};
            void* args_b[] = {&b};
            CHK(hipExtLaunchKernel((const void*)spin_kernel, dim3(256), dim3(256), args_a, 0, s, nullptr, nullptr, 0));
            CHK(hipExtLaunchKernel((const void*)stamp_kernel, dim3(256), dim3(64), args_b, 0, s, nullptr, nullptr, mode ? hipExtAnyOrderLaunch : 0));
            CHK(hipStreamSynchronize(s));
            CHK(hipMemcpy(ha.data(), a, 512 * 8, hipMemcpyDeviceToHost));
            CHK(hipMemcpy(hb.data(), b, 256 * 8, hipMemcpyDeviceToHost));
            unsigned long long a0 = ~0ull, a1 = 0, b0 = ~0ull;
            for (int i = 0; i < 256; ++i) { a0 = std::min(a0, ha[2 * i]); a1 = std::max(a1, ha[2 * i + 1]); b0 = std::min(b0, hb[i]); }
            printf("mode %s rep %d: A ran %.2f us; B's first workgroup started %.2f us after A's first (%.2f us %s A's end)\n",
                   mode ? "any-order" : "plain", rep, (a1 - a0) / 100.0, ((long long)b0 - (long long)a0) / 100.0,
                   ((long long)b0 - (long long)a1) / 100.0, b0 < a1 ? "BEFORE" : "after");
        }
    }
    return 0;
}

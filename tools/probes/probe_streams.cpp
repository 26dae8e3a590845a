// Probe: do two in-process streams run chains of short dependent kernels CONCURRENTLY on MI355X?
// Each kernel: `wgs` workgroups of 1024 threads that sleep-spin for `us` microseconds (occupies wave slots, no memory
// traffic).  Chain = n back-to-back launches on one stream.  Reports wall time of: one chain alone; two chains on two
// streams launched from one host thread (interleaved); two chains from two host threads.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <thread>

__global__ void spin_kernel(int us) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(4);
}
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    hipStream_t s[2];
    for (auto& x : s) CHK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    const int n = 400;
    for (int wgs : {128, 256}) for (int us : {4, 10, 20}) {
        auto chain = [&](hipStream_t st) { for (int i = 0; i < n; ++i) spin_kernel<<<wgs, 1024, 0, st>>>(us); };
        chain(s[0]); CHK(hipStreamSynchronize(s[0]));
        double t0 = now(); chain(s[0]); double th = now() - t0; CHK(hipStreamSynchronize(s[0])); const double one = now() - t0;
        t0 = now();
        for (int i = 0; i < n; ++i) { spin_kernel<<<wgs, 1024, 0, s[0]>>>(us); spin_kernel<<<wgs, 1024, 0, s[1]>>>(us); }
        CHK(hipStreamSynchronize(s[0])); CHK(hipStreamSynchronize(s[1]));
        const double two = now() - t0;
        t0 = now();
        std::thread a([&] { CHK(hipSetDevice(0)); chain(s[0]); CHK(hipStreamSynchronize(s[0])); });
        std::thread b([&] { CHK(hipSetDevice(0)); chain(s[1]); CHK(hipStreamSynchronize(s[1])); });
        a.join(); b.join();
        const double two_t = now() - t0;
        printf("wgs %3d x 1024 thr, %2d us kernels, %d per chain: one chain %.2f ms (host enqueue %.2f ms = %.2f us/launch); two chains 1 thread %.2f ms; two chains 2 threads %.2f ms\n",
               wgs, us, n, one * 1e3, th * 1e3, th * 1e6 / n, two * 1e3, two_t * 1e3);
    }
    return 0;
}

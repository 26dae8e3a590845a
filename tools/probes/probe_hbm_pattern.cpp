// Probe: HBM read rate of 2048 waves (512 workgroups x 4 waves, like the decode attention grid) that each keep two
// 16 KiB sets of 1 KiB wave-loads in flight, as a function of how the 16 KiB of a set are laid out in memory:
//   mode 0  two separate 8 KiB runs at random places (K run + V run of a paged KV tile)
//   mode 1  one 16 KiB run at a random place
//   mode 2  each wave streams its own contiguous region (chunk after chunk)
//   mode 3  all waves stream ONE region grid-strided (chunk i of the buffer goes to wave i % nwaves)
// Buffer 4 GiB (far beyond the 256 MiB Infinity Cache); every byte is read at most once per launch.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256, 2) reader(const uint4* __restrict__ buf, const unsigned* __restrict__ offs, int sets_per_wave, int mode,
                                                 size_t total_sets, unsigned* sink) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
    uint4 a[16], b[16];
    unsigned acc = 0;
    auto base_of = [&](int s, int half) -> size_t {  // uint4 index of the first 1 KiB of half `half` (8 KiB) of set s
        if (mode == 0) return (size_t)offs[(wave * sets_per_wave + s) * 2 + half] * 512;         // 8 KiB granules
        if (mode == 1) return (size_t)offs[(wave * sets_per_wave + s) * 2] * 512 + (size_t)half * 512;
        if (mode == 2) return ((wave * sets_per_wave + s) * 2 + half) * 512;
        return (((size_t)s * nwaves + wave) * 2 + half) * 512;
    };
    auto load = [&](int s, uint4 (&r)[16]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint4* p = buf + base_of(s, h) + lane;
#pragma unroll
            for (int i = 0; i < 8; ++i) r[h * 8 + i] = p[i * 64];
        }
    };
    auto use = [&](uint4 (&r)[16]) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc ^= r[i].x ^ r[i].w;
    };
    load(0, a);
    if (sets_per_wave > 1) load(1, b);
    for (int s = 0; s < sets_per_wave; s += 2) {
        use(a);
        if (s + 2 < sets_per_wave) load(s + 2, a);
        if (s + 1 >= sets_per_wave) break;
        use(b);
        if (s + 3 < sets_per_wave) load(s + 3, b);
    }
    if (acc == 0x12345678u) *sink = acc;
}

// reference: plain grid-stride streaming read, 16 B per lane, `unroll` loads in flight per thread
template <int UNROLL>
__global__ void __launch_bounds__(256) stream_read(const uint4* __restrict__ buf, size_t n16, unsigned* sink) {
    unsigned acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = buf[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main() {
    const size_t bytes = (size_t)4 << 30;
    uint4* buf; unsigned *offs, *sink;
    CHK(hipMalloc(&buf, bytes));
    CHK(hipMemset(buf, 1, bytes));
    CHK(hipMalloc(&sink, 4));
    {
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        for (int blocks : {512, 2048, 8192}) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                const size_t n16 = ((size_t)400 << 20) / 16;
                const uint4* win = buf + (size_t)rep * ((size_t)448 << 20) / 16;
                CHK(hipEventRecord(e0));
                stream_read<8><<<blocks, 256>>>(win, n16, sink);
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("plain streaming read of 419 MB, %d blocks x 256 threads, 8 loads in flight per thread: %.2f us  %.2f TB/s\n", blocks, best * 1e3, 419.4 / (best * 1e3));
        }
    }
    const int wgs = 512, nwaves = wgs * 4;
    for (int sets_per_wave : {3, 6, 12}) {
        const size_t total_sets = (size_t)nwaves * sets_per_wave;
        std::vector<unsigned> h(total_sets * 2);
        const size_t granules = bytes / 8192;  // 8 KiB granules
        // random distinct granules (a random permutation prefix): pairs for mode 0, even-aligned starts for mode 1
        std::vector<unsigned> perm(granules / 2);
        for (size_t i = 0; i < perm.size(); ++i) perm[i] = (unsigned)i;
        srand(1);
        for (size_t i = 0; i < total_sets * 2 && i < perm.size(); ++i) { size_t j = i + rand() % (perm.size() - i); std::swap(perm[i], perm[j]); }
        for (size_t i = 0; i < total_sets * 2; ++i) h[i] = perm[i] * 2;  // 16 KiB-aligned granule pairs: mode 0 uses h[2s], h[2s+1] (two places), mode 1 h[2s] and the next granule
        CHK(hipMalloc(&offs, h.size() * 4));
        CHK(hipMemcpy(offs, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        for (int mode = 0; mode < 4; ++mode) {
            hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                // cold data without dirty lines: modes 2/3 read a fresh window of the buffer every repetition; the random
                // modes read a random 1/8..1/2 of the whole buffer (re-touching a line across reps is rare)
                const uint4* win = (mode >= 2) ? buf + (size_t)rep * ((size_t)448 << 20) / 16 : buf;
                CHK(hipEventRecord(e0));
                reader<<<wgs, 256>>>(win, offs, sets_per_wave, mode, total_sets, sink);
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double mb = (double)total_sets * 16384 / 1e6;
            printf("sets/wave %2d (%.0f MB) mode %d: %.2f us  %.2f TB/s\n", sets_per_wave, mb, mode, best * 1e3, mb / (best * 1e3));
        }
        CHK(hipFree(offs));
    }
    return 0;
}

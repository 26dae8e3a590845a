// One-shot all-reduce for the small messages of tensor-parallel decode ([rows <= 128][hidden] f32, <= 2.6 MB; the sum
// the reference never performs: RowParallelLinear::forward, src/layers/linear.rs:184-198).
// xGMI is a point-to-point mesh: instead of a ring (tp - 1 dependent hops of a latency-bound message) every rank WRITES
// its partial straight into a slot of every peer's buffer (all links at once), raises one flag per peer, and the
// consumer -- the add-RMSNorm that follows the projection anyway -- sums the tp slots as it already sums split-K slabs.
//   push:  slot[gen][my rank] of every peer <- my partial;  system-scope fence;  flag[gen][my rank] of every peer <- seq
//   wait:  one wave polls the tp flags of this rank (bounded), then the norm kernel reads slot[gen][0..tp)
// Two generations alternate: a rank can only start call c+2 after it consumed call c+1, which needed every peer's
// flag of c+1, which the peer raised after it finished reading call c -- so nobody overwrites a slot still being read.
// Buffers are uncached / fine-grained device memory, mapped into the peers by HIP IPC (RCCL ranks) or shared directly
// (the in-process loopback test double).  Opt-in: only the loopback form has run; the IPC form needs a multi-GPU box.
#include <algorithm>

#include "device_common.h"
#include "kernels.h"

namespace nvllm {

__global__ void __launch_bounds__(256) oneshot_push_kernel(const float4* __restrict__ src, size_t n4, OneShotPeers p, int tp, int rank,
                                                           size_t slot_floats, int gen, uint32_t seq, unsigned* done) {
    const size_t slot4 = ((size_t)(gen * tp + rank) * slot_floats) >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = src[i];
        for (int q = 0; q < tp; ++q) reinterpret_cast<float4*>(p.data[q])[slot4 + i] = v;
    }
    __threadfence_system();  // this workgroup's stores are visible system-wide before it reports
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == gridDim.x - 1) {  // last workgroup of this launch: every slot write above has been fenced
            __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the next push is stream-ordered after us
            __threadfence_system();
            for (int q = 0; q < tp; ++q)
                __hip_atomic_store(p.flag[q] + gen * tp + rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// one wave: lane r waits for rank r's flag.  Bounded: a peer that never arrives leaves *err = 1 (the step's results are
// then wrong and the host reports it) instead of a kernel that never ends.
__global__ void __launch_bounds__(64) oneshot_wait_kernel(const uint32_t* flags, int tp, int gen, uint32_t seq, int* err, long long max_spins) {
    if (*reinterpret_cast<volatile int*>(err)) return;  // an earlier wait of this step already gave up: do not spin again
    if ((int)threadIdx.x < tp) {
        const uint32_t* f = flags + gen * tp + threadIdx.x;
        long long spins = 0;
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            if (++spins > max_spins) { atomicExch(err, 1); break; }
            __builtin_amdgcn_s_sleep(4);
        }
    }
}

hipError_t launch_oneshot_push(const float* src, size_t n, const OneShotPeers& p, int tp, int rank, size_t slot_floats, int gen,
                               uint32_t seq, unsigned* done, hipStream_t s) {
    if (n % 4 != 0 || n > slot_floats || tp < 2 || tp > 8) return hipErrorInvalidValue;
    const size_t n4 = n / 4;
    const int grid = (int)std::min<size_t>(64, (n4 + 255) / 256);
    oneshot_push_kernel<<<grid, 256, 0, s>>>(reinterpret_cast<const float4*>(src), n4, p, tp, rank, slot_floats, gen, seq, done);
    return hipGetLastError();
}

hipError_t launch_oneshot_wait(const uint32_t* flags, int tp, int gen, uint32_t seq, int* err, long long max_spins, hipStream_t s) {
    oneshot_wait_kernel<<<1, 64, 0, s>>>(flags, tp, gen, seq, err, max_spins);
    return hipGetLastError();
}

}  // namespace nvllm

// Device-side helpers shared by the .hip translation units of libnvllm_amd.so (kernels.hip, stream_gemm.hip):
// MFMA fragment types, bf16 hi/lo split, deferred-RMSNorm helpers, the LDS-DMA publish barrier, diagnostic stamps.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <type_traits>

#include "kernels.h"

namespace nvllm {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ---------------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint16_t bf16_bits_of(float x) { return __builtin_bit_cast(uint16_t, (__bf16)x); }
__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __builtin_bit_cast(float, (uint32_t)b << 16); }
__device__ __forceinline__ void split_bf16(float x, uint16_t& hi, uint16_t& lo) {
    const __bf16 h = (__bf16)x;
    hi = __builtin_bit_cast(uint16_t, h);
    lo = __builtin_bit_cast(uint16_t, (__bf16)(x - (float)h));
}
__device__ __forceinline__ _Float16 f16_sat(float x) { return (_Float16)fminf(fmaxf(x, -65504.f), 65504.f); }

// 24-bit V (opt-in): beside the f16 value the cache keeps the rounding residual r = x - f16(x) as the TOP BYTE
// of f16(r) -- a bf8 / e5m2 number (sign, 5 exponent bits, 2 mantissa bits), rounded to nearest even.  |r| <= ulp/2, so the
// pair carries 13..14 significant bits; the attention kernels feed the bytes to a bf8 MFMA as they are.

// 24-bit V (opt-in): beside the f16 value the cache keeps the rounding residual r = x - f16(x) as the TOP BYTE
// of f16(r) -- an e5m2 number (sign, 5 exponent bits, 2 mantissa bits), rounded to nearest even.  |r| <= ulp/2, so the pair
// carries 13..14 significant bits, and the reader rebuilds r's f16 by a byte shift (no arithmetic).
__device__ __forceinline__ void store_v24(_Float16* v, uint8_t* vlo, int t_in_block, int d, int hd, float x);

// max of three without the canonicalising v_max_f32 x, x that fmaxf() adds per operand (operands here are MFMA results
// and finite constants: no signalling NaN to quiet)
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// SiluAndMul (src/layers/activation.rs:13-18) in the GEMM epilogues: x * sigmoid(x) * up with v_rcp_f32 (1 ulp) instead
// of an IEEE division (a dozen instructions per element: 1.5 us of a 256 x 192 tile's epilogue).  exp overflow gives
// rcp(inf) = 0, i.e. -0 for very negative gates; no NaN.  The fine-seam op keeps the division.
__device__ __forceinline__ float silu_mul(float g, float u) { return (g * __builtin_amdgcn_rcpf(1.0f + __expf(-g))) * u; }

// 16 bytes of a stream that THIS CU reads once per launch (a sequence's K/V in decode): non-temporal, so the lines do not
// displace what other kernels re-read from L2 / Infinity Cache (`global_load_dwordx4 ... nt`; guide, nt-weights row)
__device__ __forceinline__ uint4 ld_stream16(const void* p) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// rinv[row] of a deferred RMSNorm (see RowNorm); 1 when the activations were normalised by the producer
__device__ __forceinline__ float rownorm_rinv(const RowNorm& rn, int row) {
    if (!rn.ssq) return 1.0f;
    float t = 0.f;
    for (int g = 0; g < rn.groups; ++g) t += rn.ssq[(size_t)g * rn.stride + row];
    return 1.0f / sqrtf(t * rn.inv_h + rn.eps);
}

// Workgroup-cooperative form for kernels whose epilogue needs rinv of up to NR rows: every thread sums a
// strided quarter of the groups of one row with independent (unrollable) loads and parks it in LDS;
// after any later barrier rownorm_rinv_lds() finishes the sum.  groups <= 64.
template <int NR>
__device__ __forceinline__ void rownorm_partials(const RowNorm& rn, int m0, int M, float* lds_part) {
    if (!rn.ssq) return;
    // every thread runs the loads (index clamped, only the LDS store is predicated): loads under a divergent branch
    // are serialised by the compiler, one vmcnt(0) round trip each
    for (int i0 = 0; i0 < 4 * NR; i0 += blockDim.x) {
        const int idx = i0 + (int)threadIdx.x;
        const int ic = min(idx, 4 * NR - 1);
        const int r = ic % NR, part = ic / NR;
        int row = min(m0 + r, M - 1);
        if (rn.row_idx) row = rn.row_idx[row];
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = rn.ssq[(size_t)min(part + 4 * i, rn.groups - 1) * rn.stride + row];
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += (part + 4 * i) < rn.groups ? v[i] : 0.f;
        if (idx < 4 * NR) lds_part[part * NR + r] = t;
    }
}
template <int NR>
__device__ __forceinline__ float rownorm_rinv_lds(const RowNorm& rn, const float* lds_part, int r) {
    if (!rn.ssq) return 1.0f;
    const float t = lds_part[r] + lds_part[NR + r] + lds_part[2 * NR + r] + lds_part[3 * NR + r];
    return 1.0f / sqrtf(t * rn.inv_h + rn.eps);
}
// one row, one wave: lane g loads group g (groups <= 64)
__device__ __forceinline__ float rownorm_rinv_wave(const RowNorm& rn, int row, int lane) {
    if (!rn.ssq) return 1.0f;
    float t = lane < rn.groups ? rn.ssq[(size_t)lane * rn.stride + row] : 0.f;
    t = wave_sum(t);
    return 1.0f / sqrtf(t * rn.inv_h + rn.eps);
}

// hipFuncSetAttribute is per device: remember on which devices a kernel's dynamic-LDS limit has been raised
// (one bit per device ordinal; idempotent, so a race between two host threads only repeats the call)
static inline void ensure_dyn_lds(const void* fn, size_t lds, std::atomic<uint64_t>& done) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        done.fetch_or(bit, std::memory_order_release);
    }
}

// KV-cache element offsets inside one (block, kv head) slab of kBlockTokens x hd f16:
// V in PV A-fragment order: token tt (0..31) of a 32-token tile sits in k-slot (g, j): tt<16: g=tt>>2, j=tt&3;
//   else g=(tt-16)>>2, j=4+(tt&3); element (tt, d) -> (((tile*(hd/16) + d/16)*64 + g*16 + d%16)*8 + j
// K in QK^T A-fragment order: 16-token tiles, each hd/32 chunks of 1 KiB: lane (g*16 + r) holds token r, dims 32c+8g..+7
__device__ __forceinline__ size_t k_packed_offset(int t_in_block, int d, int hd) {
    const int tt = t_in_block >> 4, r = t_in_block & 15;
    const int c = d >> 5, g = (d & 31) >> 3, j = d & 7;
    return ((size_t)(tt * (hd >> 5) + c) * 64 + (g << 4) + r) * 8 + j;
}

__device__ __forceinline__ size_t v_packed_offset(int t_in_block, int d, int hd) {
    const int tile = t_in_block >> 5, tt = t_in_block & 31;
    const int g = (tt & 15) >> 2, j = ((tt >> 4) << 2) | (tt & 3);
    return ((size_t)(tile * (hd >> 4) + (d >> 4)) * 64 + (g << 4) + (d & 15)) * 8 + j;
}

// 24-bit V: the residual bytes of TWO neighbouring PV fragments (features 32p..32p+15 and 32p+16..32p+31) share one 16-byte
// lane slot, [8 bytes of the even fragment][8 of the odd one], so the attention kernels fetch them with full 1 KiB wave-loads
// (512-byte loads cost the memory pipe as much per instruction as 1 KiB ones: measured +48 % kernel time for +25 % bytes).
// Byte offset inside one (block, kv head) slab of kBlockTokens x hd bytes:
__device__ __forceinline__ size_t vlo_packed_offset(int t_in_block, int d, int hd) {
    const int tile = t_in_block >> 5, tt = t_in_block & 31;
    const int g = (tt & 15) >> 2, j = ((tt >> 4) << 2) | (tt & 3);
    return ((size_t)(tile * (hd >> 5) + (d >> 5)) * 64 + (g << 4) + (d & 15)) * 16 + (size_t)(((d >> 4) & 1) << 3) + j;
}
__device__ __forceinline__ void store_v24(_Float16* v, uint8_t* vlo, int t_in_block, int d, int hd, float x) {
    const _Float16 h = f16_sat(x);
    v[v_packed_offset(t_in_block, d, hd)] = h;
    if (vlo) {
        const float xc = fminf(fmaxf(x, -65504.f), 65504.f);
        const uint32_t b = __builtin_bit_cast(uint16_t, (_Float16)(xc - (float)h));
        vlo[vlo_packed_offset(t_in_block, d, hd)] = (uint8_t)((b + 0x7Fu + ((b >> 8) & 1u)) >> 8);
    }
}

// 24-bit K (KvLayout::klo): the same residual byte per K element, two QK^T fragments (head-dim chunks c, c+1 of one 16-token
// tile) per 16-byte lane slot.  Byte offset inside one (block, kv head) slab of kBlockTokens x hd bytes:
__device__ __forceinline__ size_t klo_packed_offset(int t_in_block, int d, int hd) {
    const int tt = t_in_block >> 4, r = t_in_block & 15;
    const int c = d >> 5, g = (d & 31) >> 3, j = d & 7;
    return ((size_t)(tt * (hd >> 6) + (c >> 1)) * 64 + (g << 4) + r) * 16 + (size_t)((c & 1) << 3) + j;
}
__device__ __forceinline__ void store_k24(_Float16* k, uint8_t* klo, int t_in_block, int d, int hd, float x) {
    const _Float16 h = f16_sat(x);
    k[k_packed_offset(t_in_block, d, hd)] = h;
    if (klo) {
        const float xc = fminf(fmaxf(x, -65504.f), 65504.f);
        const uint32_t b = __builtin_bit_cast(uint16_t, (_Float16)(xc - (float)h));
        klo[klo_packed_offset(t_in_block, d, hd)] = (uint8_t)((b + 0x7Fu + ((b >> 8) & 1u)) >> 8);
    }
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Diagnostic build (make stamps, -DNVLLM_STAMPS): in-kernel time stamps of the 100 MHz constant clock, one slot per
// (workgroup, wave, point); tools/stamp_timeline.py reads them.  Compiled out of the product library.
#ifdef NVLLM_STAMPS
#define NVLLM_STAMP(args_, idx_)                                                                                          \
    do {                                                                                                                  \
        if ((args_).stamps && (threadIdx.x & 63) == 0)                                                                    \
            (args_).stamps[(((size_t)blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z)) * 16 + (threadIdx.x >> 6)) * 8 + (idx_)] = \
                __builtin_amdgcn_s_memrealtime();                                                                         \
    } while (0)
#else
#define NVLLM_STAMP(args_, idx_) do { } while (0)
#endif

// Barrier that PUBLISHES LDS-DMA data (global_load_lds): every wave first waits for its own DMA to land, then joins the
// barrier; only then may any wave ds_read fragments another wave staged.  __syncthreads() alone is not enough: hipcc
// (ROCm 7.2) emitted the loop-header barrier of gemm_kernel's chunk loop as `s_waitcnt lgkmcnt(0); s_barrier` with
// the vmcnt(0) AFTER the barrier -- each wave then waits for its own DMA only, and reads of another wave's still-in-
// flight fragments returned old LDS bytes (no stall, no fault): wrong sums under memory load (several contexts on one
// GPU), clean on a quiet chip.  The asm wait is invisible to the compiler's waitcnt pass and cannot be moved.
__device__ __forceinline__ void dma_publish_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

}  // namespace nvllm

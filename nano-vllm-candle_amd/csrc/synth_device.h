// Device-side deterministic synthetic-checkpoint generator.
//
// Independent implementation of the recipe the oracle documents (tests/test_ops_gpu.py::test_device_generator_matches_oracle_generator checks the two
// for bit-equality through the C ABI):
//   name_hash = FNV-1a-64(name) ^ (seed * 0x9E3779B97F4A7C15)
//   h         = splitmix64 finalizer of (name_hash + (idx + 1) * 0x9E3779B97F4A7C15)
//   matrix    : k = (h >> 56) - 128, value = k * 2^-12     (bf16-exact)
//   norm      : j = ((h >> 40) % 33) - 16, value = 1 + j * 2^-7   (bf16-exact)
// idx is the element's row-major index in the FULL (unsharded) HF tensor, so every TP rank
// generates exactly its shard of the same logical checkpoint.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nvllm {

constexpr int kSynthMatrix = 0;
constexpr int kSynthNorm = 1;

__host__ __device__ inline uint64_t synth_finalize(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

inline uint64_t synth_hash_name(const char* name, uint64_t seed) {
    uint64_t h = 0xCBF29CE484222325ULL;
    for (; *name; ++name) h = (h ^ (uint64_t)(unsigned char)*name) * 0x100000001B3ULL;
    return h ^ (seed * 0x9E3779B97F4A7C15ULL);
}

// bf16 bit pattern of element idx
__host__ __device__ inline uint16_t synth_bits(uint64_t name_hash, uint64_t idx, int kind) {
    const uint64_t h = synth_finalize(name_hash + (idx + 1) * 0x9E3779B97F4A7C15ULL);
    float v;
    if (kind == kSynthNorm) {
        v = 1.0f + (float)((int)((h >> 40) % 33) - 16) * 0.0078125f;
    } else {
        v = (float)((int)(h >> 56) - 128) * 0.000244140625f;
    }
    return (uint16_t)(__builtin_bit_cast(uint32_t, v) >> 16);  // exact: v is bf16-representable
}

}  // namespace nvllm

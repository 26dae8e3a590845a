/*
 * Deterministic synthetic-checkpoint generator (shared spec).
 *
 * TEST INFRASTRUCTURE (oracle side).  The same integer recipe is implemented twice, independently,
 * and tested for bit-equality (tests/test_ops_gpu.py::test_device_generator_matches_oracle_generator): here (plain C, host, feeds the oracle) and in
 * nano-vllm-candle_amd/csrc/synth_device.h (HIP, fills product weights straight into HBM).
 * The product never includes this file.
 *
 * Why: neither this container nor the GPU box holds Qwen3 weights (reference loads
 * `model.safetensors` at src/models/qwen3.rs:517-521; no network here), so every machine must be
 * able to regenerate byte-identical HF-named tensors from (seed, tensor name, element index).
 *
 * Recipe (all integer, no libm):
 *   name_hash = FNV-1a-64(name) ^ (seed * 0x9E3779B97F4A7C15)
 *   h         = splitmix64_finalizer(name_hash + (idx + 1) * 0x9E3779B97F4A7C15)
 *   matrix / embedding element : k = (h >> 56) - 128        in [-128, 127];  value = k * 2^-12
 *   norm weight element        : j = ((h >> 40) % 33) - 16  in [-16, 16];    value = 1 + j * 2^-7
 * Every value is exactly representable in bf16, so "bf16 checkpoint up-cast to f32"
 * (reference: DType::F32 load at src/models/qwen3.rs:519) is lossless, as it is for real Qwen3 files.
 */
#ifndef NVLLM_SYNTH_H
#define NVLLM_SYNTH_H
#include <stdint.h>

#define SYNTH_FN static inline

#define SYNTH_KIND_MATRIX 0
#define SYNTH_KIND_NORM 1

SYNTH_FN uint64_t synth_mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}

SYNTH_FN uint64_t synth_name_hash(const char* name, uint64_t seed) {
    uint64_t h = 0xCBF29CE484222325ULL;
    for (const char* p = name; *p; ++p) { h ^= (uint64_t)(unsigned char)*p; h *= 0x100000001B3ULL; }
    return h ^ (seed * 0x9E3779B97F4A7C15ULL);
}

/* value of element idx of the tensor whose name hashed to name_hash */
SYNTH_FN float synth_value(uint64_t name_hash, uint64_t idx, int kind) {
    uint64_t h = synth_mix64(name_hash + (idx + 1) * 0x9E3779B97F4A7C15ULL);
    if (kind == SYNTH_KIND_NORM) {
        int j = (int)((h >> 40) % 33) - 16;
        return 1.0f + (float)j * 0.0078125f; /* 2^-7 */
    }
    int k = (int)(h >> 56) - 128;
    return (float)k * 0.000244140625f; /* 2^-12 */
}

/* exact bf16 bit pattern of a synth value (value is bf16-representable: plain truncation) */
SYNTH_FN uint16_t synth_bf16_bits(uint64_t name_hash, uint64_t idx, int kind) {
    union { float f; uint32_t u; } c;
    c.f = synth_value(name_hash, idx, kind);
    return (uint16_t)(c.u >> 16);
}

#endif

// Device-side deterministic synthetic-checkpoint generator.
//
// Independent implementation of the recipe the oracle documents in oracle/synth.h (tests check the two for bit-equality
// through the C ABI: test_device_generator_matches_oracle_generator, test_heavy_profile_loaded_equals_generated):
//   name_hash = FNV-1a-64(name) ^ (seed * 0x9E3779B97F4A7C15)
//   h         = splitmix64 finalizer of (name_hash + (idx + 1) * 0x9E3779B97F4A7C15)
// profile 0 (benign):
//   matrix    : k = (h >> 56) - 128, value = k * 2^-12     (bf16-exact)
//   norm      : j = ((h >> 40) % 33) - 16, value = 1 + j * 2^-7   (bf16-exact)
// profile 1 (heavy: full mantissas, log-spread magnitudes, wide norm weights, outlier hidden channels), as bf16 bits:
//   mant = (h >> 48) & 127, z = trailing zeros of ((h >> 40) & 255 | 256)
//   matrix    : sign h>>63, exponent z/2 - 7 (+6 on an outlier channel of embed_tokens columns / o_proj, down_proj rows)
//   norm      : k = trailing zeros of ((h >> 38) & 1023 | 1024) / 2, d = (h >> 36) & 3, k2 = trailing zeros of ((h >> 33) & 7 | 8);
//               layernorms: exponent d ? min(k - 1, 4) : -2 - k2, -5 on an outlier channel; q/k-norm: d ? k - 2 : -3 - min(k2, 1)
//   outlier channels: finalizer(seed * 0xD1B54A32D192ED03 + j + 1) mod hidden_size, j = 0..3
// idx is the element's row-major index in the FULL (unsharded) HF tensor, so every TP rank
// generates exactly its shard of the same logical checkpoint.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nvllm {

constexpr int kSynthMatrix = 0;
constexpr int kSynthNorm = 1;
constexpr int kSynthQkNorm = 2;
constexpr int kSynthAxisNone = 0, kSynthAxisCol = 1, kSynthAxisRow = 2;

struct SynthSpec {
    uint64_t name_hash = 0;
    int kind = kSynthMatrix;
    int profile = 0;
    int axis = kSynthAxisNone;  // where the hidden channel of an element sits: col = idx % cols, row = idx / cols
    int64_t cols = 1;
    uint32_t ch[4] = {0, 0, 0, 0};
};

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

inline void synth_set_outliers(SynthSpec& s, uint64_t seed, int hidden_size) {
    for (int j = 0; j < 4; ++j) s.ch[j] = (uint32_t)(synth_finalize(seed * 0xD1B54A32D192ED03ULL + (uint64_t)(j + 1)) % (uint64_t)hidden_size);
}

// bf16 bit pattern of element idx whose hidden channel (when the tensor has one) is `chan`
__host__ __device__ inline uint16_t synth_bits_chan(const SynthSpec& s, uint64_t idx, uint32_t chan) {
    const uint64_t h = synth_finalize(s.name_hash + (idx + 1) * 0x9E3779B97F4A7C15ULL);
    if (s.profile == 0) {
        const float v = s.kind != kSynthMatrix ? 1.0f + (float)((int)((h >> 40) % 33) - 16) * 0.0078125f
                                               : (float)((int)(h >> 56) - 128) * 0.000244140625f;
        return (uint16_t)(__builtin_bit_cast(uint32_t, v) >> 16);  // exact: v is bf16-representable
    }
    const bool outlier = s.axis != kSynthAxisNone && (chan == s.ch[0] || chan == s.ch[1] || chan == s.ch[2] || chan == s.ch[3]);
    const uint32_t mant = (uint32_t)(h >> 48) & 0x7Fu;
    const int z = __builtin_ctz(((uint32_t)(h >> 40) & 0xFFu) | 0x100u);
    if (s.kind == kSynthMatrix)
        return (uint16_t)(((uint32_t)(h >> 63) << 15) | ((uint32_t)(120 + (z >> 1) + (outlier ? 6 : 0)) << 7) | mant);
    const int k = __builtin_ctz(((uint32_t)(h >> 38) & 0x3FFu) | 0x400u) >> 1;  // 0..5, P(k) ~ 4^-k
    const int k2 = __builtin_ctz(((uint32_t)(h >> 33) & 7u) | 8u);             // 0..3
    const bool up = ((h >> 36) & 3) != 0;
    int e;
    if (s.kind == kSynthQkNorm) e = up ? k - 2 : -3 - (k2 < 1 ? k2 : 1);
    else e = up ? (k - 1 > 4 ? 4 : k - 1) : -2 - k2;
    if (outlier) e = -5;
    return (uint16_t)(((uint32_t)(127 + e) << 7) | mant);
}

__host__ __device__ inline uint16_t synth_bits(const SynthSpec& s, uint64_t idx) {
    uint32_t chan = 0;
    if (s.profile != 0 && s.axis != kSynthAxisNone)
        chan = (uint32_t)(s.axis == kSynthAxisCol ? idx % (uint64_t)s.cols : idx / (uint64_t)s.cols);
    return synth_bits_chan(s, idx, chan);
}

// profile-0 spec from a bare hash (tuning aids fill operands with it)
inline SynthSpec synth_plain(uint64_t name_hash, int kind) {
    SynthSpec s;
    s.name_hash = name_hash; s.kind = kind;
    return s;
}

}  // namespace nvllm

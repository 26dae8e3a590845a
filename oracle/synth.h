/*
 * Deterministic synthetic-checkpoint generator (shared spec).
 *
 * TEST INFRASTRUCTURE (oracle side).  The same integer recipe is implemented twice, independently,
 * and tested for bit-equality (tests/test_ops_gpu.py::test_device_generator_matches_oracle_generator and
 * tests/test_stress_gpu.py::test_heavy_profile_loaded_equals_generated): here (plain C, host, feeds the
 * oracle) and in nano-vllm-candle_amd/csrc/synth_device.h (HIP, fills product weights straight into HBM).
 * The product never includes this file.
 *
 * Why: neither this container nor the GPU box holds Qwen3 weights (reference loads
 * `model.safetensors` at src/models/qwen3.rs:517-521; no network here), so every machine must be
 * able to regenerate byte-identical HF-named tensors from (seed, tensor name, element index).
 *
 * Recipe (all integer, no libm):
 *   name_hash = FNV-1a-64(name) ^ (seed * 0x9E3779B97F4A7C15)
 *   h         = splitmix64_finalizer(name_hash + (idx + 1) * 0x9E3779B97F4A7C15)
 *
 * PROFILE 0 ("benign", the round-1 generator):
 *   matrix / embedding element : k = (h >> 56) - 128        in [-128, 127];  value = k * 2^-12
 *   norm weight element        : j = ((h >> 40) % 33) - 16  in [-16, 16];    value = 1 + j * 2^-7
 *
 * PROFILE 1 ("heavy": the statistics real Qwen3 checkpoints have and profile 0 lacks -- full bf16 mantissas, a
 * log-spread of magnitudes, wide norm weights, a few outlier channels that put massive activations on the residual
 * stream).  Values are assembled as bf16 bit patterns  sign | exponent | 7-bit mantissa:
 *   mant = (h >> 48) & 0x7F ;  t = (h >> 40) & 0xFF ;  z = count of trailing zero bits of (t | 0x100)   (0..8)
 *   matrix / embedding : sign = h >> 63 ; k = z >> 1 (0..4, P(k) ~ 4^-k: equal energy per octave) ;
 *                        |value| = 2^(k - 7) * (1 + mant/128)            in [2^-7, 2^-2)
 *                        AMPLIFIED (x 2^6) when the element belongs to an outlier hidden channel of a tensor that WRITES the
 *                        residual stream: a column of embed_tokens, a row of o_proj or of down_proj.
 *   norm weights, positive.  z10 = trailing zero bits of (((h >> 38) & 0x3FF) | 0x400) (0..10), k = z10 >> 1 (0..5, P(k) ~ 4^-k);
 *                        d = (h >> 36) & 3 ; k2 = trailing zero bits of (((h >> 33) & 7) | 8) (0..3)
 *     input / post-attention / final layernorm : d != 0 : e = k - 1 (capped at 4) ; d == 0 : e = -2 - k2 ; value = 2^e * (1 + mant/128)
 *                        in [2^-5, 2^5): three in four weights in [0.5, 1), a quarter below, an equal-energy-per-octave tail
 *                        up to 32;  at an outlier hidden channel e = -5 (real checkpoints carry small norm weights on their
 *                        massive-activation channels)
 *     q_norm / k_norm  : d != 0 : e = k - 2 ; d == 0 : e = -3 - min(k2, 1) ;  in [2^-4, 2^4): most weights near 0.3, rare ones 8..16
 *                        (attention logits keep a standard deviation of a few units: a model whose f32 forward is itself
 *                        well-conditioned -- a 1-ulp change of the embedding moves the logits by ~1e-6 -- as trained ones are)
 *   outlier channels   : c_j = splitmix64_finalizer(seed * 0xD1B54A32D192ED03 + j + 1) mod hidden_size,  j = 0..3
 *
 * Every value is exactly representable in bf16, so "bf16 checkpoint up-cast to f32"
 * (reference: DType::F32 load at src/models/qwen3.rs:519) is lossless, as it is for real Qwen3 files.
 */
#ifndef NVLLM_SYNTH_H
#define NVLLM_SYNTH_H
#include <stdint.h>

#define SYNTH_FN static inline

#define SYNTH_KIND_MATRIX 0
#define SYNTH_KIND_NORM 1
#define SYNTH_KIND_QKNORM 2

#define SYNTH_AXIS_NONE 0
#define SYNTH_AXIS_COL 1 /* the hidden channel of element idx is idx % cols */
#define SYNTH_AXIS_ROW 2 /* ... is idx / cols */

typedef struct {
    uint64_t name_hash;
    int kind;      /* SYNTH_KIND_* */
    int profile;   /* 0 benign, 1 heavy */
    int axis;      /* SYNTH_AXIS_*: where the hidden channel sits in this tensor (profile 1 only) */
    int64_t cols;  /* row length of the FULL tensor (axis != NONE) */
    uint32_t ch[4];/* outlier hidden channels (profile 1 only) */
} synth_spec;

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

SYNTH_FN void synth_outlier_channels(uint64_t seed, int hidden_size, uint32_t ch[4]) {
    for (int j = 0; j < 4; ++j) ch[j] = (uint32_t)(synth_mix64(seed * 0xD1B54A32D192ED03ULL + (uint64_t)j + 1) % (uint64_t)hidden_size);
}

/* exact bf16 bit pattern of element idx */
SYNTH_FN uint16_t synth_spec_bits(const synth_spec* s, uint64_t idx) {
    const uint64_t h = synth_mix64(s->name_hash + (idx + 1) * 0x9E3779B97F4A7C15ULL);
    if (s->profile == 0) {
        union { float f; uint32_t u; } c;
        if (s->kind != SYNTH_KIND_MATRIX) c.f = 1.0f + (float)((int)((h >> 40) % 33) - 16) * 0.0078125f; /* 2^-7 */
        else c.f = (float)((int)(h >> 56) - 128) * 0.000244140625f;                                      /* 2^-12 */
        return (uint16_t)(c.u >> 16); /* value is bf16-representable: plain truncation */
    }
    int outlier = 0;
    if (s->axis != SYNTH_AXIS_NONE) {
        const uint32_t c = (uint32_t)(s->axis == SYNTH_AXIS_COL ? idx % (uint64_t)s->cols : idx / (uint64_t)s->cols);
        outlier = c == s->ch[0] || c == s->ch[1] || c == s->ch[2] || c == s->ch[3];
    }
    const uint32_t mant = (uint32_t)(h >> 48) & 0x7Fu;
    const uint32_t t = ((uint32_t)(h >> 40) & 0xFFu) | 0x100u;
    int z = 0;
    while (!((t >> z) & 1u)) ++z;
    if (s->kind == SYNTH_KIND_MATRIX) {
        const int e = 127 - 7 + (z >> 1) + (outlier ? 6 : 0);
        return (uint16_t)(((uint32_t)(h >> 63) << 15) | ((uint32_t)e << 7) | mant);
    }
    const uint32_t t10 = ((uint32_t)(h >> 38) & 0x3FFu) | 0x400u;
    int z10 = 0;
    while (!((t10 >> z10) & 1u)) ++z10;
    const uint32_t b = ((uint32_t)(h >> 33) & 7u) | 8u;
    int k2 = 0;
    while (!((b >> k2) & 1u)) ++k2;
    const int up = ((h >> 36) & 3) != 0;
    int e;
    if (s->kind == SYNTH_KIND_QKNORM) e = up ? (z10 >> 1) - 2 : -3 - (k2 < 1 ? k2 : 1);
    else { e = up ? (z10 >> 1) - 1 : -2 - k2; if (e > 4) e = 4; }
    if (outlier) e = -5;
    return (uint16_t)(((uint32_t)(127 + e) << 7) | mant);
}

SYNTH_FN float synth_spec_value(const synth_spec* s, uint64_t idx) {
    union { float f; uint32_t u; } c;
    c.u = (uint32_t)synth_spec_bits(s, idx) << 16;
    return c.f;
}

#endif

/*
 * qwen3_oracle.c -- CPU restatement (plain C, f32) of the reference's Qwen3 forward path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (nano-vllm-candle_amd/, include/) may include,
 * link or call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Parity status: OP-LEVEL PINNED by the reference's own weight-free known answers
 * (tests/test_oracle.py: activation.rs:26-36, layernorm.rs:68-109, linear.rs:232-354,
 * rotary_embedding.rs:115-137, tests/layer_test.rs:440-503, tests/debug_layer_test.rs:38-70,
 * tp.rs:94-98).  FULL-MODEL PARITY IS UNPINNED by the reference: its full-model tests need the
 * author's local Qwen3-0.6B files (tests/layer_test.rs:12) and the reference cannot be built here
 * (no cargo/rustc; arithmetic lives in crates.io candle-core/candle-nn 0.9.1 + gemm 0.17.1, not in
 * /root/reference).  Secondary cross-check (this container only): oracle/validate_vs_hf.py compares
 * this file with transformers' Qwen3 on a random tiny config.
 *
 * Everything below follows the reference operation for operation, in the reference's own mode:
 * dense, NO KV cache, whole sequence re-fed, additive -1e9 causal mask, f32 softmax, LM head on
 * all positions (src/models/qwen3.rs:402 "无 KV cache"; src/engine/llm_engine.rs:60-71,161-169).
 * f32 summation order inside candle's gemm is implementation-defined, so parity with the
 * reference is tolerance-based (1e-3 relative on logits), never bit-exact.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "synth.h"

#define OQ3_API __attribute__((visibility("default")))

/* mirrors Qwen3Config, src/models/qwen3.rs:20-34 */
typedef struct {
    int32_t vocab_size, hidden_size, head_dim, num_hidden_layers;
    int32_t num_attention_heads, num_key_value_heads, intermediate_size, max_position_embeddings;
    double rms_norm_eps;
    double rope_theta;
    int32_t bos_token_id, eos_token_id;
} oq3_config;

typedef struct {
    float *w_qkv;   /* [(nh+2kv)*hd, H]  = cat(q,k,v) dim 0, qwen3.rs:171 */
    float *w_o;     /* [H, nh*hd] */
    float *w_gu;    /* [2I, H] = cat(gate,up) dim 0, qwen3.rs:310 */
    float *w_down;  /* [H, I] */
    float *ln1, *ln2, *qn, *kn;
} oq3_layer;

typedef struct {
    oq3_config cfg;
    float *embed;   /* [V,H] */
    float *lm_head; /* [V,H] (reference keeps the transposed view [H,V], qwen3.rs:528) */
    float *norm;    /* [H] */
    oq3_layer *layers;
    /* optional per-layer taps: (h, residual) returned by every decoder layer, qwen3.rs:398 */
    float *trace_h, *trace_res;
} oq3_model;

/* worker threads of every parallel loop below; returns the count in effect.  The caller sizes it to the CPUs it
 * may really use (a container's cgroup quota can be far below the visible cores: oracle.py usable_cpus) */
OQ3_API int oq3_set_threads(int n) {
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
}

/* ------------------------------------------------------------------------------------------ */
/* ops                                                                                        */
/* ------------------------------------------------------------------------------------------ */

/* y[M,N] = x[M,K] . W[N,K]^T (+ bias[N]) -- candle_nn::Linear, src/layers/linear.rs:35-36,72-77,184-198.
 * Every output is a plain f32 dot product over k (candle's gemm leaves the f32 summation order
 * implementation-defined, see the header).  Blocking below is for speed only (the parity tests run whole
 * Qwen3-0.6B forwards): 4 rows of x against 2 rows of W per inner loop, 8 k-lanes of partial sums each,
 * rows of x taken in chunks that stay cache-resident while the W rows stream past. */
typedef float oq3_v8 __attribute__((vector_size(32)));
static inline oq3_v8 oq3_ld8(const float* p) { oq3_v8 v; memcpy(&v, p, sizeof v); return v; }
static inline float oq3_hsum8(oq3_v8 v) { return ((v[0] + v[4]) + (v[2] + v[6])) + ((v[1] + v[5]) + (v[3] + v[7])); }

/* one dot product with the SAME arithmetic as an element of the blocked kernel below (8 k-lanes of partial sums,
 * fixed reduction tree, scalar tail), so an output value never depends on how its row was blocked */
static inline float oq3_dot(const float* a, const float* b, int K) {
    const int K8 = K & ~7;
    oq3_v8 v = {0};
    for (int k = 0; k < K8; k += 8) v += oq3_ld8(a + k) * oq3_ld8(b + k);
    float acc = oq3_hsum8(v);
    for (int k = K8; k < K; ++k) acc += a[k] * b[k];
    return acc;
}

OQ3_API void oq3_linear(const float* x, const float* W, const float* bias, int M, int K, int N, float* y) {
    enum { MB = 4, NB = 2, MCHUNK = 128 };
    const int K8 = K & ~7;
    for (int m0 = 0; m0 < M; m0 += MCHUNK) {
        const int m1 = m0 + MCHUNK < M ? m0 + MCHUNK : M;
        const int mfull = m0 + ((m1 - m0) / MB) * MB;
#pragma omp parallel for schedule(static)
        for (int n0 = 0; n0 < N; n0 += NB) {
            const int nn = n0 + NB <= N ? NB : N - n0;
            if (nn == NB) {
                const float* w0 = W + (size_t)n0 * K;
                const float* w1 = w0 + K;
                for (int m = m0; m < mfull; m += MB) {
                    const float *x0 = x + (size_t)m * K, *x1 = x0 + K, *x2 = x1 + K, *x3 = x2 + K;
                    oq3_v8 a00 = {0}, a01 = {0}, a10 = {0}, a11 = {0}, a20 = {0}, a21 = {0}, a30 = {0}, a31 = {0};
                    for (int k = 0; k < K8; k += 8) {
                        const oq3_v8 b0 = oq3_ld8(w0 + k), b1 = oq3_ld8(w1 + k);
                        oq3_v8 xv = oq3_ld8(x0 + k); a00 += xv * b0; a01 += xv * b1;
                        xv = oq3_ld8(x1 + k); a10 += xv * b0; a11 += xv * b1;
                        xv = oq3_ld8(x2 + k); a20 += xv * b0; a21 += xv * b1;
                        xv = oq3_ld8(x3 + k); a30 += xv * b0; a31 += xv * b1;
                    }
                    float r[MB][NB] = {{oq3_hsum8(a00), oq3_hsum8(a01)}, {oq3_hsum8(a10), oq3_hsum8(a11)},
                                       {oq3_hsum8(a20), oq3_hsum8(a21)}, {oq3_hsum8(a30), oq3_hsum8(a31)}};
                    for (int i = 0; i < MB; ++i)
                        for (int j = 0; j < NB; ++j) {
                            float acc = r[i][j];
                            const float* xr = x + (size_t)(m + i) * K;
                            const float* wr = W + (size_t)(n0 + j) * K;
                            for (int k = K8; k < K; ++k) acc += xr[k] * wr[k];
                            y[(size_t)(m + i) * N + n0 + j] = bias ? acc + bias[n0 + j] : acc;
                        }
                }
            }
            /* leftover rows of the chunk (and a last odd W row): one dot product per output */
            for (int j = 0; j < nn; ++j) {
                const float* wr = W + (size_t)(n0 + j) * K;
                for (int m = (nn == NB ? mfull : m0); m < m1; ++m) {
                    const float acc = oq3_dot(x + (size_t)m * K, wr, K);
                    y[(size_t)m * N + n0 + j] = bias ? acc + bias[n0 + j] : acc;
                }
            }
        }
    }
}

/* RMSNorm::forward, src/layers/layernorm.rs:44-60.
 * s = x (+ res); var = mean(s^2); y = (s * (1/sqrt(var+eps))) * w; returns (y, s).  rows x n. */
OQ3_API void oq3_rmsnorm(const float* x, const float* res, const float* w, double eps, int rows, int n,
                         float* y, float* res_out) {
    const float epsf = (float)eps;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r) {
        const float* xr = x + (size_t)r * n;
        const float* rr = res ? res + (size_t)r * n : NULL;
        float* yr = y + (size_t)r * n;
        float* so = res_out ? res_out + (size_t)r * n : NULL;
        float ss = 0.f;
        for (int i = 0; i < n; ++i) {
            float s = rr ? xr[i] + rr[i] : xr[i];
            yr[i] = s; /* stash */
            ss += s * s;
        }
        float var = ss / (float)n;
        float rinv = 1.0f / sqrtf(var + epsf); /* (var+eps).sqrt().recip(), layernorm.rs:56 */
        for (int i = 0; i < n; ++i) {
            float s = yr[i];
            if (so) so[i] = s;
            yr[i] = (s * rinv) * w[i];
        }
    }
}

/* SiluAndMul, src/layers/activation.rs:13-18: split last dim in two, silu(a)*b.  x [rows, 2n] -> [rows, n] */
OQ3_API void oq3_silu_mul(const float* x, int rows, int n, float* y) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r) {
        const float* a = x + (size_t)r * 2 * n;
        const float* b = a + n;
        float* yr = y + (size_t)r * n;
        for (int i = 0; i < n; ++i) yr[i] = (a[i] / (1.0f + expf(-a[i]))) * b[i];
    }
}

/* RotaryEmbedding::build_cos_sin, src/layers/rotary_embedding.rs:56-80 (f32 throughout) */
OQ3_API void oq3_rope_table(int hd, float base, int t, float* cosv, float* sinv) {
    int half = hd / 2;
    for (int pos = 0; pos < t; ++pos)
        for (int j = 0; j < half; ++j) {
            float exponent = (2.0f * (float)j) / (float)hd;
            float inv_freq = 1.0f / powf(base, exponent);
            float ang = (float)pos * inv_freq;
            cosv[(size_t)pos * half + j] = cosf(ang);
            sinv[(size_t)pos * half + j] = sinf(ang);
        }
}

/* RotaryEmbedding::apply on one tensor, rotary_embedding.rs:82-107: x [B,heads,T,hd] in place;
 * y1 = x1*cos - x2*sin ; y2 = x2*cos + x1*sin ; positions 0..T */
OQ3_API void oq3_rope_apply(float* x, int B, int heads, int T, int hd, float base) {
    int half = hd / 2;
    float* cosv = (float*)malloc(sizeof(float) * (size_t)T * half);
    float* sinv = (float*)malloc(sizeof(float) * (size_t)T * half);
    oq3_rope_table(hd, base, T, cosv, sinv);
#pragma omp parallel for schedule(static)
    for (int bh = 0; bh < B * heads; ++bh)
        for (int t = 0; t < T; ++t) {
            float* v = x + ((size_t)bh * T + t) * hd;
            for (int j = 0; j < half; ++j) {
                float c = cosv[(size_t)t * half + j], s = sinv[(size_t)t * half + j];
                float x1 = v[j], x2 = v[j + half];
                v[j] = x1 * c - x2 * s;
                v[j + half] = x2 * c + x1 * s;
            }
        }
    free(cosv); free(sinv);
}

/* NUMERICS STUDY knob (tests/tools only; 0 = the reference's arithmetic, the default and the only mode parity tests use):
 * rounds chosen operands of the attention core the way a reduced-precision KV cache would, so the share of each rounding
 * in a logits error can be measured on the CPU at full model size (tools/numerics_study.py).
 * bit 0: K -> f16, bit 1: V -> f16, bit 2: unnormalised P = exp(s - max) -> f16, bit 3: K -> bf16, bit 4: V -> bf16,
 * bit 5: K -> f16 hi + f16 lo (22 bits), bit 6: V -> f16 hi + f16 lo, bit 7: P -> bf16 hi + bf16 lo,
 * bit 8: V -> f16 hi + e5m2 lo (the residual's f16 bits cut to their top byte, round to nearest even): 24-bit V,
 * bit 9: K -> f16 hi + e5m2 lo: 24-bit K. */
static int g_study = 0;
OQ3_API void oq3_set_study(int flags) { g_study = flags; }
static float study_f16(float x) {
    if (!(fabsf(x) < 65504.f)) return x > 0 ? 65504.f : (x < 0 ? -65504.f : x);
    if (fabsf(x) < 6.103515625e-05f) return rintf(x * 16777216.f) / 16777216.f; /* subnormal f16: multiples of 2^-24 */
    union { float f; uint32_t u; } c; c.f = x;
    c.u += 0xFFFu + ((c.u >> 13) & 1u); c.u &= ~0x1FFFu;
    return c.f;
}
static float study_bf16(float x) {
    union { float f; uint32_t u; } c; c.f = x;
    c.u += 0x7FFFu + ((c.u >> 16) & 1u); c.u &= 0xFFFF0000u;
    return c.f;
}
static float study_e5m2(float r) { /* f16(r) rounded to its top 8 bits (sign, 5 exponent, 2 mantissa) */
    r = study_f16(r);
    if (fabsf(r) < 6.103515625e-05f) return rintf(r * 65536.f) / 65536.f; /* f16 subnormals: top byte keeps multiples of 2^-16 */
    union { float f; uint32_t u; } c; c.f = r;
    c.u += 0xFFFFFu + ((c.u >> 21) & 1u); c.u &= ~0x1FFFFFu;
    return c.f;
}
static float study_round(float x, int f16, int bf16, int f16x2) {
    if (f16x2) { const float h = study_f16(x); return h + study_f16(x - h); }
    if (f16) return study_f16(x);
    if (bf16) return study_bf16(x);
    return x;
}

/* attention core, src/models/qwen3.rs:236-277: q [B,nh,T,hd], k,v [B,kv,T,hd] -> ctx [B*T, nh*hd].
 * GQA interleaved expand (q head h uses kv head h/(nh/kv), :241-256), scores*scale (:259),
 * additive mask -1e9 for j>i (:260-271), f32 softmax (:272-273), p.v (:275). */
OQ3_API void oq3_attention(const float* q, const float* k, const float* v, int B, int nh, int kv, int T, int hd,
                           float* ctx) {
    const float scale = powf((float)hd, -0.5f); /* qwen3.rs:134 */
    const int rep = nh / kv;
    float *ks = NULL, *vs = NULL;
    if (g_study & (1 | 8 | 32 | 512)) {
        const size_t n = (size_t)B * kv * T * hd;
        ks = (float*)malloc(sizeof(float) * n);
        for (size_t i = 0; i < n; ++i) {
            if (g_study & 512) { const float h = study_f16(k[i]); ks[i] = h + study_e5m2(k[i] - h); }
            else ks[i] = study_round(k[i], g_study & 1, g_study & 8, g_study & 32);
        }
        k = ks;
    }
    if (g_study & (2 | 16 | 64 | 256)) {
        const size_t n = (size_t)B * kv * T * hd;
        vs = (float*)malloc(sizeof(float) * n);
        for (size_t i = 0; i < n; ++i) {
            if (g_study & 256) { const float h = study_f16(v[i]); vs[i] = h + study_e5m2(v[i] - h); }
            else vs[i] = study_round(v[i], g_study & 2, g_study & 16, g_study & 64);
        }
        v = vs;
    }
    const int study_p = g_study & (4 | 128);
#pragma omp parallel
    {
        float* sc = (float*)malloc(sizeof(float) * (size_t)T);
#pragma omp for collapse(2) schedule(static)
        for (int b = 0; b < B; ++b)
            for (int h = 0; h < nh; ++h) {
                const float* qh = q + ((size_t)(b * nh + h) * T) * hd;
                const float* kh = k + ((size_t)(b * kv + h / rep) * T) * hd;
                const float* vh = v + ((size_t)(b * kv + h / rep) * T) * hd;
                for (int i = 0; i < T; ++i) {
                    /* Keys j > i carry the additive -1e9 mask (:260-271): after the row maximum is subtracted their
                     * exp() is exactly 0 in f32 (the unmasked j == i score is finite), so they contribute nothing to
                     * the sum or to p.v -- the loops below stop at j == i instead of computing those zeros. */
                    float mx = -INFINITY;
                    for (int j = 0; j <= i; ++j) {
                        const float acc = oq3_dot(qh + (size_t)i * hd, kh + (size_t)j * hd, hd) * scale;
                        sc[j] = acc;
                        if (acc > mx) mx = acc;
                    }
                    float sum = 0.f;
                    for (int j = 0; j <= i; ++j) { sc[j] = expf(sc[j] - mx); sum += sc[j]; }
                    if (study_p) /* the sum stays f32 (as an f32 row sum of the unrounded P would); only the P.V operand is rounded */
                        for (int j = 0; j <= i; ++j) {
                            if (study_p & 4) sc[j] = study_f16(sc[j]);
                            else { const float h = study_bf16(sc[j]); sc[j] = h + study_bf16(sc[j] - h); }
                        }
                    float* out = ctx + ((size_t)(b * T + i) * nh + h) * hd;
                    for (int d = 0; d < hd; ++d) out[d] = 0.f;
                    for (int j = 0; j <= i; ++j) {
                        const float p = sc[j] / sum;
                        if (p == 0.f) continue;
                        const float* vr = vh + (size_t)j * hd;
#pragma omp simd
                        for (int d = 0; d < hd; ++d) out[d] += p * vr[d];
                    }
                }
            }
        free(sc);
    }
    free(ks); free(vs);
}

/* TPConfig shard arithmetic, src/tp.rs:59-65 */
OQ3_API int64_t oq3_tp_shard_size(int64_t total, int size) { return total / size; }
OQ3_API int64_t oq3_tp_shard_offset(int64_t total, int size, int rank) { return rank * (total / size); }

/* last-max argmax: Iterator::max_by returns the LAST maximal element, llm_engine.rs:135-142 */
OQ3_API int oq3_argmax_last(const float* v, int n) {
    int best = 0;
    for (int i = 1; i < n; ++i)
        if (v[i] >= v[best]) best = i;
    return best;
}

/* ------------------------------------------------------------------------------------------ */
/* model                                                                                      */
/* ------------------------------------------------------------------------------------------ */

static float* falloc(size_t n) {
    float* p = (float*)malloc(sizeof(float) * n);
    if (!p) { fprintf(stderr, "oq3: out of memory (%zu floats)\n", n); abort(); }
    return p;
}

OQ3_API oq3_model* oq3_create(const oq3_config* cfg) {
    oq3_model* m = (oq3_model*)calloc(1, sizeof(oq3_model));
    m->cfg = *cfg;
    const size_t H = cfg->hidden_size, V = cfg->vocab_size, hd = cfg->head_dim, I = cfg->intermediate_size;
    const size_t nh = cfg->num_attention_heads, kv = cfg->num_key_value_heads;
    m->embed = falloc(V * H);
    m->lm_head = falloc(V * H);
    m->norm = falloc(H);
    m->layers = (oq3_layer*)calloc(cfg->num_hidden_layers, sizeof(oq3_layer));
    for (int l = 0; l < cfg->num_hidden_layers; ++l) {
        oq3_layer* L = &m->layers[l];
        L->w_qkv = falloc((nh + 2 * kv) * hd * H);
        L->w_o = falloc(H * nh * hd);
        L->w_gu = falloc(2 * I * H);
        L->w_down = falloc(H * I);
        L->ln1 = falloc(H); L->ln2 = falloc(H); L->qn = falloc(hd); L->kn = falloc(hd);
    }
    return m;
}

OQ3_API void oq3_destroy(oq3_model* m) {
    if (!m) return;
    for (int l = 0; l < m->cfg.num_hidden_layers; ++l) {
        oq3_layer* L = &m->layers[l];
        free(L->w_qkv); free(L->w_o); free(L->w_gu); free(L->w_down);
        free(L->ln1); free(L->ln2); free(L->qn); free(L->kn);
    }
    free(m->layers); free(m->embed); free(m->lm_head); free(m->norm); free(m);
}

/* Resolve an HF tensor name (qwen3.rs:150,156,162,168,178,184,304,308,313,353,360,432,444,526) to
 * its destination inside the (fused) oracle tensors.  Returns NULL on unknown name. */
static float* resolve_ex(oq3_model* m, const char* name, size_t* numel, int* kind, int* axis, int64_t* cols) {
    const oq3_config* c = &m->cfg;
    const size_t H = c->hidden_size, V = c->vocab_size, hd = c->head_dim, I = c->intermediate_size;
    const size_t nh = c->num_attention_heads, kv = c->num_key_value_heads;
    *kind = SYNTH_KIND_MATRIX;
    *axis = SYNTH_AXIS_NONE; *cols = 1;  /* where the hidden channel sits (oracle/synth.h, profile 1) */
    if (!strcmp(name, "model.embed_tokens.weight")) { *numel = V * H; *axis = SYNTH_AXIS_COL; *cols = (int64_t)H; return m->embed; }
    if (!strcmp(name, "lm_head.weight")) { *numel = V * H; return m->lm_head; }
    if (!strcmp(name, "model.norm.weight")) { *numel = H; *kind = SYNTH_KIND_NORM; *axis = SYNTH_AXIS_COL; *cols = (int64_t)H; return m->norm; }
    int l = -1, off = 0;
    if (sscanf(name, "model.layers.%d.%n", &l, &off) != 1 || l < 0 || l >= c->num_hidden_layers) return NULL;
    const char* s = name + off;
    oq3_layer* L = &m->layers[l];
    if (!strcmp(s, "self_attn.q_proj.weight")) { *numel = nh * hd * H; return L->w_qkv; }
    if (!strcmp(s, "self_attn.k_proj.weight")) { *numel = kv * hd * H; return L->w_qkv + nh * hd * H; }
    if (!strcmp(s, "self_attn.v_proj.weight")) { *numel = kv * hd * H; return L->w_qkv + (nh + kv) * hd * H; }
    if (!strcmp(s, "self_attn.o_proj.weight")) { *numel = H * nh * hd; *axis = SYNTH_AXIS_ROW; *cols = (int64_t)(nh * hd); return L->w_o; }
    if (!strcmp(s, "mlp.gate_proj.weight")) { *numel = I * H; return L->w_gu; }
    if (!strcmp(s, "mlp.up_proj.weight")) { *numel = I * H; return L->w_gu + I * H; }
    if (!strcmp(s, "mlp.down_proj.weight")) { *numel = H * I; *axis = SYNTH_AXIS_ROW; *cols = (int64_t)I; return L->w_down; }
    *kind = SYNTH_KIND_NORM;
    if (!strcmp(s, "input_layernorm.weight")) { *numel = H; *axis = SYNTH_AXIS_COL; *cols = (int64_t)H; return L->ln1; }
    if (!strcmp(s, "post_attention_layernorm.weight")) { *numel = H; *axis = SYNTH_AXIS_COL; *cols = (int64_t)H; return L->ln2; }
    *kind = SYNTH_KIND_QKNORM;
    if (!strcmp(s, "self_attn.q_norm.weight")) { *numel = hd; return L->qn; }
    if (!strcmp(s, "self_attn.k_norm.weight")) { *numel = hd; return L->kn; }
    return NULL;
}
static float* resolve(oq3_model* m, const char* name, size_t* numel, int* kind) {
    int axis; int64_t cols;
    return resolve_ex(m, name, numel, kind, &axis, &cols);
}

/* load one HF-named tensor from f32 host data ([out,in] row-major); 0 = ok */
OQ3_API int oq3_set_tensor(oq3_model* m, const char* name, const float* data, int64_t numel) {
    size_t n; int kind;
    float* dst = resolve(m, name, &n, &kind);
    if (!dst || (int64_t)n != numel) return -1;
    memcpy(dst, data, sizeof(float) * n);
    return 0;
}

/* read one HF-named tensor back (for fixtures / cross-checks) */
OQ3_API int oq3_get_tensor(oq3_model* m, const char* name, float* out, int64_t numel) {
    size_t n; int kind;
    float* src = resolve(m, name, &n, &kind);
    if (!src || (int64_t)n != numel) return -1;
    memcpy(out, src, sizeof(float) * n);
    return 0;
}

static void fill_one(oq3_model* m, const char* name, uint64_t seed, int profile) {
    size_t n;
    synth_spec sp;
    float* dst = resolve_ex(m, name, &n, &sp.kind, &sp.axis, &sp.cols);
    if (!dst) { fprintf(stderr, "oq3: bad tensor name %s\n", name); abort(); }
    sp.name_hash = synth_name_hash(name, seed);
    sp.profile = profile;
    synth_outlier_channels(seed, m->cfg.hidden_size, sp.ch);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) dst[i] = synth_spec_value(&sp, i);
}

/* fill every tensor from the deterministic generator (oracle/synth.h); profile 0 = benign, 1 = heavy */
OQ3_API void oq3_fill_synthetic_profile(oq3_model* m, uint64_t seed, int profile) {
    char name[128];
    fill_one(m, "model.embed_tokens.weight", seed, profile);
    fill_one(m, "lm_head.weight", seed, profile);
    fill_one(m, "model.norm.weight", seed, profile);
    static const char* per_layer[] = {"self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight",
                                      "self_attn.o_proj.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight",
                                      "mlp.down_proj.weight", "input_layernorm.weight",
                                      "post_attention_layernorm.weight", "self_attn.q_norm.weight",
                                      "self_attn.k_norm.weight"};
    for (int l = 0; l < m->cfg.num_hidden_layers; ++l)
        for (size_t k = 0; k < sizeof(per_layer) / sizeof(per_layer[0]); ++k) {
            snprintf(name, sizeof name, "model.layers.%d.%s", l, per_layer[k]);
            fill_one(m, name, seed, profile);
        }
}
OQ3_API void oq3_fill_synthetic(oq3_model* m, uint64_t seed) { oq3_fill_synthetic_profile(m, seed, 0); }

/* raw generator access for tests (bit-equality with the device generator) */
OQ3_API void oq3_synth_bf16(const char* name, uint64_t seed, int kind, int64_t first, int64_t count, uint16_t* out) {
    synth_spec sp;
    memset(&sp, 0, sizeof sp);
    sp.name_hash = synth_name_hash(name, seed); sp.kind = kind;
    for (int64_t i = 0; i < count; ++i) out[i] = synth_spec_bits(&sp, (uint64_t)(first + i));
}
/* the same for any profile: axis / cols / hidden_size say where the hidden channel of an element sits (synth.h) */
OQ3_API void oq3_synth_bf16_spec(const char* name, uint64_t seed, int kind, int profile, int axis, int64_t cols, int hidden_size,
                                 int64_t first, int64_t count, uint16_t* out) {
    synth_spec sp;
    memset(&sp, 0, sizeof sp);
    sp.name_hash = synth_name_hash(name, seed); sp.kind = kind; sp.profile = profile; sp.axis = axis; sp.cols = cols > 0 ? cols : 1;
    synth_outlier_channels(seed, hidden_size > 0 ? hidden_size : 1, sp.ch);
    for (int64_t i = 0; i < count; ++i) out[i] = synth_spec_bits(&sp, (uint64_t)(first + i));
}

OQ3_API void oq3_set_trace(oq3_model* m, float* h_per_layer, float* res_per_layer) {
    m->trace_h = h_per_layer; m->trace_res = res_per_layer;
}

/* Qwen3Attention::forward, qwen3.rs:202-280.  x [B*T,H] -> out [B*T,H] */
static void attn_forward(const oq3_model* m, const oq3_layer* L, const float* x, int B, int T, float* out) {
    const oq3_config* c = &m->cfg;
    const int H = c->hidden_size, hd = c->head_dim, nh = c->num_attention_heads, kv = c->num_key_value_heads;
    const int M = B * T, q_size = nh * hd, kv_size = kv * hd, NQ = q_size + 2 * kv_size;
    float* qkv = falloc((size_t)M * NQ);
    oq3_linear(x, L->w_qkv, NULL, M, H, NQ, qkv); /* :205 */
    /* narrow + reshape + transpose(1,2) -> [B,heads,T,hd]  (:208-222) */
    float* q = falloc((size_t)M * q_size);
    float* k = falloc((size_t)M * kv_size);
    float* v = falloc((size_t)M * kv_size);
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < T; ++t) {
            const float* row = qkv + (size_t)(b * T + t) * NQ;
            for (int h = 0; h < nh; ++h)
                memcpy(q + ((size_t)(b * nh + h) * T + t) * hd, row + h * hd, sizeof(float) * hd);
            for (int h = 0; h < kv; ++h) {
                memcpy(k + ((size_t)(b * kv + h) * T + t) * hd, row + q_size + h * hd, sizeof(float) * hd);
                memcpy(v + ((size_t)(b * kv + h) * T + t) * hd, row + q_size + kv_size + h * hd, sizeof(float) * hd);
            }
        }
    /* q_norm / k_norm over head_dim BEFORE RoPE and BEFORE GQA expand (:224-232) */
    oq3_rmsnorm(q, NULL, L->qn, c->rms_norm_eps, B * nh * T, hd, q, NULL);
    oq3_rmsnorm(k, NULL, L->kn, c->rms_norm_eps, B * kv * T, hd, k, NULL);
    /* RoPE (:234), base = rope_theta as f32 (:135,:196) */
    oq3_rope_apply(q, B, nh, T, hd, (float)c->rope_theta);
    oq3_rope_apply(k, B, kv, T, hd, (float)c->rope_theta);
    float* ctx = falloc((size_t)M * q_size);
    oq3_attention(q, k, v, B, nh, kv, T, hd, ctx);            /* :236-277 */
    oq3_linear(ctx, L->w_o, NULL, M, q_size, H, out);         /* :278, RowParallelLinear no all-reduce */
    free(qkv); free(q); free(k); free(v); free(ctx);
}

/* Qwen3ForCausalLM::forward -> Qwen3Model::forward, qwen3.rs:538-540, :458-499.
 * ids [B,T] (u32) -> hidden [B*T,H] (after the final norm). */
OQ3_API int oq3_forward(const oq3_model* m, const uint32_t* ids, int B, int T, float* hidden) {
    const oq3_config* c = &m->cfg;
    const int H = c->hidden_size, I = c->intermediate_size, M = B * T;
    float* h = falloc((size_t)M * H);
    float* res = falloc((size_t)M * H);
    float* xn = falloc((size_t)M * H);
    float* att = falloc((size_t)M * H);
    float* gu = falloc((size_t)M * 2 * I);
    float* act = falloc((size_t)M * I);
    for (int i = 0; i < M; ++i) { /* embedding gather :465-468 */
        if (ids[i] >= (uint32_t)c->vocab_size) { free(h); free(res); free(xn); free(att); free(gu); free(act); return -1; }
        memcpy(h + (size_t)i * H, m->embed + (size_t)ids[i] * H, sizeof(float) * H);
    }
    for (int l = 0; l < c->num_hidden_layers; ++l) { /* layer loop :481-493; layer = :374-399 */
        const oq3_layer* L = &m->layers[l];
        if (l == 0) { /* residual None: normed = norm(x), residual = x (:382-386) */
            oq3_rmsnorm(h, NULL, L->ln1, c->rms_norm_eps, M, H, xn, NULL);
            memcpy(res, h, sizeof(float) * (size_t)M * H);
        } else {
            oq3_rmsnorm(h, res, L->ln1, c->rms_norm_eps, M, H, xn, res); /* :378 */
        }
        attn_forward(m, L, xn, B, T, att);                                /* :390 */
        oq3_rmsnorm(att, res, L->ln2, c->rms_norm_eps, M, H, xn, res);    /* :393 */
        oq3_linear(xn, L->w_gu, NULL, M, H, 2 * I, gu);                   /* Qwen3MLP :323-327 */
        oq3_silu_mul(gu, M, I, act);
        oq3_linear(act, L->w_down, NULL, M, I, H, h);
        if (m->trace_h) memcpy(m->trace_h + (size_t)l * M * H, h, sizeof(float) * (size_t)M * H);
        if (m->trace_res) memcpy(m->trace_res + (size_t)l * M * H, res, sizeof(float) * (size_t)M * H);
    }
    oq3_rmsnorm(h, res, m->norm, c->rms_norm_eps, M, H, hidden, NULL);    /* :497 */
    free(h); free(res); free(xn); free(att); free(gu); free(act);
    return 0;
}

/* Qwen3ForCausalLM::compute_logits, qwen3.rs:542-550: hidden [rows,H] -> logits [rows,V] */
OQ3_API void oq3_compute_logits(const oq3_model* m, const float* hidden, int rows, float* logits) {
    oq3_linear(hidden, m->lm_head, NULL, rows, m->cfg.hidden_size, m->cfg.vocab_size, logits);
}

/* Qwen3ModelRunner::{build_batch, run} with the greedy (argmax) fallback path,
 * src/engine/llm_engine.rs:60-95,145-189,135-142.
 *   - right-pad every sequence's FULL token list to max_len with eos (:80-90)
 *   - forward + logits; take row len-1 of each sequence (:177-187)
 *   - next id = last maximal logit (:135-142)
 * all_rows != 0 : LM head over all B*T rows like the reference (:169) -- used for the timed CPU baseline
 * all_rows == 0 : LM head on the used rows only (same values; keeps parity tests fast)
 * last_logits (optional) [n_seqs, V].  Returns 0, or -1 on a bad token id. */
OQ3_API int oq3_run_greedy(const oq3_model* m, int n_seqs, const uint32_t* const* tokens, const int32_t* lens,
                           int all_rows, uint32_t* next_ids, float* last_logits) {
    const oq3_config* c = &m->cfg;
    const int H = c->hidden_size, V = c->vocab_size;
    int T = 0;
    for (int i = 0; i < n_seqs; ++i) if (lens[i] > T) T = lens[i];
    if (T == 0) T = 1;
    uint32_t* ids = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n_seqs * T);
    for (int i = 0; i < n_seqs; ++i)
        for (int t = 0; t < T; ++t) ids[(size_t)i * T + t] = t < lens[i] ? tokens[i][t] : (uint32_t)c->eos_token_id;
    float* hidden = falloc((size_t)n_seqs * T * H);
    int rc = oq3_forward(m, ids, n_seqs, T, hidden);
    if (rc == 0) {
        float* row_logits = falloc((size_t)V);
        float* all = NULL;
        if (all_rows) { all = falloc((size_t)n_seqs * T * V); oq3_compute_logits(m, hidden, n_seqs * T, all); }
        for (int i = 0; i < n_seqs; ++i) {
            int last = lens[i] > 0 ? lens[i] - 1 : 0; /* saturating_sub(1), :181 */
            const float* lg;
            if (all) lg = all + ((size_t)i * T + last) * V;
            else { oq3_compute_logits(m, hidden + ((size_t)i * T + last) * H, 1, row_logits); lg = row_logits; }
            next_ids[i] = (uint32_t)oq3_argmax_last(lg, V);
            if (last_logits) memcpy(last_logits + (size_t)i * V, lg, sizeof(float) * V);
        }
        free(row_logits); free(all);
    }
    free(ids); free(hidden);
    return rc;
}

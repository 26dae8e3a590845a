/*
 * nvllm_amd.h -- C ABI of the MI355X-native Qwen3 forward path for nano-vllm-candle.
 *
 * The reference (Rust) has no FFI today; this header is what a maintainer binds with
 * `extern "C"` so that the reference's LLMEngine / Scheduler / BlockManager stay untouched
 * (INTEGRATION.md shows the Rust stub).  Two seams are served:
 *
 *   coarse seam  trait ModelRunner { fn run(&mut self, seqs, is_prefill) -> Vec<usize> }
 *                (src/engine/llm_engine.rs:16-18, impl :145-189)          -> nvllm_step()
 *   fine seam    layers::{RMSNorm, RotaryEmbedding, SiluAndMul, *ParallelLinear, Attention}
 *                (src/layers/) and tp::TPConfig (src/tp.rs)            -> nvllm_op_*()
 *
 * Conventions: plain C; opaque handles; every function returns 0 on success, a negative
 * NVLLM_E* code otherwise and records a message for nvllm_last_error(); no exceptions cross
 * the boundary; host buffers are caller-owned; device buffers are library-owned unless a
 * function takes raw device pointers (nvllm_op_*).  A context is NOT thread-safe (the reference
 * engine is single-threaded: Rc<RefCell<..>>, llm_engine.rs:17).  One process per GPU; tensor
 * parallelism = one process per rank (TP_RANK/TP_SIZE as src/tp.rs:21-31) joined by an RCCL
 * unique id.  All tensors are row-major, weights are [out_features, in_features] exactly as the
 * reference loads them (qwen3.rs:147-170).
 */
#ifndef NVLLM_AMD_H
#define NVLLM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NVLLM_OK 0
#define NVLLM_EINVAL (-1)   /* bad argument / shape / unknown tensor name */
#define NVLLM_EHIP (-2)     /* HIP runtime error (message has the hipError string) */
#define NVLLM_ENOMEM (-3)   /* KV block pool or device memory exhausted */
#define NVLLM_ESTATE (-4)   /* call order violated (e.g. step before finalize / kv_alloc) */
#define NVLLM_ERCCL (-5)    /* RCCL error */

#define NVLLM_DTYPE_F32 0
#define NVLLM_DTYPE_BF16 1

#define NVLLM_RCCL_ID_BYTES 128

typedef struct nvllm_ctx nvllm_ctx;
typedef struct nvllm_model nvllm_model;

/* mirrors Qwen3Config, src/models/qwen3.rs:20-34 (hidden_act is always "silu") */
typedef struct nvllm_qwen3_config {
    int32_t vocab_size;
    int32_t hidden_size;
    int32_t head_dim;
    int32_t num_hidden_layers;
    int32_t num_attention_heads;
    int32_t num_key_value_heads;
    int32_t intermediate_size;
    int32_t max_position_embeddings;
    double rms_norm_eps;
    double rope_theta;
    int32_t bos_token_id;
    int32_t eos_token_id;
} nvllm_qwen3_config;

/* ---- context (replaces candle_core::Device selection, src/main.rs:46-82, + tp::get_tp, src/tp.rs:68-70) */

/* message of the last failure on this context (ctx may be NULL: last failure of a create call) */
const char* nvllm_last_error(const nvllm_ctx* ctx);
/* rank 0 makes an id and ships it to the other ranks out of band (file, socket, torch.distributed) */
int nvllm_rccl_unique_id(void* out_id /* NVLLM_RCCL_ID_BYTES */);
/* tp_size==1: rccl_id may be NULL.  tp_rank >= tp_size is folded to 0 like src/tp.rs:24-29. */
int nvllm_ctx_create(int device_ordinal, int tp_rank, int tp_size, const void* rccl_id, nvllm_ctx** out);
/* TEST-ONLY communicator: the tp_size contexts created with the same `group` name by different host threads of
 * one process exchange through host memory instead of RCCL (exercises every TP code path on a single GPU). */
int nvllm_ctx_create_loopback(int device_ordinal, int tp_rank, int tp_size, const char* group, nvllm_ctx** out);
int nvllm_ctx_destroy(nvllm_ctx* ctx);
int nvllm_ctx_synchronize(nvllm_ctx* ctx);
/* hipStream_t every library launch goes to (time it with HIP events recorded on THIS stream) */
void* nvllm_ctx_stream(nvllm_ctx* ctx);
int nvllm_ctx_tp_rank(const nvllm_ctx* ctx);
int nvllm_ctx_tp_size(const nvllm_ctx* ctx);
/* HIP-event timer on the library stream: start; ...launches...; stop returns elapsed ms */
int nvllm_timer_start(nvllm_ctx* ctx);
int nvllm_timer_stop(nvllm_ctx* ctx, float* elapsed_ms);

/* ---- model (replaces Qwen3ForCausalLM::from_hf_dir, src/models/qwen3.rs:515-536) */

int nvllm_model_create(nvllm_ctx* ctx, const nvllm_qwen3_config* cfg, nvllm_model** out);
int nvllm_model_destroy(nvllm_model* m);
/* One HF-named tensor (names as qwen3.rs:150,156,162,168,178,184,304,308,313,353,360,432,444,526),
 * FULL (unsharded) shape, host pointer, f32 or bf16.  The library concatenates q/k/v and gate/up
 * (qwen3.rs:171,310), takes this rank's TP shard, converts to bf16 (exact for bf16 checkpoints) and
 * repacks into its MFMA tile layout.  lm_head.weight missing at finalize => tied to embed_tokens. */
int nvllm_model_load_tensor(nvllm_model* m, const char* hf_name, const void* host_data, int dtype,
                            const int64_t* shape, int ndim);
/* Fill every tensor from the deterministic generator (oracle/synth.h documents the recipe) directly
 * in HBM -- bench / test input, identical values on every machine and every TP rank layout. */
int nvllm_model_fill_synthetic(nvllm_model* m, uint64_t seed);
/* The same with a choice of value statistics: profile 0 = the call above (256 distinct small values, norm weights near 1);
 * profile 1 = "heavy": full bf16 mantissas over five octaves, norm weights in [2^-5, 2^5), q/k-norm weights in [2^-4, 2^4),
 * four outlier hidden channels amplified x64 in embed_tokens / o_proj / down_proj (massive activations on the residual
 * stream) -- the statistics real Qwen3 checkpoints have; parity stress input (tests/test_stress_gpu.py). */
int nvllm_model_fill_synthetic_profile(nvllm_model* m, uint64_t seed, int profile);
int nvllm_model_finalize(nvllm_model* m);
/* bytes of weights this rank reads per decode step (layers + final norm + LM head; embedding excluded) */
int64_t nvllm_model_weight_bytes(const nvllm_model* m);

/* Host-only helper (no GPU): the region {row0, col0, rows, cols} of the FULL HF tensor that rank tp_rank of
 * tp_size owns -- column-parallel q/k/v/gate/up/lm_head shard rows, row-parallel o/down shard columns, norms and
 * the embedding are replicated.  This is the working version of the reference's shard arithmetic
 * (src/tp.rs:59-65; src/layers/linear.rs:80-89,201-210 only work for rank 0). */
int nvllm_tp_shard(const nvllm_qwen3_config* cfg, int tp_size, int tp_rank, const char* hf_name, int64_t out_region[4]);

/* ---- KV block pool (the real allocation the reference's stub BlockManager lacks,
 *      src/engine/block_manager.rs:24-29,64-98).  block_size must be 256 (src/engine/sequence.rs:35).
 *      max_batched_tokens bounds the rows processed per internal chunk (prefill is chunked). */
int nvllm_kv_alloc(nvllm_model* m, int num_blocks, int block_size, int max_seqs, int max_batched_tokens);
int nvllm_kv_num_free_blocks(const nvllm_model* m);
/* bytes one cached token occupies on this rank (K+V, all layers) */
int64_t nvllm_kv_bytes_per_token(const nvllm_model* m);
/* release a sequence's blocks (call where the reference calls BlockManager::deallocate,
 * src/engine/scheduler.rs:207,239).  Unknown seq_id is not an error. */
int nvllm_seq_free(nvllm_model* m, int64_t seq_id);

/* ---- step: the ModelRunner::run contract (src/engine/llm_engine.rs:145-189)
 * For each of n_seqs sequences: its id (Sequence.seq_id), its FULL token list (prompt+completion,
 * llm_engine.rs:60-71) and length.  is_prefill != 0: the sequence is (re)started, all tokens are
 * processed and cached.  is_prefill == 0: only tokens not yet cached (normally one) are processed.
 * next_ids[i] = greedy token of sequence i = LAST maximal logit of its last row
 * (llm_engine.rs:135-142,177-187).  last_logits (nullable) receives those rows, [n_seqs, vocab] f32,
 * for parity checks and host-side temperature sampling (llm_engine.rs:97-133).
 * On failure nothing is written; the caller maps non-zero to the reference's "eos for all" convention
 * (llm_engine.rs:153-175). */
int nvllm_step(nvllm_model* m, int n_seqs, const int64_t* seq_ids, const uint32_t* const* tokens,
               const int32_t* lens, int is_prefill, uint32_t* next_ids, float* last_logits);

/* The same step with the reference's sample_token on the device (src/engine/llm_engine.rs:97-133) instead of the greedy
 * arg-max: next_ids[i] is drawn from softmax(logits_i / max(temperatures[i], 1e-6)); a row whose weights do not form a
 * distribution falls back to the arg-max (last max), as the reference does.  The draw is the Gumbel-max form over a
 * counter RNG keyed by (seed, seq_id, position): reproducible for a given seed and independent of batch composition
 * (the reference uses an unseeded thread RNG, so only the distribution can be compared with it).  last_logits as above. */
int nvllm_step_sample(nvllm_model* m, int n_seqs, const int64_t* seq_ids, const uint32_t* const* tokens,
                      const int32_t* lens, int is_prefill, const float* temperatures, uint64_t seed,
                      uint32_t* next_ids, float* last_logits);

/* Decode fast path for benchmarks and engines that keep ids on the device: one more decode step for
 * the same batch as the previous nvllm_step / nvllm_decode_next call, feeding each sequence the id
 * produced by that call.  No host<->device traffic besides next_ids (nullable). */
int nvllm_decode_next(nvllm_model* m, uint32_t* next_ids);

/* Pipelined form of nvllm_decode_next: enqueue puts one more decode step on the stream and returns at once (up to 4
 * in flight); collect waits for the OLDEST enqueued step and returns its ids.  Enqueueing step t+1 before collecting
 * step t hides the host round trip between steps. */
int nvllm_decode_enqueue(nvllm_model* m);
int nvllm_decode_collect(nvllm_model* m, uint32_t* next_ids);

/* per-decode-step algorithmic HBM bytes of the LAST step on this rank:
 * weight_bytes + sum_seq ctx*kv_tok + n_seqs*kv_tok (+ 4*n_seqs*vocab when logits left the device) */
int64_t nvllm_last_step_bytes(const nvllm_model* m);
/* ---- fine seam: single ops on raw DEVICE pointers, launched on the context stream.
 * Shapes in elements; all f32 unless noted; tensors dense row-major. */

/* y[M,N] = x[M,K] . W[N,K]^T (+bias[N])   -- candle_nn::Linear as used by src/layers/linear.rs:33-37,
 * 72-77,171-175,184-198.  `w` is an opaque packed weight made by nvllm_op_pack_weight. */
typedef struct nvllm_weight nvllm_weight;
int nvllm_op_pack_weight(nvllm_ctx* ctx, const void* host_w, int dtype, int N, int K, nvllm_weight** out);
int nvllm_op_free_weight(nvllm_ctx* ctx, nvllm_weight* w);
int nvllm_op_linear(nvllm_ctx* ctx, const float* x, const nvllm_weight* w, const float* bias, int M, float* y);

/* RMSNorm::forward(x, residual?) -> (y, new_residual?)  src/layers/layernorm.rs:44-60.
 * residual/residual_out nullable together.  weight [n]. */
int nvllm_op_rmsnorm(nvllm_ctx* ctx, const float* x, const float* residual, const float* weight, double eps,
                     int rows, int n, float* y, float* residual_out);

/* SiluAndMul::forward  src/layers/activation.rs:13-18.  x [rows, 2n] -> y [rows, n] */
int nvllm_op_silu_mul(nvllm_ctx* ctx, const float* x, int rows, int n, float* y);

/* RotaryEmbedding::apply  src/layers/rotary_embedding.rs:93-107.  q [B,nh,T,hd], k [B,kv,T,hd] in
 * place, positions 0..T, half-split rotation, inv_freq = 1/base^(2j/hd) in f32. */
int nvllm_op_rope(nvllm_ctx* ctx, float* q, float* k, int B, int nh, int kv, int T, int hd, float base);

/* The attention layers::Attention::forward(q,k,v) should have been (src/layers/attention.rs:4-16 is a
 * dead sdpa wrapper; the live math is src/models/qwen3.rs:236-277): causal GQA attention,
 * q [B,nh,T,hd], k/v [B,kv,T,hd] -> ctx [B*T, nh*hd].  Internally writes k/v into a scratch paged
 * cache and runs the same paged kernel the step path uses. */
int nvllm_op_attention(nvllm_ctx* ctx, const float* q, const float* k, const float* v, int B, int nh, int kv,
                       int T, int hd, float scale, float* out);

/* embedding gather  src/models/qwen3.rs:465-468: table [V,H] f32 (device), ids [n] u32 (device) */
int nvllm_op_embedding(nvllm_ctx* ctx, const float* table, const uint32_t* ids, int n, int V, int H, float* y);

/* argmax with the reference's tie rule (LAST max wins, llm_engine.rs:135-142): logits [rows,V] -> ids */
int nvllm_op_argmax(nvllm_ctx* ctx, const float* logits, int rows, int V, uint32_t* ids);

/* sum-all-reduce of a device f32 buffer across the TP group (absent in the reference:
 * RowParallelLinear::forward has no all-reduce, src/layers/linear.rs:184-198) */
int nvllm_op_allreduce(nvllm_ctx* ctx, float* buf, int64_t count);

/* device generator access for tests: count bf16 elements of tensor `name` starting at `first`
 * into a HOST buffer (generated on the GPU, copied back) */
int nvllm_op_synth_bf16(nvllm_ctx* ctx, const char* name, uint64_t seed, int kind, int64_t first, int64_t count,
                        uint16_t* host_out);

/* device memory helpers so a non-HIP host (Rust, ctypes) can feed the nvllm_op_* calls */
int nvllm_dev_alloc(nvllm_ctx* ctx, size_t bytes, void** out);
int nvllm_dev_free(nvllm_ctx* ctx, void* p);
int nvllm_dev_upload(nvllm_ctx* ctx, void* dst, const void* host_src, size_t bytes);
int nvllm_dev_download(nvllm_ctx* ctx, void* host_dst, const void* src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* NVLLM_AMD_H */

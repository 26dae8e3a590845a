/*
 * nvllm_amd_debug.h -- profiling, parity-debug and kernel-tuning entry points of libnvllm_amd.so.
 *
 * NOT part of the boundary a maintainer of the reference binds (that is include/nvllm_amd.h: the ModelRunner::run
 * seam, src/engine/llm_engine.rs:16-18,145-189, and the layer surface, src/layers/).  These exports exist for
 * bench.py (per-kernel HIP-event timing), the parity tests (per-layer taps, the Qwen3DecoderLayer::forward
 * (h, residual) pair of src/models/qwen3.rs:374-399) and the tools/ tuning scripts.  Same conventions as the main header.
 */
#ifndef NVLLM_AMD_DEBUG_H
#define NVLLM_AMD_DEBUG_H

#include "nvllm_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Projection aid: rank tp_rank of a tp_size group WITHOUT a communicator.  Models created on it take the rank's shard
 * shapes and run the rank's kernels; every collective is skipped, so a step's results are meaningless and only its
 * duration (per-rank compute time of a tensor-parallel step) is of use.  bench.py reports it as projected_not_measured. */
int nvllm_ctx_create_null_comm(int device_ordinal, int tp_rank, int tp_size, nvllm_ctx** out);

/* HIP-event timing of one kernel class on the library stream (bench.py's roofline leg):
 * kind 0 off, 1 paged attention, 2 layer GEMMs, 3 add+RMSNorm, 4 qk-norm/RoPE/KV-write, 5 SwiGLU, 6 LM head,
 * 7 calibration (an event pair around NO launch, recorded where the decode attention launch sits: the elapsed
 * time of an empty bracket between busy kernels, to be subtracted from the bracketed durations).
 * While a kind is set every launch of that class is bracketed by two events; read returns the summed
 * elapsed ms and the number of launches since the last read. */
int nvllm_profile_kernel(nvllm_model* m, int kind);
int nvllm_profile_read(nvllm_model* m, double* total_ms, int64_t* launches);
/* Per-model options (A/B runs, tests, and two deployment knobs).  The library reads no environment variable.
 *   "kv_v_bits" (16 | 24; takes effect at the next nvllm_kv_alloc): 24 keeps V as f16 + a bf8 residual byte (13..14 bits;
 *       V bytes x1.5; head_dim 128 only) -- the knob for holding the 1e-3 logits bar on heavy-tailed checkpoints;
 *   "kv_k_bits" (16 | 24; at the next nvllm_kv_alloc, which refuses it without "kv_v_bits" = 24): the same for K;
 *   "oneshot_allreduce" (0 | 1; at the next nvllm_kv_alloc, every rank alike): one-shot all-reduce for decode messages;
 *       "oneshot_spins": bound of its wait kernel's poll; "oneshot_skip_push": test hook (this rank skips N pushes);
 *   "stream_combine" (default 0): let the fused forward use the streaming GEMM's in-launch split-K combine epilogues (fewer
 *       launches; measured slower on MI355X than slabs + a consumer launch);
 *   "tile_min_wgs" (default 192; 0 never, 1 whenever the shape fits), "tile_fuse_qk" (default 1): the prefill tile GEMM;
 *   "no_fused", "no_xpack", "no_rowpar", "no_attn_prologue" (default 0): force the generic decode path / row-major planes /
 *       no whole-K row-parallel GEMMs / q,k-norm + RoPE + KV write in their own row kernel. */
int nvllm_debug_set_option(nvllm_model* m, const char* name, int value);

/* counters for tests: "oneshot_calls" = all-reduces run on the one-shot device path by this model's context;
 * "tile_gemm_launches" = projections run by the prefill tile GEMM; "kv_f16_saturated" = elements of the K/V pool at the
 * f16 clamp (every cache write saturates at +-65504; a scan of the pool, off the hot path); "kv_f16_absmax_bits" = the
 * largest magnitude the pool holds, as f16 bits */
int nvllm_debug_get_counter(nvllm_model* m, const char* name, int64_t* value);

/* Diagnostic build only (make -C nano-vllm-candle_amd/csrc stamps -> libnvllm_amd_stamps.so, select it with NVLLM_LIB):
 * in-kernel time stamps (100 MHz constant clock) of every launch of the fused decode path.  nvllm_debug_stamps arms /
 * disarms the recording; nvllm_debug_stamps_read copies launch `launch` of the last step, [1024 workgroups][16 waves][8
 * points] u64, 0 = not written (131072 u64).  The product library returns NVLLM_ESTATE: no stamp code is compiled in. */
int nvllm_debug_stamps(nvllm_model* m, int enable);
int nvllm_debug_stamps_read(nvllm_model* m, int launch, uint64_t* out, int64_t capacity_u64);
/* copy per-layer taps of the last step to the host (debug/parity): what = 0 layer output h,
 * 1 residual; [rows, hidden] f32 of layer `layer`; rows = rows of the last step's last chunk */
int nvllm_debug_layer_tap(nvllm_model* m, int layer, int what, float* out, int64_t capacity_floats);
int nvllm_debug_enable_taps(nvllm_model* m, int enable);

/* device generator access for tests, any profile (nvllm_op_synth_bf16 is profile 0): kind 0 matrix / 1 norm / 2 q,k-norm;
 * axis 0 none / 1 hidden channel = idx mod cols / 2 = idx / cols; outlier channels derive from (seed, hidden_size) */
int nvllm_debug_synth_bf16_spec(nvllm_ctx* ctx, const char* name, uint64_t seed, int kind, int profile, int axis, int64_t cols,
                                int hidden_size, int64_t first, int64_t count, uint16_t* host_out);

/* tuning aid: time one decomposition (n-tiles per wave, waves per workgroup, K splits; 0 = planner's choice)
 * of y[M,N] = x[M,K].W^T on synthetic operands; returns microseconds per launch */
int nvllm_debug_gemm_bench(nvllm_ctx* ctx, int M, int N, int K, int nt, int nw, int n_split, int iters,
                           float* us_per_call);

/* tuning aid v2: explicit m-tiles per workgroup, epilogue mode (0 slabs, 2 SwiGLU with N = 2I) and `rot` weight
 * copies cycled per launch (cold HBM like the model; 1 = cache-warm) */
int nvllm_debug_gemm_bench2(nvllm_ctx* ctx, int M, int N, int K, int mt, int nt, int nw, int n_split, int mode, int rot,
                            int iters, float* us_per_call);

/* parity + timing of the prefill tile GEMM (256-row x 256/192-feature workgroup tiles, both operands in fragment order)
 * against the chunked kernel on the same synthetic operands.  mode 0: f32 output; 2: SwiGLU planes (N = 2I), compared as
 * hi + lo; act_packed: the tile kernel writes its planes in fragment order. */
int nvllm_debug_gemm_tile_check(nvllm_ctx* ctx, int M, int N, int K, int mode, int act_packed, int iters,
                                float* max_abs_diff, float* max_abs_ref, float* us_chunked, float* us_tile);

/* tuning aid: time the decode attention (one new token per sequence, ctx_lens[B] cached tokens each) on a
 * synthetic cache; part_tokens > 0 splits every context into workgroups of that many tokens */
int nvllm_debug_attn_bench(nvllm_ctx* ctx, int B, int nh, int kv, int hd, const int32_t* ctx_lens, int part_tokens,
                           int iters, float* us_per_call);

/* debug: XCC_ID (which of the 8 XCDs) every workgroup of a (gx,gy,gz) grid lands on; out[linear workgroup id] */
int nvllm_debug_xcc_map(nvllm_ctx* ctx, int gx, int gy, int gz, int threads, int32_t* out);

#ifdef __cplusplus
}
#endif
#endif /* NVLLM_AMD_DEBUG_H */

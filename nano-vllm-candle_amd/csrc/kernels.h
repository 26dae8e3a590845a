// Launchers of the hand-written gfx950 kernels (kernels.hip).  Host C++; every launcher enqueues on
// the given stream and returns hipGetLastError().  Data layouts are documented in DESIGN.md §3.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "synth_device.h"

namespace nvllm {

typedef uint16_t bf16_bits;  // raw bf16 storage
typedef uint16_t f16_bits;   // raw f16 storage

constexpr int kBlockTokens = 256;  // KV block size, fixed by the reference (src/engine/sequence.rs:35)

// ---- packed weight: MFMA 16x16x32 A-fragment tiles -------------------------------------------
// W[N][K] (row-major, bf16) is stored as [N/16][K/32][64 lanes][8 bf16]; lane l of tile (nt,kt)
// holds W[nt*16 + (l&15)][kt*32 + 8*(l>>4) + 0..7].  One wave-load = one contiguous 1 KiB tile.
struct PackedW {
    uint4* data = nullptr;
    int N = 0, K = 0;  // logical sizes (N % 16 == 0, K % 32 == 0)
    size_t bytes() const { return (size_t)N * K * 2; }
};

// pack rows [row0, row0+rows) of dst from a row-major bf16 source (ld = elements between rows)
// ileave: -1 = rows land at row0.. ; 0/1 = gate(0)/up(1) rows interleaved in 16-row tiles (SwiGLU epilogue layout)
hipError_t launch_pack_rows(const PackedW& dst, int row0, int rows, const bf16_bits* src, int64_t ld, int ileave,
                            hipStream_t s);
// generate rows [row0,row0+rows) of dst from the synthetic tensor `spec` (synth_device.h): logical element
// (r, k) of the destination = source element (src_row0 + r, src_col0 + k) of a [*, src_ld] tensor
hipError_t launch_synth_packed(const PackedW& dst, int row0, int rows, const SynthSpec& spec, int64_t src_row0,
                               int64_t src_col0, int64_t src_ld, int ileave, hipStream_t s);
// row-major synthetic fill: dst bf16 [count] = elements [first, first+count)
hipError_t launch_synth_rowmajor_bf16(bf16_bits* dst, const SynthSpec& spec, int64_t first, int64_t count, hipStream_t s);
hipError_t launch_synth_rowmajor_f32(float* dst, const SynthSpec& spec, int64_t first, int64_t count, hipStream_t s);

// debug scan of an f16 buffer (n % 8 == 0): *sat += elements at or above the f16_sat clamp (|x| >= 65504), *absmax_bits =
// max(*absmax_bits, largest magnitude as f16 bits)
hipError_t launch_f16_scan(const f16_bits* data, int64_t n, unsigned long long* sat, unsigned* absmax_bits, hipStream_t s);

// ---- conversions ---------------------------------------------------------------------------------
hipError_t launch_f32_to_bf16(const float* src, bf16_bits* dst, int64_t n, hipStream_t s);
hipError_t launch_bf16_to_f32(const bf16_bits* src, float* dst, int64_t n, hipStream_t s);
// x f32 [rows][n] -> hi/lo bf16 planes (hi = bf16(x), lo = bf16(x - hi))
hipError_t launch_split_hilo(const float* x, bf16_bits* hi, bf16_bits* lo, int64_t n, hipStream_t s);

// ---- deferred RMSNorm (fused decode path) -------------------------------------------------------------
// A row-parallel GEMM epilogue (gemm_rowpar) writes the new residual s, x' = w_next (.) s (the NEXT norm's weight
// folded in, but NOT the 1/rms factor) and per-row partial sums of squares ssq[group][row].  Because the
// consumer is linear in x, it applies rinv[row] = 1/sqrt(sum_g ssq[g][row]/H + eps) to its OUTPUT instead:
//   norm(s).W^T = rinv * ((w (.) s).W^T)          (layernorm.rs:55-57 semantics, f32)
// ssq == nullptr means the activations are already normalised (rinv = 1).
// Packed activation layout (MFMA B-fragment order, one 1 KiB wave-load per 16 rows x 32 k):
//   [rows/16][K/32][64 lanes][8] bf16, lane = ((k % 32) / 8) * 16 + row % 16.   KT = K / 32.
// The fused decode path keeps its hi/lo activation planes in this order when every consumer reads fragments
// (row-major planes cost the register-direct GEMMs 3x per byte: 16 half-used lines per wave-load).
__host__ __device__ inline size_t xpack_off(int row, int k, int KT) {
    return (((size_t)(row >> 4) * KT + (k >> 5)) << 9) + (size_t)(((((k & 31) >> 3) << 4) + (row & 15)) << 3) + (k & 7);
}

struct RowNorm {
    const float* ssq = nullptr;  // [groups][stride]
    int groups = 0;
    int stride = 0;              // rows allocated per group
    float inv_h = 0.f;           // 1 / hidden_size
    float eps = 0.f;
    const int* row_idx = nullptr;  // GEMM only: x row (and ssq row) of output row r
};

// ---- GEMM: out[ks][M][N] = x[M][K-slice ks] . W[N][K-slice ks]^T  (f32 slabs, ks < n_split) ------
struct GemmPlan {
    int mt, nt, nw, kc;  // m-tiles per WG, n-tiles per wave, waves per WG, k-tiles per LDS chunk
    int n_split;         // K splits across workgroups (slabs)
    int kt_per_split;
    int lm_nt = 0;       // > 0: launch_gemm_argmax runs the streaming LM-head kernel with lm_nt n-tiles per wave
};
GemmPlan plan_gemm(int M, int N, int K, int max_split);  // K % 128 == 0 required
// streaming GEMM for big decode matrices (>= 24 MB, 17..64 rows): f32 slabs out[gemm_stream_splits][M][N];
// gemm_stream_splits == 0 -> shape not covered (use launch_gemm / launch_gemm_rowpar)
int gemm_stream_splits(int M, int N, int K);
// x_packed: the planes are in xpack_off order (ldx = K either way)
hipError_t launch_gemm_stream(const bf16_bits* xh, const bf16_bits* xl, int ldx, const PackedW& w, float* out, int M, int x_packed,
                              hipStream_t s);
// The same kernel with an epilogue on COMPLETE sums (stream_gemm.hip): epi 1 plain f32 [M][N], 2 SwiGLU (w = interleaved
// gate/up, act hi/lo [M][N/2], input norm rn deferred), 3 residual + next-norm prep (as gemm_rowpar epilogue 0).  K slices
// combine inside the launch through `slabs` ([gemm_stream_slab_floats] f32 scratch) and `tickets` ([n-groups] unsigned,
// zero between launches).  Any matrix size; 17..64 rows.
struct StreamArgs {
    const bf16_bits* xh = nullptr;
    const bf16_bits* xl = nullptr;
    int ldx = 0, x_packed = 0, M = 0;
    float* slabs = nullptr;
    unsigned* tickets = nullptr;
    float* out = nullptr;                                  // epi 1
    RowNorm rn;                                            // epi 2
    bf16_bits* oh = nullptr;                               // epi 2: act planes; epi 3: x' planes
    bf16_bits* ol = nullptr;
    int o_packed = 0;
    const float* resid_in = nullptr;                       // epi 3
    float* resid_out = nullptr;
    const float* next_w = nullptr;
    float* ssq = nullptr;                                  // [gemm_stream_groups][ssq_stride]
    int ssq_stride = 0;
};
bool gemm_stream_ok(int M, int N, int K, int epi);
int gemm_stream_groups(int M, int N, int K, int epi);          // n-groups = tickets needed = ssq groups of epilogue 3
size_t gemm_stream_slab_floats(int M, int N, int K, int epi);  // f32 scratch the combine needs
hipError_t launch_gemm_stream_epi(const StreamArgs& a, const PackedW& w, int epi, hipStream_t s);
// LM head (greedy arg-max epilogue): the streaming kernel for <= 64 rows, else plan_gemm(M, N, K, 1)
GemmPlan plan_lmhead(int M, int N, int K);
void set_split(GemmPlan& p, int KT, int want);
hipError_t launch_gemm(const GemmPlan& p, const bf16_bits* xh, const bf16_bits* xl, int ldx, const PackedW& w,
                       float* out, int M, hipStream_t s);
// same GEMM (n_split must be 1) whose epilogue also emits per-wave partial arg-max (LAST max wins):
// part_val/part_idx [gemm_argmax_parts(p, N)][M]; `out` may be nullptr (ids only, logits never stored).
// rn (nullable): deferred RMSNorm of the input rows, applied to the sums (RowNorm)
hipError_t launch_gemm_argmax(const GemmPlan& p, const bf16_bits* xh, const bf16_bits* xl, int ldx, const PackedW& w,
                              float* out, int M, float* part_val, int* part_idx, const RowNorm* rn, hipStream_t s);
int gemm_argmax_parts(const GemmPlan& p, int N);
// gate/up projection with the SiLU*mul epilogue (weight interleaved by launch_pack_rows(..., ileave))
GemmPlan plan_gemm_swiglu(int M, int N2, int K);
hipError_t launch_gemm_swiglu(const GemmPlan& p, const bf16_bits* xh, const bf16_bits* xl, int ldx, const PackedW& w,
                              int M, bf16_bits* act_hi, bf16_bits* act_lo, const RowNorm* rn, hipStream_t s);
// scratch: 8*(M+1) bytes, zero before the first use (the kernel leaves it zero again)
// finish of the streaming LM head (plan.lm_nt > 0): partials are [row][n_parts], one workgroup per row
hipError_t launch_argmax_rows(const float* part_val, const int* part_idx, int n_parts, int M, uint32_t* ids, float* maxval, hipStream_t s);
hipError_t launch_argmax_parts(const float* part_val, const int* part_idx, int n_parts, int M, void* scratch,
                               uint32_t* ids, float* maxval, hipStream_t s);

// ---- (embed +) add + RMSNorm ----------------------------------------------------------------------
struct NormArgs {
    const float* in = nullptr;        // [n_slabs][rows_in][H] f32 (slab sum) ; or nullptr with ids/embed
    int n_slabs = 1;
    int64_t slab_stride = 0;          // floats between slabs
    const uint32_t* ids = nullptr;    // embedding mode: token id per row
    const bf16_bits* embed = nullptr; // [V][H] bf16
    const float* residual_in = nullptr;
    float* residual_out = nullptr;    // may alias residual_in
    const int* row_idx = nullptr;     // gather: input row of output row r (outputs are compact)
    const float* weight = nullptr;
    float eps = 1e-6f;
    int H = 0;
    bf16_bits* xh = nullptr;          // outputs (nullable individually)
    bf16_bits* xl = nullptr;
    float* y = nullptr;
    float* ssq_out = nullptr;         // prep mode (fused decode path): outputs are w (.) s (no 1/rms) and ssq_out[r] = sum s^2
    int out_packed = 0;               // xh/xl in xpack_off order
};
hipError_t launch_add_rmsnorm(const NormArgs& a, int rows, hipStream_t s);

// ---- q/k norm + RoPE + KV cache write -------------------------------------------------------------
struct KvLayout {
    f16_bits* k = nullptr;  // [num_blocks][kv_l][16 tiles][hd/32][64 lanes][8] f16 (QK^T A-fragment packed)
    f16_bits* v = nullptr;  // [num_blocks][kv_l][8 tiles][hd/16][64 lanes][8] f16 (PV A-fragment packed)
    uint8_t* vlo = nullptr; // optional 24-bit V: e5m2 rounding residual of every V element, same element order (1 byte each)
    uint8_t* klo = nullptr; // optional 24-bit K (only together with vlo): the same for K (device_common.h: klo_packed_offset)
    int kv_l = 0, hd = 0;
};
struct QkvArgs {
    const float* qkv = nullptr;  // [n_slabs][rows][(nh_l+2kv_l)*hd]
    int n_slabs = 1;
    int64_t slab_stride = 0;
    const float* qn = nullptr;   // [hd]
    const float* kn = nullptr;
    float eps = 1e-6f;
    const float* cos = nullptr;  // [max_pos][hd/2]
    const float* sin = nullptr;
    const int* pos = nullptr;    // [rows]
    const int* slot = nullptr;   // [rows] row -> sequence slot
    const int* block_tables = nullptr;
    int max_blocks = 0;
    int nh_l = 0;
    float q_scale = 1.f;         // folded into q: head_dim^-0.5 * log2(e)
    float* q_out = nullptr;      // [rows][nh_l*hd] f32
    KvLayout kv;
    RowNorm rn;                  // deferred input norm (qkv sums are multiplied by rinv[row] first)
};
hipError_t launch_qk_norm_rope_kvwrite(const QkvArgs& a, int rows, hipStream_t s);

// ---- prefill GEMM (tile_gemm.hip): 256 x 256/192 output tile, both operands in fragment order -------------------
// x planes in xpack_off order; mode 0: out f32 [M][N]; mode 2: w = interleaved gate/up, act hi/lo [M][N/2] (row-major or
// xpack_off order).  min_wgs: smallest grid the kernel is used for (a 256-row tile needs many rows to fill 256 CUs).
// hipErrorNotSupported = shape not covered.
bool gemm_tile_ok(int M, int N, int K, int mode, int min_wgs);
// mode 0 with n_slabs != nullptr may split K (narrow outputs, 128-row blocks): out = [*n_slabs][M][N], up to max_split slabs
int gemm_tile_splits(int M, int N, int K, int min_wgs, int max_split);  // 0 = shape not covered
hipError_t launch_gemm_tile(const bf16_bits* xh, const bf16_bits* xl, const PackedW& w, int M, int mode, float* out,
                            bf16_bits* act_hi, bf16_bits* act_lo, int act_packed, int min_wgs, int max_split, int* n_slabs,
                            hipStream_t s);
// mode 3 = the QKV projection of a prompt chunk with q/k-norm + RoPE + KV-cache write + q output in the epilogue
// (qwen3.rs:205-234): needs 256-wide blocks and head_dim 128 (one wave tile = one head); qa.qkv is ignored.
bool gemm_tile_qkv_ok(int M, int N, int K, int hd, int min_wgs);
hipError_t launch_gemm_tile_qkv(const bf16_bits* xh, const bf16_bits* xl, const PackedW& w, int M, const QkvArgs& qa, int min_wgs,
                                hipStream_t s);
#ifdef NVLLM_STAMPS
void tile_gemm_stamps_arm(unsigned long long* base, int max_launches);  // diagnostic build
int tile_gemm_stamps_count();
void stream_gemm_set_ablate(int v);  // diagnostic build: ablation variant of the next streaming GEMM launches (stream_gemm.hip ABL)
#endif
hipError_t launch_xpack_plane(const bf16_bits* src, bf16_bits* dst, int M, int K, hipStream_t s);

// ---- one-shot all-reduce for TP decode (oneshot.hip) -----------------------------------------------------------
// every rank's buffer: data [2 generations][tp][slot_floats] f32, flag [2][tp] u32; data[q] / flag[q] = rank q's buffers
// as mapped into THIS process
struct OneShotPeers {
    float* data[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    uint32_t* flag[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};
hipError_t launch_oneshot_push(const float* src, size_t n, const OneShotPeers& p, int tp, int rank, size_t slot_floats, int gen,
                               uint32_t seq, unsigned* done, hipStream_t s);
// max_spins bounds the poll (each spin sleeps ~64 cycles): a flag that never arrives sets *err instead of hanging the stream
hipError_t launch_oneshot_wait(const uint32_t* flags, int tp, int gen, uint32_t seq, int* err, long long max_spins, hipStream_t s);

// ---- paged attention (prefill tiles and decode rows alike) ----------------------------------------
struct AttnArgs {
    const float* q = nullptr;          // [rows][nh_l*hd], already scaled by q_scale
    KvLayout kv;
    const int* block_tables = nullptr;
    int max_blocks = 0;
    const int* tile_row0 = nullptr;    // per q-tile: first row, #rows, sequence slot
    const int* tile_nrows = nullptr;
    const int* tile_slot = nullptr;
    const int* group_order = nullptr;  // optional (prefill kernel): tile groups (4 q-tiles) sorted longest context first; the
                                       // workgroups beyond the resident set then are the short ones, not whatever came last
    const int* tile_last = nullptr;    // optional (prefill kernel): position of the tile's last row, -1 for an empty tile;
                                       // saves the kernel a chain of dependent metadata loads per workgroup
    const int* pos = nullptr;          // [rows]
    // load balance (decode): tile_order[rank] = tile index, longest context first; a grid row (one kv head) walks it
    // forwards when the row starts in an even 256-workgroup round and backwards when in an odd one, so that co-resident
    // workgroups (ids i and i+256 share a CU when two fit) pair a long sequence with a short one.  The direction is per
    // ROW: every row must visit every tile exactly once whatever the batch size.  nullptr = launch order.
    const int* tile_order = nullptr;
    int nh_l = 0, gqa = 1;
    bf16_bits* out_hi = nullptr;       // [rows][nh_l*hd]
    bf16_bits* out_lo = nullptr;
    float* out_f32 = nullptr;          // optional f32 copy (fine-seam op)
    int out_packed = 0;                // out_hi/out_lo in xpack_off order
    // fused decode prologue (qkv != nullptr; every tile must be a single row): the kernel sums the QKV GEMM's
    // slabs itself, applies q/k RMSNorm + RoPE, writes the row's K/V into the cache, then attends
    const float* qkv = nullptr;        // [n_slabs][rows][ldqkv]
    int n_slabs = 1, ldqkv = 0;
    int64_t slab_stride = 0;
    const float* qn = nullptr;
    const float* kn = nullptr;
    const float* cos = nullptr;
    const float* sin = nullptr;
    float eps = 1e-6f, q_scale = 1.f;
    RowNorm rn;                        // deferred input norm for the fused prologue
    // split-KV (decode): each workgroup covers part_tiles 32-token tiles; partials merged by a combine pass
    int part_tiles = 0, max_parts = 0;
    float* part_o = nullptr;           // [rows][nh_l][max_parts][hd]
    float* part_ml = nullptr;          // [rows][nh_l][max_parts][2]
#ifdef NVLLM_STAMPS
    unsigned long long* stamps = nullptr;  // diagnostic build only: [workgroup][16 waves][8] s_memrealtime stamps
#endif
};
// qt = q sub-tiles (of 16 MFMA rows) per workgroup: 1 (decode) or 2 (prefill)
hipError_t launch_attn_paged(const AttnArgs& a, int n_tiles, int qt, int rows, int n_parts_max, hipStream_t s);
inline int attn_tokens_per_tile(int gqa, int qt) { return (16 / gqa) * qt; }
// prefill form with LDS-staged K/V shared by four q-tiles (qt = 2) of one sequence: n_tiles is a multiple of 4, tiles
// 4b..4b+3 belong to one sequence (tile_slot[4b]), a sequence's last group is padded with empty tiles (tile_nrows = 0)
constexpr int kPrefillTileGroup = 4;
hipError_t launch_attn_prefill(const AttnArgs& a, int n_tiles, hipStream_t s);

// ---- row-parallel projection with residual + next-norm epilogue (decode, tp == 1) ---------------------
// resid_out = resid_in + x.W^T ; x' = next_w (.) resid_out as bf16 hi/lo ; ssq[group][row] partial sums of squares.
// One workgroup owns 16 rows x (NWN*16) features for the WHOLE K (waves split K, reduced through LDS), all loads
// issued up front.  Returns hipErrorNotSupported when (K, N) has no supported decomposition.
struct RowParArgs {
    const bf16_bits* xh = nullptr;
    const bf16_bits* xl = nullptr;
    int ldx = 0;
    const float* resid_in = nullptr;
    float* resid_out = nullptr;
    const float* next_w = nullptr;
    bf16_bits* oh = nullptr;
    bf16_bits* ol = nullptr;
    float* ssq = nullptr;  // [groups][ssq_stride]
    int ssq_stride = 0;
    int M = 0;
    // epilogue 1 (SwiGLU): w = interleaved gate/up, writes act hi/lo [M][N/2] (oh/ol), input norm rn deferred
    // epilogue 2 (plain): out f32 [M][N]
    float* out = nullptr;
    RowNorm rn;
    int x_packed = 0, o_packed = 0;  // xh/xl resp. oh/ol in xpack_off order (register-direct kernel only)
#ifdef NVLLM_STAMPS
    unsigned long long* stamps = nullptr;  // diagnostic build only: [workgroup][16 waves][8] s_memrealtime stamps
#endif
};
bool gemm_rowpar_supported(int N, int K);
// true when launch_gemm_rowpar would run the register-direct kernel (the one that accepts packed planes)
bool gemm_rowdir_ok(int N, int K, int epi, int M);
bool gemm_rowpar_ok(int N, int K, int epi, int M);
int gemm_rowpar_splits(int N, int K, int epi, int M);  // f32 slabs the plain epilogue leaves (1 = complete sums)
int gemm_rowpar_groups(int N, int K);
// epi: 0 residual + next-norm prep, 1 SwiGLU, 2 plain f32 output
hipError_t launch_gemm_rowpar(const RowParArgs& a, const PackedW& w, int epi, hipStream_t s);

// ---- SwiGLU -----------------------------------------------------------------------------------------
// gu [n_slabs][rows][2*I] -> act hi/lo [rows][I] (and/or f32 y)
hipError_t launch_silu_mul(const float* gu, int n_slabs, int64_t slab_stride, int rows, int I, bf16_bits* hi,
                           bf16_bits* lo, float* y, hipStream_t s);

// gu in the packed weight's interleaved 16-row-tile order (gate tile, up tile, ...) -> act hi/lo [rows][I]
// out_packed: hi/lo in xpack_off order (the streaming GEMM's input)
hipError_t launch_silu_mul_interleaved(const float* gu, int n_slabs, int64_t slab_stride, int rows, int I, bf16_bits* hi,
                                       bf16_bits* lo, int out_packed, hipStream_t s);

// ---- misc --------------------------------------------------------------------------------------------
// y[r][:] = sum_s in[s][r][:] (+ bias)  -- finishes a split-K GEMM for the fine-seam op
hipError_t launch_slab_sum(const float* in, int n_slabs, int64_t slab_stride, const float* bias, int rows, int N,
                           float* y, hipStream_t s);
hipError_t launch_slab_sum_ld(const float* in, int n_slabs, int64_t slab_stride, int64_t ld_in, const float* bias,
                              int rows, int N, float* y, int64_t ld_out, hipStream_t s);
// generic-shape fallbacks used only by the fine-seam ops
hipError_t launch_rmsnorm_generic(const float* x, const float* res, const float* w, float eps, int rows, int n, float* y,
                                  float* res_out, hipStream_t s);
hipError_t launch_silu_mul_generic(const float* x, int rows, int n, float* y, hipStream_t s);
hipError_t launch_split_hilo_pad(const float* x, int rows, int K, int Kpad, bf16_bits* hi, bf16_bits* lo, hipStream_t s);
// argmax with LAST-max tie rule; idx_offset added to the result (vocab-parallel shards)
hipError_t launch_argmax(const float* logits, int rows, int V, int64_t ld, uint32_t* ids, float* maxval,
                         hipStream_t s);
// temperature sampling, one row each (llm_engine.rs:97-133): ids[r] = Gumbel-max draw from softmax(logits[r] / max(temps[r], 1e-6)),
// arg-max (last max) when the weights do not form a distribution; keys[r] seeds row r's counter RNG; idx_offset = first
// global vocabulary id of this shard (the random stream is indexed by global id); best_score (nullable): the winning score
hipError_t launch_sample_rows(const float* logits, int rows, int V, int64_t ld, const float* temps, const uint64_t* keys, int idx_offset,
                              uint32_t* ids, float* best_score, hipStream_t s);
hipError_t launch_embedding_f32(const float* table, const uint32_t* ids, int n, int V, int H, float* y,
                                hipStream_t s);
// fine-seam RoPE on [B,heads,T,hd] f32 in place, positions 0..T
hipError_t launch_rope_bhtd(float* x, int B, int heads, int T, int hd, const float* cos, const float* sin,
                            hipStream_t s);
// [B,heads,T,hd] -> [B*T][heads*hd] (optionally scaled)
hipError_t launch_bhtd_to_rows(const float* x, int B, int heads, int T, int hd, float scale, float* y,
                               hipStream_t s);
// write k,v rows ([rows][kv*hd] f32) into the paged cache at (slot,pos) without norm/rope
hipError_t launch_kv_write_plain(const float* k, const float* v, int rows, const int* pos, const int* slot,
                                 const int* block_tables, int max_blocks, KvLayout kv, hipStream_t s);
struct PrefetchRange { const void* ptr; size_t bytes; };
// touch every 128-byte line of the ranges (device array) so they are cache-resident for the next consumer
hipError_t launch_prefetch_ranges(const PrefetchRange* d_ranges, int n_ranges, unsigned* sink, int blocks_x, hipStream_t s);
// debug: XCC_ID of every workgroup of a (gx,gy,gz) grid, out[linear workgroup id]
hipError_t launch_xcc_map(int gx, int gy, int gz, int threads, int* out, hipStream_t s);
// token feedback for nvllm_decode_next: ids[i] = next[i]; pos[i] += 1
hipError_t launch_advance_decode(uint32_t* ids, const uint32_t* next, int* pos, int n, hipStream_t s);

}  // namespace nvllm

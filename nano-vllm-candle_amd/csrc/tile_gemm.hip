// Prefill GEMM for gfx950: out[M][N] = x[M][K] . W[N][K]^T with x as bf16 hi + lo planes
// (replaces candle_nn::Linear::forward on prompt chunks: src/layers/linear.rs:35-36,72-77,184-198 via
//  src/models/qwen3.rs:205 (QKV), :278 (o_proj), :324-326 (gate/up + SiluAndMul, down)).
//
// MFMA-bound shape (hundreds to thousands of rows), so the design is about operand reuse and keeping the matrix
// pipe fed, not about HBM:
//   * one workgroup = 256 rows x 32*NTW features, 8 waves as 4 (rows) x 2 (features): a wave owns 64 rows x 16*NTW
//     features = 4 x NTW accumulator tiles (128 registers at NTW = 8), two waves per SIMD;
//   * BOTH operands are already in MFMA fragment order in HBM (W: PackedW; x: xpack_off planes written by the
//     producing kernel), so every stage is plain 1 KiB LDS-DMA wave-copies into a lane-linear image: no swizzle,
//     no ds_write, no bank conflicts, and each W fragment read from LDS feeds 8 MFMAs (4 row tiles x hi/lo), each
//     x fragment NTW;
//   * K advances one 32-deep k-tile per stage through a ring of three LDS stages: the DMA of stage s+2 is issued
//     right after the barrier that opens stage s, so it has a whole stage of MFMAs (64 per wave) to land; ONE
//     barrier per stage.  Inside a stage the fragments of the next sub-step (and, in the last one, the first
//     fragments of stage s+1, which the opening barrier already published) are read while the current MFMAs run.
//   * workgroups that share an XCD (blockIdx % 8) share their x rows or their W features (host picks), so the
//     operand re-reads hit that XCD's L2.
// MODE 0: f32 [M][N].  MODE 2: W is the gate/up matrix interleaved in 16-row tiles; the epilogue writes
// silu(gate)*up as hi/lo planes [M][N/2] (row-major or xpack_off order) -- SiluAndMul, activation.rs:13-18.
// MODE 3 (NTW = 8, head_dim 128): W is the fused QKV matrix, so a wave's 64 rows x 128 features are 64 tokens of ONE
// head and the epilogue is the whole of qwen3.rs:208-234 on registers: RMS norm over the head (q_norm / k_norm BEFORE
// RoPE), half-split RoPE (rotary_embedding.rs:82-107: features d and d+64 sit in the same lane, accumulators j and j+4),
// then q (scaled) to the f32 q buffer, K and V as f16 into the paged cache in their fragment orders.  No f32 QKV
// round trip through HBM and no separate row kernel.
#include <algorithm>

#include "device_common.h"
#include "kernels.h"

namespace nvllm {

struct TileArgs {
    const uint4* xh = nullptr;  // packed planes [ceil(M/16)][KT][64] uint4
    const uint4* xl = nullptr;
    const uint4* wp = nullptr;  // [N/16][KT][64] uint4
    float* out = nullptr;       // MODE 0
    uint16_t* act_hi = nullptr; // MODE 2
    uint16_t* act_lo = nullptr;
    int act_packed = 0;
    int M = 0, N = 0, KT = 0;
    int kts = 0;                // k-tiles per K split (blockIdx.y); split y writes slab out + y*M*N (MODE 0 only)
    int MB = 0, NB = 0;         // row blocks, feature blocks
    int map = 0;                // 0: id -> (nb fastest); 1: blocks of one XCD share rows; 2: share features
    QkvArgs q;                  // MODE 3
#ifdef NVLLM_STAMPS
    unsigned long long* stamps = nullptr;  // diagnostic build: [workgroup][16 waves][8 points] of s_memrealtime
#endif
};

#ifdef NVLLM_STAMPS
// diagnostic build only: the next tile launches record their stamps here, one kStride block per launch (tools/stamp_tile_gemm.py)
static unsigned long long* g_tile_stamps = nullptr;
static int g_tile_stamp_launch = 0, g_tile_stamp_max = 0;
void tile_gemm_stamps_arm(unsigned long long* base, int max_launches) { g_tile_stamps = base; g_tile_stamp_launch = 0; g_tile_stamp_max = max_launches; }
int tile_gemm_stamps_count() { return g_tile_stamp_launch; }
#endif

// WN = waves along the features (2: 256 rows x 32*NTW features; 4: 128 rows x 64*NTW features -- the narrow-output shape,
// N = hidden size, where 256-row tiles cannot fill the chip)
template <int NTW, int WN, int MODE>
__global__ void __launch_bounds__(512) gemm_tile_kernel(TileArgs a) {
    constexpr int WM = 8 / WN;             // waves along the rows, 4 row tiles (64 rows) each
    constexpr int XT = WM * 4;             // row tiles per workgroup
    constexpr int FRW = WN * NTW;          // 1 KiB fragments per stage: W ...
    constexpr int FR = FRW + 2 * XT;       // ... then [plane][row tile] of x
    constexpr int NI = (FR + 7) / 8;       // DMA wave-copies per wave per stage
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint4* lds = reinterpret_cast<uint4*>(smem_raw);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // uniform: DMA sources / LDS targets stay scalar
    const int l15 = lane & 15, grp = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;
    const int KT = a.kts;                  // stages of this workgroup
    const int kt0 = blockIdx.y * a.kts;

    // workgroup -> (row block, feature block); blockIdx % 8 labels the workgroups that land on one XCD
    int mb, nb;
    {
        const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
        if (a.map == 1) { nb = slot % a.NB; mb = (slot / a.NB) * 8 + xcd; }
        else if (a.map == 2) { mb = slot % a.MB; nb = (slot / a.MB) * 8 + xcd; }
        else { nb = id % a.NB; mb = id / a.NB; }
        if (mb >= a.MB || nb >= a.NB) return;  // whole workgroup, before any barrier
    }
    const int mtiles = (a.M + 15) >> 4;
    NVLLM_STAMP(a, 0);
    // MODE 3: position and cache block of this lane's four tokens, fetched NOW (two dependent loads each) so that they
    // land under the K loop; in the epilogue they were 8 us of latency chain per workgroup (stamps)
    [[maybe_unused]] int e_pos[4], e_blk[4], e_slot0 = 0;
    [[maybe_unused]] bool e_one_seq = true;  // this lane's four tokens belong to one sequence
    if constexpr (MODE == 3) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int rowc = min(mb * (WM * 64) + wm * 64 + b * 16 + l15, a.M - 1);
            const int sl = a.q.slot[rowc];
            if (b == 0) e_slot0 = sl;
            e_one_seq = e_one_seq && sl == e_slot0;
            e_pos[b] = a.q.pos[rowc];
            e_blk[b] = a.q.block_tables[(size_t)sl * a.q.max_blocks + (e_pos[b] >> 8)];
        }
    }

    // DMA sources of this wave's copies at k-tile 0, without the lane part (advance: 64 uint4 per k-tile).  Row tiles
    // past the end are clamped to the last one: their products land in columns of D that are never stored.
    const uint4* src[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int f = min(wave + 8 * i, FR - 1);
        if (f < FRW) {
            // MODE 3: the block's two heads are nb and NB + nb, not neighbours: a V head (whose epilogue is scattered
            // 2-byte cache stores) then always shares its workgroup with a q head, and the V tail spreads over twice the CUs
            const int ntile = MODE == 3 ? (f < NTW ? nb : a.NB + nb) * NTW + f % NTW : nb * FRW + f;
            src[i] = a.wp + ((size_t)ntile * a.KT + kt0) * 64;
        } else {
            const int g = f - FRW, plane = g / XT, mt = min(mb * XT + (g % XT), mtiles - 1);
            src[i] = (plane ? a.xl : a.xh) + ((size_t)mt * a.KT + kt0) * 64;
        }
    }
    // k-tile s (clamped: the two issues past the end re-fetch the last k-tile into a stage nobody reads, so the
    // stage body has no branch) -> LDS stage `buf`
    auto issue = [&](int s, int buf) {
        const size_t koff = (size_t)min(s, KT - 1) * 64;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int f = wave + 8 * i;
            if (FR % 8 == 0 || f < FR)
                __builtin_amdgcn_global_load_lds((gptr_t)(src[i] + koff + lane), (lptr_t)(lds + (size_t)(buf * FR + f) * 64), 16, 0, 0);
        }
    };
    // LDS byte addresses this lane reads in the three stages: W fragments of this wave's features, x fragments of its
    // rows.  The fragment reads are inline asm with hand-counted lgkmcnt waits: left to hipcc (ROCm 7.2) every wait in
    // this loop came out as lgkmcnt(0), i.e. each sub-step also waited for the reads it had just issued for the next one.
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem_raw;
    uint32_t w_cur = lds0 + (uint32_t)((wn * NTW) * 64 + lane) * 16;
    uint32_t x_cur = lds0 + (uint32_t)((FRW + wm * 4) * 64 + lane) * 16;
    uint32_t w_nxt = w_cur + FR * 1024, x_nxt = x_cur + FR * 1024, w_fre = w_cur + 2 * FR * 1024, x_fre = x_cur + 2 * FR * 1024;
    int fre = 2, cur_i = 0, nxt_i = 1;  // stage indices (scalar) for the DMA targets
#define NVLLM_LDSR(DST_, ADDR_, OFF_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST_) : "v"(ADDR_), "n"(OFF_))
#define NVLLM_WF(DST_, P_, J_) NVLLM_LDSR(DST_, P_, (J_) * 1024)
#define NVLLM_XF(DST_, P_, PL_, MT_) NVLLM_LDSR(DST_, P_, ((PL_) * XT + (MT_)) * 1024)
#define NVLLM_LGKM(N_) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N_) : "memory")

    f32x4 acc[NTW][4];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[j][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // every wave waits for its own DMA, then the barrier publishes all of them (and says: the stage read two
    // barriers ago is free).  One asm statement: the compiler can neither split it nor move LDS reads across it.
    auto publish = [&]() { asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); };

    issue(0, 0);
    issue(1, 1);
    publish();
    NVLLM_STAMP(a, 1);
    bf16x8 WA[NTW], WB[NTW], xh0, xl0, xh1, xl1;
#pragma unroll
    for (int j = 0; j < NTW; ++j) NVLLM_WF(WA[j], w_cur, j);
    NVLLM_XF(xh0, x_cur, 0, 0);
    NVLLM_XF(xl0, x_cur, 1, 0);

#define NVLLM_SB __builtin_amdgcn_sched_barrier(0)
#define NVLLM_TILE_MFMA_HALF(W_, X_, MT_)                                                                    \
    do {                                                                                                     \
        _Pragma("unroll") for (int j = 0; j < NTW; ++j)                                                      \
            acc[j][MT_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W_[j], X_, acc[j][MT_], 0, 0, 0);          \
    } while (0)
    // One stage = k-tile s: Wc (registers) x the four row tiles of this wave; Wn <- W fragments of stage s+1.
    // Each sub-step: issue the reads of the NEXT sub-step, wait until only those are outstanding (LDS returns in
    // order), run this sub-step's 2*NTW MFMAs; the sched_barriers keep the MFMAs on their side of the waits.
    // The DMA of stage s+2 is issued between the hi and lo MFMA groups of the first sub-step (its scalar address
    // arithmetic issues in the shadow of the MFMAs instead of between the barrier and the first MFMA).
    constexpr int W1 = NTW / 3, W2 = 2 * NTW / 3;
#define NVLLM_TILE_STAGE(Wc_, Wn_, S_)                                                                       \
    do {                                                                                                     \
        publish();                                                                                           \
        NVLLM_XF(xh1, x_cur, 0, 1); NVLLM_XF(xl1, x_cur, 1, 1);                                              \
        _Pragma("unroll") for (int j = 0; j < W1; ++j) NVLLM_WF(Wn_[j], w_nxt, j);                           \
        NVLLM_LGKM(2 + W1);                                                                                  \
        NVLLM_SB;                                                                                            \
        NVLLM_TILE_MFMA_HALF(Wc_, xh0, 0);                                                                   \
        issue((S_) + 2, fre);                                                                                \
        NVLLM_TILE_MFMA_HALF(Wc_, xl0, 0);                                                                   \
        NVLLM_SB;                                                                                            \
        NVLLM_XF(xh0, x_cur, 0, 2); NVLLM_XF(xl0, x_cur, 1, 2);                                              \
        _Pragma("unroll") for (int j = W1; j < W2; ++j) NVLLM_WF(Wn_[j], w_nxt, j);                          \
        NVLLM_LGKM(2 + W2 - W1);                                                                             \
        NVLLM_SB;                                                                                            \
        NVLLM_TILE_MFMA_HALF(Wc_, xh1, 1); NVLLM_TILE_MFMA_HALF(Wc_, xl1, 1);                                \
        NVLLM_SB;                                                                                            \
        NVLLM_XF(xh1, x_cur, 0, 3); NVLLM_XF(xl1, x_cur, 1, 3);                                              \
        _Pragma("unroll") for (int j = W2; j < NTW; ++j) NVLLM_WF(Wn_[j], w_nxt, j);                         \
        NVLLM_LGKM(2 + NTW - W2);                                                                            \
        NVLLM_SB;                                                                                            \
        NVLLM_TILE_MFMA_HALF(Wc_, xh0, 2); NVLLM_TILE_MFMA_HALF(Wc_, xl0, 2);                                \
        NVLLM_SB;                                                                                            \
        NVLLM_XF(xh0, x_nxt, 0, 0); NVLLM_XF(xl0, x_nxt, 1, 0);                                              \
        NVLLM_LGKM(2);                                                                                       \
        NVLLM_SB;                                                                                            \
        NVLLM_TILE_MFMA_HALF(Wc_, xh1, 3); NVLLM_TILE_MFMA_HALF(Wc_, xl1, 3);                                \
        NVLLM_SB;                                                                                            \
        { const uint32_t t_ = w_cur; w_cur = w_nxt; w_nxt = w_fre; w_fre = t_; }                             \
        { const uint32_t t_ = x_cur; x_cur = x_nxt; x_nxt = x_fre; x_fre = t_; }                             \
        { const int t_ = cur_i; cur_i = nxt_i; nxt_i = fre; fre = t_; }                                      \
    } while (0)

    for (int s = 0; s < KT; s += 2) {  // KT is even (host)
        NVLLM_TILE_STAGE(WA, WB, s);
        NVLLM_TILE_STAGE(WB, WA, s + 1);
    }
#undef NVLLM_TILE_STAGE
#undef NVLLM_TILE_MFMA_HALF
#undef NVLLM_SB
#undef NVLLM_WF
#undef NVLLM_XF
#undef NVLLM_LDSR
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the two issues past the end, the last prefetch reads
#undef NVLLM_LGKM
    NVLLM_STAMP(a, 2);

    // D[feature 4*grp + r][token l15]
    const int nt0 = nb * FRW + wn * NTW;
    const int row0 = mb * (WM * 64) + wm * 64;
    if constexpr (MODE == 0) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int row = row0 + b * 16 + l15;
            if (row < a.M) {
                float* o = a.out + (size_t)blockIdx.y * a.M * a.N + (size_t)row * a.N + (size_t)nt0 * 16 + grp * 4;
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    const f32x4 v = acc[j][b];
                    *reinterpret_cast<float4*>(o + j * 16) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    } else if constexpr (MODE == 3) {
        static_assert(MODE != 3 || (NTW == 8 && WN == 2), "one wave tile = one 128-wide head");
        const QkvArgs& q = a.q;
        const int hh = wn == 0 ? nb : a.NB + nb;  // this wave's head: [q heads | k heads | v heads]
        const int nh = q.nh_l, kvl = q.kv.kv_l;
        // V heads: in the cache's PV fragment order TOKENS are the fast index (8 per 16-byte slot), while a lane here
        // holds 4 features of ONE token -- stored directly that is 128 two-byte stores per lane, and the V workgroups set
        // the kernel's end (stamps: epilogue 20 us against 13).  When the wave's 64 rows are consecutive positions of one
        // sequence (the usual case) the tile goes through LDS instead -- the ring is free now -- as Vt[feature][token +
        // pos % 4], so that every aligned group of four positions is one 8-byte word: 34 eight-byte stores per lane.
        asm volatile("s_barrier" ::: "memory");  // every wave has finished reading the ring (uniform: all modes-3 waves)
        bool v_done = false;
        if (hh >= nh + kvl) {
            const int nvalid = min(64, a.M - row0);
            const int P0 = __builtin_amdgcn_readfirstlane(e_pos[0]);
            bool okl = e_one_seq && e_slot0 == __builtin_amdgcn_readfirstlane(e_slot0);
#pragma unroll
            for (int b = 0; b < 4; ++b) okl = okl && (b * 16 + l15 >= nvalid || e_pos[b] == P0 + b * 16 + l15);
            if (nvalid > 0 && __builtin_amdgcn_ballot_w64(!okl) == 0) {
                constexpr int LDV = 72;  // halves per feature row: 64 tokens + 3 shift, padded to a multiple of 4
                _Float16* vt = reinterpret_cast<_Float16*>(smem_raw) + (size_t)wm * (128 * LDV);
                const int sh = P0 & 3;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float ri = rownorm_rinv(q.rn, min(row0 + b * 16 + l15, a.M - 1));
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) vt[(j * 16 + grp * 4 + r) * LDV + b * 16 + l15 + sh] = f16_sat(acc[j][b][r] * ri);
                }
                const int blkA = __builtin_amdgcn_readfirstlane(e_blk[0]);                       // block of position P0
                const int blkB = __builtin_amdgcn_readlane(e_blk[(nvalid - 1) >> 4], (nvalid - 1) & 15);  // ... of the last valid token
                const int kvh = hh - nh - kvl;
                for (int cq = 0; cq < 17; ++cq) {
                    const int tfirst = 4 * cq - sh;  // token of the word's first column
                    if (tfirst + 3 < 0 || tfirst >= nvalid) continue;
                    const int p = P0 + tfirst;       // multiple of 4
                    const int blk = (p >> 8) == (P0 >> 8) ? blkA : blkB;
                    _Float16* vdst = reinterpret_cast<_Float16*>(q.kv.v) + (size_t)(blk * kvl + kvh) * kBlockTokens * 128;
                    const int tib = p & 255, tile32 = tib >> 5, tt = tib & 31;
                    const bool full = tfirst >= 0 && tfirst + 3 < nvalid;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int d = lane * 2 + e;
                        const uint2 w = *reinterpret_cast<const uint2*>(vt + d * LDV + 4 * cq);
                        const size_t off = ((size_t)(tile32 * 8 + (d >> 4)) * 64 + ((tt & 15) >> 2) * 16 + (d & 15)) * 8 + (tt >> 4) * 4;
                        if (full) {
                            *reinterpret_cast<uint2*>(vdst + off) = w;
                        } else {
                            const uint16_t hv[4] = {(uint16_t)(w.x & 0xffff), (uint16_t)(w.x >> 16), (uint16_t)(w.y & 0xffff), (uint16_t)(w.y >> 16)};
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (tfirst + i >= 0 && tfirst + i < nvalid) reinterpret_cast<uint16_t*>(vdst)[off + i] = hv[i];
                        }
                    }
                }
                v_done = true;
            }
        }
        if (hh < nh + kvl) {
            // q / k head.  cos / sin rows of the NEXT 16 tokens are fetched while the current 16 are rotated (two register
            // sets, static indices); hoisting the norm weights as well spilled (256 registers + scratch)
            const float* w = hh < nh ? q.qn : q.kn;
            auto load_cs = [&](int pos, float4 (&cs)[4], float4 (&sn)[4]) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    cs[j] = *reinterpret_cast<const float4*>(q.cos + (size_t)pos * 64 + j * 16 + grp * 4);
                    sn[j] = *reinterpret_cast<const float4*>(q.sin + (size_t)pos * 64 + j * 16 + grp * 4);
                }
            };
            auto tokens16 = [&](auto bc, const float4 (&cs)[4], const float4 (&sn)[4]) {
                constexpr int b = decltype(bc)::value;
                const int row = row0 + b * 16 + l15;
                const int pos = e_pos[b];
                const float ri = rownorm_rinv(q.rn, min(row, a.M - 1));  // deferred input norm (1 when the planes were normalised)
                // whole-vector arithmetic on the f32x4 accumulators: hipcc pairs it into v_pk_mul / v_pk_fma (two lanes' worth
                // of flops per instruction) -- this epilogue is VALU-bound (every wave of the chip is in it at once)
                f32x4 ss4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f32x4 x = acc[j][b] * ri;
                    acc[j][b] = x;
                    ss4 += x * x;
                }
                float ss = (ss4[0] + ss4[1]) + (ss4[2] + ss4[3]);
                ss += __shfl_xor(ss, 16);  // the four lane groups hold the other 96 features of this token
                ss += __shfl_xor(ss, 32);
                const float rinv = 1.0f / sqrtf(ss / 128.0f + q.eps);
                const float osc = hh < nh ? q.q_scale : 1.0f;  // q leaves scaled; folded into the norm weight
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 w1 = __builtin_bit_cast(f32x4, *reinterpret_cast<const float4*>(w + j * 16 + grp * 4)) * (rinv * osc);
                    const f32x4 w2 = __builtin_bit_cast(f32x4, *reinterpret_cast<const float4*>(w + 64 + j * 16 + grp * 4)) * (rinv * osc);
                    const f32x4 c = __builtin_bit_cast(f32x4, cs[j]), sn_ = __builtin_bit_cast(f32x4, sn[j]);
                    const f32x4 n1 = acc[j][b] * w1, n2 = acc[j + 4][b] * w2;
                    acc[j][b] = n1 * c - n2 * sn_;
                    acc[j + 4][b] = n2 * c + n1 * sn_;
                }
                if (row < a.M) {
                    if (hh < nh) {
                        float* qo = q.q_out + (size_t)row * (nh * 128) + (size_t)hh * 128 + grp * 4;
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            *reinterpret_cast<float4*>(qo + j * 16) = make_float4(acc[j][b][0], acc[j][b][1], acc[j][b][2], acc[j][b][3]);
                    } else {
                        _Float16* k = reinterpret_cast<_Float16*>(q.kv.k) + (size_t)(e_blk[b] * kvl + (hh - nh)) * kBlockTokens * 128;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
                            const f16x4 v = {f16_sat(acc[j][b][0]), f16_sat(acc[j][b][1]), f16_sat(acc[j][b][2]), f16_sat(acc[j][b][3])};
                            *reinterpret_cast<f16x4*>(k + k_packed_offset(pos & 255, j * 16 + grp * 4, 128)) = v;
                        }
                    }
                }
            };
            float4 csA[4], snA[4], csB[4], snB[4];
            load_cs(e_pos[0], csA, snA);
            load_cs(e_pos[1], csB, snB);
            tokens16(std::integral_constant<int, 0>{}, csA, snA);
            load_cs(e_pos[2], csA, snA);
            tokens16(std::integral_constant<int, 1>{}, csB, snB);
            load_cs(e_pos[3], csB, snB);
            tokens16(std::integral_constant<int, 2>{}, csA, snA);
            tokens16(std::integral_constant<int, 3>{}, csB, snB);
        } else if (!v_done) {  // V head, rows of several sequences in this wave: element-wise copy into the PV fragment order
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int row = row0 + b * 16 + l15;
                if (row < a.M) {
                    const float ri = rownorm_rinv(q.rn, row);
                    const int pos = e_pos[b];
                    _Float16* v = reinterpret_cast<_Float16*>(q.kv.v) + (size_t)(e_blk[b] * kvl + (hh - nh - kvl)) * kBlockTokens * 128;
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[v_packed_offset(pos & 255, j * 16 + grp * 4 + r, 128)] = f16_sat(acc[j][b][r] * ri);
                }
            }
        }
    } else {
        static_assert(MODE != 2 || NTW % 2 == 0, "gate and up tiles of a feature live in one wave");
        const int I = a.N >> 1;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int row = row0 + b * 16 + l15;
            if (row < a.M) {
#pragma unroll
                for (int pr = 0; pr < NTW / 2; ++pr) {
                    const int f0 = ((nt0 >> 1) + pr) * 16 + grp * 4;
                    uint16_t h[4], l[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float g = acc[2 * pr][b][r], u = acc[2 * pr + 1][b][r];
                        split_bf16(silu_mul(g, u), h[r], l[r]);
                    }
                    const size_t xo = a.act_packed ? xpack_off(row, f0, I >> 5) : (size_t)row * I + f0;
                    *reinterpret_cast<uint2*>(a.act_hi + xo) = make_uint2(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16));
                    *reinterpret_cast<uint2*>(a.act_lo + xo) = make_uint2(l[0] | ((uint32_t)l[1] << 16), l[2] | ((uint32_t)l[3] << 16));
                }
            }
        }
    }
    NVLLM_STAMP(a, 3);
}

// ---- host side -------------------------------------------------------------------------------------
// Shape of the launch for an [M][N] output: block geometry (NTW, WN) and K splits.  Candidates: 256 x 256 and 256 x 192
// (WN 2) and 128 x 256 (WN 4, NTW 4) for outputs too narrow or too short to fill the chip with 256-row blocks; mode 0
// may split K (f32 slabs, summed by the consumer like the chunked kernel's).  Cost in microseconds = chip rounds x one
// block's MFMA time (hi + lo planes, half the dense peak as the sustained rate, the narrow block charged 25 % more for
// its LDS reads per MFMA) + the slabs' write and re-read when K is split.  ntw == 0: not covered, or fewer than
// min_wgs workgroups.
struct TileShape { int ntw = 0, wn = 0, ks = 1; };
static TileShape tile_shape(int M, int N, int K, int mode, int min_wgs, int max_split) {
    TileShape best;
    if (M < 1 || K % 64 != 0 || K < 128 || N % 32 != 0) return best;
    const int KT = K / 32;
    double best_cost = 0;
    const int cand[3][2] = {{8, 2}, {6, 2}, {4, 4}};
    for (const auto& c : cand) {
        const int ntw = c[0], wn = c[1], cols = wn * ntw * 16, rows = (8 / wn) * 64;
        if (N % cols != 0 || (mode == 2 && ntw % 2)) continue;
        const int64_t blocks = (int64_t)((M + rows - 1) / rows) * (N / cols);
        for (int ks = 1; ks <= (mode == 0 ? max_split : 1); ks *= 2) {
            if (KT % (2 * ks) != 0 || (ks > 1 && KT / ks < 8)) continue;
            const int64_t wgs = blocks * ks;
            if (wgs < min_wgs) continue;
            const double block_us = 4.0 * rows * cols * (double)(K / ks) / (2.5e15 * 0.5 / 256) * 1e6 * (wn == 4 ? 1.25 : 1.0);
            const double slab_us = ks > 1 ? 2.0 * ks * (double)M * N * 4 / 4e12 * 1e6 : 0.0;
            const double cost = (double)((wgs + 255) / 256) * block_us + slab_us;
            if (!best.ntw || cost < best_cost) { best.ntw = ntw; best.wn = wn; best.ks = ks; best_cost = cost; }
        }
    }
    return best;
}

bool gemm_tile_ok(int M, int N, int K, int mode, int min_wgs) { return tile_shape(M, N, K, mode, min_wgs, 1).ntw != 0; }
int gemm_tile_splits(int M, int N, int K, int min_wgs, int max_split) { return tile_shape(M, N, K, 0, min_wgs, max_split).ntw ? tile_shape(M, N, K, 0, min_wgs, max_split).ks : 0; }

template <int NTW, int WN, int MODE>
static hipError_t tile_launch_t(TileArgs& a, int ks, hipStream_t s) {
    constexpr int rows = (8 / WN) * 64, FR = WN * NTW + 2 * (8 / WN) * 4;
    a.NB = a.N / (WN * NTW * 16);
    a.MB = (a.M + rows - 1) / rows;
    a.kts = a.KT / ks;
    int grid;
    if (a.MB >= 8) { a.map = 1; grid = ((a.MB + 7) / 8) * 8 * a.NB; }
    else if (a.NB >= 8) { a.map = 2; grid = ((a.NB + 7) / 8) * 8 * a.MB; }
    else { a.map = 0; grid = a.MB * a.NB; }
    const size_t lds = (size_t)3 * FR * 1024;
    static std::atomic<uint64_t> lds_set{0};
    ensure_dyn_lds(reinterpret_cast<const void*>(gemm_tile_kernel<NTW, WN, MODE>), lds, lds_set);
#ifdef NVLLM_STAMPS
    if (g_tile_stamps && g_tile_stamp_launch < g_tile_stamp_max && (size_t)grid * ks <= 1024)
        a.stamps = g_tile_stamps + (size_t)(g_tile_stamp_launch++) * ((size_t)1024 * 16 * 8);
#endif
    gemm_tile_kernel<NTW, WN, MODE><<<dim3(grid, ks), 512, lds, s>>>(a);
    return hipGetLastError();
}

// n_slabs (nullable): K splits the launch wrote (out = [n_slabs][M][N]); nullptr -> never split
hipError_t launch_gemm_tile(const bf16_bits* xh, const bf16_bits* xl, const PackedW& w, int M, int mode, float* out,
                            bf16_bits* act_hi, bf16_bits* act_lo, int act_packed, int min_wgs, int max_split, int* n_slabs,
                            hipStream_t s) {
    const TileShape t = tile_shape(M, w.N, w.K, mode, min_wgs, n_slabs ? max_split : 1);
    if (!t.ntw) return hipErrorNotSupported;
    if (mode == 2 && act_packed && (w.N / 2) % 32 != 0) return hipErrorNotSupported;
    TileArgs a;
    a.xh = reinterpret_cast<const uint4*>(xh); a.xl = reinterpret_cast<const uint4*>(xl); a.wp = w.data;
    a.out = out; a.act_hi = act_hi; a.act_lo = act_lo; a.act_packed = act_packed;
    a.M = M; a.N = w.N; a.KT = w.K / 32;
    if (n_slabs) *n_slabs = t.ks;
    if (mode == 0) {
        if (t.wn == 4) return tile_launch_t<4, 4, 0>(a, t.ks, s);
        return t.ntw == 8 ? tile_launch_t<8, 2, 0>(a, t.ks, s) : tile_launch_t<6, 2, 0>(a, t.ks, s);
    }
    if (mode == 2) {
        if (t.wn == 4) return tile_launch_t<4, 4, 2>(a, 1, s);
        return t.ntw == 8 ? tile_launch_t<8, 2, 2>(a, 1, s) : tile_launch_t<6, 2, 2>(a, 1, s);
    }
    return hipErrorInvalidValue;
}

bool gemm_tile_qkv_ok(int M, int N, int K, int hd, int min_wgs) {
    const TileShape t = tile_shape(M, N, K, 0, min_wgs, 1);
    return hd == 128 && t.ntw == 8 && t.wn == 2;
}

hipError_t launch_gemm_tile_qkv(const bf16_bits* xh, const bf16_bits* xl, const PackedW& w, int M, const QkvArgs& qa, int min_wgs,
                                hipStream_t s) {
    if (!gemm_tile_qkv_ok(M, w.N, w.K, qa.kv.hd, min_wgs) || w.N != (qa.nh_l + 2 * qa.kv.kv_l) * 128) return hipErrorNotSupported;
    if (qa.rn.ssq && qa.rn.groups != 1) return hipErrorNotSupported;
    TileArgs a;
    a.xh = reinterpret_cast<const uint4*>(xh); a.xl = reinterpret_cast<const uint4*>(xl); a.wp = w.data;
    a.M = M; a.N = w.N; a.KT = w.K / 32; a.q = qa;
    return tile_launch_t<8, 2, 3>(a, 1, s);
}

// row-major plane [M][K] -> xpack_off order (tests and the tuning bench; the product's producers write the order directly)
__global__ void __launch_bounds__(256) xpack_plane_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int M, int K) {
    const int KT = K >> 5;
    const int64_t n = (int64_t)((M + 15) >> 4) * KT * 64;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        const int64_t t = i >> 6;
        const int kt = (int)(t % KT), mt = (int)(t / KT);
        const int row = min(mt * 16 + (lane & 15), M - 1);
        dst[i] = src[((size_t)row * K + (size_t)kt * 32 + (lane >> 4) * 8) >> 3];
    }
}
hipError_t launch_xpack_plane(const bf16_bits* src, bf16_bits* dst, int M, int K, hipStream_t s) {
    if (K % 32 != 0 || M < 1) return hipErrorInvalidValue;
    const int64_t n = (int64_t)((M + 15) / 16) * (K / 32) * 64;
    xpack_plane_kernel<<<(int)std::min<int64_t>((n + 255) / 256, 8192), 256, 0, s>>>(reinterpret_cast<const uint4*>(src),
                                                                                      reinterpret_cast<uint4*>(dst), M, K);
    return hipGetLastError();
}

}  // namespace nvllm

// Streaming GEMM of the decode path for matrices too big (or too oddly shaped) for the whole-K register-direct kernel:
// Qwen3-8B / 32B layers, tensor-parallel shards.  y[M][N] = x[M][K] . W[N][K]^T for 17..64 rows
// (replaces candle_nn::Linear::forward, src/layers/linear.rs:35-36,72-77,184-198; src/models/qwen3.rs:205,278,324,326).
//
// One workgroup of 8 waves per (n-group, K slice): a wave owns NT n-tiles over its slice and keeps two SC-k-tile weight
// sets in flight (HBM -> VGPR, 1 KiB wave-loads, refilled right after use); all 64 rows of x ride through LDS in
// KC-k-tile chunks (LDS-DMA, double buffered).  After the first chunk the barrier waits only for the wave's own x stage
// (counted vmcnt), so the weight stream is never drained.  K is cut into `ks` slices over blockIdx.y so that about one
// workgroup lands on every CU.
//
// Epilogues (EPI):
//   0  every slice leaves an f32 slab out[ks][M][N]; the consumer sums them (add_rmsnorm, attention prologue, ...)
//   1  complete sums [M][N] f32                       (QKV; row-parallel o_proj / down_proj partials before an all-reduce)
//   2  SwiGLU: silu(gate) * up as bf16 hi/lo [M][N/2] (activation.rs:13-18), input norm deferred (RowNorm)
//   3  residual add + next-norm prep: resid += y; x' = w_next (.) resid as bf16 hi/lo; ssq[n-group][row]
//      (layernorm.rs:44-60 with the 1/rms factor left to the consumer, see RowNorm)
// Epilogues 1-3 with ks > 1 combine INSIDE the launch: every slice stores its slab, the workgroup that arrives last at the
// n-group's ticket (agent-scope release before the ticket, acquire after: cdna guide, in-launch split-K reduction) sums
// all slabs in slice order -- deterministic -- and runs the epilogue.  That removes the separate sum / SiLU / norm launch
// (about 4 us of boundary + start-up each) at the price of one workgroup per n-group reading ks slabs.
#include <algorithm>
#include <cstdlib>

#include "device_common.h"

namespace nvllm {

// NTL: weight loads non-temporal.  Every weight byte is read once, by one workgroup, so it need not displace the x planes
// and slabs other workgroups re-read; measured (interleaved A/B, batch 64): Qwen3-32B 19.1 -> 18.1 ms/step, Qwen3-8B
// 6.37 -> 6.43 -- it pays on the big matrices, so the host sets it from the matrix size.
// NW: waves per workgroup.  8 everywhere but for slab outputs (EPI 0), where the host may pick 10 or 13 so that the n-tiles
// of a hidden size with a factor 5 (Qwen3-32B: 5120 = 320 tiles) or a very wide matrix (3200 tiles) still make ~256
// workgroups: with 8 waves those shapes land on 200 of the 256 CUs.
// ABL (diagnostic build only, tools/ablate_stream.py): 1 = no MFMAs and no LDS fragment reads, 2 = no x staging, 3 = no weight
// loads, 4 = no slab stores -- what is left of the kernel's time says which part bounds it.  0 in every product launch.
template <int NT, int SC, int KC, int EPI, bool NTL, int NW = 8, int ABL = 0>
__global__ void __launch_bounds__(NW * 64) gemm_stream_kernel(StreamArgs a, const uint4* __restrict__ wp, int N, int KT, int kts) {
    constexpr int MT = 4;
    constexpr int FRAGS = 2 * MT * KC;          // 1 KiB fragments per x chunk: [2 planes][MT][KC]
    constexpr int PER_WAVE = (FRAGS + NW - 1) / NW;  // LDS-DMA loads per wave and chunk (fragment f = wave + i * NW while f < FRAGS)
    static_assert(KC == 8 || KC == 4, "x chunks of 8 or 4 k-tiles");
    static_assert((KC == 8 && (SC == 4 || SC == 2)) || (KC == 4 && SC == 2), "the two weight sets alternate inside a chunk");
    static_assert(EPI != 2 || NT % 2 == 0, "SwiGLU pairs the gate and up tiles of a feature in one wave");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint4* lds = reinterpret_cast<uint4*>(smem_raw);  // [2][FRAGS][64]; after the K loop: epilogue scratch
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, grp = lane >> 4;
    const int ntiles = N >> 4, M = a.M, ks = (int)gridDim.y;
    const int nt0 = ((int)blockIdx.x * NW + wave) * NT;
    const int kt_begin = (int)blockIdx.y * kts;
    // this wave stages fragments f = wave + i * NW of every chunk: f -> k-tile f % KC of (plane, row block) f / KC
    unsigned xoff[PER_WAVE];
    const unsigned xstep = a.x_packed ? 512u : 32u;
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int f = min(wave + i * NW, FRAGS - 1), kst = f % KC, pb = f / KC, b = pb % MT;
        xoff[i] = a.x_packed ? (unsigned)((b * (a.ldx >> 5) + kt_begin + kst) * 512 + lane * 8)
                             : (unsigned)min(b * 16 + l15, M - 1) * (unsigned)a.ldx + (unsigned)((kt_begin + kst) * 32 + grp * 8);
    }
    f32x4 acc[NT][MT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[t][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto stage = [&](int c, int buf) {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int f = wave + i * NW, plane = (f / KC) / MT;
            if (ABL != 2 && (FRAGS % NW == 0 || f < FRAGS)) {  // wave-uniform
                const uint16_t* src = (plane ? a.xl : a.xh) + (size_t)(xoff[i] + (unsigned)(c * KC) * xstep);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + (size_t)(buf * FRAGS + f) * 64), 16, 0, 0);
            }
        }
    };
    uint4 wA[NT][SC], wB[NT][SC];
    auto issue_w = [&](int k_rel, uint4 (&w)[NT][SC]) {
        const int kt = kt_begin + min(k_rel, kts - SC);  // past the slice: a harmless re-read of its last set
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int ntc = min(nt0 + t, ntiles - 1);
#pragma unroll
            for (int j = 0; j < SC; ++j) {
                const uint4* pw = wp + ((size_t)ntc * KT + kt + j) * 64 + lane;
                if constexpr (ABL == 3) w[t][j] = make_uint4(0, 0, 0, 0);
                else w[t][j] = NTL ? ld_stream16(pw) : *pw;
            }
        }
    };
    auto compute = [&](int k0, int buf, const uint4 (&w)[NT][SC]) {
        if constexpr (ABL == 1) {  // the weight registers stay live (their loads are waited for), nothing is computed
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < SC; ++j) asm volatile("" ::"v"(w[t][j].x), "v"(w[t][j].y), "v"(w[t][j].z), "v"(w[t][j].w));
            return;
        }
#pragma unroll
        for (int j = 0; j < SC; ++j)
#pragma unroll
            for (int plane = 0; plane < 2; ++plane) {
                bf16x8 bx[MT];
#pragma unroll
                for (int b = 0; b < MT; ++b)
                    bx[b] = __builtin_bit_cast(bf16x8, lds[(size_t)(buf * FRAGS + (plane * MT + b) * KC + k0 + j) * 64 + lane]);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const bf16x8 wv = __builtin_bit_cast(bf16x8, w[t][j]);
#pragma unroll
                    for (int b = 0; b < MT; ++b) acc[t][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, bx[b], acc[t][b], 0, 0, 0);
                }
            }
    };
    const int nchunks = kts / KC;
    stage(0, 0);
    issue_w(0, wA);
    issue_w(SC, wB);
    for (int c = 0; c < nchunks; ++c) {
        // chunk c is in LDS and every wave is done with the other buffer.  After the first chunk only this wave's own x
        // stage is waited for: the KC * NT weight loads issued after it (vmcnt retires in order) stay in flight
        if (c == 0) dma_publish_barrier();
        else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(KC * NT) : "memory");
        if (c + 1 < nchunks) stage(c + 1, (c + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);  // the weight loads below are issued AFTER the stage (the count above relies on it)
        const int base = c * KC, buf = c & 1;
#define NVLLM_ST_STEP(k0_, set_, next_)            \
    compute(k0_, buf, set_);                       \
    __builtin_amdgcn_sched_barrier(0);             \
    issue_w(next_, set_);                          \
    __builtin_amdgcn_sched_barrier(0);
        if constexpr (KC == 8 && SC == 4) {
            NVLLM_ST_STEP(0, wA, base + 8)
            NVLLM_ST_STEP(4, wB, base + 12)
        } else if constexpr (KC == 8 && SC == 2) {
            NVLLM_ST_STEP(0, wA, base + 4)
            NVLLM_ST_STEP(2, wB, base + 6)
            NVLLM_ST_STEP(4, wA, base + 8)
            NVLLM_ST_STEP(6, wB, base + 10)
        } else {
            static_assert(KC == 8 || SC == 2, "4-k-tile chunks run 2-k-tile sets");
            NVLLM_ST_STEP(0, wA, base + 4)
            NVLLM_ST_STEP(2, wB, base + 6)
        }
#undef NVLLM_ST_STEP
    }
    // ---- slabs / in-launch combine ------------------------------------------------------------------------------
    const size_t slab_stride = (size_t)M * N;
    if (ABL != 4 && (EPI == 0 || ks > 1)) {
        float* o = a.slabs + (size_t)blockIdx.y * slab_stride;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (nt0 + t >= ntiles) continue;
#pragma unroll
            for (int b = 0; b < MT; ++b) {
                const int row = b * 16 + l15;
                if (row < M) {
                    const f32x4 v = acc[t][b];
                    *reinterpret_cast<float4*>(o + (size_t)row * N + (size_t)(nt0 + t) * 16 + grp * 4) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
    if constexpr (EPI == 0) return;
    unsigned* flag = reinterpret_cast<unsigned*>(smem_raw);  // LDS word 0 (the x buffers are dead: every wave passes the barrier below first)
    if (ks > 1) {
        // publish: every wave's slab stores are complete, then ONE lane releases at agent scope and draws the ticket
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the fence's own wait may be dropped by the compiler (guide, G16 pitfall 12)
            const unsigned t = __hip_atomic_fetch_add(a.tickets + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned last = t == (unsigned)(ks - 1) ? 1u : 0u;
            if (last) {
                __hip_atomic_store(a.tickets + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the next launch
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            *flag = last;
        }
        __syncthreads();
        if (*flag == 0u) return;  // uniform for the workgroup
        __syncthreads();          // everyone has read the flag word before the epilogue reuses LDS
        // the last arriver: sum ALL slices in slice order (its own slab included: the order must not depend on arrival)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int b = 0; b < MT; ++b) acc[t][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        // U slabs per trip, their loads independent (slab index clamped, the sum masked): one load per loop trip would
        // cost a round trip per slab
        constexpr int U = NT <= 2 ? 4 : 2;
        size_t eo[NT][MT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int b = 0; b < MT; ++b)  // clamped: rows >= M / tiles past the end are loaded but never stored
                eo[t][b] = (size_t)min(b * 16 + l15, M - 1) * N + (size_t)min(nt0 + t, ntiles - 1) * 16 + grp * 4;
        for (int sl0 = 0; sl0 < ks; sl0 += U) {
            float4 v[U][NT][MT];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float* o = a.slabs + (size_t)min(sl0 + u, ks - 1) * slab_stride;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int b = 0; b < MT; ++b) v[u][t][b] = *reinterpret_cast<const float4*>(o + eo[t][b]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float mk = sl0 + u < ks ? 1.f : 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int b = 0; b < MT; ++b) {
                        acc[t][b][0] += mk * v[u][t][b].x; acc[t][b][1] += mk * v[u][t][b].y;
                        acc[t][b][2] += mk * v[u][t][b].z; acc[t][b][3] += mk * v[u][t][b].w;
                    }
            }
        }
    } else {
        __syncthreads();  // the x buffers are dead from here on (epilogue scratch)
    }
    // ---- epilogues on complete sums ------------------------------------------------------------------------------
    float* lds_f = reinterpret_cast<float*>(smem_raw);
    if constexpr (EPI == 1) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (nt0 + t >= ntiles) continue;
#pragma unroll
            for (int b = 0; b < MT; ++b) {
                const int row = b * 16 + l15;
                if (row < M) {
                    const f32x4 v = acc[t][b];
                    *reinterpret_cast<float4*>(a.out + (size_t)row * N + (size_t)(nt0 + t) * 16 + grp * 4) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    } else if constexpr (EPI == 2) {
        rownorm_partials<MT * 16>(a.rn, 0, M, lds_f);
        __syncthreads();
        const int I = N >> 1;
#pragma unroll
        for (int p = 0; p < NT / 2; ++p) {
            if (nt0 + 2 * p + 1 >= ntiles) continue;
            const int f0 = ((nt0 >> 1) + p) * 16 + grp * 4;  // activation feature of acc[..][..][0]
#pragma unroll
            for (int b = 0; b < MT; ++b) {
                const int row = b * 16 + l15;
                if (row >= M) continue;
                const float ri = rownorm_rinv_lds<MT * 16>(a.rn, lds_f, row);
                uint16_t h[4], l[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float g = acc[2 * p][b][r] * ri, u = acc[2 * p + 1][b][r] * ri;
                    split_bf16(silu_mul(g, u), h[r], l[r]);
                }
                const size_t o = a.o_packed ? xpack_off(row, f0, I >> 5) : (size_t)row * I + f0;
                *reinterpret_cast<uint2*>(a.oh + o) = make_uint2(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16));
                *reinterpret_cast<uint2*>(a.ol + o) = make_uint2(l[0] | ((uint32_t)l[1] << 16), l[2] | ((uint32_t)l[3] << 16));
            }
        }
    } else {
        // residual + next-norm prep; per-row partial sums of squares over this workgroup's NW * NT * 16 columns
        float ssq_row[MT];
#pragma unroll
        for (int b = 0; b < MT; ++b) ssq_row[b] = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bool ok = nt0 + t < ntiles;
            const int ntc = min(nt0 + t, ntiles - 1);
            const float4 nw4 = *reinterpret_cast<const float4*>(a.next_w + (size_t)ntc * 16 + grp * 4);
#pragma unroll
            for (int b = 0; b < MT; ++b) {
                const int row = b * 16 + l15;
                if (!ok || row >= M) continue;
                const size_t o = (size_t)row * N + (size_t)ntc * 16 + grp * 4;
                const float4 r4 = *reinterpret_cast<const float4*>(a.resid_in + o);
                const float s0 = acc[t][b][0] + r4.x, s1 = acc[t][b][1] + r4.y, s2 = acc[t][b][2] + r4.z, s3 = acc[t][b][3] + r4.w;
                *reinterpret_cast<float4*>(a.resid_out + o) = make_float4(s0, s1, s2, s3);
                uint16_t h0, h1, h2, h3, l0, l1, l2, l3;
                split_bf16(s0 * nw4.x, h0, l0); split_bf16(s1 * nw4.y, h1, l1); split_bf16(s2 * nw4.z, h2, l2); split_bf16(s3 * nw4.w, h3, l3);
                const size_t ob = a.o_packed ? xpack_off(row, ntc * 16 + grp * 4, N >> 5) : o;
                *reinterpret_cast<uint2*>(a.oh + ob) = make_uint2(h0 | ((uint32_t)h1 << 16), h2 | ((uint32_t)h3 << 16));
                *reinterpret_cast<uint2*>(a.ol + ob) = make_uint2(l0 | ((uint32_t)l1 << 16), l2 | ((uint32_t)l3 << 16));
                ssq_row[b] += s0 * s0 + s1 * s1 + s2 * s2 + s3 * s3;
            }
        }
#pragma unroll
        for (int b = 0; b < MT; ++b) {
            ssq_row[b] += __shfl_xor(ssq_row[b], 16);
            ssq_row[b] += __shfl_xor(ssq_row[b], 32);
            if (grp == 0) lds_f[wave * (MT * 16) + b * 16 + l15] = ssq_row[b];
        }
        __syncthreads();
        if (threadIdx.x < MT * 16 && (int)threadIdx.x < M) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += lds_f[w * (MT * 16) + threadIdx.x];
            a.ssq[(size_t)blockIdx.x * a.ssq_stride + threadIdx.x] = t;
        }
    }
}

// (n-tiles per wave, K slices, x chunk depth): about one 8-wave workgroup per CU, at most one round of the chip
struct StreamShape { int nt, ks, kc, nw; };
static StreamShape stream_shape(int M, int N, int K, int epi, bool any_size) {
    StreamShape none{0, 0, 0, 8};
    if (M <= 16 || M > 64 || N % 16 || K % 32) return none;
    if (!any_size && (size_t)N * K * 2 < ((size_t)24 << 20)) return none;  // small matrices: whole-K kernels when they apply
    if (epi == 2 && (N / 16) % 2) return none;
    const int KT = K / 32, ntiles = N / 16;
    StreamShape best = none;
    int best_wgs = 0;
    for (int nw : {8, 10, 13}) {  // 8 first: a wider workgroup must bring strictly more workgroups to be taken
        if (nw != 8 && (epi != 0 || M <= 16)) continue;  // only the slab form is instantiated for 10 / 13 waves
        for (int kc : {8, 4})
            for (int nt = 4; nt >= 1; --nt) {
                if (epi == 2 && nt % 2) continue;
                if (epi != 2 && nt == 4) continue;  // 4 tiles per wave only where SwiGLU needs pairs and 2 leave too many workgroups
                if (nw == 13 && nt > 2) continue;   // 13 waves: 4 on one SIMD, <= 128 registers
                for (int ks = 1; ks <= 16; ++ks) {
                    if (KT % ks || (KT / ks) % kc || KT / ks < 2 * kc) continue;
                    const int ngroups = ((ntiles + nt - 1) / nt + nw - 1) / nw;
                    if (epi == 3 && ngroups > 64) continue;  // deferred-norm consumers sum <= 64 ssq groups
                    const int wgs = ngroups * ks;
                    // most workgroups within one round of the chip; ties: deep x chunks, then fewer slices (fewer slabs)
                    if (wgs <= 256 && (wgs > best_wgs || (wgs == best_wgs && nw == best.nw && kc == best.kc && ks < best.ks))) { best_wgs = wgs; best = StreamShape{nt, ks, kc, nw}; }
                }
            }
    }
    return best_wgs >= (any_size ? 48 : 128) ? best : none;
}
int gemm_stream_splits(int M, int N, int K) { return stream_shape(M, N, K, 0, false).ks; }
bool gemm_stream_ok(int M, int N, int K, int epi) { return stream_shape(M, N, K, epi, true).nt != 0; }
int gemm_stream_groups(int M, int N, int K, int epi) {
    const StreamShape sh = stream_shape(M, N, K, epi, true);
    return sh.nt ? ((N / 16 + sh.nt - 1) / sh.nt + sh.nw - 1) / sh.nw : 0;
}
size_t gemm_stream_slab_floats(int M, int N, int K, int epi) {
    const StreamShape sh = stream_shape(M, N, K, epi, true);
    return sh.nt ? (size_t)sh.ks * M * N : 0;
}

#ifdef NVLLM_STAMPS
static int g_stream_ablate = 0;
void stream_gemm_set_ablate(int v) { g_stream_ablate = v; }
#endif
template <int NT, int SC, int KC, int EPI, int NW>
static hipError_t stream_launch_t(const StreamShape& sh, const StreamArgs& a, const PackedW& w, hipStream_t s) {
    const size_t lds = (size_t)2 * (2 * 4 * KC) * 1024;
    const int waves = (w.N / 16 + NT - 1) / NT;
    dim3 grid((waves + NW - 1) / NW, sh.ks);
#ifdef NVLLM_STAMPS
    if constexpr (EPI == 0 && NT == 1) {
        if (g_stream_ablate) {
            static std::atomic<uint64_t> lds_set_ab{0};
#define NVLLM_AB(V_)                                                                                                             \
    if (g_stream_ablate == V_) {                                                                                                 \
        ensure_dyn_lds(reinterpret_cast<const void*>(gemm_stream_kernel<NT, SC, KC, EPI, true, NW, V_>), lds, lds_set_ab);       \
        gemm_stream_kernel<NT, SC, KC, EPI, true, NW, V_><<<grid, NW * 64, lds, s>>>(a, w.data, w.N, w.K / 32, w.K / 32 / sh.ks); \
        return hipGetLastError();                                                                                                \
    }
            NVLLM_AB(1) NVLLM_AB(2) NVLLM_AB(3) NVLLM_AB(4)
#undef NVLLM_AB
        }
    }
#endif
    constexpr size_t nt_min_bytes = (size_t)80 << 20;  // measured: 32B shards gain from 80 MB up, 8B matrices below lose
    if (w.bytes() >= nt_min_bytes) {  // big matrix: non-temporal weight stream
        static std::atomic<uint64_t> lds_set_nt{0};
        ensure_dyn_lds(reinterpret_cast<const void*>(gemm_stream_kernel<NT, SC, KC, EPI, true, NW>), lds, lds_set_nt);
        gemm_stream_kernel<NT, SC, KC, EPI, true, NW><<<grid, NW * 64, lds, s>>>(a, w.data, w.N, w.K / 32, w.K / 32 / sh.ks);
        return hipGetLastError();
    }
    static std::atomic<uint64_t> lds_set{0};
    ensure_dyn_lds(reinterpret_cast<const void*>(gemm_stream_kernel<NT, SC, KC, EPI, false, NW>), lds, lds_set);
    gemm_stream_kernel<NT, SC, KC, EPI, false, NW><<<grid, NW * 64, lds, s>>>(a, w.data, w.N, w.K / 32, w.K / 32 / sh.ks);
    return hipGetLastError();
}
template <int EPI>
static hipError_t stream_dispatch(const StreamShape& sh, const StreamArgs& a, const PackedW& w, hipStream_t s) {
    // set depth made no difference on MI355X (2- vs 4-k-tile sets, tools/probe_lm.py): the shallow ones use fewer registers
#define NVLLM_ST(NT_, SC_, KC_) if (sh.nt == NT_ && sh.kc == KC_ && sh.nw == 8) return stream_launch_t<NT_, SC_, KC_, EPI, 8>(sh, a, w, s);
#define NVLLM_STW(NT_, SC_, KC_, NW_) if (sh.nt == NT_ && sh.kc == KC_ && sh.nw == NW_) return stream_launch_t<NT_, SC_, KC_, EPI, NW_>(sh, a, w, s);
    if constexpr (EPI != 2) { NVLLM_ST(1, 4, 8) NVLLM_ST(1, 2, 4) NVLLM_ST(3, 2, 8) NVLLM_ST(3, 2, 4) }
    NVLLM_ST(2, 2, 8) NVLLM_ST(2, 2, 4)
    if constexpr (EPI == 2) { NVLLM_ST(4, 2, 8) NVLLM_ST(4, 2, 4) }
    if constexpr (EPI == 0) {
        NVLLM_STW(1, 4, 8, 10) NVLLM_STW(1, 2, 4, 10) NVLLM_STW(2, 2, 8, 10) NVLLM_STW(2, 2, 4, 10) NVLLM_STW(3, 2, 8, 10) NVLLM_STW(3, 2, 4, 10)
        NVLLM_STW(1, 4, 8, 13) NVLLM_STW(1, 2, 4, 13) NVLLM_STW(2, 2, 8, 13) NVLLM_STW(2, 2, 4, 13)
    }
#undef NVLLM_ST
#undef NVLLM_STW
    return hipErrorNotSupported;
}
hipError_t launch_gemm_stream_epi(const StreamArgs& a, const PackedW& w, int epi, hipStream_t s) {
    const StreamShape sh = stream_shape(a.M, w.N, w.K, epi, true);
    if (!sh.nt || a.ldx != w.K) return hipErrorNotSupported;
    if (epi != 0 && sh.ks > 1 && (!a.tickets || !a.slabs)) return hipErrorInvalidValue;
    if (epi == 0) return stream_dispatch<0>(sh, a, w, s);
    if (epi == 1) return stream_dispatch<1>(sh, a, w, s);
    if (epi == 2) return stream_dispatch<2>(sh, a, w, s);
    if (epi == 3) return stream_dispatch<3>(sh, a, w, s);
    return hipErrorInvalidValue;
}
hipError_t launch_gemm_stream(const bf16_bits* xh, const bf16_bits* xl, int ldx, const PackedW& w, float* out, int M, int x_packed,
                              hipStream_t s) {
    if (!stream_shape(M, w.N, w.K, 0, false).nt) return hipErrorNotSupported;
    StreamArgs a;
    a.xh = xh; a.xl = xl; a.ldx = ldx; a.x_packed = x_packed; a.M = M; a.slabs = out;
    return launch_gemm_stream_epi(a, w, 0, s);
}

}  // namespace nvllm

// C ABI (include/nvllm_amd.h) over the gfx950 kernels: context, model (weights in MFMA tile layout),
// native KV block pool, the ModelRunner::run-shaped step, and the fine-seam ops.
// Host C++ compiled with hipcc.  No CPU compute fallback exists anywhere in this file: every
// arithmetic result comes from a kernel in kernels.hip.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/nvllm_amd.h"
#include "../../include/nvllm_amd_debug.h"
#include "kernels.h"
#include "synth_device.h"

using namespace nvllm;

// ---------------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------------
// In-process loopback communicator (tests only): contexts created with the same group name by different
// host threads of ONE process exchange through host memory.  It exercises every tensor-parallel code path
// (sharded load, per-rank kernels, all-reduce placement, vocab-parallel arg-max) on a single GPU; production
// uses RCCL.
struct LoopGroup {
    std::mutex mu;
    std::condition_variable cv;
    int size = 0, arrived = 0;
    uint64_t generation = 0;
    std::vector<float> acc;
    std::vector<unsigned char> gather;
    std::vector<float*> os_data;      // one-shot all-reduce buffers of the ranks (same process, same device: plain pointers)
    std::vector<uint32_t*> os_flag;
};
static std::mutex g_groups_mu;
static std::map<std::string, std::shared_ptr<LoopGroup>> g_groups;

struct nvllm_ctx {
    std::shared_ptr<LoopGroup> loop;
    bool null_comm = false;  // projection mode: tp_size > 1 shard shapes, every collective skipped (results meaningless)
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int tp_rank = 0, tp_size = 1;
    ncclComm_t comm = nullptr;
    std::string err;
    // one-shot all-reduce (oneshot.hip), opt-in; empty when not set up
    bool oneshot = false;
    float* os_data = nullptr;     // [2 generations][tp][os_slot] f32, uncached device memory
    uint32_t* os_flag = nullptr;  // [2][tp]
    unsigned* os_done = nullptr;  // push kernel's workgroup counter
    int* os_err = nullptr;        // set by a wait that timed out
    int* os_agree = nullptr;      // [tp] error words gathered at the end of a step: every rank takes the same decision
    size_t os_slot = 0;
    uint64_t os_calls = 0;
    OneShotPeers os_peers;
    std::vector<void*> os_ipc;    // peer mappings opened with hipIpcOpenMemHandle
};

static thread_local std::string g_create_err;

static int fail(nvllm_ctx* ctx, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_err = buf;
    return code;
}

#define HIPCHK(ctx, expr)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(ctx, NVLLM_EHIP, "%s -> %s", #expr, hipGetErrorString(e_)); \
    } while (0)
#define NCCLCHK(ctx, expr)                                                                               \
    do {                                                                                                 \
        ncclResult_t r_ = (expr);                                                                        \
        if (r_ != ncclSuccess) return fail(ctx, NVLLM_ERCCL, "%s -> %s", #expr, ncclGetErrorString(r_)); \
    } while (0)

// ---- collectives: RCCL on the library stream, or the in-process loopback group ----------------------
static int comm_allreduce_sum(nvllm_ctx* ctx, float* buf, size_t n) {
    if (ctx->tp_size == 1 || ctx->null_comm) return NVLLM_OK;
    if (!ctx->loop) {
        NCCLCHK(ctx, ncclAllReduce(buf, buf, n, ncclFloat, ncclSum, ctx->comm, ctx->stream));
        return NVLLM_OK;
    }
    std::vector<float> host(n);
    HIPCHK(ctx, hipMemcpyAsync(host.data(), buf, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    LoopGroup& g = *ctx->loop;
    {
        std::unique_lock<std::mutex> lk(g.mu);
        const uint64_t gen = g.generation;
        if (g.arrived == 0) g.acc.assign(n, 0.f);
        // ranks add in arrival order: like RCCL, the order of the sum is not the TP=1 order
        for (size_t i = 0; i < n; ++i) g.acc[i] += host[i];
        if (++g.arrived == g.size) { g.arrived = 0; ++g.generation; g.cv.notify_all(); }
        else g.cv.wait(lk, [&] { return g.generation != gen; });
        host = g.acc;  // every rank reads the same bytes
    }
    // second phase so nobody re-initialises acc while a slow rank still copies it
    {
        std::unique_lock<std::mutex> lk(g.mu);
        const uint64_t gen = g.generation;
        if (++g.arrived == g.size) { g.arrived = 0; ++g.generation; g.cv.notify_all(); }
        else g.cv.wait(lk, [&] { return g.generation != gen; });
    }
    HIPCHK(ctx, hipMemcpyAsync(buf, host.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return NVLLM_OK;
}

// recv[r*bytes .. ) = rank r's send (send may alias recv + rank*bytes)
static int comm_allgather(nvllm_ctx* ctx, const void* send, void* recv, size_t bytes) {
    if (ctx->null_comm) return NVLLM_OK;  // the rank's own slot already holds its part
    if (!ctx->loop) {
        NCCLCHK(ctx, ncclAllGather(send, recv, bytes, ncclUint8, ctx->comm, ctx->stream));
        return NVLLM_OK;
    }
    std::vector<unsigned char> host(bytes);
    HIPCHK(ctx, hipMemcpyAsync(host.data(), send, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    LoopGroup& g = *ctx->loop;
    std::vector<unsigned char> all;
    {
        std::unique_lock<std::mutex> lk(g.mu);
        const uint64_t gen = g.generation;
        if (g.arrived == 0) g.gather.assign(bytes * g.size, 0);
        memcpy(g.gather.data() + (size_t)ctx->tp_rank * bytes, host.data(), bytes);
        if (++g.arrived == g.size) { g.arrived = 0; ++g.generation; g.cv.notify_all(); }
        else g.cv.wait(lk, [&] { return g.generation != gen; });
        all = g.gather;
    }
    {
        std::unique_lock<std::mutex> lk(g.mu);
        const uint64_t gen = g.generation;
        if (++g.arrived == g.size) { g.arrived = 0; ++g.generation; g.cv.notify_all(); }
        else g.cv.wait(lk, [&] { return g.generation != gen; });
    }
    HIPCHK(ctx, hipMemcpyAsync(recv, all.data(), all.size(), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return NVLLM_OK;
}

// group barrier of the loopback communicator (host threads)
static void loop_barrier(LoopGroup& g) {
    std::unique_lock<std::mutex> lk(g.mu);
    const uint64_t gen = g.generation;
    if (++g.arrived == g.size) { g.arrived = 0; ++g.generation; g.cv.notify_all(); }
    else g.cv.wait(lk, [&] { return g.generation != gen; });
}

// Loopback test double only: its ranks are host threads of ONE process whose streams share a handful of hardware
// queues, so a wait kernel enqueued before a peer's push could sit IN FRONT of that push in the same queue and spin
// until its timeout.  A host barrier between "every rank has enqueued its push" and "anybody enqueues a wait" removes
// that order; the data / flag protocol on the device is unchanged.  RCCL ranks (one process, one GPU each) skip it.
static void oneshot_loopback_order(nvllm_ctx* ctx) {
    if (ctx->loop) loop_barrier(*ctx->loop);
}

static void oneshot_free(nvllm_ctx* ctx) {
    for (void* p : ctx->os_ipc) (void)hipIpcCloseMemHandle(p);
    ctx->os_ipc.clear();
    if (ctx->os_data) (void)hipFree(ctx->os_data);
    if (ctx->os_flag) (void)hipFree(ctx->os_flag);
    if (ctx->os_done) (void)hipFree(ctx->os_done);
    if (ctx->os_err) (void)hipFree(ctx->os_err);
    if (ctx->os_agree) (void)hipFree(ctx->os_agree);
    ctx->os_data = nullptr; ctx->os_flag = nullptr; ctx->os_done = nullptr; ctx->os_err = nullptr; ctx->os_agree = nullptr;
    ctx->oneshot = false; ctx->os_slot = 0; ctx->os_peers = OneShotPeers();
}

// Buffers + peer mappings of the one-shot all-reduce.  Collective: every rank of the group calls it with the same
// slot size.  RCCL ranks exchange HIP IPC handles through ncclAllGather; loopback ranks exchange pointers through the
// group.  Any failure leaves the context on the RCCL / loopback all-reduce (oneshot == false) and is reported in err.
static int oneshot_setup(nvllm_ctx* ctx, size_t slot_floats) {
    if (ctx->tp_size < 2 || ctx->tp_size > 8 || ctx->null_comm) return NVLLM_OK;
    if (ctx->oneshot && ctx->os_slot >= slot_floats) return NVLLM_OK;
    oneshot_free(ctx);
    const int tp = ctx->tp_size;
    slot_floats = (slot_floats + 3) / 4 * 4;
    const size_t data_bytes = (size_t)2 * tp * slot_floats * 4, flag_bytes = (size_t)2 * tp * 4;
    auto alloc = [&](void** p, size_t bytes) {  // uncached: peers' writes must be what this GPU reads next, not an L2 line
        hipError_t e = hipExtMallocWithFlags(p, bytes, hipDeviceMallocUncached);
        if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags(p, bytes, hipDeviceMallocFinegrained); }
        return e;
    };
    bool ok = alloc((void**)&ctx->os_data, data_bytes) == hipSuccess && alloc((void**)&ctx->os_flag, std::max<size_t>(flag_bytes, 256)) == hipSuccess &&
              hipMalloc((void**)&ctx->os_done, 256) == hipSuccess && hipMalloc((void**)&ctx->os_err, 256) == hipSuccess &&
              hipMalloc((void**)&ctx->os_agree, 256) == hipSuccess;
    if (ok) ok = hipMemset(ctx->os_flag, 0, std::max<size_t>(flag_bytes, 256)) == hipSuccess && hipMemset(ctx->os_done, 0, 256) == hipSuccess &&
                 hipMemset(ctx->os_err, 0, 256) == hipSuccess && hipMemset(ctx->os_agree, 0, 256) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    int all_ok = ok ? 1 : 0;
    if (ctx->loop) {
        LoopGroup& g = *ctx->loop;
        {
            std::lock_guard<std::mutex> lk(g.mu);
            g.os_data.resize(tp, nullptr); g.os_flag.resize(tp, nullptr);
            g.os_data[ctx->tp_rank] = ok ? ctx->os_data : nullptr;
            g.os_flag[ctx->tp_rank] = ok ? ctx->os_flag : nullptr;
        }
        loop_barrier(g);
        {
            std::lock_guard<std::mutex> lk(g.mu);
            for (int r = 0; r < tp; ++r) {
                ctx->os_peers.data[r] = g.os_data[r]; ctx->os_peers.flag[r] = g.os_flag[r];
                if (!g.os_data[r] || !g.os_flag[r]) all_ok = 0;
            }
        }
        loop_barrier(g);  // nobody re-registers before everybody has read
    } else {
        // [tp][2] IPC handles through the communicator that exists anyway
        struct Rec { hipIpcMemHandle_t data, flag; int ok; int pad[3]; };
        Rec mine; memset(&mine, 0, sizeof mine);
        mine.ok = ok && hipIpcGetMemHandle(&mine.data, ctx->os_data) == hipSuccess && hipIpcGetMemHandle(&mine.flag, ctx->os_flag) == hipSuccess;
        Rec* d_all = nullptr;
        std::vector<Rec> all(tp);
        if (hipMalloc((void**)&d_all, sizeof(Rec) * tp) != hipSuccess) { oneshot_free(ctx); return fail(ctx, NVLLM_EHIP, "one-shot all-reduce: staging allocation failed"); }
        (void)hipMemcpy(d_all + ctx->tp_rank, &mine, sizeof mine, hipMemcpyHostToDevice);
        ncclResult_t r = ncclAllGather(d_all + ctx->tp_rank, d_all, sizeof(Rec), ncclUint8, ctx->comm, ctx->stream);
        if (r == ncclSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess &&
            hipMemcpy(all.data(), d_all, sizeof(Rec) * tp, hipMemcpyDeviceToHost) == hipSuccess) {
            for (int q = 0; q < tp; ++q) all_ok = all_ok && all[q].ok;
            for (int q = 0; q < tp && all_ok; ++q) {
                if (q == ctx->tp_rank) { ctx->os_peers.data[q] = ctx->os_data; ctx->os_peers.flag[q] = ctx->os_flag; continue; }
                void *pd = nullptr, *pf = nullptr;
                if (hipIpcOpenMemHandle(&pd, all[q].data, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { all_ok = 0; break; }
                ctx->os_ipc.push_back(pd);
                if (hipIpcOpenMemHandle(&pf, all[q].flag, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { all_ok = 0; break; }
                ctx->os_ipc.push_back(pf);
                ctx->os_peers.data[q] = (float*)pd; ctx->os_peers.flag[q] = (uint32_t*)pf;
            }
        } else {
            all_ok = 0;
        }
        // every rank must take the same decision: a rank on the one-shot path would wait for flags nobody raises
        int* d_ok = reinterpret_cast<int*>(d_all);  // reuse the staging buffer (freed below)
        int h_ok = all_ok;
        (void)hipMemcpy(d_ok, &h_ok, 4, hipMemcpyHostToDevice);
        if (ncclAllReduce(d_ok, d_ok, 1, ncclInt, ncclMin, ctx->comm, ctx->stream) != ncclSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess ||
            hipMemcpy(&h_ok, d_ok, 4, hipMemcpyDeviceToHost) != hipSuccess) h_ok = 0;
        all_ok = h_ok;
        (void)hipFree(d_all);
    }
    if (!all_ok) {
        oneshot_free(ctx);
        return fail(ctx, NVLLM_OK, "one-shot all-reduce not available on this group (allocation or IPC mapping failed): using the communicator's all-reduce");
    }
    ctx->os_slot = slot_floats;
    ctx->os_calls = 0;
    ctx->oneshot = true;
    return NVLLM_OK;
}

extern "C" const char* nvllm_last_error(const nvllm_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

extern "C" int nvllm_rccl_unique_id(void* out_id) {
    static_assert(sizeof(ncclUniqueId) <= NVLLM_RCCL_ID_BYTES, "id size");
    if (!out_id) return fail(nullptr, NVLLM_EINVAL, "out_id is NULL");
    ncclUniqueId id;
    NCCLCHK(nullptr, ncclGetUniqueId(&id));
    memset(out_id, 0, NVLLM_RCCL_ID_BYTES);
    memcpy(out_id, &id, sizeof id);
    return NVLLM_OK;
}

extern "C" int nvllm_ctx_create(int device_ordinal, int tp_rank, int tp_size, const void* rccl_id, nvllm_ctx** out) {
    if (!out) return fail(nullptr, NVLLM_EINVAL, "out is NULL");
    if (tp_size < 1) return fail(nullptr, NVLLM_EINVAL, "tp_size must be >= 1");
    // src/tp.rs:24-29 folds a bad rank to 0, which is harmless there (no communicator exists); with a real RCCL
    // group two "rank 0" processes would hang the rendezvous, so an out-of-range rank is an error here and the
    // fold stays in the host mirror only (TPConfig.from_env)
    if (tp_size > 1 && (tp_rank < 0 || tp_rank >= tp_size))
        return fail(nullptr, NVLLM_EINVAL, "tp_rank %d outside [0, %d)", tp_rank, tp_size);
    if (tp_rank < 0 || tp_rank >= tp_size) tp_rank = 0;
    int ndev = 0;
    HIPCHK(nullptr, hipGetDeviceCount(&ndev));
    if (device_ordinal < 0 || device_ordinal >= ndev)
        return fail(nullptr, NVLLM_EINVAL, "device %d not present (%d devices)", device_ordinal, ndev);
    HIPCHK(nullptr, hipSetDevice(device_ordinal));
    nvllm_ctx* c = new nvllm_ctx();
    c->device = device_ordinal;
    c->tp_rank = tp_rank;
    c->tp_size = tp_size;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e != hipSuccess) {
        delete c;
        return fail(nullptr, NVLLM_EHIP, "stream/event create -> %s", hipGetErrorString(e));
    }
    if (tp_size > 1) {
        if (!rccl_id) { delete c; return fail(nullptr, NVLLM_EINVAL, "tp_size>1 needs an RCCL unique id"); }
        ncclUniqueId id;
        memcpy(&id, rccl_id, sizeof id);
        ncclResult_t r = ncclCommInitRank(&c->comm, tp_size, id, tp_rank);
        if (r != ncclSuccess) {
            delete c;
            return fail(nullptr, NVLLM_ERCCL, "ncclCommInitRank -> %s", ncclGetErrorString(r));
        }
    }
    *out = c;
    return NVLLM_OK;
}

extern "C" int nvllm_ctx_create_loopback(int device_ordinal, int tp_rank, int tp_size, const char* group, nvllm_ctx** out) {
    if (!out || !group || tp_size < 1 || tp_rank < 0 || tp_rank >= tp_size) return fail(nullptr, NVLLM_EINVAL, "bad loopback arguments");
    int rc = nvllm_ctx_create(device_ordinal, 0, 1, nullptr, out);
    if (rc) return rc;
    (*out)->tp_rank = tp_rank;
    (*out)->tp_size = tp_size;
    std::lock_guard<std::mutex> lk(g_groups_mu);
    auto& g = g_groups[group];
    if (!g) { g = std::make_shared<LoopGroup>(); g->size = tp_size; }
    if (g->size != tp_size) { nvllm_ctx_destroy(*out); *out = nullptr; return fail(nullptr, NVLLM_EINVAL, "loopback group size mismatch"); }
    (*out)->loop = g;
    return NVLLM_OK;
}

// Projection mode (bench.py tp_projection): one rank of a tp_size group with NO communicator.  The model takes the
// rank's shard shapes and runs the rank's kernels; all-reduce / all-gather are skipped, so the numbers a step produces
// are meaningless -- only its duration (the per-rank compute time of a TP step) is.
extern "C" int nvllm_ctx_create_null_comm(int device_ordinal, int tp_rank, int tp_size, nvllm_ctx** out) {
    if (!out || tp_size < 1 || tp_rank < 0 || tp_rank >= tp_size) return fail(nullptr, NVLLM_EINVAL, "bad null-comm arguments");
    int rc = nvllm_ctx_create(device_ordinal, 0, 1, nullptr, out);
    if (rc) return rc;
    (*out)->tp_rank = tp_rank;
    (*out)->tp_size = tp_size;
    (*out)->null_comm = true;
    return NVLLM_OK;
}

extern "C" int nvllm_ctx_destroy(nvllm_ctx* c) {
    if (!c) return NVLLM_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    oneshot_free(c);
    if (c->comm) ncclCommDestroy(c->comm);
    (void)hipEventDestroy(c->ev0);
    (void)hipEventDestroy(c->ev1);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return NVLLM_OK;
}
extern "C" int nvllm_ctx_synchronize(nvllm_ctx* c) {
    if (!c) return NVLLM_EINVAL;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return NVLLM_OK;
}
extern "C" void* nvllm_ctx_stream(nvllm_ctx* c) { return c ? (void*)c->stream : nullptr; }
extern "C" int nvllm_ctx_tp_rank(const nvllm_ctx* c) { return c ? c->tp_rank : 0; }
extern "C" int nvllm_ctx_tp_size(const nvllm_ctx* c) { return c ? c->tp_size : 1; }
extern "C" int nvllm_timer_start(nvllm_ctx* c) {
    if (!c) return NVLLM_EINVAL;
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    return NVLLM_OK;
}
extern "C" int nvllm_timer_stop(nvllm_ctx* c, float* ms) {
    if (!c || !ms) return NVLLM_EINVAL;
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    HIPCHK(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return NVLLM_OK;
}

// device memory helpers -------------------------------------------------------------------------------
extern "C" int nvllm_dev_alloc(nvllm_ctx* c, size_t bytes, void** out) {
    if (!c || !out) return NVLLM_EINVAL;
    HIPCHK(c, hipMalloc(out, bytes ? bytes : 16));
    return NVLLM_OK;
}
extern "C" int nvllm_dev_free(nvllm_ctx* c, void* p) {
    if (!c) return NVLLM_EINVAL;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(p));
    return NVLLM_OK;
}
extern "C" int nvllm_dev_upload(nvllm_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c) return NVLLM_EINVAL;
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return NVLLM_OK;
}
extern "C" int nvllm_dev_download(nvllm_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c) return NVLLM_EINVAL;
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return NVLLM_OK;
}

template <typename T>
static int dmalloc(nvllm_ctx* c, T** p, size_t count) {
    HIPCHK(c, hipMalloc((void**)p, std::max<size_t>(count * sizeof(T), 16)));
    return NVLLM_OK;
}

// ---------------------------------------------------------------------------------------------------
// model
// ---------------------------------------------------------------------------------------------------
constexpr int kFusedMaxRows = 128;  // the fused decode path (deferred norms, row-parallel epilogues) handles up to this many rows
constexpr int kMaxPending = 4;   // decode steps that may be enqueued before one is collected
constexpr int kAttnMaxParts = 64;  // split-KV partitions per sequence (partition grows with context beyond 8K)

struct LayerW {
    PackedW qkv, o, gu, down;
    float *ln1 = nullptr, *ln2 = nullptr, *qn = nullptr, *kn = nullptr;
};

struct SeqState {
    int slot = -1;
    int cached = 0;  // tokens whose K/V are in the cache
    std::vector<int> blocks;
};

struct nvllm_model {
    nvllm_ctx* ctx = nullptr;
    nvllm_qwen3_config cfg{};
    int H = 0, hd = 0, L = 0, nh_l = 0, kv_l = 0, I_l = 0, V_l = 0, gqa = 1;
    bf16_bits* embed = nullptr;  // [V][H] row-major bf16 (gather table, replicated)
    PackedW lm_head;             // [V_l][H]
    float* norm = nullptr;
    std::vector<LayerW> layers;
    std::unordered_map<std::string, bool> loaded;
    bool finalized = false;
    float *cosv = nullptr, *sinv = nullptr;
    int rope_len = 0;

    // KV pool
    int num_blocks = 0, max_seqs = 0, max_blocks = 0, max_rows = 0;
    std::vector<f16_bits*> kcache, vcache;
    std::vector<uint8_t*> vlocache;  // 24-bit V (opt_kv_v_bits == 24): one residual byte per V element, else empty
    std::vector<uint8_t*> klocache;  // 24-bit K (opt_kv_k_bits == 24, only with 24-bit V): the same for K
    std::vector<int> free_blocks, free_slots;
    std::unordered_map<int64_t, SeqState> seqs;
    int* d_block_tables = nullptr;
    std::vector<int> h_block_tables;

    // step buffers
    uint32_t* d_ids = nullptr;
    int *d_pos = nullptr, *d_slot = nullptr, *d_tile_row0 = nullptr, *d_tile_nrows = nullptr, *d_tile_slot = nullptr,
        *d_last_rows = nullptr, *d_tile_order = nullptr, *d_tile_last = nullptr, *d_group_order = nullptr;
    float *resid = nullptr, *slabs = nullptr, *qbuf = nullptr, *logits = nullptr, *d_maxval = nullptr, *red = nullptr;
    bf16_bits *xh = nullptr, *xl = nullptr, *xh2 = nullptr, *xl2 = nullptr, *ctxh = nullptr, *ctxl = nullptr;
    float *ssqA = nullptr, *ssqB = nullptr;  // deferred-norm partial sums of squares [groups][kFusedMaxRows]
    uint64_t* d_keys = nullptr;              // device sampling: per-row RNG keys and temperatures
    float* d_temps = nullptr;
    // The streaming GEMM's in-launch split-K combine (epilogues 1-3) is OFF by default: measured slower than slabs + a
    // consumer launch on MI355X (Qwen3-8B batch 64: 8.6 vs 6.4 ms/step; 32B TP=8 shard: 8.9 vs 6.5 ms) -- one workgroup
    // per n-group reads every slab behind an agent-scope release of all the others.  nvllm_debug_set_option turns it on.
    int opt_stream_combine = 0;
    int opt_oneshot_allreduce = 0;  // TP decode: one-shot all-reduce instead of the communicator's (set before kv_alloc; opt-in)
    int opt_kv_v_bits = 16;            // 24: V as f16 + e5m2 residual (13..14 significant bits, V bytes x1.5); read at kv_alloc
    int opt_kv_k_bits = 16;            // 24: K likewise (needs kv_v_bits = 24: K/V bytes x1.5 together); read at kv_alloc
    int opt_oneshot_skip_push = 0;     // test hook: this rank "forgets" its next N pushes (the give-up path of its peers' waits)
    int opt_oneshot_spins = 20000000;  // bound of the one-shot wait kernel's poll (~ seconds): a missing peer is an error code
    int opt_no_fused = 0, opt_no_xpack = 0, opt_no_rowpar = 0;  // A/B switches (nvllm_debug_set_option): force the generic paths
    int opt_no_attn_prologue = 0;  // A/B: decode q/k-norm + RoPE + KV write in their own row kernel, not in the attention prologue
    int opt_tile_fuse_qk = 1;    // QKV tile GEMM with the q/k-norm + RoPE + KV-write epilogue (head_dim 128, 256-wide blocks)
    int opt_tile_min_wgs = 192;  // prefill tile GEMM: smallest grid it is used for (256-row tiles need rows to fill 256 CUs)
    unsigned* tickets = nullptr;             // arrival counters of the streaming GEMM's in-launch combine (zero between launches)
    float* qkv_out = nullptr;                // complete QKV sums of the streaming GEMM (its combine reads the slabs in m->slabs)
    uint32_t* d_next = nullptr;
    float *attn_po = nullptr, *attn_pml = nullptr;  // split-KV partials [max_seqs][nh_l][kAttnMaxParts][hd] / [..][2]
    int attn_part_tiles = 4, attn_parts_max = 1;   // this step's split geometry (decode only)
    float* part_val = nullptr;  // fused LM-head arg-max partials [V_l/16][max_seqs]
    int* part_idx = nullptr;
    unsigned long long* argmax_scratch = nullptr;  // ticket + per-row keys of argmax_parts_kernel
    bool want_logits = false;   // this step stores the last-row logits
    int cur_n = 0;              // sequences of the step in flight
    size_t slab_floats = 0;
    void* h_stage = nullptr;  // pinned
    size_t h_stage_bytes = 0;
    float *tap_h = nullptr, *tap_res = nullptr;
    bool taps = false;
    int tap_rows = 0;

    // last step (for nvllm_decode_next / byte accounting)
    std::vector<int64_t> last_ids;
    std::vector<int> last_lens;
    int64_t last_bytes = 0;
    bool decode_resident = false;  // device metadata describes a pure-decode batch == last_ids
    // pipelined decode (nvllm_decode_enqueue / _collect)
    uint32_t* pin_ids = nullptr;
    hipEvent_t pend_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<int> pending;
    int pend_next = 0;

    // diagnostic build (-DNVLLM_STAMPS): in-kernel time stamps of every launch of the fused decode path
    unsigned long long* stamps = nullptr;
    bool stamps_on = false;
    int stamp_launch = 0;
    int64_t tile_launches = 0;  // projections run by the tile GEMM so far (tests assert the path they force was taken)

    // per-kernel-class HIP-event timing (bench roofline leg); 0 = off
    int prof_kind = 0;
    std::vector<hipEvent_t> prof_ev;
    size_t prof_used = 0;
};

#ifdef NVLLM_STAMPS
constexpr int kStampLaunches = 160;               // launches recorded per step (fused decode path: 5 per layer)
constexpr size_t kStampStride = (size_t)1024 * 16 * 8;  // u64 per launch: [<= 1024 workgroups][16 waves][8 points]
#define STAMPS(args_, m_)                                                                                        \
    do {                                                                                                         \
        if ((m_)->stamps_on && (m_)->stamp_launch < kStampLaunches) (args_).stamps = (m_)->stamps + (size_t)((m_)->stamp_launch++) * kStampStride; \
    } while (0)
#else
#define STAMPS(args_, m_) do { } while (0)
#endif

enum { PROF_ATTN = 1, PROF_GEMM = 2, PROF_NORM = 3, PROF_QK = 4, PROF_SILU = 5, PROF_LMHEAD = 6, PROF_EMPTY = 7 };

// bracket one launch with HIP events on the library stream when its class is being profiled
#define PROF(m_, kind_, expr_)                                                           \
    do {                                                                                  \
        if ((m_)->prof_kind == (kind_)) {                                                 \
            if ((m_)->prof_used + 2 > (m_)->prof_ev.size()) {                             \
                for (int i_ = 0; i_ < 256; ++i_) {                                        \
                    hipEvent_t evn_;                                                      \
                    HIPCHK((m_)->ctx, hipEventCreate(&evn_));                              \
                    (m_)->prof_ev.push_back(evn_);                                        \
                }                                                                         \
            }                                                                             \
            HIPCHK((m_)->ctx, hipEventRecord((m_)->prof_ev[(m_)->prof_used], (m_)->ctx->stream));     \
            HIPCHK((m_)->ctx, expr_);                                                     \
            HIPCHK((m_)->ctx, hipEventRecord((m_)->prof_ev[(m_)->prof_used + 1], (m_)->ctx->stream)); \
            (m_)->prof_used += 2;                                                         \
        } else {                                                                          \
            HIPCHK((m_)->ctx, expr_);                                                     \
        }                                                                                 \
    } while (0)

static int model_fail(nvllm_model* m, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    m->ctx->err = buf;
    return code;
}

static const char* kLayerTensors[] = {"self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight",
                                      "self_attn.o_proj.weight", "mlp.gate_proj.weight",    "mlp.up_proj.weight",
                                      "mlp.down_proj.weight",    "input_layernorm.weight",  "post_attention_layernorm.weight",
                                      "self_attn.q_norm.weight", "self_attn.k_norm.weight"};

extern "C" int nvllm_model_create(nvllm_ctx* ctx, const nvllm_qwen3_config* cfg, nvllm_model** out) {
    if (!ctx || !cfg || !out) return fail(ctx, NVLLM_EINVAL, "NULL argument");
    const int tp = ctx->tp_size;
    const nvllm_qwen3_config& c = *cfg;
    if (c.vocab_size <= 0 || c.hidden_size <= 0 || c.num_hidden_layers <= 0 || c.num_attention_heads <= 0 ||
        c.num_key_value_heads <= 0 || c.intermediate_size <= 0 || c.head_dim <= 0)
        return fail(ctx, NVLLM_EINVAL, "config has non-positive sizes");
    if (c.head_dim != 64 && c.head_dim != 128) return fail(ctx, NVLLM_EINVAL, "head_dim %d unsupported (64 or 128)", c.head_dim);
    if (c.num_attention_heads % c.num_key_value_heads) return fail(ctx, NVLLM_EINVAL, "num_heads %% num_kv_heads != 0");
    if (c.num_attention_heads / c.num_key_value_heads > 16) return fail(ctx, NVLLM_EINVAL, "GQA group > 16 unsupported");
    if (c.num_key_value_heads % tp || c.num_attention_heads % tp)
        return fail(ctx, NVLLM_EINVAL, "heads (%d q / %d kv) not divisible by tp_size %d", c.num_attention_heads, c.num_key_value_heads, tp);
    if (c.hidden_size % 128 || c.hidden_size > 8192) return fail(ctx, NVLLM_EINVAL, "hidden_size must be a multiple of 128 and <= 8192");
    if ((c.num_attention_heads / tp * c.head_dim) % 128) return fail(ctx, NVLLM_EINVAL, "per-rank q width (heads/tp * head_dim) must be a multiple of 128");
    if (c.intermediate_size % (128 * tp)) return fail(ctx, NVLLM_EINVAL, "intermediate_size must be a multiple of 128*tp");
    if (c.vocab_size % (16 * tp)) return fail(ctx, NVLLM_EINVAL, "vocab_size must be a multiple of 16*tp");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    nvllm_model* m = new nvllm_model();
    m->ctx = ctx;
    m->cfg = c;
    m->H = c.hidden_size; m->hd = c.head_dim; m->L = c.num_hidden_layers;
    m->nh_l = c.num_attention_heads / tp; m->kv_l = c.num_key_value_heads / tp;
    m->I_l = c.intermediate_size / tp; m->V_l = c.vocab_size / tp;
    m->gqa = c.num_attention_heads / c.num_key_value_heads;
    const int H = m->H, hd = m->hd;
    auto alloc_w = [&](PackedW& w, int N, int K) -> int {
        w.N = N; w.K = K;
        HIPCHK(ctx, hipMalloc((void**)&w.data, w.bytes()));
        return NVLLM_OK;
    };
    int rc = dmalloc(ctx, &m->embed, (size_t)c.vocab_size * H);
    if (!rc) rc = alloc_w(m->lm_head, m->V_l, H);
    if (!rc) rc = dmalloc(ctx, &m->norm, H);
    m->layers.resize(m->L);
    for (int l = 0; l < m->L && !rc; ++l) {
        LayerW& w = m->layers[l];
        rc = alloc_w(w.qkv, (m->nh_l + 2 * m->kv_l) * hd, H);
        if (!rc) rc = alloc_w(w.o, H, m->nh_l * hd);
        if (!rc) rc = alloc_w(w.gu, 2 * m->I_l, H);
        if (!rc) rc = alloc_w(w.down, H, m->I_l);
        if (!rc) rc = dmalloc(ctx, &w.ln1, H);
        if (!rc) rc = dmalloc(ctx, &w.ln2, H);
        if (!rc) rc = dmalloc(ctx, &w.qn, hd);
        if (!rc) rc = dmalloc(ctx, &w.kn, hd);
    }
    if (rc) { nvllm_model_destroy(m); return rc; }
    *out = m;
    return NVLLM_OK;
}

static void free_kv(nvllm_model* m) {
    for (auto p : m->kcache) (void)hipFree(p);
    for (auto p : m->vcache) (void)hipFree(p);
    for (auto p : m->vlocache) (void)hipFree(p);
    for (auto p : m->klocache) (void)hipFree(p);
    m->kcache.clear(); m->vcache.clear(); m->vlocache.clear(); m->klocache.clear();
    void* ptrs[] = {m->d_block_tables, m->d_ids, m->d_pos, m->d_slot, m->d_tile_row0, m->d_tile_nrows, m->d_tile_slot,
                    m->d_last_rows, m->d_tile_order, m->d_tile_last, m->d_group_order, m->resid, m->slabs, m->qbuf, m->logits, m->d_maxval, m->red, m->xh, m->xl, m->xh2, m->xl2, m->ctxh, m->ctxl, m->ssqA, m->ssqB, m->d_next,
                    m->part_val, m->part_idx, m->argmax_scratch, m->attn_po, m->attn_pml, m->tap_h, m->tap_res, m->cosv, m->sinv, m->tickets, m->qkv_out, m->d_keys, m->d_temps};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    m->d_block_tables = nullptr; m->d_ids = nullptr; m->d_pos = m->d_slot = m->d_tile_row0 = m->d_tile_nrows = m->d_tile_slot = m->d_last_rows = m->d_tile_order = m->d_tile_last = m->d_group_order = nullptr;
    m->resid = m->slabs = m->qbuf = m->logits = m->d_maxval = m->red = nullptr; m->xh = m->xl = m->xh2 = m->xl2 = m->ctxh = m->ctxl = nullptr; m->ssqA = m->ssqB = nullptr; m->d_next = nullptr; m->part_val = nullptr; m->part_idx = nullptr; m->argmax_scratch = nullptr; m->attn_po = m->attn_pml = nullptr;
    m->tap_h = m->tap_res = nullptr; m->cosv = m->sinv = nullptr; m->tickets = nullptr; m->qkv_out = nullptr; m->d_keys = nullptr; m->d_temps = nullptr;
    if (m->h_stage) (void)hipHostFree(m->h_stage);
    m->h_stage = nullptr;
    m->num_blocks = 0;
}

extern "C" int nvllm_model_destroy(nvllm_model* m) {
    if (!m) return NVLLM_OK;
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    free_kv(m);
    if (m->pin_ids) (void)hipHostFree(m->pin_ids);
    for (hipEvent_t e : m->pend_ev) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : m->prof_ev) (void)hipEventDestroy(e);
    (void)hipFree(m->embed); (void)hipFree(m->lm_head.data); (void)hipFree(m->norm);
    for (auto& w : m->layers) {
        (void)hipFree(w.qkv.data); (void)hipFree(w.o.data); (void)hipFree(w.gu.data); (void)hipFree(w.down.data);
        (void)hipFree(w.ln1); (void)hipFree(w.ln2); (void)hipFree(w.qn); (void)hipFree(w.kn);
    }
    delete m;
    return NVLLM_OK;
}

// Where an HF tensor lands: a region (rows r0.., cols c0..) of the full tensor goes to rows
// [dst_row0, dst_row0+rows) of a packed matrix / or to a plain f32 / bf16 row-major buffer.
struct Target {
    enum Kind { PACKED, F32VEC, BF16ROWS } kind = PACKED;
    PackedW* w = nullptr;
    float* f = nullptr;
    bf16_bits* b = nullptr;
    int64_t full_rows = 0, full_cols = 0;  // HF shape
    int64_t r0 = 0, c0 = 0, rows = 0, cols = 0;  // shard region
    int dst_row0 = 0;
    int ileave = -1;  // gate (0) / up (1): interleaved 16-row tiles
    int synth_kind = kSynthMatrix;
    int synth_axis = kSynthAxisNone;  // where the hidden channel sits in the FULL tensor (heavy synthetic profile: outlier channels)
};

static bool resolve(nvllm_model* m, const char* name, Target& t) {
    const nvllm_qwen3_config& c = m->cfg;
    const int rank = m->ctx->tp_rank;
    const int64_t H = m->H, hd = m->hd, nh = c.num_attention_heads, kv = c.num_key_value_heads, I = c.intermediate_size, V = c.vocab_size;
    if (!strcmp(name, "model.embed_tokens.weight")) {
        t.kind = Target::BF16ROWS; t.b = m->embed; t.full_rows = V; t.full_cols = H; t.rows = V; t.cols = H; t.synth_axis = kSynthAxisCol; return true;
    }
    if (!strcmp(name, "lm_head.weight")) {
        t.w = &m->lm_head; t.full_rows = V; t.full_cols = H; t.r0 = (int64_t)rank * m->V_l; t.rows = m->V_l; t.cols = H; return true;
    }
    if (!strcmp(name, "model.norm.weight")) {
        t.kind = Target::F32VEC; t.f = m->norm; t.full_rows = 1; t.full_cols = H; t.rows = 1; t.cols = H; t.synth_kind = kSynthNorm; t.synth_axis = kSynthAxisCol; return true;
    }
    int l = -1, off = 0;
    if (sscanf(name, "model.layers.%d.%n", &l, &off) != 1 || l < 0 || l >= m->L || off == 0) return false;
    const char* s = name + off;
    LayerW& w = m->layers[l];
    const int64_t qr = (int64_t)m->nh_l * hd, kr = (int64_t)m->kv_l * hd;
    if (!strcmp(s, "self_attn.q_proj.weight")) { t.w = &w.qkv; t.full_rows = nh * hd; t.full_cols = H; t.r0 = rank * qr; t.rows = qr; t.cols = H; t.dst_row0 = 0; return true; }
    if (!strcmp(s, "self_attn.k_proj.weight")) { t.w = &w.qkv; t.full_rows = kv * hd; t.full_cols = H; t.r0 = rank * kr; t.rows = kr; t.cols = H; t.dst_row0 = (int)qr; return true; }
    if (!strcmp(s, "self_attn.v_proj.weight")) { t.w = &w.qkv; t.full_rows = kv * hd; t.full_cols = H; t.r0 = rank * kr; t.rows = kr; t.cols = H; t.dst_row0 = (int)(qr + kr); return true; }
    if (!strcmp(s, "self_attn.o_proj.weight")) { t.w = &w.o; t.full_rows = H; t.full_cols = nh * hd; t.c0 = rank * qr; t.rows = H; t.cols = qr; t.synth_axis = kSynthAxisRow; return true; }
    if (!strcmp(s, "mlp.gate_proj.weight")) { t.w = &w.gu; t.full_rows = I; t.full_cols = H; t.r0 = (int64_t)rank * m->I_l; t.rows = m->I_l; t.cols = H; t.dst_row0 = 0; t.ileave = 0; return true; }
    if (!strcmp(s, "mlp.up_proj.weight")) { t.w = &w.gu; t.full_rows = I; t.full_cols = H; t.r0 = (int64_t)rank * m->I_l; t.rows = m->I_l; t.cols = H; t.dst_row0 = 0; t.ileave = 1; return true; }
    if (!strcmp(s, "mlp.down_proj.weight")) { t.w = &w.down; t.full_rows = H; t.full_cols = I; t.c0 = (int64_t)rank * m->I_l; t.rows = H; t.cols = m->I_l; t.synth_axis = kSynthAxisRow; return true; }
    t.kind = Target::F32VEC; t.synth_kind = kSynthNorm; t.full_rows = 1; t.rows = 1;
    if (!strcmp(s, "input_layernorm.weight")) { t.f = w.ln1; t.full_cols = t.cols = H; t.synth_axis = kSynthAxisCol; return true; }
    if (!strcmp(s, "post_attention_layernorm.weight")) { t.f = w.ln2; t.full_cols = t.cols = H; t.synth_axis = kSynthAxisCol; return true; }
    t.synth_kind = kSynthQkNorm;
    if (!strcmp(s, "self_attn.q_norm.weight")) { t.f = w.qn; t.full_cols = t.cols = hd; return true; }
    if (!strcmp(s, "self_attn.k_norm.weight")) { t.f = w.kn; t.full_cols = t.cols = hd; return true; }
    return false;
}

static inline uint16_t host_f32_to_bf16(float f) {  // round-to-nearest-even (exact for bf16-valued inputs)
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

extern "C" int nvllm_model_load_tensor(nvllm_model* m, const char* hf_name, const void* host_data, int dtype,
                                       const int64_t* shape, int ndim) {
    if (!m || !hf_name || !host_data || !shape) return fail(m ? m->ctx : nullptr, NVLLM_EINVAL, "NULL argument");
    nvllm_ctx* ctx = m->ctx;
    if (dtype != NVLLM_DTYPE_F32 && dtype != NVLLM_DTYPE_BF16) return fail(ctx, NVLLM_EINVAL, "dtype %d unsupported", dtype);
    Target t;
    if (!resolve(m, hf_name, t)) return fail(ctx, NVLLM_EINVAL, "unknown tensor name '%s'", hf_name);
    int64_t rows = ndim == 2 ? shape[0] : 1, cols = ndim == 2 ? shape[1] : (ndim == 1 ? shape[0] : -1);
    if (ndim == 2 && t.kind == Target::F32VEC && shape[0] == 1) { rows = 1; cols = shape[1]; }  // [1,H] norm weight (qwen3.rs:180)
    if (cols != t.full_cols || rows != t.full_rows)
        return fail(ctx, NVLLM_EINVAL, "tensor '%s': shape mismatch (got %lld x %lld, want %lld x %lld)", hf_name,
                    (long long)rows, (long long)cols, (long long)t.full_rows, (long long)t.full_cols);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)t.rows * t.cols;
    if (t.kind == Target::F32VEC) {
        std::vector<float> tmp(n);
        for (size_t i = 0; i < n; ++i) {
            if (dtype == NVLLM_DTYPE_F32) tmp[i] = ((const float*)host_data)[i];
            else { uint32_t u = (uint32_t)((const uint16_t*)host_data)[i] << 16; memcpy(&tmp[i], &u, 4); }
        }
        HIPCHK(ctx, hipMemcpy(t.f, tmp.data(), n * 4, hipMemcpyHostToDevice));
    } else {
        std::vector<uint16_t> tmp(n);
        for (int64_t r = 0; r < t.rows; ++r)
            for (int64_t c = 0; c < t.cols; ++c) {
                const size_t si = (size_t)(t.r0 + r) * t.full_cols + (t.c0 + c);
                tmp[(size_t)r * t.cols + c] = dtype == NVLLM_DTYPE_F32 ? host_f32_to_bf16(((const float*)host_data)[si])
                                                                        : ((const uint16_t*)host_data)[si];
            }
        if (t.kind == Target::BF16ROWS) {
            HIPCHK(ctx, hipMemcpy(t.b, tmp.data(), n * 2, hipMemcpyHostToDevice));
        } else {
            if (t.cols != t.w->K) return fail(ctx, NVLLM_EINVAL, "internal: K mismatch for %s", hf_name);
            bf16_bits* dtmp = nullptr;
            HIPCHK(ctx, hipMalloc((void**)&dtmp, n * 2));
            hipError_t e = hipMemcpy(dtmp, tmp.data(), n * 2, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = launch_pack_rows(*t.w, t.dst_row0, (int)t.rows, dtmp, t.cols, t.ileave, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            (void)hipFree(dtmp);
            HIPCHK(ctx, e);
        }
    }
    m->loaded[hf_name] = true;
    return NVLLM_OK;
}

static int synth_one(nvllm_model* m, const char* name, uint64_t seed, int profile) {
    nvllm_ctx* ctx = m->ctx;
    Target t;
    if (!resolve(m, name, t)) return fail(ctx, NVLLM_EINVAL, "internal: bad tensor name %s", name);
    SynthSpec sp;
    sp.name_hash = synth_hash_name(name, seed);
    sp.kind = t.synth_kind; sp.profile = profile; sp.axis = t.synth_axis; sp.cols = t.full_cols;
    synth_set_outliers(sp, seed, m->H);
    if (t.kind == Target::F32VEC) {
        HIPCHK(ctx, launch_synth_rowmajor_f32(t.f, sp, 0, t.cols, ctx->stream));
    } else if (t.kind == Target::BF16ROWS) {
        HIPCHK(ctx, launch_synth_rowmajor_bf16(t.b, sp, 0, t.rows * t.cols, ctx->stream));
    } else {
        HIPCHK(ctx, launch_synth_packed(*t.w, t.dst_row0, (int)t.rows, sp, t.r0, t.c0, t.full_cols, t.ileave, ctx->stream));
    }
    m->loaded[name] = true;
    return NVLLM_OK;
}

// Host-only: which region of the FULL HF tensor `hf_name` rank `tp_rank` of `tp_size` owns
// (row0, col0, rows, cols).  Same table nvllm_model_load_tensor / fill_synthetic use; no GPU needed.
extern "C" int nvllm_tp_shard(const nvllm_qwen3_config* cfg, int tp_size, int tp_rank, const char* hf_name,
                              int64_t out_region[4]) {
    if (!cfg || !hf_name || !out_region || tp_size < 1 || tp_rank < 0 || tp_rank >= tp_size) return NVLLM_EINVAL;
    if (cfg->num_attention_heads % tp_size || cfg->num_key_value_heads % tp_size || cfg->intermediate_size % tp_size ||
        cfg->vocab_size % tp_size)
        return NVLLM_EINVAL;
    nvllm_ctx fake_ctx;
    fake_ctx.tp_rank = tp_rank;
    fake_ctx.tp_size = tp_size;
    nvllm_model fake;
    fake.ctx = &fake_ctx;
    fake.cfg = *cfg;
    fake.H = cfg->hidden_size; fake.hd = cfg->head_dim; fake.L = cfg->num_hidden_layers;
    fake.nh_l = cfg->num_attention_heads / tp_size; fake.kv_l = cfg->num_key_value_heads / tp_size;
    fake.I_l = cfg->intermediate_size / tp_size; fake.V_l = cfg->vocab_size / tp_size;
    fake.layers.resize(fake.L);
    Target t;
    if (!resolve(&fake, hf_name, t)) return NVLLM_EINVAL;
    out_region[0] = t.r0; out_region[1] = t.c0; out_region[2] = t.rows; out_region[3] = t.cols;
    return NVLLM_OK;
}

extern "C" int nvllm_model_fill_synthetic_profile(nvllm_model* m, uint64_t seed, int profile) {
    if (!m) return NVLLM_EINVAL;
    if (profile != 0 && profile != 1) return fail(m->ctx, NVLLM_EINVAL, "synthetic profile %d unknown (0 benign, 1 heavy)", profile);
    HIPCHK(m->ctx, hipSetDevice(m->ctx->device));
    int rc = synth_one(m, "model.embed_tokens.weight", seed, profile);
    if (!rc) rc = synth_one(m, "lm_head.weight", seed, profile);
    if (!rc) rc = synth_one(m, "model.norm.weight", seed, profile);
    char name[160];
    for (int l = 0; l < m->L && !rc; ++l)
        for (const char* s : kLayerTensors) {
            snprintf(name, sizeof name, "model.layers.%d.%s", l, s);
            rc = synth_one(m, name, seed, profile);
            if (rc) break;
        }
    if (rc) return rc;
    HIPCHK(m->ctx, hipStreamSynchronize(m->ctx->stream));
    return NVLLM_OK;
}
extern "C" int nvllm_model_fill_synthetic(nvllm_model* m, uint64_t seed) { return nvllm_model_fill_synthetic_profile(m, seed, 0); }

extern "C" int nvllm_model_finalize(nvllm_model* m) {
    if (!m) return NVLLM_EINVAL;
    nvllm_ctx* ctx = m->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    char name[160];
    if (!m->loaded.count("model.embed_tokens.weight")) return fail(ctx, NVLLM_ESTATE, "missing tensor model.embed_tokens.weight");
    if (!m->loaded.count("model.norm.weight")) return fail(ctx, NVLLM_ESTATE, "missing tensor model.norm.weight");
    for (int l = 0; l < m->L; ++l)
        for (const char* s : kLayerTensors) {
            snprintf(name, sizeof name, "model.layers.%d.%s", l, s);
            if (!m->loaded.count(name)) return fail(ctx, NVLLM_ESTATE, "missing tensor %s", name);
        }
    if (!m->loaded.count("lm_head.weight")) {
        // tied embeddings: LM head = this rank's vocab rows of the embedding table
        HIPCHK(ctx, launch_pack_rows(m->lm_head, 0, m->V_l, m->embed + (size_t)ctx->tp_rank * m->V_l * m->H, m->H, -1, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    m->finalized = true;
    return NVLLM_OK;
}

extern "C" int64_t nvllm_model_weight_bytes(const nvllm_model* m) {
    if (!m) return 0;
    int64_t b = (int64_t)m->lm_head.bytes() + (int64_t)m->H * 4;
    for (const auto& w : m->layers)
        b += (int64_t)(w.qkv.bytes() + w.o.bytes() + w.gu.bytes() + w.down.bytes()) + (int64_t)(2 * m->H + 2 * m->hd) * 4;
    return b;
}

// ---------------------------------------------------------------------------------------------------
// KV pool
// ---------------------------------------------------------------------------------------------------
extern "C" int nvllm_kv_alloc(nvllm_model* m, int num_blocks, int block_size, int max_seqs, int max_batched_tokens) {
    if (!m) return NVLLM_EINVAL;
    nvllm_ctx* ctx = m->ctx;
    if (block_size != kBlockTokens) return fail(ctx, NVLLM_EINVAL, "block_size must be %d (src/engine/sequence.rs:35)", kBlockTokens);
    if (num_blocks < 1 || max_seqs < 1 || max_batched_tokens < 1) return fail(ctx, NVLLM_EINVAL, "non-positive pool sizes");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    free_kv(m);
    // a second kv_alloc starts from a clean slate: no sequence, no resident decode batch, no step in flight
    m->seqs.clear();
    m->last_ids.clear();
    m->last_lens.clear();
    m->pending.clear();
    m->pend_next = 0;
    m->decode_resident = false;
    if (m->pin_ids) { (void)hipHostFree(m->pin_ids); m->pin_ids = nullptr; }  // sized by the old max_seqs
    m->num_blocks = num_blocks;
    m->max_seqs = max_seqs;
    // every row-sized buffer holds max_rows rows; a decode step has one row per sequence, so max_seqs rows must fit
    m->max_rows = std::max({max_batched_tokens, max_seqs, 16});
    if (m->opt_oneshot_allreduce && ctx->tp_size > 1 && !ctx->null_comm) {
        // collective: every rank of the group reaches this with the same option (slot = the largest fused decode message)
        (void)oneshot_setup(ctx, (size_t)kFusedMaxRows * m->H);
    } else if (ctx->oneshot) {
        oneshot_free(ctx);  // option switched off: back to the communicator's all-reduce (same decision on every rank)
    }
    const int by_pos = (m->cfg.max_position_embeddings + kBlockTokens - 1) / kBlockTokens;
    m->max_blocks = std::max(1, std::min(num_blocks, by_pos));
    const size_t per_layer = (size_t)num_blocks * m->kv_l * kBlockTokens * m->hd;
    if (m->opt_kv_v_bits == 24 && m->hd != 128) return fail(ctx, NVLLM_EINVAL, "kv_v_bits = 24 needs head_dim 128");
    if (m->opt_kv_k_bits == 24 && m->opt_kv_v_bits != 24) return fail(ctx, NVLLM_EINVAL, "kv_k_bits = 24 needs kv_v_bits = 24 as well");
    m->kcache.assign(m->L, nullptr);
    m->vcache.assign(m->L, nullptr);
    if (m->opt_kv_v_bits == 24) m->vlocache.assign(m->L, nullptr);
    if (m->opt_kv_k_bits == 24) m->klocache.assign(m->L, nullptr);
    for (int l = 0; l < m->L; ++l) {
        HIPCHK(ctx, hipMalloc((void**)&m->kcache[l], per_layer * 2));
        HIPCHK(ctx, hipMalloc((void**)&m->vcache[l], per_layer * 2));
        if (!m->vlocache.empty()) {
            HIPCHK(ctx, hipMalloc((void**)&m->vlocache[l], per_layer));
            HIPCHK(ctx, hipMemsetAsync(m->vlocache[l], 0, per_layer, ctx->stream));
        }
        if (!m->klocache.empty()) {
            HIPCHK(ctx, hipMalloc((void**)&m->klocache[l], per_layer));
            HIPCHK(ctx, hipMemsetAsync(m->klocache[l], 0, per_layer, ctx->stream));
        }
        // zero once: masked lanes multiply P = 0 with whatever the block holds; 0 * finite = 0 needs finite data
        HIPCHK(ctx, hipMemsetAsync(m->kcache[l], 0, per_layer * 2, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(m->vcache[l], 0, per_layer * 2, ctx->stream));
    }
    m->free_blocks.resize(num_blocks);
    for (int i = 0; i < num_blocks; ++i) m->free_blocks[i] = num_blocks - 1 - i;  // pop_back hands out 0,1,2,...
    m->free_slots.resize(max_seqs);
    for (int i = 0; i < max_seqs; ++i) m->free_slots[i] = max_seqs - 1 - i;
    m->h_block_tables.assign((size_t)max_seqs * m->max_blocks, 0);
    int rc = dmalloc(ctx, &m->d_block_tables, (size_t)max_seqs * m->max_blocks);
    if (!rc) HIPCHK(ctx, hipMemsetAsync(m->d_block_tables, 0, (size_t)max_seqs * m->max_blocks * 4, ctx->stream));
    const size_t R = m->max_rows;
    const size_t wide = std::max<size_t>({(size_t)m->H, (size_t)m->nh_l * m->hd, (size_t)m->I_l});
    const size_t nmax = std::max<size_t>({(size_t)(m->nh_l + 2 * m->kv_l) * m->hd, (size_t)m->H, (size_t)2 * m->I_l});
    m->slab_floats = std::max<size_t>(R * nmax, (size_t)32 * std::min<size_t>(R, 128) * nmax);
    if (!rc) rc = dmalloc(ctx, &m->d_ids, R);
    if (!rc) rc = dmalloc(ctx, &m->d_pos, R);
    if (!rc) rc = dmalloc(ctx, &m->d_slot, R);
    // q-tiles: at most one per row, plus the empty tiles that pad every sequence's list to a multiple of 4 (prefill)
    if (!rc) rc = dmalloc(ctx, &m->d_tile_row0, 4 * R);
    if (!rc) rc = dmalloc(ctx, &m->d_tile_nrows, 4 * R);
    if (!rc) rc = dmalloc(ctx, &m->d_tile_slot, 4 * R);
    if (!rc) rc = dmalloc(ctx, &m->d_last_rows, (size_t)max_seqs);
    if (!rc) rc = dmalloc(ctx, &m->d_tile_order, 4 * R);
    if (!rc) rc = dmalloc(ctx, &m->d_tile_last, 4 * R);
    if (!rc) rc = dmalloc(ctx, &m->d_group_order, R + 4);
    if (!rc) rc = dmalloc(ctx, &m->resid, R * m->H);
    if (!rc) rc = dmalloc(ctx, &m->slabs, m->slab_floats);
    if (!rc) rc = dmalloc(ctx, &m->qbuf, R * m->nh_l * m->hd);
    if (!rc) rc = dmalloc(ctx, &m->red, R * m->H);
    if (!rc) rc = dmalloc(ctx, &m->logits, (size_t)max_seqs * m->cfg.vocab_size);  // full vocab: TP gathers here
    if (!rc) rc = dmalloc(ctx, &m->d_maxval, (size_t)max_seqs * std::max(1, ctx->tp_size) * 2);
    const size_t R16 = (R + 15) / 16 * 16;  // packed activation planes are read in whole 16-row blocks
    if (!rc) rc = dmalloc(ctx, &m->xh, R16 * wide);
    if (!rc) rc = dmalloc(ctx, &m->xl, R16 * wide);
    if (!rc) { HIPCHK(ctx, hipMemsetAsync(m->xh, 0, R16 * wide * 2, ctx->stream)); HIPCHK(ctx, hipMemsetAsync(m->xl, 0, R16 * wide * 2, ctx->stream)); }
    if (!rc) rc = dmalloc(ctx, &m->ssqA, (size_t)64 * kFusedMaxRows);  // deferred-norm partials: <= 64 groups
    if (!rc) rc = dmalloc(ctx, &m->ssqB, (size_t)64 * kFusedMaxRows);
    if (!rc) rc = dmalloc(ctx, &m->d_keys, (size_t)max_seqs);
    if (!rc) rc = dmalloc(ctx, &m->d_temps, (size_t)max_seqs);
    if (!rc) rc = dmalloc(ctx, &m->tickets, (size_t)1024);
    if (!rc) HIPCHK(ctx, hipMemsetAsync(m->tickets, 0, 1024 * sizeof(unsigned), ctx->stream));
    if (!rc) rc = dmalloc(ctx, &m->qkv_out, (size_t)std::min<size_t>(R, kFusedMaxRows) * (m->nh_l + 2 * m->kv_l) * m->hd);
    if (!rc) rc = dmalloc(ctx, &m->ctxh, R16 * (size_t)m->nh_l * m->hd);
    if (!rc) rc = dmalloc(ctx, &m->ctxl, R16 * (size_t)m->nh_l * m->hd);
    if (!rc) rc = dmalloc(ctx, &m->xh2, R16 * (size_t)m->I_l);
    if (!rc) rc = dmalloc(ctx, &m->xl2, R16 * (size_t)m->I_l);
    if (!rc) {
        HIPCHK(ctx, hipMemsetAsync(m->ctxh, 0, R16 * (size_t)m->nh_l * m->hd * 2, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(m->ctxl, 0, R16 * (size_t)m->nh_l * m->hd * 2, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(m->xh2, 0, R16 * (size_t)m->I_l * 2, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(m->xl2, 0, R16 * (size_t)m->I_l * 2, ctx->stream));
    }
    if (!rc) rc = dmalloc(ctx, &m->d_next, (size_t)max_seqs * (ctx->tp_size + 1));
    if (!rc) rc = dmalloc(ctx, &m->attn_po, (size_t)max_seqs * m->nh_l * kAttnMaxParts * m->hd);
    if (!rc) rc = dmalloc(ctx, &m->attn_pml, (size_t)max_seqs * m->nh_l * kAttnMaxParts * 2);
    // one partial per LM-head wave: at most V_l/16 n-tiles rounded up to whole workgroups (<= 8 waves each)
    if (!rc) rc = dmalloc(ctx, &m->part_val, (size_t)(m->V_l / 16 + 16) * max_seqs);
    if (!rc) rc = dmalloc(ctx, &m->part_idx, (size_t)(m->V_l / 16 + 16) * max_seqs);
    if (!rc) rc = dmalloc(ctx, &m->argmax_scratch, (size_t)max_seqs + 1);
    if (!rc) HIPCHK(ctx, hipMemsetAsync(m->argmax_scratch, 0, ((size_t)max_seqs + 1) * 8, ctx->stream));
    if (rc) return rc;
    m->h_stage_bytes = (R * 24 + (size_t)max_seqs * 4) * sizeof(int) + 256;
    HIPCHK(ctx, hipHostMalloc(&m->h_stage, m->h_stage_bytes, hipHostMallocDefault));
    // RoPE table: rotary_embedding.rs:56-80 (f32: inv_freq = 1/base^(2j/hd); angle = pos * inv_freq)
    m->rope_len = std::min(m->cfg.max_position_embeddings, m->max_blocks * kBlockTokens);
    const int half = m->hd / 2;
    std::vector<float> hc((size_t)m->rope_len * half), hs((size_t)m->rope_len * half);
    const float base = (float)m->cfg.rope_theta;  // "rope_theta as f32", qwen3.rs:135,196
    for (int j = 0; j < half; ++j) {
        const float exponent = (2.0f * (float)j) / (float)m->hd;
        const float inv_freq = 1.0f / powf(base, exponent);
        for (int p = 0; p < m->rope_len; ++p) {
            const float ang = (float)p * inv_freq;
            hc[(size_t)p * half + j] = cosf(ang);
            hs[(size_t)p * half + j] = sinf(ang);
        }
    }
    rc = dmalloc(ctx, &m->cosv, hc.size());
    if (!rc) rc = dmalloc(ctx, &m->sinv, hs.size());
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(m->cosv, hc.data(), hc.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(m->sinv, hs.data(), hs.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return NVLLM_OK;
}

extern "C" int nvllm_kv_num_free_blocks(const nvllm_model* m) { return m ? (int)m->free_blocks.size() : 0; }
extern "C" int64_t nvllm_kv_bytes_per_token(const nvllm_model* m) {
    if (!m) return 0;
    const int64_t v_bytes = m->vlocache.empty() ? 2 : 3;  // 24-bit V: + one residual byte per element
    const int64_t k_bytes = m->klocache.empty() ? 2 : 3;
    return (int64_t)m->kv_l * m->hd * (k_bytes + v_bytes) * m->L;
}

static void release_seq(nvllm_model* m, SeqState& s) {
    for (int b : s.blocks) m->free_blocks.push_back(b);
    s.blocks.clear();
    s.cached = 0;
}

extern "C" int nvllm_seq_free(nvllm_model* m, int64_t seq_id) {
    if (!m) return NVLLM_EINVAL;
    auto it = m->seqs.find(seq_id);
    if (it == m->seqs.end()) return NVLLM_OK;
    release_seq(m, it->second);
    m->free_slots.push_back(it->second.slot);
    m->seqs.erase(it);
    m->last_ids.clear();
    m->decode_resident = false;
    return NVLLM_OK;
}

extern "C" int nvllm_debug_enable_taps(nvllm_model* m, int enable) {
    if (!m) return NVLLM_EINVAL;
    if (enable && !m->tap_h) {
        if (!m->max_rows) return fail(m->ctx, NVLLM_ESTATE, "kv_alloc first");
        int rc = dmalloc(m->ctx, &m->tap_h, (size_t)m->L * m->max_rows * m->H);
        if (!rc) rc = dmalloc(m->ctx, &m->tap_res, (size_t)m->L * m->max_rows * m->H);
        if (rc) return rc;
    }
    m->taps = enable != 0;
    return NVLLM_OK;
}
extern "C" int nvllm_debug_layer_tap(nvllm_model* m, int layer, int what, float* out, int64_t capacity_floats) {
    if (!m || !out) return NVLLM_EINVAL;
    if (!m->tap_h || layer < 0 || layer >= m->L) return fail(m->ctx, NVLLM_ESTATE, "taps not enabled or bad layer");
    const size_t n = (size_t)m->tap_rows * m->H;
    if ((int64_t)n > capacity_floats) return fail(m->ctx, NVLLM_EINVAL, "tap needs %zu floats", n);
    const float* src = (what ? m->tap_res : m->tap_h) + (size_t)layer * m->max_rows * m->H;
    HIPCHK(m->ctx, hipMemcpy(out, src, n * 4, hipMemcpyDeviceToHost));
    return NVLLM_OK;
}

// ---------------------------------------------------------------------------------------------------
// forward over one chunk of rows (all layers)
// ---------------------------------------------------------------------------------------------------
// Sum split-K slabs into m->red and all-reduce across the TP group; returns the buffer the next
// norm should read (with n_slabs = 1).  Row-parallel outputs only (o_proj, down_proj): the
// all-reduce the reference's RowParallelLinear lacks (src/layers/linear.rs:184-198).
// *stride: floats between the slabs the consumer has to sum
static int tp_reduce(nvllm_model* m, int rows, int ns, const float** in, int* n_slabs, int64_t* stride) {
    nvllm_ctx* ctx = m->ctx;
    *stride = (int64_t)rows * m->H;
    if (ctx->tp_size == 1) { *in = m->slabs; *n_slabs = ns; return NVLLM_OK; }
    float* buf = m->slabs;  // a single complete partial is reduced where it lies
    if (ns > 1) {
        HIPCHK(ctx, launch_slab_sum(m->slabs, ns, (int64_t)rows * m->H, nullptr, rows, m->H, m->red, ctx->stream));
        buf = m->red;
    }
    if (ctx->oneshot && (size_t)rows * m->H <= ctx->os_slot && ((size_t)rows * m->H) % 4 == 0) {
        // one-shot all-reduce (oneshot.hip): the consumer sums the tp slots like split-K slabs
        const int gen = (int)(ctx->os_calls & 1);
        const uint32_t seq = (uint32_t)(++ctx->os_calls);
        if (m->opt_oneshot_skip_push > 0) --m->opt_oneshot_skip_push;
        else HIPCHK(ctx, launch_oneshot_push(buf, (size_t)rows * m->H, ctx->os_peers, ctx->tp_size, ctx->tp_rank, ctx->os_slot, gen, seq, ctx->os_done, ctx->stream));
        oneshot_loopback_order(ctx);
        HIPCHK(ctx, launch_oneshot_wait(ctx->os_flag, ctx->tp_size, gen, seq, ctx->os_err, m->opt_oneshot_spins, ctx->stream));
        *in = ctx->os_data + (size_t)gen * ctx->tp_size * ctx->os_slot;
        *n_slabs = ctx->tp_size;
        *stride = (int64_t)ctx->os_slot;
        return NVLLM_OK;
    }
    int rc_ = comm_allreduce_sum(ctx, buf, (size_t)rows * m->H);
    if (rc_) return rc_;
    *in = buf;
    *n_slabs = 1;
    return NVLLM_OK;
}

// ---------------------------------------------------------------------------------------------------
// Fused forward (R <= kFusedMaxRows rows: decode steps, short prefill chunks): 5 launches per layer at tp = 1,
// 7 launches + 2 all-reduces at tp > 1, instead of the generic path's 8..13.
//   QKV GEMM (complete f32 sums) -> attention (q/k-norm + RoPE + KV write in the prologue for single-row tiles)
//   -> o_proj -> gate/up GEMM (+ SiLU*mul) -> down_proj.
//   tp = 1: o_proj / down_proj epilogues add the residual and prepare the next norm (deferred RMSNorm: they write
//           x' = w_next (.) resid and the row's partial sums of squares; every consumer multiplies its -- linear -- result
//           by rinv[row], RowNorm), so no norm launch exists.
//   tp > 1: o_proj / down_proj leave the rank's partial [R, H] f32 in ONE buffer -> all-reduce(sum) in place (the
//           collective RowParallelLinear::forward lacks, src/layers/linear.rs:184-198) -> one row kernel adds the
//           residual and prepares the next norm the same deferred way.
// Each GEMM takes the whole-K register-direct / row-parallel kernel when its shape allows, else the streaming kernel with
// the matching in-launch-combine epilogue (stream_gemm.hip: 17..64 rows), else -- where slabs are acceptable -- the
// generic kernel.  fused_plan() says whether every GEMM of the layer has such a form at this row count.
// ---------------------------------------------------------------------------------------------------
enum { G_NONE = 0, G_ROW, G_STREAM, G_GENERIC };
struct FusedPlan {
    int qkv = G_NONE, o = G_NONE, gu = G_NONE, down = G_NONE;
    int packed = 0;  // activation planes in MFMA fragment order between the kernels
};
static bool fused_plan(const nvllm_model* m, int R, FusedPlan& p) {
    const int H = m->H, hd = m->hd, NQ = (m->nh_l + 2 * m->kv_l) * hd, KO = m->nh_l * hd, I2 = 2 * m->I_l;
    const bool tp = m->ctx->tp_size > 1;
    if (R > kFusedMaxRows || m->taps || m->opt_no_fused) return false;
    auto fits = [&](int N, int K, int epi) {
        return m->opt_stream_combine && gemm_stream_ok(R, N, K, epi) && gemm_stream_slab_floats(R, N, K, epi) <= m->slab_floats;
    };
    // QKV: complete sums preferred; the generic kernel's slabs are summed by the attention prologue / qk kernel
    if (gemm_rowpar_ok(NQ, H, 2, R) && gemm_rowpar_splits(NQ, H, 2, R) == 1) p.qkv = G_ROW;
    else if (fits(NQ, H, 1)) p.qkv = G_STREAM;
    else p.qkv = G_GENERIC;
    // o_proj / down_proj: tp = 1 needs the residual + norm-prep epilogue; tp > 1 one complete partial buffer
    auto rowlin = [&](int N, int K) -> int {
        if (!tp) {
            if (gemm_rowpar_ok(N, K, 0, R) && gemm_rowpar_groups(N, K) <= 64 && gemm_rowpar_groups(N, K) > 0) return G_ROW;
            if (fits(N, K, 3)) return G_STREAM;
            return G_NONE;
        }
        if (gemm_rowpar_ok(N, K, 2, R) && gemm_rowpar_splits(N, K, 2, R) == 1) return G_ROW;
        if (fits(N, K, 1)) return G_STREAM;
        return G_GENERIC;  // slabs + one sum launch before the all-reduce
    };
    p.o = rowlin(H, KO);
    p.down = rowlin(H, m->I_l);
    if (p.o == G_NONE || p.down == G_NONE) return false;
    // tp > 1 without a single-buffer kernel for a row-parallel projection: the generic path (streaming slabs + sum) is the
    // faster one (Qwen3-32B TP=8 shard shapes, per rank: 6.2 vs 7.4 ms/step)
    if (tp && (p.o == G_GENERIC || p.down == G_GENERIC)) return false;
    if (gemm_rowpar_ok(I2, H, 1, R)) p.gu = G_ROW;
    else if (fits(I2, H, 2)) p.gu = G_STREAM;
    else if (gemm_stream_splits(R, I2, H) > 0) return false;  // big gate/up: streaming slabs + SiLU launch beat the unsplit generic kernel
    else p.gu = G_GENERIC;  // generic kernel with the SwiGLU epilogue (no K split)
    const bool no_xpack = m->opt_no_xpack != 0;
    auto takes_packed = [&](int kind, int N, int K, int epi) { return kind == G_STREAM || (kind == G_ROW && gemm_rowdir_ok(N, K, epi, R)); };
    // fewer than 16 rows: a fragment-order plane is 15/16 padding (a 1-row step would load 1 KiB per k-tile where a
    // row-major plane serves 64 bytes to all 16 lanes): batch 1 measured 960 -> 1017 tokens/s with row-major planes
    p.packed = !no_xpack && R >= 16 && m->I_l % 32 == 0 && takes_packed(p.qkv, NQ, H, 2) && takes_packed(p.o, H, KO, tp ? 2 : 0) &&
               takes_packed(p.gu, I2, H, 1) && takes_packed(p.down, H, m->I_l, tp ? 2 : 0);
    return true;
}

static int forward_chunk_fused(nvllm_model* m, const FusedPlan& fp, int R, int n_tiles, int qt, int n_last, int logits_row0) {
    nvllm_ctx* ctx = m->ctx;
    hipStream_t s = ctx->stream;
    const int H = m->H, hd = m->hd;
    const float eps = (float)m->cfg.rms_norm_eps;
    const int NQ = (m->nh_l + 2 * m->kv_l) * hd;
    const int KO = m->nh_l * hd;
    const bool tp = ctx->tp_size > 1;
    const int packed = fp.packed;
    m->tap_rows = R;
    m->stamp_launch = 0;
    RowNorm rn;  // the norm pending on xh/xl
    rn.stride = kFusedMaxRows; rn.inv_h = 1.0f / (float)H; rn.eps = eps;
    auto stream_args = [&](const bf16_bits* xh, const bf16_bits* xl, int K) {
        StreamArgs sa;
        sa.xh = xh; sa.xl = xl; sa.ldx = K; sa.x_packed = packed; sa.M = R; sa.slabs = m->slabs; sa.tickets = m->tickets;
        return sa;
    };
    // prep: resid (+)= in; x' = w (.) resid as hi/lo; ssq[row] (one group).  in == nullptr: layer-0 embedding rows
    auto prep = [&](const float* in, const float* w, int out_packed, int in_slabs = 1, int64_t in_stride = 0) -> int {
        NormArgs na;
        if (in) { na.in = in; na.n_slabs = in_slabs; na.slab_stride = in_stride; na.residual_in = m->resid; }
        else { na.ids = m->d_ids; na.embed = m->embed; }
        na.weight = w; na.eps = eps; na.H = H; na.xh = m->xh; na.xl = m->xl; na.residual_out = m->resid; na.ssq_out = m->ssqB;
        na.out_packed = out_packed;
        PROF(m, PROF_NORM, launch_add_rmsnorm(na, R, s));
        rn.ssq = m->ssqB; rn.groups = 1;
        return NVLLM_OK;
    };
    // row-parallel projection (o_proj, down_proj) of input planes (xh_, xl_) [R, K]: tp = 1 epilogue or partial + all-reduce + prep
    auto row_linear = [&](int kind, const PackedW& w, const bf16_bits* xh_, const bf16_bits* xl_, int K, const float* next_w,
                          float* ssq_buf, int o_packed) -> int {
        if (!tp) {
            if (kind == G_ROW) {
                RowParArgs ra;
                ra.xh = xh_; ra.xl = xl_; ra.ldx = K; ra.resid_in = m->resid; ra.resid_out = m->resid; ra.next_w = next_w;
                ra.oh = m->xh; ra.ol = m->xl; ra.ssq = ssq_buf; ra.ssq_stride = kFusedMaxRows; ra.M = R;
                ra.x_packed = packed; ra.o_packed = o_packed;
                STAMPS(ra, m);
                PROF(m, PROF_GEMM, launch_gemm_rowpar(ra, w, 0, s));
                rn.ssq = ssq_buf; rn.groups = gemm_rowpar_groups(H, K);
            } else {
                StreamArgs sa = stream_args(xh_, xl_, K);
                sa.resid_in = m->resid; sa.resid_out = m->resid; sa.next_w = next_w; sa.oh = m->xh; sa.ol = m->xl; sa.o_packed = o_packed;
                sa.ssq = ssq_buf; sa.ssq_stride = kFusedMaxRows;
                PROF(m, PROF_GEMM, launch_gemm_stream_epi(sa, w, 3, s));
                rn.ssq = ssq_buf; rn.groups = gemm_stream_groups(R, H, K, 3);
            }
            return NVLLM_OK;
        }
        if (kind == G_ROW) {
            RowParArgs ra;
            ra.xh = xh_; ra.xl = xl_; ra.ldx = K; ra.out = m->red; ra.M = R; ra.x_packed = packed;
            PROF(m, PROF_GEMM, launch_gemm_rowpar(ra, w, 2, s));
        } else if (kind == G_STREAM) {
            StreamArgs sa = stream_args(xh_, xl_, K);
            sa.out = m->red;
            PROF(m, PROF_GEMM, launch_gemm_stream_epi(sa, w, 1, s));
        } else {
            GemmPlan pg = plan_gemm(R, H, K, 8);
            PROF(m, PROF_GEMM, launch_gemm(pg, xh_, xl_, K, w, m->slabs, R, s));
            HIPCHK(ctx, launch_slab_sum(m->slabs, pg.n_split, (int64_t)R * H, nullptr, R, H, m->red, s));
        }
        if (ctx->oneshot && (size_t)R * H <= ctx->os_slot && (R * H) % 4 == 0) {
            // one-shot all-reduce: push the partial into every peer's slot, wait for the tp flags, and let the norm prep
            // sum the tp slots like split-K slabs (every rank adds them in rank order: identical bits on all ranks)
            const int gen = (int)(ctx->os_calls & 1);
            const uint32_t seq = (uint32_t)(++ctx->os_calls);
            if (m->opt_oneshot_skip_push > 0) --m->opt_oneshot_skip_push;
            else HIPCHK(ctx, launch_oneshot_push(m->red, (size_t)R * H, ctx->os_peers, ctx->tp_size, ctx->tp_rank, ctx->os_slot, gen, seq, ctx->os_done, s));
            oneshot_loopback_order(ctx);
            HIPCHK(ctx, launch_oneshot_wait(ctx->os_flag, ctx->tp_size, gen, seq, ctx->os_err, m->opt_oneshot_spins, s));
            return prep(ctx->os_data + (size_t)gen * ctx->tp_size * ctx->os_slot, next_w, o_packed, ctx->tp_size, (int64_t)ctx->os_slot);
        }
        int rc = comm_allreduce_sum(ctx, m->red, (size_t)R * H);
        if (rc) return rc;
        return prep(m->red, next_w, o_packed);
    };
    // layer 0 input: residual = embedding row, x' = ln1 (.) row, ssq (qwen3.rs:382-386, 465-468)
    int rc0 = prep(nullptr, m->layers[0].ln1, packed);
    if (rc0) return rc0;
    for (int l = 0; l < m->L; ++l) {
        const LayerW& w = m->layers[l];
        QkvArgs qa;
        qa.n_slabs = 1;
        if (fp.qkv == G_ROW) {
            RowParArgs rq;
            rq.xh = m->xh; rq.xl = m->xl; rq.ldx = H; rq.out = m->slabs; rq.M = R; rq.x_packed = packed;
            STAMPS(rq, m);
            PROF(m, PROF_GEMM, launch_gemm_rowpar(rq, w.qkv, 2, s));
            qa.qkv = m->slabs;
        } else if (fp.qkv == G_STREAM) {
            StreamArgs sa = stream_args(m->xh, m->xl, H);
            sa.out = m->qkv_out;
            PROF(m, PROF_GEMM, launch_gemm_stream_epi(sa, w.qkv, 1, s));
            qa.qkv = m->qkv_out;
        } else {
            GemmPlan pq = plan_gemm(R, NQ, H, 8);
            PROF(m, PROF_GEMM, launch_gemm(pq, m->xh, m->xl, H, w.qkv, m->slabs, R, s));
            qa.qkv = m->slabs; qa.n_slabs = pq.n_split;
        }
        qa.slab_stride = (int64_t)R * NQ; qa.qn = w.qn; qa.kn = w.kn; qa.eps = eps;
        qa.cos = m->cosv; qa.sin = m->sinv; qa.pos = m->d_pos; qa.slot = m->d_slot; qa.block_tables = m->d_block_tables;
        qa.max_blocks = m->max_blocks; qa.nh_l = m->nh_l;
        qa.q_scale = powf((float)hd, -0.5f) * 1.4426950408889634f;
        qa.q_out = m->qbuf; qa.rn = rn;
        qa.kv.k = m->kcache[l]; qa.kv.v = m->vcache[l]; qa.kv.kv_l = m->kv_l; qa.kv.hd = hd;
        qa.kv.vlo = m->vlocache.empty() ? nullptr : m->vlocache[l];
        qa.kv.klo = m->klocache.empty() ? nullptr : m->klocache[l];
        const bool fuse_qk = qt == 1 && n_tiles == R && !m->opt_no_attn_prologue;
        if (!fuse_qk) PROF(m, PROF_QK, launch_qk_norm_rope_kvwrite(qa, R, s));
        AttnArgs aa;
        aa.q = m->qbuf; aa.kv = qa.kv; aa.block_tables = m->d_block_tables; aa.max_blocks = m->max_blocks;
        aa.tile_row0 = m->d_tile_row0; aa.tile_nrows = m->d_tile_nrows; aa.tile_slot = m->d_tile_slot; aa.pos = m->d_pos;
        aa.tile_last = m->d_tile_last;
        if (qt == 2) aa.group_order = m->d_group_order;
        aa.nh_l = m->nh_l; aa.gqa = m->gqa; aa.out_hi = m->ctxh; aa.out_lo = m->ctxl; aa.out_packed = packed;
        if (qt == 1) aa.tile_order = m->d_tile_order;
        if (fuse_qk) {
            aa.qkv = qa.qkv; aa.n_slabs = qa.n_slabs; aa.slab_stride = qa.slab_stride; aa.ldqkv = NQ; aa.qn = w.qn; aa.kn = w.kn;
            aa.cos = m->cosv; aa.sin = m->sinv; aa.eps = eps; aa.q_scale = qa.q_scale; aa.rn = rn;
        }
        int parts_max = 1;
        if (qt == 1 && R <= m->max_seqs) {
            aa.part_tiles = m->attn_part_tiles; aa.max_parts = kAttnMaxParts; aa.part_o = m->attn_po; aa.part_ml = m->attn_pml;
            parts_max = m->attn_parts_max;
        }
        PROF(m, PROF_EMPTY, hipSuccess);  // calibration: an event pair around nothing, at the attention launch's place
        STAMPS(aa, m);
        if (qt == 2 && !aa.kv.vlo) PROF(m, PROF_ATTN, launch_attn_prefill(aa, n_tiles, s));
        else PROF(m, PROF_ATTN, launch_attn_paged(aa, n_tiles, qt, R, parts_max, s));
        // o_proj (+ residual + post-attention norm prep; qwen3.rs:278, :393)
        int rc = row_linear(fp.o, w.o, m->ctxh, m->ctxl, KO, w.ln2, m->ssqA, packed);
        if (rc) return rc;
        // gate/up + SiLU*mul (qwen3.rs:324-325), scaled by the pending norm's rinv
        if (fp.gu == G_ROW) {
            RowParArgs rg;
            rg.xh = m->xh; rg.xl = m->xl; rg.ldx = H; rg.oh = m->xh2; rg.ol = m->xl2; rg.M = R; rg.rn = rn;
            rg.x_packed = packed; rg.o_packed = packed;
            STAMPS(rg, m);
            PROF(m, PROF_GEMM, launch_gemm_rowpar(rg, w.gu, 1, s));
        } else if (fp.gu == G_STREAM) {
            StreamArgs sa = stream_args(m->xh, m->xl, H);
            sa.rn = rn; sa.oh = m->xh2; sa.ol = m->xl2; sa.o_packed = packed;
            PROF(m, PROF_GEMM, launch_gemm_stream_epi(sa, w.gu, 2, s));
        } else {
            GemmPlan pg = plan_gemm_swiglu(R, 2 * m->I_l, H);
            PROF(m, PROF_GEMM, launch_gemm_swiglu(pg, m->xh, m->xl, H, w.gu, R, m->xh2, m->xl2, &rn, s));
        }
        // down_proj (+ residual + next layer's input norm prep; qwen3.rs:326, next layer :378; last layer: final norm :497)
        const bool last = l + 1 == m->L;
        rc = row_linear(fp.down, w.down, m->xh2, m->xl2, m->I_l, last ? m->norm : m->layers[l + 1].ln1, m->ssqB,
                        packed && !last);  // the LM head (chunked kernel) reads row-major planes
        if (rc) return rc;
    }
    if (n_last > 0) {
        // LM head on the last-token rows only (gathered by row_idx), logits scaled by the final norm's rinv
        GemmPlan pl = plan_lmhead(n_last, m->V_l, H);
        float* lg = m->want_logits ? m->logits + (size_t)logits_row0 * m->V_l : nullptr;
        rn.row_idx = m->d_last_rows;
        PROF(m, PROF_LMHEAD, launch_gemm_argmax(pl, m->xh, m->xl, H, m->lm_head, lg, n_last, m->part_val, m->part_idx, &rn, s));
        const int tps = ctx->tp_size;  // vocab-parallel: every rank leaves (id, max) of its shard for the gather in finish_logits
        uint32_t* ids_dst = tps == 1 ? m->d_next + logits_row0 : m->d_next + m->cur_n + (size_t)ctx->tp_rank * m->cur_n + logits_row0;
        float* val_dst = tps == 1 ? nullptr : m->d_maxval + (size_t)ctx->tp_rank * m->cur_n + logits_row0;
        if (pl.lm_nt > 0)
            HIPCHK(ctx, launch_argmax_rows(m->part_val, m->part_idx, gemm_argmax_parts(pl, m->V_l), n_last, ids_dst, val_dst, s));
        else
            HIPCHK(ctx, launch_argmax_parts(m->part_val, m->part_idx, gemm_argmax_parts(pl, m->V_l), n_last, m->argmax_scratch,
                                            ids_dst, val_dst, s));
    }
    return NVLLM_OK;
}

// y (f32 slabs) = x.W^T on the generic path: for few rows the row-parallel streaming kernel (every load of a
// workgroup issued up front) beats the chunked kernel on big matrices; otherwise the chunked kernel.
static bool gemm_streams(const nvllm_model* m, const PackedW& w, int R) {
    const int ss = gemm_stream_splits(R, w.N, w.K);
    return ss > 0 && (size_t)ss * R * w.N <= m->slab_floats;
}
static int gemm_slabs(nvllm_model* m, const bf16_bits* xh, const bf16_bits* xl, int ldx, const PackedW& w, float* out, int R,
                      int max_split, int* n_slabs, int x_packed = 0) {
    nvllm_ctx* ctx = m->ctx;
    if (gemm_streams(m, w, R)) {
        PROF(m, PROF_GEMM, launch_gemm_stream(xh, xl, ldx, w, out, R, x_packed, ctx->stream));
        *n_slabs = gemm_stream_splits(R, w.N, w.K);
        return NVLLM_OK;
    }
    if (x_packed) return fail(ctx, NVLLM_ESTATE, "packed activation planes reached a non-streaming GEMM");
    if (R <= kFusedMaxRows && gemm_rowpar_ok(w.N, w.K, 2, R) &&
        (size_t)gemm_rowpar_splits(w.N, w.K, 2, R) * R * w.N <= m->slab_floats && !m->opt_no_rowpar) {
        RowParArgs ra;
        ra.xh = xh; ra.xl = xl; ra.ldx = ldx; ra.out = out; ra.M = R;
        PROF(m, PROF_GEMM, launch_gemm_rowpar(ra, w, 2, ctx->stream));
        *n_slabs = gemm_rowpar_splits(w.N, w.K, 2, R);
        return NVLLM_OK;
    }
    GemmPlan p = plan_gemm(R, w.N, w.K, max_split);
    PROF(m, PROF_GEMM, launch_gemm(p, xh, xl, ldx, w, out, R, ctx->stream));
    *n_slabs = p.n_split;
    return NVLLM_OK;
}

// rows R (ids/pos/slot/tiles already on the device), n_last rows listed in d_last_rows -> logits rows
static int forward_chunk(nvllm_model* m, int R, int n_tiles, int qt, int n_last, int logits_row0) {
    FusedPlan fp;
    if (fused_plan(m, R, fp)) return forward_chunk_fused(m, fp, R, n_tiles, qt, n_last, logits_row0);
    nvllm_ctx* ctx = m->ctx;
    hipStream_t s = ctx->stream;
    const int H = m->H, hd = m->hd;
    const float eps = (float)m->cfg.rms_norm_eps;
    const int NQ = (m->nh_l + 2 * m->kv_l) * hd;
    const float* prev = nullptr;  // output of the previous layer's MLP (slabs or reduced)
    m->stamp_launch = 0;
    int prev_ns = 1;
    int64_t prev_stride = 0;
    m->tap_rows = R;
    // big-model decode: when all four projections of a layer run the streaming GEMM, the activation planes between
    // the row kernels and the GEMMs stay in MFMA fragment order (xpack_off): x staging costs 3x per byte otherwise
    const bool no_xpack = m->opt_no_xpack != 0;
    const int packed = !no_xpack && !m->layers.empty() && gemm_streams(m, m->layers[0].qkv, R) && gemm_streams(m, m->layers[0].o, R) &&
                       gemm_streams(m, m->layers[0].gu, R) && gemm_streams(m, m->layers[0].down, R) && m->I_l % 32 == 0;
    // prompt chunks (MFMA-bound): projections whose grid fills the chip run the tile GEMM (tile_gemm.hip); it reads both
    // operands in fragment order, so the kernel that produces its input planes writes them that way
    const int KO = m->nh_l * hd;
    const int tmin = m->opt_tile_min_wgs;
    const bool tile_on = !packed && tmin > 0 && R > kFusedMaxRows;
    // K splits (f32 slabs summed by the consumer): up to 4 on prompt chunks; a few hundred decode rows are MFMA-bound
    // too (hi + lo planes: 4 flops per weight byte and row) and need up to 8 to fill the chip
    const int max_ks = R <= 512 ? 8 : 4;
    const int o_ks = tile_on ? gemm_tile_splits(R, H, KO, tmin, max_ks) : 0, d_ks = tile_on ? gemm_tile_splits(R, H, m->I_l, tmin, max_ks) : 0;
    const int q_ks = tile_on ? gemm_tile_splits(R, NQ, H, tmin, max_ks) : 0;
    const bool t_qkv = q_ks > 0 && (size_t)q_ks * R * NQ <= m->slab_floats;
    // (the fused epilogue writes V as packed f16 stores only: 24-bit V keeps the row kernel)
    const bool t_qkv_fused = t_qkv && m->opt_tile_fuse_qk && m->vlocache.empty() && gemm_tile_qkv_ok(R, NQ, H, hd, tmin);
    const bool t_o = o_ks > 0 && (size_t)o_ks * R * H <= m->slab_floats;
    const bool t_gu = tile_on && gemm_tile_ok(R, 2 * m->I_l, H, 2, tmin);
    const bool t_down = t_gu && m->I_l % 32 == 0 && d_ks > 0 && (size_t)d_ks * R * H <= m->slab_floats;
    m->tile_launches += (int64_t)m->L * ((int)t_qkv + (int)t_o + (int)t_gu + (int)t_down);
    for (int l = 0; l < m->L; ++l) {
        const LayerW& w = m->layers[l];
        NormArgs na;
        na.weight = w.ln1; na.eps = eps; na.H = H; na.xh = m->xh; na.xl = m->xl; na.residual_out = m->resid; na.out_packed = packed || t_qkv;
        if (l == 0) {  // residual None: normed = norm(x), residual = x  (qwen3.rs:382-386)
            na.ids = m->d_ids; na.embed = m->embed;
        } else {       // qwen3.rs:378
            na.in = prev; na.n_slabs = prev_ns; na.slab_stride = prev_stride; na.residual_in = m->resid;
        }
        PROF(m, PROF_NORM, launch_add_rmsnorm(na, R, s));
        // QKV projection (qwen3.rs:205)
        QkvArgs qa;
        qa.qkv = m->slabs; qa.slab_stride = (int64_t)R * NQ; qa.qn = w.qn; qa.kn = w.kn; qa.eps = eps;
        qa.cos = m->cosv; qa.sin = m->sinv; qa.pos = m->d_pos; qa.slot = m->d_slot; qa.block_tables = m->d_block_tables;
        qa.max_blocks = m->max_blocks; qa.nh_l = m->nh_l;
        qa.q_scale = powf((float)hd, -0.5f) * 1.4426950408889634f;  // head_dim^-0.5 (qwen3.rs:134) * log2(e)
        qa.q_out = m->qbuf;
        qa.kv.k = m->kcache[l]; qa.kv.v = m->vcache[l]; qa.kv.kv_l = m->kv_l; qa.kv.hd = hd;
        qa.kv.vlo = m->vlocache.empty() ? nullptr : m->vlocache[l];
        qa.kv.klo = m->klocache.empty() ? nullptr : m->klocache[l];
        const bool fuse_qk = qt == 1 && n_tiles == R;  // decode: every q-tile is one row
        int rcg = NVLLM_OK;
        if (t_qkv_fused && !fuse_qk) {
            // prompt chunk: q/k-norm + RoPE + KV write + q output in the QKV GEMM's epilogue (one wave tile = one head)
            PROF(m, PROF_GEMM, launch_gemm_tile_qkv(m->xh, m->xl, w.qkv, R, qa, tmin, s));
        } else {
            if (t_qkv) PROF(m, PROF_GEMM, launch_gemm_tile(m->xh, m->xl, w.qkv, R, 0, m->slabs, nullptr, nullptr, 0, tmin, max_ks, &qa.n_slabs, s));
            else rcg = gemm_slabs(m, m->xh, m->xl, H, w.qkv, m->slabs, R, 8, &qa.n_slabs, packed);
            if (rcg) return rcg;
            if (!fuse_qk) PROF(m, PROF_QK, launch_qk_norm_rope_kvwrite(qa, R, s));
        }
        AttnArgs aa;
        aa.q = m->qbuf; aa.kv = qa.kv; aa.block_tables = m->d_block_tables; aa.max_blocks = m->max_blocks;
        aa.tile_row0 = m->d_tile_row0; aa.tile_nrows = m->d_tile_nrows; aa.tile_slot = m->d_tile_slot; aa.pos = m->d_pos;
        aa.tile_last = m->d_tile_last;
        if (qt == 2) aa.group_order = m->d_group_order;
        aa.nh_l = m->nh_l; aa.gqa = m->gqa; aa.out_hi = m->xh; aa.out_lo = m->xl; aa.out_packed = packed || t_o;
        if (qt == 1) aa.tile_order = m->d_tile_order;
        if (fuse_qk) {
            aa.qkv = qa.qkv; aa.n_slabs = qa.n_slabs; aa.slab_stride = qa.slab_stride; aa.ldqkv = NQ; aa.qn = w.qn; aa.kn = w.kn;
            aa.cos = m->cosv; aa.sin = m->sinv; aa.eps = eps; aa.q_scale = qa.q_scale;
        }
        int parts_max = 1;
        if (qt == 1 && R <= m->max_seqs) {  // decode rows: split the context over workgroups
            aa.part_tiles = m->attn_part_tiles; aa.max_parts = kAttnMaxParts; aa.part_o = m->attn_po; aa.part_ml = m->attn_pml;
            parts_max = m->attn_parts_max;
        }
        STAMPS(aa, m);
        if (qt == 2 && !aa.kv.vlo) PROF(m, PROF_ATTN, launch_attn_prefill(aa, n_tiles, s));
        else PROF(m, PROF_ATTN, launch_attn_paged(aa, n_tiles, qt, R, parts_max, s));
        // output projection (qwen3.rs:278) + TP all-reduce
        int o_slabs = 1;
        if (t_o) PROF(m, PROF_GEMM, launch_gemm_tile(m->xh, m->xl, w.o, R, 0, m->slabs, nullptr, nullptr, 0, tmin, max_ks, &o_slabs, s));
        else rcg = gemm_slabs(m, m->xh, m->xl, KO, w.o, m->slabs, R, 8, &o_slabs, packed);
        if (rcg) return rcg;
        const float* oin; int ons;
        int64_t ostride = 0;
        int rc = tp_reduce(m, R, o_slabs, &oin, &ons, &ostride);
        if (rc) return rc;
        NormArgs nb;  // post-attention add + norm (qwen3.rs:393)
        nb.in = oin; nb.n_slabs = ons; nb.slab_stride = ostride; nb.residual_in = m->resid; nb.residual_out = m->resid;
        nb.weight = w.ln2; nb.eps = eps; nb.H = H; nb.xh = m->xh; nb.xl = m->xl; nb.out_packed = packed || t_gu;
        PROF(m, PROF_NORM, launch_add_rmsnorm(nb, R, s));
        // MLP (qwen3.rs:323-327)
        if (t_gu) {
            PROF(m, PROF_GEMM, launch_gemm_tile(m->xh, m->xl, w.gu, R, 2, nullptr, m->xh2, m->xl2, t_down ? 1 : 0, tmin, 1, nullptr, s));
        } else if (!packed && R <= kFusedMaxRows && gemm_rowpar_ok(2 * m->I_l, H, 1, R)) {
            // small gate/up: whole-K row-parallel kernel with the SiLU*mul epilogue
            RowParArgs rg;
            rg.xh = m->xh; rg.xl = m->xl; rg.ldx = H; rg.oh = m->xh2; rg.ol = m->xl2; rg.M = R;
            PROF(m, PROF_GEMM, launch_gemm_rowpar(rg, w.gu, 1, s));
        } else if (packed || (R <= kFusedMaxRows && gemm_rowpar_ok(2 * m->I_l, H, 2, R) &&
                              (size_t)gemm_rowpar_splits(2 * m->I_l, H, 2, R) * R * 2 * m->I_l <= m->slab_floats && !m->opt_no_rowpar)) {
            // big gate/up at few rows: stream it with K slices (f32 slabs of the INTERLEAVED gate/up rows),
            // then SiLU*mul on the summed slabs
            int gs = 1;
            rcg = gemm_slabs(m, m->xh, m->xl, H, w.gu, m->slabs, R, 8, &gs, packed);
            if (rcg) return rcg;
            PROF(m, PROF_SILU, launch_silu_mul_interleaved(m->slabs, gs, (int64_t)R * 2 * m->I_l, R, m->I_l, m->xh2, m->xl2, packed, s));
        } else {
            // gate/up GEMM with the SiLU*mul epilogue: act hi/lo written directly, no slabs, no extra launch
            GemmPlan pg = plan_gemm_swiglu(R, 2 * m->I_l, H);
            PROF(m, PROF_GEMM, launch_gemm_swiglu(pg, m->xh, m->xl, H, w.gu, R, m->xh2, m->xl2, nullptr, s));
        }
        int d_slabs = 1;
        if (t_down) PROF(m, PROF_GEMM, launch_gemm_tile(m->xh2, m->xl2, w.down, R, 0, m->slabs, nullptr, nullptr, 0, tmin, max_ks, &d_slabs, s));
        else rcg = gemm_slabs(m, m->xh2, m->xl2, m->I_l, w.down, m->slabs, R, 8, &d_slabs, packed);
        if (rcg) return rcg;
        rc = tp_reduce(m, R, d_slabs, &prev, &prev_ns, &prev_stride);
        if (rc) return rc;
        if (m->taps) {
            HIPCHK(ctx, launch_slab_sum(prev, prev_ns, prev_stride, nullptr, R, H, m->tap_h + (size_t)l * m->max_rows * H, s));
            HIPCHK(ctx, hipMemcpyAsync(m->tap_res + (size_t)l * m->max_rows * H, m->resid, (size_t)R * H * 4, hipMemcpyDeviceToDevice, s));
        }
    }
    if (n_last > 0) {
        // final add + norm on the rows that are a sequence's last token only (qwen3.rs:497), then LM head
        // on those rows (qwen3.rs:542-550 computes all rows; only row len-1 is used, llm_engine.rs:181-183)
        NormArgs nf;
        nf.in = prev; nf.n_slabs = prev_ns; nf.slab_stride = prev_stride; nf.residual_in = m->resid; nf.row_idx = m->d_last_rows;
        nf.weight = m->norm; nf.eps = eps; nf.H = H; nf.xh = m->xh; nf.xl = m->xl;
        HIPCHK(ctx, launch_add_rmsnorm(nf, n_last, s));
        GemmPlan pl = plan_lmhead(n_last, m->V_l, H);
        // logits are stored only when the caller asked for them; the greedy id always comes from the
        // GEMM epilogue's per-wave partial arg-max (LAST max wins), finished by one small kernel
        float* lg = m->want_logits ? m->logits + (size_t)logits_row0 * m->V_l : nullptr;
        PROF(m, PROF_LMHEAD, launch_gemm_argmax(pl, m->xh, m->xl, H, m->lm_head, lg, n_last, m->part_val, m->part_idx, nullptr, s));
        const int tp = ctx->tp_size;
        uint32_t* ids_dst = tp == 1 ? m->d_next + logits_row0 : m->d_next + m->cur_n + (size_t)ctx->tp_rank * m->cur_n + logits_row0;
        float* val_dst = tp == 1 ? nullptr : m->d_maxval + (size_t)ctx->tp_rank * m->cur_n + logits_row0;
        if (pl.lm_nt > 0)
            HIPCHK(ctx, launch_argmax_rows(m->part_val, m->part_idx, gemm_argmax_parts(pl, m->V_l), n_last, ids_dst, val_dst, s));
        else
            HIPCHK(ctx, launch_argmax_parts(m->part_val, m->part_idx, gemm_argmax_parts(pl, m->V_l), n_last, m->argmax_scratch, ids_dst, val_dst, s));
    }
    return NVLLM_OK;
}

// ---------------------------------------------------------------------------------------------------
// step
// ---------------------------------------------------------------------------------------------------
static int ensure_blocks(nvllm_model* m, SeqState& s, int len) {
    const int need = (len + kBlockTokens - 1) / kBlockTokens;
    if (need > m->max_blocks) return model_fail(m, NVLLM_ENOMEM, "sequence of %d tokens exceeds %d blocks per sequence", len, m->max_blocks);
    if ((int)s.blocks.size() >= need) return NVLLM_OK;
    if ((int)m->free_blocks.size() < need - (int)s.blocks.size())
        return model_fail(m, NVLLM_ENOMEM, "KV pool exhausted (%zu free blocks, need %d more)", m->free_blocks.size(), need - (int)s.blocks.size());
    while ((int)s.blocks.size() < need) {
        const int b = m->free_blocks.back();
        m->free_blocks.pop_back();
        m->h_block_tables[(size_t)s.slot * m->max_blocks + s.blocks.size()] = b;
        s.blocks.push_back(b);
    }
    return NVLLM_OK;
}

// Split a sequence's context over several workgroups only when (sequences x kv heads) alone cannot fill
// the chip (small batches, long contexts); a full batch already has >= 2 workgroups per CU.
static void set_attn_split(nvllm_model* m, int n_seqs, int max_len) {
    const int tiles = (max_len + 31) / 32;
    const int base_wgs = std::max(1, n_seqs * m->kv_l);
    int want = std::min(kAttnMaxParts, std::max(1, 512 / base_wgs));
    // <= 256 tokens: four waves hold the whole context in one round; <= 512: two more rounds of the loop (~2.6 us) still cost
    // less than the combine launch and its boundary (4.7 + 1.4 us, batch-1 profile)
    if (tiles <= 16) want = 1;
    int part = std::max(4, (tiles + want - 1) / want);  // >= 128 tokens per workgroup
    m->attn_part_tiles = part;
    m->attn_parts_max = (tiles + part - 1) / part;
}

// A one-shot all-reduce whose wait timed out (a peer never raised its flag) has left wrong sums.  The error must not stay
// per-rank: a rank that returned early would leave its peers in the collectives that follow (the (max, index) gather of
// this very step).  So every rank contributes its error word to one small all-gather, all of them take the same
// decision, and after an error all of them leave the one-shot path (the context falls back to the communicator's
// all-reduce until the next kv_alloc sets the buffers up afresh).  A peer that is DEAD still hangs the communicator's own
// collectives -- that is RCCL's failure model, not this protocol's.
static int oneshot_check(nvllm_ctx* ctx) {
    if (!ctx->oneshot) return NVLLM_OK;
    const int tp = ctx->tp_size;
    HIPCHK(ctx, hipMemcpyAsync(ctx->os_agree + ctx->tp_rank, ctx->os_err, 4, hipMemcpyDeviceToDevice, ctx->stream));
    int rc = comm_allgather(ctx, ctx->os_agree + ctx->tp_rank, ctx->os_agree, 4);
    if (rc) return rc;
    int e[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIPCHK(ctx, hipMemcpyAsync(e, ctx->os_agree, (size_t)tp * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    int bad = -1;
    for (int r = 0; r < tp; ++r) if (e[r]) bad = r;
    if (bad >= 0) {
        (void)hipMemsetAsync(ctx->os_err, 0, 4, ctx->stream);
        ctx->oneshot = false;  // every rank reaches this line in the same step: the group leaves the device path together
        return fail(ctx, NVLLM_ERCCL, "one-shot all-reduce: a peer's flag never arrived at rank %d (bounded wait gave up); this step's results are "
                                      "invalid and the group is back on the communicator's all-reduce", bad);
    }
    return NVLLM_OK;
}

static int finish_logits(nvllm_model* m, int n, uint32_t* next_ids, float* last_logits) {
    nvllm_ctx* ctx = m->ctx;
    hipStream_t s = ctx->stream;
    const int tp = ctx->tp_size, V = m->cfg.vocab_size, Vl = m->V_l;
    { const int rc_os = oneshot_check(ctx); if (rc_os) return rc_os; }
    if (tp == 1) {
        if (next_ids) HIPCHK(ctx, hipMemcpyAsync(next_ids, m->d_next, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        if (last_logits) HIPCHK(ctx, hipMemcpyAsync(last_logits, m->logits, (size_t)n * V * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(ctx, hipStreamSynchronize(s));
        return NVLLM_OK;
    }
    // local (max, idx) -> all-gather -> pick the best; ties go to the higher global index
    uint32_t* idx_all = m->d_next + n;           // [tp][n] lives after the first n entries
    float* val_all = m->d_maxval;                // [tp][n]
    int rc_ = comm_allgather(ctx, idx_all + (size_t)ctx->tp_rank * n, idx_all, (size_t)n * 4);
    if (!rc_) rc_ = comm_allgather(ctx, val_all + (size_t)ctx->tp_rank * n, val_all, (size_t)n * 4);
    if (rc_) return rc_;
    std::vector<uint32_t> hi((size_t)tp * n);
    std::vector<float> hv((size_t)tp * n);
    HIPCHK(ctx, hipMemcpyAsync(hi.data(), idx_all, hi.size() * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipMemcpyAsync(hv.data(), val_all, hv.size() * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    std::vector<uint32_t> best(n);
    for (int i = 0; i < n; ++i) {
        float bv = -INFINITY; uint32_t bi = 0;
        for (int r = 0; r < tp; ++r) {
            const float v = hv[(size_t)r * n + i];
            const uint32_t gi = hi[(size_t)r * n + i] + (uint32_t)r * Vl;
            if (r == 0 || v > bv || (v == bv && gi > bi)) { bv = v; bi = gi; }
        }
        best[i] = bi;
    }
    HIPCHK(ctx, hipMemcpyAsync(m->d_next, best.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
    if (next_ids) memcpy(next_ids, best.data(), (size_t)n * 4);
    if (last_logits) {
        // every rank sends its [n][Vl] shard; host interleaves into [n][V]
        std::vector<float> shard((size_t)tp * n * Vl);
        float* dall = nullptr;
        HIPCHK(ctx, hipMalloc((void**)&dall, shard.size() * 4));
        int rg = comm_allgather(ctx, m->logits, dall, (size_t)n * Vl * 4);
        hipError_t e = rg == 0 ? hipMemcpyAsync(shard.data(), dall, shard.size() * 4, hipMemcpyDeviceToHost, s) : hipSuccess;
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        (void)hipFree(dall);
        if (rg) return rg;
        HIPCHK(ctx, e);
        for (int rk = 0; rk < tp; ++rk)
            for (int i = 0; i < n; ++i)
                memcpy(last_logits + (size_t)i * V + (size_t)rk * Vl, shard.data() + ((size_t)rk * n + i) * Vl, (size_t)Vl * 4);
    }
    HIPCHK(ctx, hipStreamSynchronize(s));
    return NVLLM_OK;
}

struct RowPlan {
    std::vector<uint32_t> ids;
    std::vector<int> pos, slot, tile_row0, tile_nrows, tile_slot, last_rows;
};

// upload one chunk's metadata through the pinned staging buffer (one H2D copy per array kind)
static int upload_chunk(nvllm_model* m, const RowPlan& p, int r0, int R, int t0, int T, const std::vector<int>& last_local) {
    nvllm_ctx* ctx = m->ctx;
    hipStream_t s = ctx->stream;
    HIPCHK(ctx, hipStreamSynchronize(s));  // staging buffer reuse
    int* st = (int*)m->h_stage;
    int* h_ids = st; int* h_pos = st + R; int* h_slot = st + 2 * R;
    int* h_t0 = st + 3 * R; int* h_tn = h_t0 + T; int* h_ts = h_tn + T; int* h_last = h_ts + T;
    memcpy(h_ids, p.ids.data() + r0, (size_t)R * 4);
    memcpy(h_pos, p.pos.data() + r0, (size_t)R * 4);
    memcpy(h_slot, p.slot.data() + r0, (size_t)R * 4);
    for (int i = 0; i < T; ++i) { h_t0[i] = p.tile_row0[t0 + i] - r0; h_tn[i] = p.tile_nrows[t0 + i]; h_ts[i] = p.tile_slot[t0 + i]; }
    for (size_t i = 0; i < last_local.size(); ++i) h_last[i] = last_local[i];
    // attention load balance: q-tiles ordered by context length, longest first (stable for ties); an empty padding tile
    // (prefill groups of four) has no last row: it sorts as -1, behind every real tile
    int* h_ord = h_last + m->max_seqs;
    int* h_tl = h_ord + T;  // position of each tile's last row (-1: padding tile)
    for (int i = 0; i < T; ++i) h_tl[i] = p.tile_nrows[t0 + i] > 0 ? p.pos[p.tile_row0[t0 + i] + p.tile_nrows[t0 + i] - 1] : -1;
    for (int i = 0; i < T; ++i) h_ord[i] = i;
    std::stable_sort(h_ord, h_ord + T, [&](int x, int y) { return h_tl[x] > h_tl[y]; });
    HIPCHK(ctx, hipMemcpyAsync(m->d_ids, h_ids, (size_t)R * 4, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(m->d_pos, h_pos, (size_t)R * 4, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(m->d_slot, h_slot, (size_t)R * 4, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(m->d_tile_row0, h_t0, (size_t)T * 4, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(m->d_tile_nrows, h_tn, (size_t)T * 4, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(m->d_tile_slot, h_ts, (size_t)T * 4, hipMemcpyHostToDevice, s));
    if (!last_local.empty())
        HIPCHK(ctx, hipMemcpyAsync(m->d_last_rows, h_last, last_local.size() * 4, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(m->d_tile_order, h_ord, (size_t)T * 4, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(m->d_tile_last, h_tl, (size_t)T * 4, hipMemcpyHostToDevice, s));
    if (T % kPrefillTileGroup == 0) {  // prefill attention: groups of four q-tiles, longest context first
        int* h_go = h_tl + T;
        const int G = T / kPrefillTileGroup;
        std::vector<int> glast(G);
        for (int g = 0; g < G; ++g) {
            int mx = -1;
            for (int k = 0; k < kPrefillTileGroup; ++k) mx = std::max(mx, h_tl[g * kPrefillTileGroup + k]);
            glast[g] = mx;
            h_go[g] = g;
        }
        std::stable_sort(h_go, h_go + G, [&](int x, int y) { return glast[x] > glast[y]; });
        // every other block of 256 workgroups (256 / kv heads groups) backwards: workgroups i and i + 256 tend to share
        // a CU, so the longest context gets the shortest of the next block as its neighbour and the CU to itself sooner
        const int per = std::max(1, 256 / std::max(1, m->kv_l));
        for (int b0 = per; b0 < G; b0 += 2 * per) std::reverse(h_go + b0, h_go + std::min(G, b0 + per));
        HIPCHK(ctx, hipMemcpyAsync(m->d_group_order, h_go, (size_t)G * 4, hipMemcpyHostToDevice, s));
    }
    return NVLLM_OK;
}

static int step_impl(nvllm_model* m, int n_seqs, const int64_t* seq_ids, const uint32_t* const* tokens, const int32_t* lens,
                     int is_prefill, uint32_t* next_ids, float* last_logits, const float* temperatures, uint64_t seed);

extern "C" int nvllm_step(nvllm_model* m, int n_seqs, const int64_t* seq_ids, const uint32_t* const* tokens,
                          const int32_t* lens, int is_prefill, uint32_t* next_ids, float* last_logits) {
    return step_impl(m, n_seqs, seq_ids, tokens, lens, is_prefill, next_ids, last_logits, nullptr, 0);
}

// per-row key of the sampling counter RNG: (seed, sequence id, position of the token being drawn)
static inline uint64_t sample_key(uint64_t seed, int64_t seq_id, int len) {
    return synth_finalize(synth_finalize(seed * 0x9E3779B97F4A7C15ULL + (uint64_t)seq_id) + (uint64_t)len * 0xD1B54A32D192ED03ULL);
}

extern "C" int nvllm_step_sample(nvllm_model* m, int n_seqs, const int64_t* seq_ids, const uint32_t* const* tokens,
                                 const int32_t* lens, int is_prefill, const float* temperatures, uint64_t seed,
                                 uint32_t* next_ids, float* last_logits) {
    if (m && !temperatures) return fail(m->ctx, NVLLM_EINVAL, "temperatures is NULL");
    return step_impl(m, n_seqs, seq_ids, tokens, lens, is_prefill, next_ids, last_logits, temperatures, seed);
}

static int step_impl(nvllm_model* m, int n_seqs, const int64_t* seq_ids, const uint32_t* const* tokens, const int32_t* lens,
                     int is_prefill, uint32_t* next_ids, float* last_logits, const float* temperatures, uint64_t seed) {
    if (!m) return NVLLM_EINVAL;
    nvllm_ctx* ctx = m->ctx;
    if (!m->finalized) return fail(ctx, NVLLM_ESTATE, "nvllm_model_finalize not called");
    if (!m->num_blocks) return fail(ctx, NVLLM_ESTATE, "nvllm_kv_alloc not called");
    if (!m->pending.empty()) return fail(ctx, NVLLM_ESTATE, "collect the enqueued decode steps first");
    if (n_seqs == 0) return NVLLM_OK;  // llm_engine.rs:147-149
    if (n_seqs < 0 || !seq_ids || !tokens || !lens) return fail(ctx, NVLLM_EINVAL, "bad step arguments");
    if (n_seqs > m->max_seqs) return fail(ctx, NVLLM_EINVAL, "%d sequences > max_seqs %d", n_seqs, m->max_seqs);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    for (int i = 0; i < n_seqs; ++i) {
        if (lens[i] < 1) return fail(ctx, NVLLM_EINVAL, "sequence %d is empty", i);
        if (lens[i] > m->rope_len) return fail(ctx, NVLLM_EINVAL, "sequence %d: %d tokens > max positions %d", i, lens[i], m->rope_len);
        for (int t = 0; t < lens[i]; ++t)
            if (tokens[i][t] >= (uint32_t)m->cfg.vocab_size) return fail(ctx, NVLLM_EINVAL, "sequence %d: token id %u out of range", i, tokens[i][t]);
        for (int j = 0; j < i; ++j)
            if (seq_ids[j] == seq_ids[i]) return fail(ctx, NVLLM_EINVAL, "duplicate seq_id in one step");
    }
    // sequence bookkeeping + block allocation (the part BlockManager::allocate/may_append leave undone)
    std::vector<int> start(n_seqs);
    int new_slots = 0;
    for (int i = 0; i < n_seqs; ++i) if (!m->seqs.count(seq_ids[i])) ++new_slots;
    if (new_slots > (int)m->free_slots.size()) return fail(ctx, NVLLM_ENOMEM, "no free sequence slots (max_seqs %d)", m->max_seqs);
    for (int i = 0; i < n_seqs; ++i) {
        auto it = m->seqs.find(seq_ids[i]);
        if (it == m->seqs.end()) {
            SeqState s;
            s.slot = m->free_slots.back();
            m->free_slots.pop_back();
            it = m->seqs.emplace(seq_ids[i], s).first;
        } else if (is_prefill) {
            release_seq(m, it->second);
        }
        SeqState& s = it->second;
        if (s.cached >= lens[i]) s.cached = lens[i] - 1;  // nothing new: recompute the last token
        start[i] = s.cached;
        int rc = ensure_blocks(m, s, lens[i]);
        if (rc) return rc;
    }
    HIPCHK(ctx, hipMemcpyAsync(m->d_block_tables, m->h_block_tables.data(), m->h_block_tables.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    m->want_logits = last_logits != nullptr || temperatures != nullptr;  // sampling reads the last-row logits on the device
    m->cur_n = n_seqs;
    set_attn_split(m, n_seqs, *std::max_element(lens, lens + n_seqs));
    // rows and q-tiles
    RowPlan p;
    bool all_single = true;
    for (int i = 0; i < n_seqs; ++i) if (lens[i] - start[i] != 1) all_single = false;
    const int qt = all_single ? 1 : 2;
    const int tpt = attn_tokens_per_tile(m->gqa, qt);
    std::vector<int> seq_last_row(n_seqs);
    for (int i = 0; i < n_seqs; ++i) {
        const int slot = m->seqs[seq_ids[i]].slot;
        for (int t = start[i]; t < lens[i]; ++t) {
            p.ids.push_back(tokens[i][t]); p.pos.push_back(t); p.slot.push_back(slot);
        }
        seq_last_row[i] = (int)p.ids.size() - 1;
    }
    const int total_rows = (int)p.ids.size();
    // chunks of <= max_rows rows; tiles never cross a chunk or a sequence
    int64_t kv_read_tokens = 0;
    for (int i = 0; i < n_seqs; ++i) kv_read_tokens += lens[i];
    int r0 = 0, logits_row = 0;
    std::vector<int> row_seq(total_rows);
    { int r = 0; for (int i = 0; i < n_seqs; ++i) for (int t = start[i]; t < lens[i]; ++t) row_seq[r++] = i; }
    while (r0 < total_rows) {
        const int R = std::min(m->max_rows, total_rows - r0);
        p.tile_row0.clear(); p.tile_nrows.clear(); p.tile_slot.clear();
        int r = r0;
        while (r < r0 + R) {
            int e = r;
            while (e < r0 + R && e - r < tpt && row_seq[e] == row_seq[r]) ++e;
            p.tile_row0.push_back(r); p.tile_nrows.push_back(e - r); p.tile_slot.push_back(p.slot[r]);
            // prefill attention shares K/V between the four q-tiles of a workgroup: a sequence's tiles start on a multiple
            // of four, its last group is padded with empty tiles
            if (qt == 2 && m->vlocache.empty() && (e == r0 + R || row_seq[e] != row_seq[r]))
                while (p.tile_row0.size() % kPrefillTileGroup) { p.tile_row0.push_back(r); p.tile_nrows.push_back(0); p.tile_slot.push_back(p.slot[r]); }
            r = e;
        }
        std::vector<int> last_local;
        for (int i = 0; i < n_seqs; ++i)
            if (seq_last_row[i] >= r0 && seq_last_row[i] < r0 + R) last_local.push_back(seq_last_row[i] - r0);
        int rc = upload_chunk(m, p, r0, R, 0, (int)p.tile_row0.size(), last_local);
        if (rc) return rc;
        rc = forward_chunk(m, R, (int)p.tile_row0.size(), qt, (int)last_local.size(), logits_row);
        if (rc) return rc;
        logits_row += (int)last_local.size();
        r0 += R;
    }
    if (temperatures) {
        // sample_token on the device (llm_engine.rs:97-133): overwrites the greedy ids the LM head's arg-max left.  Under
        // TP every rank draws over its vocabulary shard with the SAME random stream (indexed by global id); the best
        // (score, id) pair wins in finish_logits exactly like the greedy (max, id) pair.
        std::vector<uint64_t> keys(n_seqs);
        for (int i = 0; i < n_seqs; ++i) keys[i] = sample_key(seed, seq_ids[i], lens[i]);
        HIPCHK(ctx, hipMemcpyAsync(m->d_keys, keys.data(), (size_t)n_seqs * 8, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(m->d_temps, temperatures, (size_t)n_seqs * 4, hipMemcpyHostToDevice, ctx->stream));
        const int tps = ctx->tp_size;
        uint32_t* ids_dst = tps == 1 ? m->d_next : m->d_next + n_seqs + (size_t)ctx->tp_rank * n_seqs;
        float* val_dst = tps == 1 ? nullptr : m->d_maxval + (size_t)ctx->tp_rank * n_seqs;
        HIPCHK(ctx, launch_sample_rows(m->logits, n_seqs, m->V_l, m->V_l, m->d_temps, m->d_keys, ctx->tp_rank * m->V_l, ids_dst, val_dst, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // keys / temperatures are read from pageable host memory above
    }
    int rc = finish_logits(m, n_seqs, next_ids, last_logits);
    if (rc) return rc;
    m->last_ids.assign(seq_ids, seq_ids + n_seqs);
    m->last_lens.assign(lens, lens + n_seqs);
    m->decode_resident = all_single && total_rows <= m->max_rows;
    for (int i = 0; i < n_seqs; ++i) m->seqs[seq_ids[i]].cached = lens[i];
    m->last_bytes = nvllm_model_weight_bytes(m) + kv_read_tokens * nvllm_kv_bytes_per_token(m) +
                    (last_logits ? (int64_t)4 * n_seqs * m->cfg.vocab_size : 0);
    return NVLLM_OK;
}

// enqueue one more decode step for the resident batch (no host sync): everything up to the device arg-max
static int decode_core(nvllm_model* m) {
    nvllm_ctx* ctx = m->ctx;
    const int n = (int)m->last_ids.size();
    if (n == 0) return fail(ctx, NVLLM_ESTATE, "no previous step to continue");
    if (n > m->max_rows) return fail(ctx, NVLLM_ESTATE, "%d decode rows > %d rows of step buffers", n, m->max_rows);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    bool table_dirty = false;
    for (int i = 0; i < n; ++i) {
        auto it = m->seqs.find(m->last_ids[i]);
        if (it == m->seqs.end()) return fail(ctx, NVLLM_ESTATE, "sequence %lld of the previous step is gone", (long long)m->last_ids[i]);
        SeqState& st = it->second;
        const int len = m->last_lens[i] + 1;
        if (len > m->rope_len) return fail(ctx, NVLLM_EINVAL, "sequence %d would exceed max positions %d", i, m->rope_len);
        const size_t before = st.blocks.size();
        int rc = ensure_blocks(m, st, len);
        if (rc) return rc;
        table_dirty |= st.blocks.size() != before;
    }
    if (table_dirty)
        HIPCHK(ctx, hipMemcpyAsync(m->d_block_tables, m->h_block_tables.data(), m->h_block_tables.size() * 4, hipMemcpyHostToDevice, s));
    if (m->decode_resident) {
        // ids <- last greedy ids, pos += 1: the batch metadata is already on the device
        HIPCHK(ctx, launch_advance_decode(m->d_ids, m->d_next, m->d_pos, n, s));
    } else {
        // previous step was a prefill (or a multi-chunk step): rebuild decode metadata once
        RowPlan p;
        std::vector<int> last_local(n);
        for (int i = 0; i < n; ++i) {
            const SeqState& st = m->seqs[m->last_ids[i]];
            p.ids.push_back(0); p.pos.push_back(m->last_lens[i]); p.slot.push_back(st.slot);
            p.tile_row0.push_back(i); p.tile_nrows.push_back(1); p.tile_slot.push_back(st.slot);
            last_local[i] = i;
        }
        int rc0 = upload_chunk(m, p, 0, n, 0, n, last_local);
        if (rc0) return rc0;
        HIPCHK(ctx, hipMemcpyAsync(m->d_ids, m->d_next, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
        m->decode_resident = true;
    }
    m->want_logits = false;
    m->cur_n = n;
    set_attn_split(m, n, *std::max_element(m->last_lens.begin(), m->last_lens.end()) + 1);
    int rc = forward_chunk(m, n, n, 1, n, 0);
    if (rc) return rc;
    int64_t kv_tokens = 0;
    for (int i = 0; i < n; ++i) {
        m->last_lens[i] += 1;
        m->seqs[m->last_ids[i]].cached = m->last_lens[i];
        kv_tokens += m->last_lens[i];
    }
    m->last_bytes = nvllm_model_weight_bytes(m) + kv_tokens * nvllm_kv_bytes_per_token(m);
    return NVLLM_OK;
}

extern "C" int nvllm_decode_next(nvllm_model* m, uint32_t* next_ids) {
    if (!m) return NVLLM_EINVAL;
    if (!m->pending.empty()) return fail(m->ctx, NVLLM_ESTATE, "collect the enqueued decode steps first");
    int rc = decode_core(m);
    if (rc) return rc;
    return finish_logits(m, (int)m->last_ids.size(), next_ids, nullptr);
}

// Pipelined form: enqueue returns as soon as the step is on the stream (its ids travel to a pinned slot behind
// an event); collect waits for the OLDEST enqueued step.  Keeping one step in flight hides the host round trip
// (the reference's engine consumes ids once per step, llm_engine.rs:239-242; asynchronous scheduling sees them one
// step late).  Tensor-parallel contexts fall back to a synchronous step inside enqueue.
extern "C" int nvllm_decode_enqueue(nvllm_model* m) {
    if (!m) return NVLLM_EINVAL;
    nvllm_ctx* ctx = m->ctx;
    const int n = (int)m->last_ids.size();
    if ((int)m->pending.size() >= kMaxPending) return fail(ctx, NVLLM_ESTATE, "too many decode steps in flight (max %d)", kMaxPending);
    int rc = decode_core(m);
    if (rc) return rc;
    if (!m->pin_ids) {
        HIPCHK(ctx, hipHostMalloc((void**)&m->pin_ids, (size_t)kMaxPending * m->max_seqs * 4, hipHostMallocDefault));
        for (int i = 0; i < kMaxPending; ++i)
            if (!m->pend_ev[i]) HIPCHK(ctx, hipEventCreateWithFlags(&m->pend_ev[i], hipEventDisableTiming));
    }
    const int slot = m->pend_next;
    m->pend_next = (m->pend_next + 1) % kMaxPending;
    uint32_t* dst = m->pin_ids + (size_t)slot * m->max_seqs;
    if (ctx->tp_size > 1) {
        rc = finish_logits(m, n, dst, nullptr);  // synchronous under TP (host picks the best shard)
        if (rc) return rc;
    } else {
        HIPCHK(ctx, hipMemcpyAsync(dst, m->d_next, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(ctx, hipEventRecord(m->pend_ev[slot], ctx->stream));
    m->pending.push_back(slot);
    return NVLLM_OK;
}

extern "C" int nvllm_decode_collect(nvllm_model* m, uint32_t* next_ids) {
    if (!m) return NVLLM_EINVAL;
    if (m->pending.empty()) return fail(m->ctx, NVLLM_ESTATE, "no enqueued decode step");
    const int slot = m->pending.front();
    m->pending.erase(m->pending.begin());
    HIPCHK(m->ctx, hipEventSynchronize(m->pend_ev[slot]));
    if (next_ids) memcpy(next_ids, m->pin_ids + (size_t)slot * m->max_seqs, m->last_ids.size() * 4);
    return NVLLM_OK;
}

extern "C" int64_t nvllm_last_step_bytes(const nvllm_model* m) { return m ? m->last_bytes : 0; }

// Diagnostic build only (make -C csrc stamps): record s_memrealtime stamps inside every kernel of the fused decode path.
// enable != 0 arms the recording for the following steps; read copies launch `launch` of the LAST step:
// [1024 workgroups][16 waves][8 points] u64 (zero = not written), launches in issue order (QKV, attention, o_proj,
// gate/up, down per layer).  The product library returns NVLLM_ESTATE.
// tuning switches of a model (A/B and tests): "stream_combine" = the streaming GEMM's in-launch split-K combine;
// "tile_min_wgs" = smallest grid the prefill tile GEMM is used for (1: whenever the shape fits; 0: never)
extern "C" int nvllm_debug_set_option(nvllm_model* m, const char* name, int value) {
    if (!m || !name) return NVLLM_EINVAL;
    if (!strcmp(name, "stream_combine")) { m->opt_stream_combine = value; return NVLLM_OK; }
    if (!strcmp(name, "oneshot_allreduce")) { m->opt_oneshot_allreduce = value; return NVLLM_OK; }  // takes effect at the next kv_alloc
    if (!strcmp(name, "oneshot_spins")) { m->opt_oneshot_spins = std::max(1, value); return NVLLM_OK; }
    if (!strcmp(name, "kv_k_bits")) {  // takes effect at the next kv_alloc (which insists on kv_v_bits = 24 with it)
        if (value != 16 && value != 24) return fail(m->ctx, NVLLM_EINVAL, "kv_k_bits is 16 or 24");
        m->opt_kv_k_bits = value; return NVLLM_OK;
    }
    if (!strcmp(name, "kv_v_bits")) {  // takes effect at the next kv_alloc
        if (value != 16 && value != 24) return fail(m->ctx, NVLLM_EINVAL, "kv_v_bits is 16 or 24");
        m->opt_kv_v_bits = value; return NVLLM_OK;
    }
    if (!strcmp(name, "oneshot_skip_push")) { m->opt_oneshot_skip_push = std::max(0, value); return NVLLM_OK; }  // test hook
    if (!strcmp(name, "no_fused")) { m->opt_no_fused = value; return NVLLM_OK; }    // decode through the generic path
    if (!strcmp(name, "no_xpack")) { m->opt_no_xpack = value; return NVLLM_OK; }    // row-major activation planes
    if (!strcmp(name, "no_rowpar")) { m->opt_no_rowpar = value; return NVLLM_OK; }  // generic path: no whole-K row-parallel GEMMs
    if (!strcmp(name, "no_attn_prologue")) { m->opt_no_attn_prologue = value; return NVLLM_OK; }
    if (!strcmp(name, "tile_fuse_qk")) { m->opt_tile_fuse_qk = value; return NVLLM_OK; }
    if (!strcmp(name, "tile_min_wgs")) { m->opt_tile_min_wgs = value; return NVLLM_OK; }  // <= 0: never use the tile GEMM
    return fail(m->ctx, NVLLM_EINVAL, "unknown option '%s'", name);
}

// counters for tests: "oneshot_calls" = all-reduces this model's context has run on the one-shot device path;
// "tile_gemm_launches" = projections this model has run on the prefill tile GEMM
extern "C" int nvllm_debug_get_counter(nvllm_model* m, const char* name, int64_t* value) {
    if (!m || !name || !value) return NVLLM_EINVAL;
    if (!strcmp(name, "oneshot_calls")) { *value = (int64_t)m->ctx->os_calls; return NVLLM_OK; }
    if (!strcmp(name, "tile_gemm_launches")) { *value = m->tile_launches; return NVLLM_OK; }
    if (!strcmp(name, "kv_f16_saturated") || !strcmp(name, "kv_f16_absmax_bits")) {
        // scan of the whole K/V pool (debug, off the hot path): elements the f16_sat clamp of a cache write produced, and the
        // largest magnitude stored (f16 bits; 0x7BFF = 65504 is the clamp)
        if (m->kcache.empty()) return fail(m->ctx, NVLLM_ESTATE, "kv_alloc first");
        nvllm_ctx* ctx = m->ctx;
        HIPCHK(ctx, hipSetDevice(ctx->device));
        unsigned long long* d = nullptr;
        HIPCHK(ctx, hipMalloc((void**)&d, 16));
        hipError_t e = hipMemsetAsync(d, 0, 16, ctx->stream);
        const int64_t per_layer = (int64_t)m->num_blocks * m->kv_l * kBlockTokens * m->hd;
        for (int l = 0; l < m->L && e == hipSuccess; ++l) {
            e = launch_f16_scan(m->kcache[l], per_layer, d, reinterpret_cast<unsigned*>(d + 1), ctx->stream);
            if (e == hipSuccess) e = launch_f16_scan(m->vcache[l], per_layer, d, reinterpret_cast<unsigned*>(d + 1), ctx->stream);
        }
        unsigned long long h[2] = {0, 0};
        if (e == hipSuccess) e = hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        (void)hipFree(d);
        HIPCHK(ctx, e);
        *value = !strcmp(name, "kv_f16_saturated") ? (int64_t)h[0] : (int64_t)(h[1] & 0xFFFFu);
        return NVLLM_OK;
    }
    return fail(m->ctx, NVLLM_EINVAL, "unknown counter '%s'", name);
}

extern "C" int nvllm_debug_stamps(nvllm_model* m, int enable) {
    if (!m) return NVLLM_EINVAL;
#ifdef NVLLM_STAMPS
    HIPCHK(m->ctx, hipStreamSynchronize(m->ctx->stream));
    if (enable && !m->stamps) HIPCHK(m->ctx, hipMalloc((void**)&m->stamps, kStampLaunches * kStampStride * 8));
    if (enable) HIPCHK(m->ctx, hipMemset(m->stamps, 0, kStampLaunches * kStampStride * 8));
    m->stamps_on = enable != 0;
    // enable == 2: the tile GEMM launches of the following steps record instead of the attention / decode kernels
    // (same buffer, launch index = order of the tile launches)
    nvllm::tile_gemm_stamps_arm(enable == 2 ? m->stamps : nullptr, kStampLaunches);
    if (enable == 2) m->stamps_on = false;
    return NVLLM_OK;
#else
    return fail(m->ctx, NVLLM_ESTATE, "library built without NVLLM_STAMPS (make -C nano-vllm-candle_amd/csrc stamps)");
#endif
}
extern "C" int nvllm_debug_stamps_read(nvllm_model* m, int launch, uint64_t* out, int64_t capacity_u64) {
    if (!m || !out) return NVLLM_EINVAL;
#ifdef NVLLM_STAMPS
    if (!m->stamps || launch < 0 || launch >= kStampLaunches || capacity_u64 < (int64_t)kStampStride) return fail(m->ctx, NVLLM_EINVAL, "bad stamps_read arguments");
    HIPCHK(m->ctx, hipStreamSynchronize(m->ctx->stream));
    HIPCHK(m->ctx, hipMemcpy(out, m->stamps + (size_t)launch * kStampStride, kStampStride * 8, hipMemcpyDeviceToHost));
    return NVLLM_OK;
#else
    return fail(m->ctx, NVLLM_ESTATE, "library built without NVLLM_STAMPS");
#endif
}

extern "C" int nvllm_profile_kernel(nvllm_model* m, int kind) {
    if (!m || kind < 0 || kind > PROF_EMPTY) return NVLLM_EINVAL;
    HIPCHK(m->ctx, hipStreamSynchronize(m->ctx->stream));
    m->prof_kind = kind;
    m->prof_used = 0;
    return NVLLM_OK;
}
extern "C" int nvllm_profile_read(nvllm_model* m, double* total_ms, int64_t* launches) {
    if (!m || !total_ms || !launches) return NVLLM_EINVAL;
    HIPCHK(m->ctx, hipStreamSynchronize(m->ctx->stream));
    double t = 0;
    for (size_t i = 0; i + 1 < m->prof_used; i += 2) {
        float ms = 0;
        HIPCHK(m->ctx, hipEventElapsedTime(&ms, m->prof_ev[i], m->prof_ev[i + 1]));
        t += ms;
    }
    *total_ms = t;
    *launches = (int64_t)(m->prof_used / 2);
    m->prof_used = 0;
    return NVLLM_OK;
}

// ---------------------------------------------------------------------------------------------------
// fine seam: single ops on raw device pointers (layer-level parity tests; src/layers/*.rs surface)
// ---------------------------------------------------------------------------------------------------
struct nvllm_weight {
    PackedW w;     // padded: N to 16, K to 32 (zeros)
    int N = 0, K = 0;  // logical
};

struct TmpBufs {  // frees on scope exit (after a stream sync done by the caller)
    std::vector<void*> p;
    ~TmpBufs() { for (void* q : p) (void)hipFree(q); }
    template <typename T>
    hipError_t get(T** out, size_t count) {
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, std::max<size_t>(count * sizeof(T), 16));
        if (e == hipSuccess) { p.push_back(q); *out = (T*)q; }
        return e;
    }
};

extern "C" int nvllm_op_pack_weight(nvllm_ctx* ctx, const void* host_w, int dtype, int N, int K, nvllm_weight** out) {
    if (!ctx || !host_w || !out || N < 1 || K < 1) return fail(ctx, NVLLM_EINVAL, "bad pack_weight arguments");
    if (dtype != NVLLM_DTYPE_F32 && dtype != NVLLM_DTYPE_BF16) return fail(ctx, NVLLM_EINVAL, "dtype %d unsupported", dtype);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int Np = (N + 15) / 16 * 16, Kp = (K + 127) / 128 * 128;
    std::vector<uint16_t> tmp((size_t)Np * Kp, 0);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k)
            tmp[(size_t)n * Kp + k] = dtype == NVLLM_DTYPE_F32 ? host_f32_to_bf16(((const float*)host_w)[(size_t)n * K + k])
                                                               : ((const uint16_t*)host_w)[(size_t)n * K + k];
    nvllm_weight* w = new nvllm_weight();
    w->N = N; w->K = K; w->w.N = Np; w->w.K = Kp;
    bf16_bits* dtmp = nullptr;
    hipError_t e = hipMalloc((void**)&w->w.data, w->w.bytes());
    if (e == hipSuccess) e = hipMalloc((void**)&dtmp, tmp.size() * 2);
    if (e == hipSuccess) e = hipMemcpy(dtmp, tmp.data(), tmp.size() * 2, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_pack_rows(w->w, 0, Np, dtmp, Kp, -1, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (dtmp) (void)hipFree(dtmp);
    if (e != hipSuccess) { if (w->w.data) (void)hipFree(w->w.data); delete w; HIPCHK(ctx, e); }
    *out = w;
    return NVLLM_OK;
}
extern "C" int nvllm_op_free_weight(nvllm_ctx* ctx, nvllm_weight* w) {
    if (!w) return NVLLM_OK;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(w->w.data);
    delete w;
    return NVLLM_OK;
}

extern "C" int nvllm_op_linear(nvllm_ctx* ctx, const float* x, const nvllm_weight* w, const float* bias, int M, float* y) {
    if (!ctx || !x || !w || !y || M < 1) return fail(ctx, NVLLM_EINVAL, "bad linear arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    TmpBufs t;
    bf16_bits *hi, *lo;
    float* slabs;
    const int Kp = w->w.K, Np = w->w.N;
    GemmPlan p = plan_gemm(M, Np, Kp, 8);
    HIPCHK(ctx, t.get(&hi, (size_t)M * Kp));
    HIPCHK(ctx, t.get(&lo, (size_t)M * Kp));
    HIPCHK(ctx, t.get(&slabs, (size_t)p.n_split * M * Np));
    HIPCHK(ctx, launch_split_hilo_pad(x, M, w->K, Kp, hi, lo, s));
    HIPCHK(ctx, launch_gemm(p, hi, lo, Kp, w->w, slabs, M, s));
    HIPCHK(ctx, launch_slab_sum_ld(slabs, p.n_split, (int64_t)M * Np, Np, bias, M, w->N, y, w->N, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    return NVLLM_OK;
}

extern "C" int nvllm_op_rmsnorm(nvllm_ctx* ctx, const float* x, const float* residual, const float* weight, double eps,
                                int rows, int n, float* y, float* residual_out) {
    if (!ctx || !x || !weight || !y || rows < 1 || n < 1) return fail(ctx, NVLLM_EINVAL, "bad rmsnorm arguments");
    if ((residual == nullptr) != (residual_out == nullptr)) return fail(ctx, NVLLM_EINVAL, "residual and residual_out go together");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (n % 4 == 0 && n <= 8192) {  // the kernel the step path uses
        NormArgs a;
        a.in = x; a.residual_in = residual; a.residual_out = residual_out; a.weight = weight; a.eps = (float)eps; a.H = n; a.y = y;
        HIPCHK(ctx, launch_add_rmsnorm(a, rows, ctx->stream));
    } else {
        HIPCHK(ctx, launch_rmsnorm_generic(x, residual, weight, (float)eps, rows, n, y, residual_out, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return NVLLM_OK;
}

extern "C" int nvllm_op_silu_mul(nvllm_ctx* ctx, const float* x, int rows, int n, float* y) {
    if (!ctx || !x || !y || rows < 1 || n < 1) return fail(ctx, NVLLM_EINVAL, "bad silu_mul arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (n % 4 == 0) HIPCHK(ctx, launch_silu_mul(x, 1, 0, rows, n, nullptr, nullptr, y, ctx->stream));
    else HIPCHK(ctx, launch_silu_mul_generic(x, rows, n, y, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return NVLLM_OK;
}

static void host_rope_table(int hd, float base, int T, std::vector<float>& hc, std::vector<float>& hs) {
    const int half = hd / 2;
    hc.resize((size_t)T * half); hs.resize((size_t)T * half);
    for (int j = 0; j < half; ++j) {
        const float inv_freq = 1.0f / powf(base, (2.0f * (float)j) / (float)hd);
        for (int p = 0; p < T; ++p) {
            const float ang = (float)p * inv_freq;
            hc[(size_t)p * half + j] = cosf(ang);
            hs[(size_t)p * half + j] = sinf(ang);
        }
    }
}

extern "C" int nvllm_op_rope(nvllm_ctx* ctx, float* q, float* k, int B, int nh, int kv, int T, int hd, float base) {
    if (!ctx || !q || !k || B < 1 || nh < 1 || kv < 1 || T < 1 || hd < 2 || hd % 2) return fail(ctx, NVLLM_EINVAL, "bad rope arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::vector<float> hc, hs;
    host_rope_table(hd, base, T, hc, hs);
    TmpBufs t;
    float *dc, *ds;
    HIPCHK(ctx, t.get(&dc, hc.size()));
    HIPCHK(ctx, t.get(&ds, hs.size()));
    HIPCHK(ctx, hipMemcpyAsync(dc, hc.data(), hc.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ds, hs.data(), hs.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, launch_rope_bhtd(q, B, nh, T, hd, dc, ds, ctx->stream));
    HIPCHK(ctx, launch_rope_bhtd(k, B, kv, T, hd, dc, ds, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return NVLLM_OK;
}

extern "C" int nvllm_op_attention(nvllm_ctx* ctx, const float* q, const float* k, const float* v, int B, int nh, int kv,
                                  int T, int hd, float scale, float* out) {
    if (!ctx || !q || !k || !v || !out || B < 1 || T < 1) return fail(ctx, NVLLM_EINVAL, "bad attention arguments");
    if (hd != 64 && hd != 128) return fail(ctx, NVLLM_EINVAL, "head_dim %d unsupported (64 or 128)", hd);
    if (nh % kv || nh / kv > 16) return fail(ctx, NVLLM_EINVAL, "bad GQA shape");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int gqa = nh / kv, rows = B * T, bps = (T + kBlockTokens - 1) / kBlockTokens, nblk = B * bps;
    TmpBufs t;
    float *qr, *kr, *vr;
    int *dpos, *dslot, *dbt, *dt0, *dtn, *dts;
    KvLayout kvl;
    kvl.kv_l = kv; kvl.hd = hd;
    const size_t cache_elems = (size_t)nblk * kv * kBlockTokens * hd;
    HIPCHK(ctx, t.get(&qr, (size_t)rows * nh * hd));
    HIPCHK(ctx, t.get(&kr, (size_t)rows * kv * hd));
    HIPCHK(ctx, t.get(&vr, (size_t)rows * kv * hd));
    HIPCHK(ctx, t.get(&kvl.k, cache_elems));
    HIPCHK(ctx, t.get(&kvl.v, cache_elems));
    HIPCHK(ctx, hipMemsetAsync(kvl.k, 0, cache_elems * 2, s));
    HIPCHK(ctx, hipMemsetAsync(kvl.v, 0, cache_elems * 2, s));
    const int qt = 2, tpt = attn_tokens_per_tile(gqa, qt);
    std::vector<int> hpos(rows), hslot(rows), hbt(nblk), ht0, htn, hts;
    for (int b = 0; b < B; ++b) {
        for (int tt = 0; tt < T; ++tt) { hpos[b * T + tt] = tt; hslot[b * T + tt] = b; }
        for (int j = 0; j < bps; ++j) hbt[b * bps + j] = b * bps + j;
        for (int tt = 0; tt < T; tt += tpt) { ht0.push_back(b * T + tt); htn.push_back(std::min(tpt, T - tt)); hts.push_back(b); }
        while (ht0.size() % kPrefillTileGroup) { ht0.push_back(b * T); htn.push_back(0); hts.push_back(b); }  // launch_attn_prefill's tile groups
    }
    const int nt = (int)ht0.size();
    HIPCHK(ctx, t.get(&dpos, rows)); HIPCHK(ctx, t.get(&dslot, rows)); HIPCHK(ctx, t.get(&dbt, nblk));
    HIPCHK(ctx, t.get(&dt0, nt)); HIPCHK(ctx, t.get(&dtn, nt)); HIPCHK(ctx, t.get(&dts, nt));
    HIPCHK(ctx, hipMemcpyAsync(dpos, hpos.data(), (size_t)rows * 4, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(dslot, hslot.data(), (size_t)rows * 4, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(dbt, hbt.data(), (size_t)nblk * 4, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(dt0, ht0.data(), (size_t)nt * 4, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(dtn, htn.data(), (size_t)nt * 4, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(dts, hts.data(), (size_t)nt * 4, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, launch_bhtd_to_rows(q, B, nh, T, hd, scale * 1.4426950408889634f, qr, s));
    HIPCHK(ctx, launch_bhtd_to_rows(k, B, kv, T, hd, 1.0f, kr, s));
    HIPCHK(ctx, launch_bhtd_to_rows(v, B, kv, T, hd, 1.0f, vr, s));
    HIPCHK(ctx, launch_kv_write_plain(kr, vr, rows, dpos, dslot, dbt, bps, kvl, s));
    AttnArgs a;
    a.q = qr; a.kv = kvl; a.block_tables = dbt; a.max_blocks = bps; a.tile_row0 = dt0; a.tile_nrows = dtn; a.tile_slot = dts;
    a.pos = dpos; a.nh_l = nh; a.gqa = gqa; a.out_f32 = out;
    HIPCHK(ctx, launch_attn_prefill(a, nt, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    return NVLLM_OK;
}

extern "C" int nvllm_op_embedding(nvllm_ctx* ctx, const float* table, const uint32_t* ids, int n, int V, int H, float* y) {
    if (!ctx || !table || !ids || !y || n < 1) return fail(ctx, NVLLM_EINVAL, "bad embedding arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, launch_embedding_f32(table, ids, n, V, H, y, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return NVLLM_OK;
}

extern "C" int nvllm_op_argmax(nvllm_ctx* ctx, const float* logits, int rows, int V, uint32_t* ids) {
    if (!ctx || !logits || !ids || rows < 1 || V < 1) return fail(ctx, NVLLM_EINVAL, "bad argmax arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, launch_argmax(logits, rows, V, V, ids, nullptr, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return NVLLM_OK;
}

extern "C" int nvllm_op_allreduce(nvllm_ctx* ctx, float* buf, int64_t count) {
    if (!ctx || !buf || count < 0) return fail(ctx, NVLLM_EINVAL, "bad allreduce arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc_ = comm_allreduce_sum(ctx, buf, (size_t)count);
    if (rc_) return rc_;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return NVLLM_OK;
}

extern "C" int nvllm_op_synth_bf16(nvllm_ctx* ctx, const char* name, uint64_t seed, int kind, int64_t first, int64_t count,
                                   uint16_t* host_out) {
    if (!ctx || !name || !host_out || count < 1) return fail(ctx, NVLLM_EINVAL, "bad synth arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    TmpBufs t;
    bf16_bits* d;
    HIPCHK(ctx, t.get(&d, (size_t)count));
    HIPCHK(ctx, launch_synth_rowmajor_bf16(d, synth_plain(synth_hash_name(name, seed), kind), first, count, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(host_out, d, (size_t)count * 2, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return NVLLM_OK;
}

// the same for any generator profile (debug header): axis / cols / hidden_size say where an element's hidden channel sits
extern "C" int nvllm_debug_synth_bf16_spec(nvllm_ctx* ctx, const char* name, uint64_t seed, int kind, int profile, int axis, int64_t cols,
                                           int hidden_size, int64_t first, int64_t count, uint16_t* host_out) {
    if (!ctx || !name || !host_out || count < 1 || cols < 1 || hidden_size < 1) return fail(ctx, NVLLM_EINVAL, "bad synth arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    TmpBufs t;
    bf16_bits* d;
    HIPCHK(ctx, t.get(&d, (size_t)count));
    SynthSpec sp;
    sp.name_hash = synth_hash_name(name, seed); sp.kind = kind; sp.profile = profile; sp.axis = axis; sp.cols = cols;
    synth_set_outliers(sp, seed, hidden_size);
    HIPCHK(ctx, launch_synth_rowmajor_bf16(d, sp, first, count, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(host_out, d, (size_t)count * 2, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return NVLLM_OK;
}

// ---------------------------------------------------------------------------------------------------
// tuning aid: time one GEMM decomposition on synthetic (random-valued) operands
// ---------------------------------------------------------------------------------------------------
extern "C" int nvllm_debug_gemm_bench(nvllm_ctx* ctx, int M, int N, int K, int nt, int nw, int n_split, int iters,
                                      float* us_per_call) {
    if (!ctx || !us_per_call || M < 1 || N % 16 || K % 128 || iters < 1) return fail(ctx, NVLLM_EINVAL, "bad gemm_bench arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    GemmPlan p = plan_gemm(M, N, K, 64);
    if (nt > 0) p.nt = nt;
    if (nw > 0) p.nw = nw;
    if (n_split > 0) set_split(p, K / 32, n_split);
    TmpBufs t;
    PackedW w; w.N = N; w.K = K;
    bf16_bits *xh, *xl; float* out;
    HIPCHK(ctx, t.get(&w.data, (size_t)N * K / 8));
    HIPCHK(ctx, t.get(&xh, (size_t)M * K)); HIPCHK(ctx, t.get(&xl, (size_t)M * K));
    HIPCHK(ctx, t.get(&out, (size_t)p.n_split * M * N));
    HIPCHK(ctx, launch_synth_packed(w, 0, N, synth_plain(12345, kSynthMatrix), 0, 0, K, -1, s));
    HIPCHK(ctx, launch_synth_rowmajor_bf16(xh, synth_plain(777, kSynthMatrix), 0, (int64_t)M * K, s));
    HIPCHK(ctx, launch_synth_rowmajor_bf16(xl, synth_plain(778, kSynthMatrix), 0, (int64_t)M * K, s));
    for (int i = 0; i < 3; ++i) HIPCHK(ctx, launch_gemm(p, xh, xl, K, w, out, M, s));
    HIPCHK(ctx, hipEventRecord(ctx->ev0, s));
    for (int i = 0; i < iters; ++i) HIPCHK(ctx, launch_gemm(p, xh, xl, K, w, out, M, s));
    HIPCHK(ctx, hipEventRecord(ctx->ev1, s));
    HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *us_per_call = ms * 1e3f / iters;
    return NVLLM_OK;
}

// parity + timing of the prefill tile GEMM (tile_gemm.hip) against the chunked kernel on the same synthetic operands.
// mode 0: f32 out; mode 2: SwiGLU planes (compared as hi + lo).  act_packed: the tile kernel writes xpack_off order.
extern "C" int nvllm_debug_gemm_tile_check(nvllm_ctx* ctx, int M, int N, int K, int mode, int act_packed, int iters,
                                           float* max_abs_diff, float* max_abs_ref, float* us_chunked, float* us_tile) {
    if (!ctx || !max_abs_diff || !max_abs_ref || !us_chunked || !us_tile || M < 1 || N % 32 || K % 128 || iters < 1 || (mode != 0 && mode != 2))
        return fail(ctx, NVLLM_EINVAL, "bad gemm_tile_check arguments");
    if (!gemm_tile_ok(M, N, K, mode, 1)) return fail(ctx, NVLLM_EINVAL, "no tile shape for M=%d N=%d K=%d mode=%d", M, N, K, mode);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    TmpBufs t;
    PackedW w; w.N = N; w.K = K;
    const size_t Mp = (size_t)(M + 15) / 16 * 16;
    const int No = mode == 2 ? N / 2 : N;
    bf16_bits *xh, *xl, *ph, *pl, *ah0, *al0, *ah1, *al1; float *o0, *o1;
    HIPCHK(ctx, t.get(&w.data, (size_t)N * K / 8));
    HIPCHK(ctx, t.get(&xh, Mp * K)); HIPCHK(ctx, t.get(&xl, Mp * K)); HIPCHK(ctx, t.get(&ph, Mp * K)); HIPCHK(ctx, t.get(&pl, Mp * K));
    HIPCHK(ctx, t.get(&ah0, Mp * No)); HIPCHK(ctx, t.get(&al0, Mp * No)); HIPCHK(ctx, t.get(&ah1, Mp * No)); HIPCHK(ctx, t.get(&al1, Mp * No));
    const int ks = mode == 0 ? std::max(1, gemm_tile_splits(M, N, K, 1, 4)) : 1;  // narrow outputs: the tile kernel splits K
    HIPCHK(ctx, t.get(&o0, (size_t)M * N)); HIPCHK(ctx, t.get(&o1, (size_t)ks * M * N));
    HIPCHK(ctx, launch_synth_packed(w, 0, N, synth_plain(4321, kSynthMatrix), 0, 0, K, -1, s));
    HIPCHK(ctx, launch_synth_rowmajor_bf16(xh, synth_plain(777, kSynthMatrix), 0, (int64_t)M * K, s));
    HIPCHK(ctx, launch_synth_rowmajor_bf16(xl, synth_plain(778, kSynthMatrix), 0, (int64_t)M * K, s));
    HIPCHK(ctx, launch_xpack_plane(xh, ph, M, K, s));
    HIPCHK(ctx, launch_xpack_plane(xl, pl, M, K, s));
    HIPCHK(ctx, hipMemsetAsync(ah1, 0, Mp * No * 2, s)); HIPCHK(ctx, hipMemsetAsync(al1, 0, Mp * No * 2, s));
    GemmPlan p = mode == 2 ? plan_gemm_swiglu(M, N, K) : plan_gemm(M, N, K, 1);
    auto old_go = [&]() { return mode == 2 ? launch_gemm_swiglu(p, xh, xl, K, w, M, ah0, al0, nullptr, s) : launch_gemm(p, xh, xl, K, w, o0, M, s); };
    int ns = 1;
    auto new_go = [&]() { return launch_gemm_tile(ph, pl, w, M, mode, o1, ah1, al1, act_packed, 1, 4, mode == 0 ? &ns : nullptr, s); };
    float ms = 0;
    for (int i = 0; i < 2; ++i) HIPCHK(ctx, old_go());
    HIPCHK(ctx, hipEventRecord(ctx->ev0, s));
    for (int i = 0; i < iters; ++i) HIPCHK(ctx, old_go());
    HIPCHK(ctx, hipEventRecord(ctx->ev1, s));
    HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *us_chunked = ms * 1e3f / iters;
    for (int i = 0; i < 2; ++i) HIPCHK(ctx, new_go());
    HIPCHK(ctx, hipEventRecord(ctx->ev0, s));
    for (int i = 0; i < iters; ++i) HIPCHK(ctx, new_go());
    HIPCHK(ctx, hipEventRecord(ctx->ev1, s));
    HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *us_tile = ms * 1e3f / iters;
    double md = 0, mr = 0;
    if (mode == 0) {
        std::vector<float> a((size_t)M * N), b((size_t)M * N);
        HIPCHK(ctx, hipMemcpy(a.data(), o0, a.size() * 4, hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(b.data(), o1, b.size() * 4, hipMemcpyDeviceToHost));
        for (int sl = 1; sl < ns; ++sl) {  // K splits: slabs are summed by the consumer
            std::vector<float> c(a.size());
            HIPCHK(ctx, hipMemcpy(c.data(), o1 + (size_t)sl * M * N, c.size() * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < c.size(); ++i) b[i] += c[i];
        }
        for (size_t i = 0; i < a.size(); ++i) {
            const double d = std::fabs((double)a[i] - (double)b[i]);
            md = (d > md || d != d) ? d : md;
            mr = std::max(mr, std::fabs((double)a[i]));
        }
    } else {
        std::vector<uint16_t> h0(Mp * No), l0(Mp * No), h1(Mp * No), l1(Mp * No);
        HIPCHK(ctx, hipMemcpy(h0.data(), ah0, h0.size() * 2, hipMemcpyDeviceToHost)); HIPCHK(ctx, hipMemcpy(l0.data(), al0, l0.size() * 2, hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(h1.data(), ah1, h1.size() * 2, hipMemcpyDeviceToHost)); HIPCHK(ctx, hipMemcpy(l1.data(), al1, l1.size() * 2, hipMemcpyDeviceToHost));
        auto f = [](uint16_t b) { uint32_t u = (uint32_t)b << 16; float v; memcpy(&v, &u, 4); return (double)v; };
        for (int r = 0; r < M; ++r)
            for (int c = 0; c < No; ++c) {
                const size_t i0 = (size_t)r * No + c, i1 = act_packed ? xpack_off(r, c, No >> 5) : i0;
                const double a = f(h0[i0]) + f(l0[i0]), b = f(h1[i1]) + f(l1[i1]);
                const double d = std::fabs(a - b);
                md = (d > md || d != d) ? d : md;
                mr = std::max(mr, std::fabs(a));
            }
    }
    *max_abs_diff = (float)md;
    *max_abs_ref = (float)mr;
    return NVLLM_OK;
}

// tuning aid: time the decode attention kernel alone on a synthetic cache.  ctx_lens[B] tokens per
// sequence; part_tokens 0 = no split.  Returns microseconds per launch (attention + combine).
extern "C" int nvllm_debug_attn_bench(nvllm_ctx* ctx, int B, int nh, int kv, int hd, const int32_t* ctx_lens,
                                      int part_tokens, int iters, float* us_per_call) {
    const int nrot = 8;  // distinct cache copies cycled through: every launch reads cold HBM, like the model's per-layer caches
    if (!ctx || !ctx_lens || !us_per_call || B < 1 || iters < 1 || (hd != 64 && hd != 128) || nh % kv)
        return fail(ctx, NVLLM_EINVAL, "bad attn_bench arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    TmpBufs t;
    int max_len = 0, nblk = 0;
    for (int i = 0; i < B; ++i) max_len = std::max(max_len, (int)ctx_lens[i]);
    const int bps = (max_len + kBlockTokens - 1) / kBlockTokens;
    std::vector<int> hbt((size_t)B * bps, 0), hpos(B), hslot(B), ht0(B), htn(B, 1), hts(B);
    for (int i = 0; i < B; ++i) {
        for (int j = 0; j < (ctx_lens[i] + kBlockTokens - 1) / kBlockTokens; ++j) hbt[(size_t)i * bps + j] = nblk++;
        hpos[i] = ctx_lens[i] - 1; hslot[i] = i; ht0[i] = i; hts[i] = i;
    }
    KvLayout kvl; kvl.kv_l = kv; kvl.hd = hd;
    const size_t cache_elems = (size_t)nblk * kv * kBlockTokens * hd;
    float *q, *po, *pml; bf16_bits *oh, *ol;
    int *dbt, *dpos, *dslot, *dt0, *dtn, *dts;
    std::vector<f16_bits*> rk(nrot), rv(nrot);
    for (int r = 0; r < nrot; ++r) {
        HIPCHK(ctx, t.get(&rk[r], cache_elems)); HIPCHK(ctx, t.get(&rv[r], cache_elems));
        HIPCHK(ctx, launch_synth_rowmajor_bf16(rk[r], synth_plain(1 + r, kSynthMatrix), 0, (int64_t)cache_elems, s));
        HIPCHK(ctx, launch_synth_rowmajor_bf16(rv[r], synth_plain(100 + r, kSynthMatrix), 0, (int64_t)cache_elems, s));
    }
    kvl.k = rk[0]; kvl.v = rv[0];
    HIPCHK(ctx, t.get(&q, (size_t)B * nh * hd)); HIPCHK(ctx, t.get(&oh, (size_t)B * nh * hd)); HIPCHK(ctx, t.get(&ol, (size_t)B * nh * hd));
    HIPCHK(ctx, t.get(&po, (size_t)B * nh * kAttnMaxParts * hd)); HIPCHK(ctx, t.get(&pml, (size_t)B * nh * kAttnMaxParts * 2));
    HIPCHK(ctx, t.get(&dbt, hbt.size())); HIPCHK(ctx, t.get(&dpos, B)); HIPCHK(ctx, t.get(&dslot, B));
    HIPCHK(ctx, t.get(&dt0, B)); HIPCHK(ctx, t.get(&dtn, B)); HIPCHK(ctx, t.get(&dts, B));
    // bf16-valued synthetic bits reinterpreted as f16 are small finite numbers: fine for timing
    HIPCHK(ctx, launch_synth_rowmajor_f32(q, synth_plain(3, kSynthMatrix), 0, (int64_t)B * nh * hd, s));
    HIPCHK(ctx, hipMemcpyAsync(dbt, hbt.data(), hbt.size() * 4, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(dpos, hpos.data(), (size_t)B * 4, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(dslot, hslot.data(), (size_t)B * 4, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(dt0, ht0.data(), (size_t)B * 4, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(dtn, htn.data(), (size_t)B * 4, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(dts, hts.data(), (size_t)B * 4, hipMemcpyHostToDevice, s));
    AttnArgs a;
    a.q = q; a.kv = kvl; a.block_tables = dbt; a.max_blocks = bps; a.tile_row0 = dt0; a.tile_nrows = dtn; a.tile_slot = dts;
    a.pos = dpos; a.nh_l = nh; a.gqa = nh / kv; a.out_hi = oh; a.out_lo = ol;
    int parts_max = 1;
    if (part_tokens > 0) {
        a.part_tiles = std::max(1, part_tokens / 32); a.max_parts = kAttnMaxParts; a.part_o = po; a.part_ml = pml;
        parts_max = ((max_len + 31) / 32 + a.part_tiles - 1) / a.part_tiles;
        if (parts_max > kAttnMaxParts) return fail(ctx, NVLLM_EINVAL, "too many parts");
    }
    for (int i = 0; i < 3; ++i) HIPCHK(ctx, launch_attn_paged(a, B, 1, B, parts_max, s));
    HIPCHK(ctx, hipEventRecord(ctx->ev0, s));
    for (int i = 0; i < iters; ++i) {
        a.kv.k = rk[i % nrot]; a.kv.v = rv[i % nrot];
        HIPCHK(ctx, launch_attn_paged(a, B, 1, B, parts_max, s));
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev1, s));
    HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *us_per_call = ms * 1e3f / iters;
    return NVLLM_OK;
}

// which XCD each workgroup of a (gx,gy,gz) grid of `threads`-wide workgroups is dispatched to (out: gx*gy*gz ints)
extern "C" int nvllm_debug_xcc_map(nvllm_ctx* ctx, int gx, int gy, int gz, int threads, int32_t* out) {
    if (!ctx || !out || gx < 1 || gy < 1 || gz < 1 || threads < 64 || threads > 1024) return fail(ctx, NVLLM_EINVAL, "bad xcc_map arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    TmpBufs t;
    int* d; const size_t n = (size_t)gx * gy * gz;
    HIPCHK(ctx, t.get(&d, n));
    HIPCHK(ctx, launch_xcc_map(gx, gy, gz, threads, d, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(out, d, n * 4, hipMemcpyDeviceToHost));
    return NVLLM_OK;
}

// tuning aid, v2: mode 0 = slabs, 2 = SwiGLU epilogue (N = 2I); mt = m-tiles per workgroup (0 = planner);
// `rot` weight copies are cycled so every launch streams cold HBM like the model does (1 = cache-warm).
extern "C" int nvllm_debug_gemm_bench2(nvllm_ctx* ctx, int M, int N, int K, int mt, int nt, int nw, int n_split, int mode,
                                       int rot, int iters, float* us_per_call) {
    // diagnostic build: mode 22 / 23 + 100 * a = ablation variant a of the streaming kernel (stream_gemm.hip ABL)
    const int ablate = mode / 100;
    mode %= 100;
#ifdef NVLLM_STAMPS
    nvllm::stream_gemm_set_ablate(ablate);
#else
    if (ablate) return fail(ctx, NVLLM_ESTATE, "ablation variants exist in the diagnostic build only (make stamps)");
#endif
    if (!ctx || !us_per_call || M < 1 || (M > 128 && mode >= 10) || N % 32 || K % 128 || iters < 1 || rot < 1) return fail(ctx, NVLLM_EINVAL, "bad gemm_bench2 arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    GemmPlan p = mode == 2 ? plan_gemm_swiglu(M, N, K) : plan_gemm(M, N, K, 64);
    if (mt > 0) { p.mt = mt; p.kc = mt == 8 ? 2 : 4; }
    if (nt > 0) p.nt = nt;
    if (nw > 0) p.nw = nw;
    set_split(p, K / 32, n_split > 0 ? n_split : p.n_split);
    TmpBufs t;
    std::vector<PackedW> ws(rot);
    for (int r = 0; r < rot; ++r) {
        ws[r].N = N; ws[r].K = K;
        HIPCHK(ctx, t.get(&ws[r].data, (size_t)N * K / 8));
        HIPCHK(ctx, launch_synth_packed(ws[r], 0, N, synth_plain(12345 + r, kSynthMatrix), 0, 0, K, -1, s));
    }
    bf16_bits *xh, *xl, *ah, *al; float* out;
    HIPCHK(ctx, t.get(&xh, (size_t)M * K)); HIPCHK(ctx, t.get(&xl, (size_t)M * K));
    HIPCHK(ctx, t.get(&ah, (size_t)M * N)); HIPCHK(ctx, t.get(&al, (size_t)M * N));
    HIPCHK(ctx, t.get(&out, (size_t)p.n_split * M * N));
    HIPCHK(ctx, launch_synth_rowmajor_bf16(xh, synth_plain(777, kSynthMatrix), 0, (int64_t)M * K, s));
    HIPCHK(ctx, launch_synth_rowmajor_bf16(xl, synth_plain(778, kSynthMatrix), 0, (int64_t)M * K, s));
    // modes 10/11/12: the row-parallel decode kernel with epilogue 0/1/2 (planner shapes; mt/nt/nw/n_split ignored)
    float *resid = nullptr, *nextw = nullptr, *ssq = nullptr, *ssq_in = nullptr;
    if (mode == 22 || mode == 23) {
        if (!gemm_stream_splits(M, N, K)) return fail(ctx, NVLLM_EINVAL, "no streaming shape for M=%d N=%d K=%d", M, N, K);
        float* o2; HIPCHK(ctx, t.get(&o2, (size_t)gemm_stream_splits(M, N, K) * M * N)); out = o2;
    }
    if (mode >= 10 && mode < 20) {
        if (!gemm_rowpar_ok(N, K, mode - 10, M)) return fail(ctx, NVLLM_EINVAL, "no row-parallel shape for N=%d K=%d epi=%d", N, K, mode - 10);
        HIPCHK(ctx, t.get(&resid, (size_t)M * N)); HIPCHK(ctx, t.get(&nextw, (size_t)N));
        HIPCHK(ctx, t.get(&ssq, (size_t)1024 * 128)); HIPCHK(ctx, t.get(&ssq_in, (size_t)128));
        HIPCHK(ctx, hipMemsetAsync(resid, 0, (size_t)M * N * 4, s)); HIPCHK(ctx, hipMemsetAsync(nextw, 0, (size_t)N * 4, s));
        HIPCHK(ctx, hipMemsetAsync(ssq_in, 0, 128 * 4, s));
        float* o2; HIPCHK(ctx, t.get(&o2, (size_t)std::max(1, gemm_rowpar_splits(N, K, 2, M)) * M * N)); out = o2;
    }
    float* pv = nullptr; int* pi = nullptr;
    GemmPlan plm = plan_lmhead(M, N, K);
    if (mode == 20 || mode == 21) {  // LM head + arg-max partials: 20 = streaming kernel when applicable, 21 = chunked kernel
        if (mode == 21) plm = plan_gemm(M, N, K, 1);
        HIPCHK(ctx, t.get(&pv, (size_t)(N / 16 + 16) * M)); HIPCHK(ctx, t.get(&pi, (size_t)(N / 16 + 16) * M));
    }
    auto go = [&](int i) -> hipError_t {
        const PackedW& w = ws[i % rot];
        if (mode == 20 || mode == 21) return launch_gemm_argmax(plm, xh, xl, K, w, nullptr, M, pv, pi, nullptr, s);
        if (mode == 22 || mode == 23) return launch_gemm_stream(xh, xl, K, w, out, M, mode == 23, s);  // 23: packed x planes
        if (mode >= 10) {
            RowParArgs ra;
            ra.xh = xh; ra.xl = xl; ra.ldx = K; ra.M = M; ra.out = out; ra.resid_in = resid; ra.resid_out = resid; ra.next_w = nextw;
            ra.oh = ah; ra.ol = al; ra.ssq = ssq; ra.ssq_stride = 128;
            ra.rn.ssq = ssq_in; ra.rn.groups = 1; ra.rn.stride = 128; ra.rn.inv_h = 1.0f / K; ra.rn.eps = 1e-6f;
            return launch_gemm_rowpar(ra, w, mode - 10, s);
        }
        return mode == 2 ? launch_gemm_swiglu(p, xh, xl, K, w, M, ah, al, nullptr, s) : launch_gemm(p, xh, xl, K, w, out, M, s);
    };
    for (int i = 0; i < 3; ++i) HIPCHK(ctx, go(i));
    HIPCHK(ctx, hipEventRecord(ctx->ev0, s));
    for (int i = 0; i < iters; ++i) HIPCHK(ctx, go(i));
    HIPCHK(ctx, hipEventRecord(ctx->ev1, s));
    HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *us_per_call = ms * 1e3f / iters;
    return NVLLM_OK;
}

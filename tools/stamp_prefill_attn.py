"""In-kernel timeline of the prefill attention kernel (attn_prefill_kernel) on the last 4096-row chunk of the bench prompts,
from the stamped diagnostic build:  make -C nano-vllm-candle_amd/csrc stamps && python tools/stamp_prefill_attn.py
Stamps (s_memrealtime, 100 MHz): 0 entry, 1 prologue done (metadata, first two K/V tiles issued, q loaded), 2 K/V loop done,
3 end.  Slots 5 / 6 carry the workgroup's and the wave's iteration counts.  Read the SHARES: the stamps cost time."""
import ctypes as C
import os
import sys

import numpy as np

os.environ.setdefault("NVLLM_LIB", "libnvllm_amd_stamps.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nano_vllm_candle_amd as pkg  # noqa: E402

L = pkg._lib.lib()
ctx = pkg.Context(0)
cfg = pkg.Qwen3Config.qwen3_0_6b()
m = pkg.Qwen3ForCausalLM.from_synthetic(cfg, 0, ctx)
rng = np.random.default_rng(0)
lens = rng.integers(64, 513, size=64)
prompts = [rng.integers(0, cfg.vocab_size, size=int(n), dtype=np.uint32).tolist() for n in lens]
m.kv_alloc(num_blocks=64 * 3 + 2, max_seqs=64, max_batched_tokens=4096)
m.step(list(range(64)), prompts, True)  # warm
m.kv_alloc(num_blocks=64 * 3 + 2, max_seqs=64, max_batched_tokens=4096)
pkg._lib.check(L.nvllm_debug_stamps(m.h, 1), ctx.h)
m.step(list(range(14)), prompts[:14], True)  # one chunk (14 prompts ~ 4000 rows)
pkg._lib.check(L.nvllm_debug_stamps(m.h, 0), ctx.h)
print("rows in the chunk:", int(sum(lens[:14])))
N = 1024 * 16 * 8
for layer in (5, 6):
    buf = np.zeros(N, np.uint64)
    pkg._lib.check(L.nvllm_debug_stamps_read(m.h, layer, buf.ctypes.data_as(C.POINTER(C.c_uint64)), N), ctx.h)
    st = buf.reshape(1024, 16, 8).astype(np.int64)[:, :4]
    used = st[:, :, 0] > 0
    wg = used.any(axis=1)
    t0 = np.where(used, st[:, :, 0], np.iinfo(np.int64).max).min(axis=1)[wg]
    t1 = np.where(used, st[:, :, 1], 0).max(axis=1)[wg]
    t2 = np.where(used, st[:, :, 2], 0).max(axis=1)[wg]
    t3 = np.where(used, st[:, :, 3], 0).max(axis=1)[wg]
    iters = st[:, 0, 5][wg]
    k0 = t0.min()
    print(f"layer {layer}: {int(wg.sum())} workgroups, kernel {(t3.max() - k0) / 100:.1f} us; start skew max {(t0.max() - k0) / 100:.1f} us "
          f"(workgroups starting > 2 us late: {int(((t0 - k0) > 200).sum())})")
    print(f"   prologue median {np.median(t1 - t0) / 100:.2f} max {(t1 - t0).max() / 100:.2f} us; epilogue median {np.median(t3 - t2) / 100:.2f} max {(t3 - t2).max() / 100:.2f} us")
    loop = (t2 - t1) / 100.0
    for lo, hi in ((1, 2), (3, 4), (5, 8), (9, 12), (13, 16), (17, 32)):
        sel = (iters >= lo) & (iters <= hi)
        if sel.any():
            print(f"   {lo:2d}-{hi:2d} iterations: {int(sel.sum()):4d} workgroups, loop median {np.median(loop[sel]):6.2f} us = {np.median(loop[sel] / iters[sel]):5.2f} us per iteration, "
                  f"ends at median {np.median((t3[sel] - k0)) / 100:5.1f} max {(t3[sel] - k0).max() / 100:5.1f} us")

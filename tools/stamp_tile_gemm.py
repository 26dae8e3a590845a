"""In-kernel timeline of the prefill tile GEMM launches of one 4096-row chunk (Qwen3-0.6B), from the stamped diagnostic
build:  make -C nano-vllm-candle_amd/csrc stamps && python tools/stamp_tile_gemm.py
Stamps (s_memrealtime, 100 MHz): 0 entry, 1 first two stages landed (pipeline filled), 2 K loop done, 3 end (epilogue done).
Read the SHARES: the stamps cost time themselves."""
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
n_seq = 15
m.kv_alloc(num_blocks=64 * 3 + 2, max_seqs=64, max_batched_tokens=4096)
m.step(list(range(n_seq)), prompts[:n_seq], True)  # warm
m.kv_alloc(num_blocks=64 * 3 + 2, max_seqs=64, max_batched_tokens=4096)
pkg._lib.check(L.nvllm_debug_stamps(m.h, 2), ctx.h)  # 2: the tile GEMM launches record
m.step(list(range(n_seq)), prompts[:n_seq], True)
pkg._lib.check(L.nvllm_debug_stamps(m.h, 0), ctx.h)
print("rows in the chunk:", int(sum(lens[:n_seq])))
N = 1024 * 16 * 8
names = ["qkv (+ q/k-norm, RoPE, KV write)", "o_proj", "gate_up (+ SiLU*mul)", "down_proj"]
for layer in (5,):
    for k, name in enumerate(names):
        buf = np.zeros(N, np.uint64)
        pkg._lib.check(L.nvllm_debug_stamps_read(m.h, layer * 4 + k, buf.ctypes.data_as(C.POINTER(C.c_uint64)), N), ctx.h)
        st = buf.reshape(1024, 16, 8).astype(np.int64)[:, :8]
        used = st[:, :, 0] > 0
        wg = used.any(axis=1)
        big = np.iinfo(np.int64).max
        t0 = np.where(used, st[:, :, 0], big).min(axis=1)[wg]
        t1 = np.where(used, st[:, :, 1], 0).max(axis=1)[wg]
        t2 = np.where(used, st[:, :, 2], 0).max(axis=1)[wg]
        t3 = np.where(used, st[:, :, 3], 0).max(axis=1)[wg]
        k0 = t0.min()
        late = (t0 - k0) > 200
        print(f"layer {layer} {name}: {int(wg.sum())} workgroups ({int(late.sum())} start > 2 us late), kernel {(t3.max() - k0) / 100:.1f} us")
        for tag, sel in (("first wave", ~late), ("late starters", late)):
            if sel.any():
                print(f"   {tag:13s}: fill {np.median((t1 - t0)[sel]) / 100:5.2f}  K loop {np.median((t2 - t1)[sel]) / 100:6.2f}  epilogue {np.median((t3 - t2)[sel]) / 100:5.2f} "
                      f"(max {((t3 - t2)[sel]).max() / 100:5.2f}) us; start at median {np.median((t0 - k0)[sel]) / 100:5.1f}, end at median {np.median((t3 - k0)[sel]) / 100:5.1f} max {((t3 - k0)[sel]).max() / 100:5.1f} us")

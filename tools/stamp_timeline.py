"""In-kernel timeline of one decode step of the bench workload (Qwen3-0.6B, 64 sequences, prompts U[64,512] seed 0) from
the stamped diagnostic build:  make -C nano-vllm-candle_amd/csrc stamps && NVLLM_LIB=libnvllm_amd_stamps.so python
tools/stamp_timeline.py.   Stamps are s_memrealtime (100 MHz): 0 entry, 1 loads issued (GEMM) / prologue done (attention),
2 MFMAs done (GEMM) / barrier passed (attention), 3 LDS reduce barrier passed (GEMM) / KV loop done (attention), 4 end.
Read the SHARES, not the length: the stamps cost time themselves."""
import ctypes as C
import os
import sys

import numpy as np

os.environ.setdefault("NVLLM_LIB", "libnvllm_amd_stamps.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nano_vllm_candle_amd as pkg  # noqa: E402

import argparse  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--prompt-min", type=int, default=64)
ap.add_argument("--prompt-max", type=int, default=512)
ap.add_argument("--option", action="append", default=[])
args = ap.parse_args()
B = args.batch
L = pkg._lib.lib()
ctx = pkg.Context(0)
cfg = pkg.Qwen3Config.qwen3_0_6b()
m = pkg.Qwen3ForCausalLM.from_synthetic(cfg, 0, ctx)
for o in args.option:
    name, _, val = o.partition("=")
    m.set_option(name, int(val or 1))
rng = np.random.default_rng(0)
lens = rng.integers(args.prompt_min, args.prompt_max + 1, size=B)
prompts = [rng.integers(0, cfg.vocab_size, size=int(n), dtype=np.uint32).tolist() for n in lens]
m.kv_alloc(num_blocks=B * 3 + 2, max_seqs=B, max_batched_tokens=4096)
m.step(list(range(B)), prompts, True)
for _ in range(40):
    m.decode_next()
pkg._lib.check(L.nvllm_debug_stamps(m.h, 1), ctx.h)
m.decode_next()
pkg._lib.check(L.nvllm_debug_stamps(m.h, 0), ctx.h)
ctx_lens = lens + 42

N = 1024 * 16 * 8
names = ["qkv", "attn", "o_proj", "gate_up", "down"]
prev_end = None
for layer in (10, 11, 12):
    for k, name in enumerate(names):
        buf = np.zeros(N, np.uint64)
        pkg._lib.check(L.nvllm_debug_stamps_read(m.h, layer * 5 + k, buf.ctypes.data_as(C.POINTER(C.c_uint64)), N), ctx.h)
        st = buf.reshape(1024, 16, 8).astype(np.int64)
        used = st[:, :, 0] > 0                         # (workgroup, wave) slots that ran
        wg = used.any(axis=1)
        t0 = np.where(used, st[:, :, 0], np.iinfo(np.int64).max).min(axis=1)[wg]   # per workgroup: first entry
        t4 = np.where(used, st[:, :, 4], 0).max(axis=1)[wg]
        k0, k4 = t0.min(), t4.max()
        seg = []
        for a_, b_ in ((0, 1), (1, 2), (2, 3), (3, 4)):
            d = (st[:, :, b_] - st[:, :, a_])[used & (st[:, :, b_] > 0) & (st[:, :, a_] > 0)]
            seg.append((np.median(d) / 100.0, d.max() / 100.0) if d.size else (0.0, 0.0))
        gap = (k0 - prev_end) / 100.0 if prev_end is not None else float("nan")
        print(f"layer {layer} {name:8s}: {int(wg.sum()):4d} WGs, kernel {(k4 - k0) / 100.0:6.2f} us (gap before {gap:5.2f} us), start skew {(t0.max() - k0) / 100.0:5.2f}, "
              f"per-WG lifetime median {np.median(t4 - t0) / 100.0:5.2f} max {(t4 - t0).max() / 100.0:5.2f}; "
              "segments median/max us: " + "  ".join(f"{i}->{i + 1} {a_:.2f}/{b_:.2f}" for i, (a_, b_) in enumerate(seg)))
        if name == "attn" and B == 64:
            # workgroup (x = rank in the longest-first order, y = kv head): lifetime vs context length
            order = np.argsort(-ctx_lens, kind="stable")
            life = np.zeros(64)
            for r in range(64):
                ws = [r + 64 * h for h in range(8)]
                x = st[ws][:, :4]
                life[r] = (x[:, :, 4].max() - x[:, :, 0][x[:, :, 0] > 0].min()) / 100.0
            q = [0, 8, 16, 32, 48, 63]
            if layer == 11:
                # per rank: when (relative to the kernel's first entry) each stamp is reached, median over heads and waves
                for r in q:
                    x = st[[r + 64 * h for h in range(8)]][:, :4, :5].astype(np.float64)
                    x[x == 0] = np.nan
                    rel = (x - k0) / 100.0
                    print(f"      rank #{r} ctx {ctx_lens[order[r]]}: stamps 0..4 at (median over heads/waves) "
                          + "  ".join(f"{np.nanmedian(rel[:, :, i]):.1f}" for i in range(5))
                          + "  | per wave end of loop (stamp 3), head 0: " + " ".join(f"{v:.1f}" for v in rel[0, :, 3]))
            print("      attention lifetime by rank (ctx): " + "  ".join(f"#{r} ({ctx_lens[order[r]]}) {life[r]:.1f}" for r in q),
                  f"| end of last WG by rank: " + "  ".join(f"#{r} {(st[[r + 64 * h for h in range(8)]][:, :4, 4].max() - k0) / 100.0:.1f}" for r in q))
        prev_end = k4

"""Cold-weight GEMM sweep for a model's decode shapes: python tools/tune_gemm3.py [M] [8b|32b|0.6b]"""
import ctypes as C
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nano_vllm_candle_amd as pkg

M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
which = sys.argv[2] if len(sys.argv) > 2 else "8b"
cfg = {"8b": pkg.Qwen3Config.qwen3_8b(), "32b": pkg.Qwen3Config.qwen3_32b(), "0.6b": pkg.Qwen3Config.qwen3_0_6b()}[which]
H, I, nq = cfg.hidden_size, cfg.intermediate_size, (cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * cfg.head_dim
ctx = pkg.Context(0)
L = pkg._lib.lib()


def run(N, K, mt, nt, nw, ns, mode, rot):
    us = C.c_float()
    rc = L.nvllm_debug_gemm_bench2(ctx.h, M, N, K, mt, nt, nw, ns, mode, rot, 30, C.byref(us))
    return us.value if rc == 0 else None


shapes = {"qkv": (nq, H, 0), "o": (H, cfg.num_attention_heads * cfg.head_dim, 0), "down": (H, I, 0), "gate_up swiglu": (2 * I, H, 2)}
for name, (N, K, mode) in shapes.items():
    rot = max(2, int(400e6 / (N * K * 2)) + 1)
    res = []
    grid = itertools.product([0], [2] if mode == 2 else [1, 2], [8], [1] if mode == 2 else [1, 2, 4, 8, 16])
    for mt, nt, nw, ns in grid:
        if (K // 32) // ns < 8:
            continue
        us = run(N, K, mt, nt, nw, ns, mode, rot)
        if us:
            res.append((us, nt, nw, ns))
    res.sort()
    print(f"{name} M={M} N={N} K={K} weight={N*K*2/1e6:.1f}MB rot={rot}", flush=True)
    for us, nt, nw, ns in res[:5]:
        print(f"   nt={nt} nw={nw} split={ns}: {us:8.2f} us  {N*K*2/us/1e3:7.1f} GB/s", flush=True)

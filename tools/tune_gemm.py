"""Sweep decompositions of the generic decode GEMM (m-tiles per workgroup, n-tiles per wave, waves per workgroup, K splits)
on cold weights -- the matrix is rotated through more copies than the Infinity Cache holds, like the model's per-layer
weights -- for one model's decode shapes, SwiGLU mode included (nvllm_debug_gemm_bench2).
    python tools/tune_gemm.py [M] [0.6b|8b|32b]    -> microseconds per launch and achieved weight GB/s, best six per shape"""
import ctypes as C
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nano_vllm_candle_amd as pkg  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
which = sys.argv[2] if len(sys.argv) > 2 else "0.6b"
cfg = {"8b": pkg.Qwen3Config.qwen3_8b(), "32b": pkg.Qwen3Config.qwen3_32b(), "0.6b": pkg.Qwen3Config.qwen3_0_6b()}[which]
H, I = cfg.hidden_size, cfg.intermediate_size
nq = (cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * cfg.head_dim
ctx = pkg.Context(0)
L = pkg._lib.lib()


def run(N, K, mt, nt, nw, ns, mode, rot):
    us = C.c_float()
    rc = L.nvllm_debug_gemm_bench2(ctx.h, M, N, K, mt, nt, nw, ns, mode, rot, 40, C.byref(us))
    return us.value if rc == 0 else None


shapes = {"qkv": (nq, H, 0), "o": (H, cfg.num_attention_heads * cfg.head_dim, 0), "down": (H, I, 0),
          "gate_up slabs": (2 * I, H, 0), "gate_up swiglu": (2 * I, H, 2)}
for name, (N, K, mode) in shapes.items():
    rot = max(2, int(400e6 / (N * K * 2)) + 1)
    res = []
    grid = itertools.product([0, 4], [2] if mode == 2 else [1, 2], [2, 4, 8], [1] if mode == 2 else [1, 2, 4, 8, 16])
    for mt, nt, nw, ns in grid:
        if (K // 32) // ns < 4:
            continue
        us = run(N, K, mt, nt, nw, ns, mode, rot)
        if us is not None:
            res.append((us, mt, nt, nw, ns))
    res.sort()
    print(f"{name} M={M} N={N} K={K} weight={N * K * 2 / 1e6:.1f} MB, {rot} cold copies", flush=True)
    for us, mt, nt, nw, ns in res[:6]:
        print(f"   mt={mt} nt={nt} nw={nw} split={ns}: {us:7.2f} us  {N * K * 2 / us / 1e3:7.1f} GB/s", flush=True)

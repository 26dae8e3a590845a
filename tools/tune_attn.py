"""Time the decode attention kernel alone for several context-length distributions / split sizes.
The bench cycles through 8 cache copies (cold HBM every launch, like the model's per-layer caches).  PARTS=0,256 (this
script's own variable): split sizes in tokens to try."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nano_vllm_candle_amd as pkg

ctx = pkg.Context(0)
L = pkg._lib.lib()


def run(lens, part, nh=16, kv=8, hd=128, iters=40):
    lens = np.ascontiguousarray(lens, np.int32)
    us = C.c_float()
    rc = L.nvllm_debug_attn_bench(ctx.h, len(lens), nh, kv, hd, lens.ctypes.data_as(C.POINTER(C.c_int32)), part, iters, C.byref(us))
    if rc:
        return None
    byts = float(lens.sum()) * kv * hd * 2 * 2
    return us.value, byts / us.value / 1e3


rng = np.random.default_rng(0)
bench_lens = rng.integers(64, 513, 64) + 20
cases = {"bench U[64,512]+20": bench_lens, "uniform 333": np.full(64, 333), "uniform 512": np.full(64, 512),
         "B=1 ctx 4096": np.full(1, 4096), "B=8 ctx 1024": np.full(8, 1024)}
parts = [int(p) for p in os.environ.get("PARTS", "0,256").split(",")]
for name, lens in cases.items():
    for part in parts:
        r = run(lens, part)
        if r:
            print(f"{name:22s} part={part:4d}: {r[0]:8.2f} us  {r[1]:7.1f} GB/s", flush=True)

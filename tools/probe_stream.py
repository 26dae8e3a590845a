import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import nano_vllm_candle_amd as pkg
ctx = pkg.Context(0); L = pkg._lib.lib()
for name, (N, K) in {"8B qkv": (6144, 4096), "8B o": (4096, 4096), "8B gate_up": (24576, 4096), "8B down": (4096, 12288),
                     "32B qkv": (10240, 5120), "32B o": (5120, 8192), "32B gate_up": (51200, 5120), "32B down": (5120, 25600),
                     "32B/8 gate_up": (6400, 5120), "32B/8 down": (5120, 3200), "32B/2 qkv": (5120, 5120), "32B/2 down": (5120, 12800)}.items():
    mb = N * K * 2 / 1e6
    us = C.c_float()
    rc = L.nvllm_debug_gemm_bench2(ctx.h, 64, N, K, 0, 0, 0, 0, 23, max(2, int(700 / mb)), 20, C.byref(us))
    print(f"{name:14s} N={N} K={K} {mb:6.1f} MB: rc={rc} {us.value:7.2f} us {mb / max(us.value, 1e-3):6.2f} TB/s", flush=True)

"""CPU, world_size 2 over gloo: the tensor-parallel decomposition the HIP path uses -- shard regions from the
C ABI (nvllm_tp_shard), column-parallel q/k/v/gate/up, row-parallel o/down, ONE all-reduce(sum) after o_proj and
one after down_proj -- reproduces the unsharded layer.  The per-rank arithmetic is done with the oracle ops (this
is a test of the decomposition and of where the collectives sit, not of the kernels; the GPU data path uses the
same regions and RCCL in place of gloo)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import nano_vllm_candle_amd as pkg
    from nano_vllm_candle_amd.tp import shard_region
    from oracle import oracle as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = pkg.Qwen3Config.tiny(num_attention_heads=4, num_key_value_heads=2, head_dim=64)
        H, hd, I = cfg.hidden_size, cfg.head_dim, cfg.intermediate_size
        nh, kv = cfg.num_attention_heads, cfg.num_key_value_heads
        from tests.util import oracle_config

        om = O.Model(oracle_config(O, cfg)).fill_synthetic(0)
        shapes = cfg.hf_tensor_shapes()

        def full(name):
            return om.get_tensor(name, shapes[name])

        def shard(name):
            r0, c0, rows, cols = shard_region(cfg, world, rank, name)
            w = full(name)
            w2 = w if w.ndim == 2 else w.reshape(1, -1)
            return np.ascontiguousarray(w2[r0:r0 + rows, c0:c0 + cols])

        rng = np.random.default_rng(5)
        B, T = 2, 7
        x = rng.standard_normal((B * T, H)).astype(np.float32)
        p = "model.layers.0."
        # ---- attention block, this rank's heads only
        nh_l, kv_l = nh // world, kv // world
        q = O.linear(x, shard(p + "self_attn.q_proj.weight")).reshape(B, T, nh_l, hd).transpose(0, 2, 1, 3)
        k = O.linear(x, shard(p + "self_attn.k_proj.weight")).reshape(B, T, kv_l, hd).transpose(0, 2, 1, 3)
        v = O.linear(x, shard(p + "self_attn.v_proj.weight")).reshape(B, T, kv_l, hd).transpose(0, 2, 1, 3)
        qn, _ = O.rmsnorm(q.reshape(-1, hd), full(p + "self_attn.q_norm.weight"), cfg.rms_norm_eps)
        kn, _ = O.rmsnorm(k.reshape(-1, hd), full(p + "self_attn.k_norm.weight"), cfg.rms_norm_eps)
        q = O.rope_apply(qn.reshape(B, nh_l, T, hd), cfg.rope_theta)
        k = O.rope_apply(kn.reshape(B, kv_l, T, hd), cfg.rope_theta)
        ctx = O.attention(q, k, np.ascontiguousarray(v))
        part = torch.from_numpy(O.linear(ctx, shard(p + "self_attn.o_proj.weight")))
        dist.all_reduce(part)  # the all-reduce RowParallelLinear::forward lacks (linear.rs:184-198)
        attn_out = part.numpy()
        # ---- MLP block, this rank's intermediate columns only
        g = O.linear(x, shard(p + "mlp.gate_proj.weight"))
        u = O.linear(x, shard(p + "mlp.up_proj.weight"))
        act = O.silu_mul(np.concatenate([g, u], 1))
        part = torch.from_numpy(O.linear(act, shard(p + "mlp.down_proj.weight")))
        dist.all_reduce(part)
        mlp_out = part.numpy()
        # ---- vocab-parallel LM head: local (max, idx) -> gather -> best, ties to the higher global index
        lg = O.linear(x[:3], shard("lm_head.weight"))
        Vl = lg.shape[1]
        loc = torch.tensor([[float(r.max()), float(np.flatnonzero(r == r.max())[-1] + rank * Vl)] for r in lg])
        allp = [torch.zeros_like(loc) for _ in range(world)]
        dist.all_gather(allp, loc)
        ids = []
        for i in range(3):
            best = max(((float(a[i, 0]), int(a[i, 1])) for a in allp))
            ids.append(best[1])
        if rank == 0:
            # unsharded reference on the same inputs
            qf = O.linear(x, full(p + "self_attn.q_proj.weight")).reshape(B, T, nh, hd).transpose(0, 2, 1, 3)
            kf = O.linear(x, full(p + "self_attn.k_proj.weight")).reshape(B, T, kv, hd).transpose(0, 2, 1, 3)
            vf = O.linear(x, full(p + "self_attn.v_proj.weight")).reshape(B, T, kv, hd).transpose(0, 2, 1, 3)
            qn, _ = O.rmsnorm(qf.reshape(-1, hd), full(p + "self_attn.q_norm.weight"), cfg.rms_norm_eps)
            kn, _ = O.rmsnorm(kf.reshape(-1, hd), full(p + "self_attn.k_norm.weight"), cfg.rms_norm_eps)
            cf = O.attention(O.rope_apply(qn.reshape(B, nh, T, hd), cfg.rope_theta),
                             O.rope_apply(kn.reshape(B, kv, T, hd), cfg.rope_theta), np.ascontiguousarray(vf))
            ref_attn = O.linear(cf, full(p + "self_attn.o_proj.weight"))
            gu = O.linear(x, np.concatenate([full(p + "mlp.gate_proj.weight"), full(p + "mlp.up_proj.weight")], 0))
            ref_mlp = O.linear(O.silu_mul(gu), full(p + "mlp.down_proj.weight"))
            ref_ids = [O.argmax_last(r) for r in O.linear(x[:3], full("lm_head.weight"))]
            out_q.put((float(np.abs(attn_out - ref_attn).max() / np.abs(ref_attn).max()),
                   float(np.abs(mlp_out - ref_mlp).max() / np.abs(ref_mlp).max()), ids, ref_ids))
    finally:
        dist.destroy_process_group()


def test_tp2_decomposition_matches_unsharded_layer():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    e_attn, e_mlp, ids, ref_ids = q.get(timeout=5)
    # summation order differs across ranks, so tolerance not bit-equality (SURVEY §7 hard parts)
    assert e_attn < 1e-5 and e_mlp < 1e-5, (e_attn, e_mlp)
    assert ids == ref_ids


def test_bench_launches_its_own_ranks_from_a_plain_command():
    # `python bench.py --gpus 2` with no torch.distributed.run around it: the launcher starts one rank process per GPU
    # (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in their env) and relays rank 0's single JSON line; --launch-check stops
    # after the gloo rendezvous, so this runs without a GPU
    import json
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                        capture_output=True, text=True, timeout=240)
    assert cp.returncode == 0, cp.stderr[-2000:]
    lines = [l for l in cp.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"launch_check": True, "world": 2, "max_rank": 1}

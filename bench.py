#!/usr/bin/env python3
"""bench.py -- decode tokens/s of the MI355X-native Qwen3 forward path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--model qwen3-0.6b] [--batch 64]

One "step" = one decode step of the whole batch through the HIP path (libnvllm_amd.so, C ABI):
every live sequence gets one new token; the greedy ids of every step come back to the host (the reference's
LLMEngine.step consumes them, src/engine/llm_engine.rs:239-264), one step late: the loop keeps one step
enqueued ahead (nvllm_decode_enqueue / nvllm_decode_collect).

N = 1  Qwen3-0.6B shapes, 64 live sequences (BASELINE.json configs[2] steady state: prompt lengths
       uniform 64..512, seed 0), synthetic bf16 weights generated in HBM, prefill untimed.
N > 1  sequences are independent units: every GPU serves its own 64 sequences with the TP=1 path, no
       data-path collective => `value` = N x per-replica rate, "weak" scaling (--parallel tp serves ONE batch
       tensor-parallel instead).
Every run also carries `tp_scaling`: decode tokens/s of Qwen3-32B (configs[4]: batch 64, prompt 128) at TP = N
over RCCL (row/column-parallel linears, 2 all-reduces per layer, vocab-parallel LM head), so the N = 1, 2, 4, 8
lines together are the TP=1/2/4/8 scaling curve the north_star asks for.

Prints ONE JSON line on rank 0 (driver contract) with `roofline` (dominant kernel, HIP events on the
library stream) and `cpu_baseline` (the C oracle in the reference's no-KV-cache mode on host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--model", default="qwen3-0.6b", choices=["qwen3-0.6b", "qwen3-8b", "qwen3-32b", "tiny"])
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--prompt-min", type=int, default=64)
    ap.add_argument("--prompt-max", type=int, default=512)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max-batched-tokens", type=int, default=4096,
                    help="rows per internal prefill chunk (SchedulerConfig.max_num_batched_tokens, scheduler.rs:10-56: default 4096)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seqs", type=int, default=12, help="sequences of the batch the CPU baseline re-runs (about 10-20 s of CPU work)")
    ap.add_argument("--profile-steps", type=int, default=8, help="extra steps for the per-kernel HIP-event pass")
    ap.add_argument("--parallel", default="dp", choices=["dp", "tp"],
                    help="N>1: dp = one TP=1 replica per GPU (weak scaling, no collective); tp = one TP=N group")
    ap.add_argument("--skip-tp-leg", action="store_true", help="do not run the tensor-parallel leg (tp_scaling)")
    ap.add_argument("--tp-model", default="qwen3-32b", choices=["qwen3-0.6b", "qwen3-8b", "qwen3-32b", "tiny"])
    ap.add_argument("--tp-batch", type=int, default=64)
    ap.add_argument("--tp-prompt", type=int, default=128)
    ap.add_argument("--tp-steps", type=int, default=32)
    ap.add_argument("--tp-timeout", type=float, default=300.0)
    ap.add_argument("--force-device", type=int, default=-1, help="testing aid: every rank uses this GPU ordinal")
    ap.add_argument("--tp-leg-child", action="store_true", help=argparse.SUPPRESS)  # internal: run only the TP leg
    ap.add_argument("--tp-proj-child", type=int, default=0, help=argparse.SUPPRESS)  # internal: one projection point (TP degree)
    ap.add_argument("--skip-tp-projection", action="store_true",
                    help="do not run the per-rank projection of the TP leg at TP = 2/4/8 (null communicator, one GPU)")
    ap.add_argument("--prefill-only", action="store_true",
                    help="time the prefill of the batch (cold call + warmed repeats) and stop: the MFMA-side profile run")
    ap.add_argument("--prefill-reps", type=int, default=3)
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="skip the rocprofv3 --pmc FETCH_SIZE child pass that fills roofline.traffic (N = 1 only; ~20 s)")
    ap.add_argument("--profile-meta", action="store_true",
                    help="add `all_decode_steps` to the line (tools/summarize_prof.py: bytes per launch averaged over EVERY decode step "
                         "of the process, what a rocprofv3 --stats average covers); off by default: one byte count per kernel in the line")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="A/B aid: nvllm_debug_set_option on the headline model before kv_alloc (repeatable)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="plain `python bench.py --gpus N`: overall limit of the rank processes")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous only (no GPU): every rank joins the gloo group, rank 0 prints the max rank seen")
    return ap.parse_args()


def model_config(pkg, name):
    return {"qwen3-0.6b": pkg.Qwen3Config.qwen3_0_6b, "qwen3-8b": pkg.Qwen3Config.qwen3_8b,
            "qwen3-32b": pkg.Qwen3Config.qwen3_32b, "tiny": pkg.Qwen3Config.tiny}[name]()


def make_prompts(cfg, batch, lo, hi, seed):
    import numpy as np

    rng = np.random.default_rng(seed)
    lens = rng.integers(lo, hi + 1, size=batch)
    return [rng.integers(0, cfg.vocab_size, size=int(n), dtype=np.uint32).tolist() for n in lens]


def cpu_baseline(cfg, prompts, n_seqs, seed):
    """The reference's CPU path restated (oracle/qwen3_oracle.c): no KV cache, whole sequences re-fed,
    LM head on all rows -- one engine step for the first n_seqs sequences of the batch."""
    from oracle import oracle as O

    ocfg = O.make_config(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, head_dim=cfg.head_dim,
                         num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                         num_key_value_heads=cfg.num_key_value_heads, intermediate_size=cfg.intermediate_size,
                         max_position_embeddings=cfg.max_position_embeddings, rms_norm_eps=cfg.rms_norm_eps,
                         rope_theta=cfg.rope_theta, bos_token_id=cfg.bos_token_id, eos_token_id=cfg.eos_token_id)
    om = O.Model(ocfg).fill_synthetic(seed)
    sample = [list(p) for p in prompts[:n_seqs]]
    t0 = time.perf_counter()
    om.run_greedy(sample, all_rows=True, want_logits=False)
    dt = time.perf_counter() - t0
    cores = O.threads  # worker threads the oracle really used (CPU quota of the container, not visible cores)
    return {"value": n_seqs / dt, "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"1 engine step of {n_seqs} of the {len(prompts)} sequences (lens {[len(s) for s in sample]}), "
                      f"reference mode: no KV cache, full re-forward, LM head on all rows; {dt:.1f} s"}


def run_tp_extra(pkg, torch, dist, a, rank, world, local_rank):
    """Tensor-parallel leg (north_star: column/row-parallel linears over RCCL, TP=1/2/4/8 scaling on Qwen3-32B):
    decode tokens/s of `--tp-model` at TP = world on synthetic prompts, all ranks in one RCCL group."""
    import numpy as np

    rccl_id = None
    if world > 1:
        box = [pkg.Context.make_rccl_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        rccl_id = box[0]
    ctx = pkg.Context(local_rank, tp_rank=rank, tp_size=world, rccl_id=rccl_id)
    cfg = model_config(pkg, a.tp_model)
    model = pkg.Qwen3ForCausalLM.from_synthetic(cfg, seed=a.seed, ctx=ctx)
    B, P, steps, warm = a.tp_batch, a.tp_prompt, a.tp_steps, 4
    rng = np.random.default_rng(a.seed + 1)
    prompts = [rng.integers(0, cfg.vocab_size, size=P, dtype=np.uint32).tolist() for _ in range(B)]
    model.kv_alloc(num_blocks=B * (-(-(P + steps + warm + 4) // 256)) + 2, max_seqs=B, max_batched_tokens=4096)
    model.step(list(range(B)), prompts, is_prefill=True)
    for _ in range(warm):
        model.decode_next()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    byts = 0
    for _ in range(steps):
        model.decode_next()
        byts += model.last_step_bytes
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    res = {"model": a.tp_model, "tp": world, "batch": B, "prompt": P, "steps": steps, "tokens_per_s": B * steps / dt,
           "ms_per_step": dt * 1e3 / steps, "per_rank_bytes_per_step": byts / steps,
           "per_rank_hbm_frac": (byts / steps) / (dt / steps) / 1e9 / HBM_PEAK_GBS,
           "collectives": "2 RCCL all-reduce(sum) of [batch, hidden] f32 per layer + (max,idx) all-gather" if world > 1 else "none"}
    if world > 1 and rank == 0:
        print("TP_LEG_RESULT " + json.dumps(res), flush=True)  # on record before the optional trial below
    if world > 1 and not os.environ.get("NVLLM_BENCH_SKIP_ONESHOT"):
        # Trial of the opt-in one-shot all-reduce (csrc/oneshot.hip) on the SAME group: greedy ids of a replay must match
        # the RCCL replay, then the same timed loop.  Everything here may fail or time out without touching `res` above.
        try:
            def replay(n):
                model.kv_alloc(num_blocks=B * (-(-(P + steps + warm + 4) // 256)) + 2, max_seqs=B, max_batched_tokens=4096)
                first = model.step(list(range(B)), prompts, is_prefill=True)[0].copy()
                return [first] + [model.decode_next()[:B].copy() for _ in range(n)]

            ref_ids = replay(8)
            model.set_option("oneshot_allreduce", 1)
            os_ids = replay(8)
            calls = model.counter("oneshot_calls")
            match = float(np.mean([np.mean(x == y) for x, y in zip(ref_ids, os_ids)]))
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                model.decode_next()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            dt2 = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([dt2], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt2 = float(t.item())
            res["oneshot_allreduce_trial"] = {
                "active": calls > 0, "device_allreduces": calls, "ms_per_step": dt2 * 1e3 / steps, "tokens_per_s": B * steps / dt2,
                "greedy_ids_equal_to_rccl_run": match,
                "note": "opt-in path (NVLLM_ONESHOT_AR=1); ids of prefill + 8 decode steps compared with the RCCL replay (sum order differs: rare ties may flip)"}
        except Exception as e:  # noqa: BLE001
            res["oneshot_allreduce_trial"] = {"error": repr(e)[:300]}
    model.close()
    ctx.close()
    return res


def spawn_ranks(n, timeout_s=1500.0):
    """Launcher for `python bench.py --gpus N` without torch.distributed.run: N child processes with
    RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in their environment, started BEFORE anything in this process initialises
    the GPU (no exec of a GPU-initialised process anywhere).  Rank 0's stdout (the one JSON line) is relayed.
    Every child is polled: as soon as one exits non-zero, or after timeout_s, the rest are terminated and the exit
    code is 1 (a rank that dies before the rendezvous would otherwise leave its peers in gloo's 30-minute timeout)."""
    import socket
    import subprocess
    import tempfile

    # the listening socket stays open (SO_REUSEADDR) until the children are started, so nobody else is handed the port
    sk = socket.socket()
    sk.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    out0 = tempfile.TemporaryFile(mode="w+")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if r == 0:
            sk.close()  # rank 0 binds the port next; the others only connect to it
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, text=True))
    deadline = time.monotonic() + timeout_s
    bad = []
    while True:
        rcs = [p.poll() for p in procs]
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad or all(rc == 0 for rc in rcs):
            break
        if time.monotonic() > deadline:
            bad = [(r, "timeout") for r, rc in enumerate(rcs) if rc is None]
            break
        time.sleep(0.2)
    for p in procs:  # a failed or late run: stop what is still alive (exact PIDs we started)
        if p.poll() is None:
            p.terminate()
    for p in procs:
        try:
            p.wait(timeout=20)
        except subprocess.TimeoutExpired:
            p.kill()
    out0.seek(0)
    sys.stdout.write(out0.read())
    sys.stdout.flush()
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


def run_tp_projection_point(pkg, a, tp):
    """Per-rank compute time of one TP decode step WITHOUT a multi-GPU node: rank 0 of a tp-way group on a null
    communicator (the rank's shard shapes and kernels, every collective skipped).  Results of such a step are
    meaningless; its duration bounds what TP = tp can reach: step(tp) >= this + 2 * layers all-reduces."""
    import numpy as np

    ctx = pkg.Context(0, tp_rank=0, tp_size=tp, null_comm=True)
    cfg = model_config(pkg, a.tp_model)
    model = pkg.Qwen3ForCausalLM.from_synthetic(cfg, seed=a.seed, ctx=ctx)
    B, P, steps, warm = a.tp_batch, a.tp_prompt, min(a.tp_steps, 16), 4
    rng = np.random.default_rng(a.seed + 1)
    prompts = [rng.integers(0, cfg.vocab_size, size=P, dtype=np.uint32).tolist() for _ in range(B)]
    model.kv_alloc(num_blocks=B * (-(-(P + steps + warm + 4) // 256)) + 2, max_seqs=B, max_batched_tokens=4096)
    model.step(list(range(B)), prompts, is_prefill=True)
    for _ in range(warm):
        model.decode_next()
    ctx.synchronize()
    ctx.timer_start()
    byts = 0
    for _ in range(steps):
        model.decode_next()
        byts += model.last_step_bytes
    ms = ctx.timer_stop() / steps
    res = {"tp": tp, "per_rank_ms_per_step": ms, "per_rank_bytes_per_step": byts / steps,
           "per_rank_hbm_frac": (byts / steps) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    model.close()
    ctx.close()
    return res


def live_traffic(a, dom_kernel):
    """roofline.traffic measured in THIS invocation: the same decode workload once more in a child process under
    `rocprofv3 --pmc FETCH_SIZE` (counters in a pass of their own, as /opt/skills/guides/MI355X_MICROARCH.md prescribes:
    FETCH_SIZE is in KB and reads half the bytes of wide streaming loads on gfx950 -> x 1024 x 2).  The child runs the same
    steps, so the average is over the same launches as its own algorithmic byte count.  Any failure -> None + the reason."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 not on PATH"
    if any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_CTOR")):
        return None, "already running under a profiler (no nested counter pass)"
    tmp = tempfile.mkdtemp(prefix="nvllm_pmc_", dir="/tmp")
    cmd = ["rocprofv3", "--pmc", "FETCH_SIZE", "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "r", "--",
           sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(a.steps), "--warmup", str(a.warmup),
           "--model", a.model, "--batch", str(a.batch), "--prompt-min", str(a.prompt_min), "--prompt-max", str(a.prompt_max),
           "--seed", str(a.seed), "--profile-steps", str(a.profile_steps), "--no-cpu-baseline", "--skip-tp-leg", "--no-live-traffic",
           "--profile-meta"] + [x for o in a.option for x in ("--option", o)]
    try:
        cp = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=240)
        line = [l for l in cp.stdout.splitlines() if l.startswith('{"metric"')]
        files = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
        if cp.returncode != 0 or not line or not files:
            return None, f"counter pass failed (exit {cp.returncode}): " + (cp.stderr.strip().splitlines() or ["no output"])[-1][:200]
        child = json.loads(line[-1])
        n, tot = 0, 0.0
        with open(files[0]) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") == "FETCH_SIZE" and dom_kernel in row["Kernel_Name"].replace("nvllm::", ""):
                    n += 1
                    tot += float(row["Counter_Value"])
        if n == 0:
            return None, f"no dispatch of {dom_kernel} in the counter pass"
        fetch = 2.0 * 1024.0 * tot / n
        alg = child.get("all_decode_steps", {}).get("attn_algorithmic_bytes_per_launch") if dom_kernel.startswith("attn_paged") else None
        return {"bytes_per_launch": fetch, "launches": n, "over_algorithmic_of_that_pass": (fetch / alg) if alg else None,
                "how": "child pass: rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 bench.py <same workload>; mean FETCH_SIZE [KB] x 1024 x 2 "
                       "(gfx950 wide-stream correction)"}, None
    except Exception as e:  # noqa: BLE001
        return None, repr(e)[:200]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    a = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if a.force_device >= 0:
        local_rank = a.force_device
    if world == 1 and a.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher (it never imports torch or touches a
        # GPU) and starts one fresh rank process per GPU, exactly what torch.distributed.run would have started
        sys.exit(spawn_ranks(a.gpus, a.launch_timeout))
    if world != a.gpus:
        sys.exit(f"bench.py --gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    import numpy as np
    import torch

    import nano_vllm_candle_amd as pkg

    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane (id exchange, barrier, max over ranks) on gloo; the data path is RCCL inside the library
        import datetime

        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    if a.launch_check:
        t = torch.tensor([float(rank)], dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"launch_check": True, "world": world, "max_rank": int(t.item())}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    torch.cuda.set_device(local_rank)
    if a.tp_proj_child:
        print("TP_PROJ_RESULT " + json.dumps(run_tp_projection_point(pkg, a, a.tp_proj_child)), flush=True)
        return
    if a.tp_leg_child:
        # child process of a bench rank: only the tensor-parallel leg, result as one JSON line on stdout (rank 0)
        res = run_tp_extra(pkg, torch, dist, a, rank, world, local_rank)
        if rank == 0:
            print("TP_LEG_RESULT " + json.dumps(res), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    # Headline workload.  --parallel tp: the batch is served tensor-parallel over all ranks (one RCCL group).
    # --parallel dp (default for N > 1): sequences are independent units, so every GPU serves its own 64
    # sequences with the TP=1 path and no data-path collective ("weak" scaling); the TP leg below measures the
    # collective path on the model it is meant for.
    use_tp = a.parallel == "tp" and world > 1
    rccl_id = None
    if use_tp:
        box = [pkg.Context.make_rccl_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        rccl_id = box[0]
    ctx = pkg.Context(local_rank, tp_rank=rank if use_tp else 0, tp_size=world if use_tp else 1, rccl_id=rccl_id)
    tpw = world if use_tp else 1
    cfg = model_config(pkg, a.model)
    model = pkg.Qwen3ForCausalLM.from_synthetic(cfg, seed=a.seed, ctx=ctx)
    for opt in a.option:
        name, _, val = opt.partition("=")
        model.set_option(name, int(val or 1))
    prompts = make_prompts(cfg, a.batch, a.prompt_min, a.prompt_max, a.seed + (0 if use_tp else rank))
    total_steps = a.warmup + a.steps + a.profile_steps + 2
    max_len = max(len(p) for p in prompts) + total_steps
    blocks = sum(-(-(len(p) + total_steps) // 256) for p in prompts) + 2
    model.kv_alloc(num_blocks=blocks, max_seqs=a.batch, max_batched_tokens=a.max_batched_tokens)
    seq_ids = list(range(a.batch))

    # prefill (not part of `value`; timed separately for the MFMA-side statement), chunked by max_batched_tokens
    torch.cuda.synchronize()
    tp0 = time.perf_counter()
    model.step(seq_ids, prompts, is_prefill=True)
    torch.cuda.synchronize()
    prefill_cold_s = time.perf_counter() - tp0
    prefill_tokens = sum(len(p) for p in prompts)
    # warmed figure: the same prefill again (is_prefill restarts the sequences and re-uses their blocks); the first call
    # also pays one-time costs (kernel code upload, LDS-size attributes, first touch of the step buffers)
    reps = a.prefill_reps if a.prefill_only else 1
    tw = []
    for _ in range(reps):
        torch.cuda.synchronize()
        tp0 = time.perf_counter()
        model.step(seq_ids, prompts, is_prefill=True)
        torch.cuda.synchronize()
        tw.append(time.perf_counter() - tp0)
    prefill_s = min(tw)
    # prefill is the MFMA-bound side.  Algorithmic flops of what the kernels compute: the four layer projections on every
    # prompt token, the LM head on the LAST row of every sequence only (compute_logits is needed for row len-1,
    # llm_engine.rs:181-183; the reference runs it on all rows, qwen3.rs:548), attention reported as its own term.  The
    # kernels issue 2x the projection flops on the MFMA pipe because activations are bf16 hi + lo.
    layer_params = cfg.hidden_size * (cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * cfg.head_dim \
        + cfg.hidden_size * cfg.num_attention_heads * cfg.head_dim + 3 * cfg.hidden_size * cfg.intermediate_size
    proj_flops = 2.0 * cfg.num_hidden_layers * layer_params * prefill_tokens
    lm_flops = 2.0 * cfg.vocab_size * cfg.hidden_size * len(prompts)
    attn_flops = 4.0 * cfg.num_attention_heads * cfg.head_dim * cfg.num_hidden_layers * sum(len(p) * (len(p) + 1) / 2 for p in prompts)
    pf_tflops = (proj_flops + lm_flops) / prefill_s / 1e12
    prefill_info = {"tokens": prefill_tokens, "ms": prefill_s * 1e3, "ms_first_call": prefill_cold_s * 1e3,
                    "tokens_per_s": prefill_tokens / prefill_s, "algorithmic_tflops": pf_tflops,
                    "algorithmic_flops": {"layer_projections": proj_flops, "lm_head_last_rows": lm_flops,
                                          "attention_causal_not_in_frac": attn_flops},
                    "mfma_peak_tflops_bf16_dense": 2500.0 * tpw, "frac_of_mfma_peak": pf_tflops / (2500.0 * tpw),
                    "frac_of_mfma_peak_with_attention": (proj_flops + lm_flops + attn_flops) / prefill_s / 1e12 / (2500.0 * tpw),
                    "note": "warmed (repeat of the same prefill); ms_first_call includes one-time setup; frac = (projections on "
                            "all tokens + LM head on last rows) / time / dense bf16 peak"}
    if a.prefill_only:
        if rank == 0:
            print(json.dumps({"metric": "prefill tokens/sec", "value": prefill_tokens / prefill_s, "unit": "tokens/s",
                              "n_gpus": world, "config": {"workload": f"{a.model} prefill of {a.batch} prompts U[{a.prompt_min},{a.prompt_max}] seed {a.seed}"},
                              "prefill": prefill_info}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    ctx_lens = np.array([len(p) + 1 for p in prompts], dtype=np.int64)  # tokens attended by the next decode step

    def barrier():
        if dist is not None:
            dist.barrier()

    all_step_tokens = []  # attended tokens of EVERY decode step this process runs (what a rocprofv3 average is taken over)
    for _ in range(a.warmup):
        all_step_tokens.append(int(ctx_lens.sum()))
        model.decode_next()
        ctx_lens += 1
    barrier()
    torch.cuda.synchronize()
    bytes_total = 0
    ctx.timer_start()
    t0 = time.perf_counter()
    # pipelined decode: step t+1 is enqueued before step t's ids are collected, so the GPU never waits for the
    # host between steps; every step's ids still reach the host (asynchronous scheduling, one step late)
    model.decode_enqueue()
    bytes_total += model.last_step_bytes
    for _ in range(a.steps - 1):
        model.decode_enqueue()
        bytes_total += model.last_step_bytes
        model.decode_collect()
    model.decode_collect()
    all_step_tokens += [int(ctx_lens.sum()) + i * a.batch for i in range(a.steps)]
    ctx_lens += a.steps
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    ev_ms = ctx.timer_stop()
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt * 1e3 / a.steps
    replicas = 1 if use_tp else world
    tok_s = replicas * a.batch * a.steps / dt  # whole job: every replica serves its own batch
    mean_ctx = float(ctx_lens.mean()) - a.steps / 2

    # per-kernel HIP-event pass (outside the timed region; same process, same resident state)
    kern = {}
    pass_steps = max(1, a.profile_steps // 4)
    attn_ctx = []
    for kind in ("attn", "gemm", "lm_head", "norm", "qk", "silu", "empty"):
        model.profile_kernel(kind)
        for _ in range(pass_steps):
            if kind == "attn":
                attn_ctx.append(int(ctx_lens.sum()))
            all_step_tokens.append(int(ctx_lens.sum()))
            model.decode_next()
            ctx_lens += 1
        kern[kind] = model.profile_read()
    model.profile_kernel(None)
    # an event pair around nothing, recorded between busy kernels: what every bracket adds to a launch's duration
    # A bracket around a launch adds less than a whole empty bracket (the second event's processing overlaps the
    # kernel's tail).  Calibration against rocprofv3 --kernel-trace of this command, compared on the SAME step window
    # (round 1 compared a bracketed figure of the later, longer-context profile steps with rocprof's all-step average and
    # took 0.8): bracketed 26.79 us on the profile steps (95.9 MB per launch), empty bracket 5.89 us, rocprof 22.70 us over
    # all steps (87.1 MB) = 24.3 us scaled to the profile steps' bytes (7 us fixed + a part proportional to the bytes):
    # excess 2.5 us = 0.42 of the empty bracket.  0.4 keeps the figure conservative (a slightly longer kernel).
    empty_ms, empty_n = kern.pop("empty")
    bracket_us = 0.4 * empty_ms / max(empty_n, 1) * 1e3
    per_step = {k: ms / pass_steps for k, (ms, n) in kern.items()}
    kv_layer = model.kv_bytes_per_token // cfg.num_hidden_layers  # K+V bytes of one token in one layer (this rank)
    # dominant kernel = the class that moves the most algorithmic bytes per step (the path is HBM-bound); the
    # HIP-event brackets add ~2 us per launch, so ranking by bracketed time would favour many-launch classes
    lm_bytes = cfg.hidden_size * (cfg.vocab_size // tpw) * 2
    step_bytes_by_kind = {"attn": float(np.mean(attn_ctx)) * kv_layer * cfg.num_hidden_layers,
                          "gemm": float(model.weight_bytes - lm_bytes), "lm_head": float(lm_bytes)}
    dom = max(step_bytes_by_kind, key=step_bytes_by_kind.get)
    dom_ms, dom_n = kern[dom]
    if dom == "attn":
        # algorithmic bytes of one launch = every attended token's K and V of this layer, read once
        dom_bytes = float(np.mean(attn_ctx)) * kv_layer
        _o = dict(o.partition("=")[::2] for o in a.option)  # the cache-width options select the kernel instantiation
        _vlo = 2 if _o.get("kv_k_bits") == "24" else (1 if _o.get("kv_v_bits") == "24" else 0)
        dom_name = f"attn_paged_kernel<128, 1, 4, true, {_vlo}>"
    elif dom == "lm_head":
        dom_bytes = float(lm_bytes)
        dom_name = "lmhead_kernel"
    elif dom == "gemm":
        dom_bytes = float(model.weight_bytes - lm_bytes) / (4 * cfg.num_hidden_layers)
        dom_name = "layer projection GEMMs (mean of qkv / o_proj / gate_up / down launches)"
    else:
        dom_bytes = None
        dom_name = dom
    step_gbs = (bytes_total / a.steps) / (ev_ms / a.steps * 1e-3) / 1e9
    raw_us = dom_ms / max(dom_n, 1) * 1e3
    launch_us = max(raw_us - bracket_us, 1e-3)  # bracketed duration minus the calibrated empty bracket
    roof = {"bound": "hbm", "kernel": dom_name, "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
            "avg_launch_us": launch_us, "avg_launch_us_bracketed": raw_us, "bracket_correction_us": bracket_us}
    # `traffic` (PMC HBM bytes of THIS run) needs rocprofv3 around the process, so it is null in the live line; the
    # committed counter pass of the same command is quoted beside it with its own algorithmic bytes (a different run,
    # hence a different context: never mixed into `achieved`)
    default_workload = a.model == "qwen3-0.6b" and a.batch == 64 and a.prompt_min == 64 and a.prompt_max == 512 and a.seed == 0
    for pmc_path in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_fetch_size.json") and default_workload), reverse=True):
        pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_path))).get(dom_name)
        if pmc:
            # the committed counter pass of this same command: only the RATIO is quoted here (its byte counts live in the file)
            roof["traffic_from_profile"] = {"file": "profiles/" + pmc_path,
                                            "hbm_fetch_over_algorithmic": pmc["fetch_bytes_per_launch"] / pmc["algorithmic_bytes_per_launch"]}
            break
    if rank == 0 and world == 1 and not a.no_live_traffic and dom in ("attn", "lm_head"):
        tr, why = live_traffic(a, dom_name)
        if tr:
            roof["traffic"] = tr["bytes_per_launch"]
            roof["traffic_detail"] = {k: v for k, v in tr.items() if k != "bytes_per_launch"}
        else:
            roof["traffic_unavailable"] = why
    if dom_bytes is not None:
        roof["achieved"] = dom_bytes / (launch_us * 1e-6) / 1e9
        roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
        roof["bytes_per_launch"] = dom_bytes
    opts = dict(o.partition("=")[::2] for o in a.option)
    kv_desc = {("16", "16"): "f16", ("16", "24"): "f16 K / 24-bit V (f16 + bf8 residual)", ("24", "24"): "24-bit K and V (f16 + bf8 residual)"}.get(
        (opts.get("kv_k_bits", "16"), opts.get("kv_v_bits", "16")), "f16")
    out = {
        "metric": "decode tokens/sec", "value": tok_s, "unit": "tokens/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong" if use_tp else "weak", "vs_baseline": None,
        "dtype": f"bf16 weights, {kv_desc} KV cache, bf16x2(hi+lo)/f16 MFMA operands, f32 accumulate",
        "data": "synthetic",
        "config": {"workload": f"{a.model} decode, {a.batch} live sequences, prompts U[{a.prompt_min},{a.prompt_max}] seed {a.seed}",
                   **({"options": list(a.option)} if a.option else {}),
                   "batch": a.batch, "global_batch": a.batch * replicas, "mean_context": round(mean_ctx, 1),
                   "parallelism": f"tp{world}" if use_tp else (f"dp{world} x tp1 (independent replicas, no collective)" if world > 1 else "tp1")},
        "roofline": roof,
        "step_roofline": {"bound": "hbm", "achieved": step_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": step_gbs / HBM_PEAK_GBS, "bytes_per_step": bytes_total / a.steps,
                          "event_ms_per_step": ev_ms / a.steps},
        "kernel_ms_per_step": {k: round(v, 4) for k, v in per_step.items()},
    }
    if a.profile_meta:
        # for tools/summarize_prof.py: the attention kernel's algorithmic bytes per launch averaged over ALL decode steps of
        # this process (warm-up + timed + per-kernel pass) -- the launches a rocprofv3 --stats average covers
        out["all_decode_steps"] = {"steps": len(all_step_tokens), "attn_algorithmic_bytes_per_launch": float(np.mean(all_step_tokens)) * kv_layer if all_step_tokens else None,
                                   "weight_bytes": {"qkv": 2 * cfg.hidden_size * (cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * cfg.head_dim // tpw,
                                                    "o_proj": 2 * cfg.hidden_size * cfg.num_attention_heads * cfg.head_dim // tpw,
                                                    "gate_up": 4 * cfg.hidden_size * cfg.intermediate_size // tpw,
                                                    "down": 2 * cfg.hidden_size * cfg.intermediate_size // tpw, "lm_head": lm_bytes}}
    out["prefill"] = prefill_info
    if rank == 0 and world == 1 and not a.no_cpu_baseline:  # reported at N=1 only (driver contract)
        out["cpu_baseline"] = cpu_baseline(cfg, prompts, a.cpu_seqs, a.seed)
    # ---- tensor-parallel leg, in a CHILD process per rank (own rendezvous port): an RCCL crash or a collective that
    # ---- never completes costs the tp_scaling entry, never the headline line
    if not a.skip_tp_leg:
        import subprocess

        model.close()
        env = dict(os.environ)
        env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 17)
        env["MASTER_ADDR"] = os.environ.get("MASTER_ADDR", "127.0.0.1")
        # torchrun makes its ranks clients of the launcher's store; on the new port child rank 0 must host its own
        env["TORCHELASTIC_USE_AGENT_STORE"] = "False"
        cmd = [sys.executable, os.path.abspath(__file__), "--tp-leg-child", "--gpus", str(world), "--tp-model", a.tp_model,
               "--tp-batch", str(a.tp_batch), "--tp-prompt", str(a.tp_prompt), "--tp-steps", str(a.tp_steps),
               "--seed", str(a.seed)]
        if a.force_device >= 0:
            cmd += ["--force-device", str(a.force_device)]
        tp_res = {"model": a.tp_model, "tp": world, "error": "no result"}
        try:
            cp = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=a.tp_timeout)
            for line in cp.stdout.splitlines():
                if line.startswith("TP_LEG_RESULT "):  # the last one wins: it carries the one-shot trial when that finished
                    tp_res = json.loads(line[len("TP_LEG_RESULT "):])
            if "error" in tp_res and rank == 0:
                tp_res["error"] = f"child exit {cp.returncode}: " + (cp.stderr.strip().splitlines() or ["?"])[-1][:300]
        except subprocess.TimeoutExpired as e:
            tail = (e.stderr.decode(errors="replace") if isinstance(e.stderr, bytes) else (e.stderr or "")).strip().splitlines()
            tp_res = {"model": a.tp_model, "tp": world, "error": f"timed out after {a.tp_timeout} s: " + (tail[-1][:200] if tail else "")}
            so = e.stdout.decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or "")
            for line in so.splitlines():  # the RCCL result is printed before the one-shot trial: a trial that hangs costs only itself
                if line.startswith("TP_LEG_RESULT "):
                    tp_res = json.loads(line[len("TP_LEG_RESULT "):])
                    tp_res["oneshot_allreduce_trial"] = {"error": f"child timed out after {a.tp_timeout} s during the trial"}
        except Exception as e:  # noqa: BLE001
            tp_res = {"model": a.tp_model, "tp": world, "error": repr(e)[:300]}
        out["tp_scaling"] = tp_res
        # projection of the same leg at TP = 2/4/8 on THIS GPU (rank 0's shard, null communicator): N = 1 only
        if world == 1 and not a.skip_tp_projection and "ms_per_step" in tp_res:
            proj = {"label": "projected_not_measured: rank 0's shard shapes and kernels on one GPU, every collective skipped; "
                             "a real TP step adds 2 all-reduces of [batch, hidden] f32 per layer + the (max, index) gather",
                    "model": a.tp_model, "batch": a.tp_batch, "prompt": a.tp_prompt, "tp1_ms_per_step": tp_res["ms_per_step"], "points": []}
            for tpd in (2, 4, 8):
                cmdp = [sys.executable, os.path.abspath(__file__), "--tp-proj-child", str(tpd), "--tp-model", a.tp_model,
                        "--tp-batch", str(a.tp_batch), "--tp-prompt", str(a.tp_prompt), "--tp-steps", str(a.tp_steps), "--seed", str(a.seed)]
                pt = {"tp": tpd, "error": "no result"}
                try:
                    cpp = subprocess.run(cmdp, env=dict(os.environ), capture_output=True, text=True, timeout=a.tp_timeout)
                    for line in cpp.stdout.splitlines():
                        if line.startswith("TP_PROJ_RESULT "):
                            pt = json.loads(line[len("TP_PROJ_RESULT "):])
                    if "error" in pt:
                        pt["error"] = f"child exit {cpp.returncode}: " + (cpp.stderr.strip().splitlines() or ["?"])[-1][:200]
                except Exception as e:  # noqa: BLE001
                    pt = {"tp": tpd, "error": repr(e)[:200]}
                if "per_rank_ms_per_step" in pt:
                    pt["compute_only_speedup_vs_tp1"] = tp_res["ms_per_step"] / pt["per_rank_ms_per_step"]
                proj["points"].append(pt)
            out["tp_projection"] = proj
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

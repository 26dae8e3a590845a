"""Turn the raw rocprofv3 output of tools/profile_round.sh into the committed summaries under profiles/.
usage: python tools/summarize_prof.py TAG ROUND   (reads gpurun_out/{prof,pmc,prof_prefill,pmc_prefill,bench}_TAG*, writes profiles/rROUND_*)

Every fraction quoted in DESIGN.md / README.md is a row of profiles/rROUND_decode_kernels.csv: average duration of the
kernel over ALL its launches of the profiled run (rocprofv3 --kernel-trace --stats), algorithmic bytes per launch averaged
over the SAME launches (bench.py prints them: all_decode_steps), FETCH_SIZE of the separate counter pass (x2: the gfx950
wide-stream correction of MI355X_MICROARCH.md, HBM section) with that pass's own algorithmic bytes beside it."""
import csv
import glob
import json
import os
import re
import sys

tag, rnd = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
HBM_PEAK = 8000.0  # GB/s
G = os.path.join(root, "gpurun_out")


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("nvllm::", "")


def bench_line(log):
    lines = [l for l in open(log).read().splitlines() if l.startswith('{"metric"')]
    return json.loads(lines[-1]) if lines else None


def find(pattern):
    r = glob.glob(os.path.join(G, pattern), recursive=True)
    return r[0] if r else None


def fetch_per_kernel(path):
    acc = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != "FETCH_SIZE":
                continue
            a = acc.setdefault(short(row["Kernel_Name"]), [0, 0.0])
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    return {k: 2 * tot / n * 1024 for k, (n, tot) in acc.items()}  # bytes per launch, corrected


# ---- decode: kernel stats + per-kernel roofline table ----------------------------------------------------------
ks = find(f"prof_{tag}/**/*kernel_stats.csv")
prof_bench = bench_line(os.path.join(G, f"prof_{tag}.log"))
pmc = find(f"pmc_{tag}/**/*counter_collection.csv")
pmc_bench = bench_line(os.path.join(G, f"pmc_{tag}.log")) if os.path.exists(os.path.join(G, f"pmc_{tag}.log")) else None
if ks and prof_bench:
    dst = os.path.join(out, f"r{rnd}_bench_decode_kernel_stats.csv")
    open(dst, "w").write(open(ks).read())
    print("wrote", dst)
    info = prof_bench["all_decode_steps"]
    wb = info["weight_bytes"]
    # which projection a register-direct instantiation is, by its template arguments <NT, 16, TK, EPI> (Qwen3-0.6B shapes)
    alg = {"attn_paged_kernel<128, 1, 4, true, 0>": info["attn_algorithmic_bytes_per_launch"],
           "gemm_rowdir_kernel<4, 16, 2, 2>": wb["qkv"], "gemm_rowdir_kernel<1, 16, 4, 0>": wb["o_proj"],
           "gemm_rowdir_kernel<6, 16, 2, 1>": wb["gate_up"], "gemm_rowdir_kernel<1, 16, 6, 0>": wb["down"],
           "lmhead_kernel<4, 5, 4>": wb["lm_head"]}
    fetch = fetch_per_kernel(pmc) if pmc else {}
    pmc_alg = dict(alg)
    if pmc_bench:
        pmc_alg["attn_paged_kernel<128, 1, 4, true, 0>"] = pmc_bench["all_decode_steps"]["attn_algorithmic_bytes_per_launch"]
    rows = []
    with open(ks) as f:
        for row in csv.DictReader(f):
            k = short(row["Name"])
            if k in alg:
                avg_us = float(row["AverageNs"]) / 1e3
                gbs = alg[k] / (avg_us * 1e-6) / 1e9
                rows.append((k, int(row["Calls"]), avg_us, alg[k], gbs, gbs / HBM_PEAK, fetch.get(k), pmc_alg.get(k)))
    dst = os.path.join(out, f"r{rnd}_decode_kernels.csv")
    with open(dst, "w") as g:
        g.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --skip-tp-leg (the driver's command; MI355X, {prof_bench['config']['workload']})\n"
                f"# avg_us: all launches of the run ({info['steps']} decode steps); algorithmic_bytes: mean over the SAME launches; frac = GB/s / {HBM_PEAK:.0f}\n"
                "# fetch_bytes: separate --pmc FETCH_SIZE pass (x2 gfx950 correction), quoted with that pass's own algorithmic bytes\n"
                "kernel,launches,avg_us,algorithmic_bytes_per_launch,achieved_GBps,frac_of_hbm_peak,fetch_bytes_per_launch_pmc_pass,algorithmic_bytes_pmc_pass\n")
        for r in sorted(rows, key=lambda r: -r[2] * r[1]):
            g.write(f"\"{r[0]}\",{r[1]},{r[2]:.2f},{r[3]:.0f},{r[4]:.0f},{r[5]:.3f},{'' if r[6] is None else f'{r[6]:.0f}'},{'' if r[7] is None else f'{r[7]:.0f}'}\n")
        step_us = sum(r[2] * r[1] for r in rows) / info["steps"]
        g.write(f"# sum of these kernels per decode step: {step_us:.1f} us; bench ms_per_step of the same run: {prof_bench['ms_per_step']:.4f}\n")
    print("wrote", dst)
    dom = "attn_paged_kernel<128, 1, 4, true, 0>"
    if dom in fetch and pmc_bench:
        js = {"_comment": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --skip-tp-leg "
                          "(the driver's command) on MI355X; FETCH_SIZE KB x 2 (gfx950 wide-stream correction, MI355X_MICROARCH.md HBM section) x 1024; algorithmic = mean over "
                          f"the same run's launches. Full table: r{rnd}_decode_kernels.csv",
              dom: {"fetch_bytes_per_launch": int(round(fetch[dom])), "algorithmic_bytes_per_launch": int(pmc_alg[dom])}}
        dst = os.path.join(out, f"r{rnd}_pmc_fetch_size.json")
        json.dump(js, open(dst, "w"), indent=1)
        print("wrote", dst, js[dom])

# ---- prefill: kernel stats + matrix-core counters ---------------------------------------------------------------
pks = find(f"prof_prefill_{tag}/**/*kernel_stats.csv")
pb = bench_line(os.path.join(G, f"prof_prefill_{tag}.log")) if os.path.exists(os.path.join(G, f"prof_prefill_{tag}.log")) else None
if pks:
    dst = os.path.join(out, f"r{rnd}_prefill_kernel_stats.csv")
    open(dst, "w").write(open(pks).read())
    print("wrote", dst)
ppmc = find(f"pmc_prefill_{tag}/**/*counter_collection.csv")
if ppmc:
    acc, calls = {}, {}
    with open(ppmc) as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            acc.setdefault(k, {}).setdefault(row["Counter_Name"], 0.0)
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                calls[k] = calls.get(k, 0) + 1
    dst = os.path.join(out, f"r{rnd}_prefill_mfma.csv")
    with open(dst, "w") as g:
        g.write("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace -- "
                "python3 bench.py --no-cpu-baseline --skip-tp-leg --prefill-only --prefill-reps 1 (MI355X)\n"
                "# mfma_busy_pct = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs): rocprofv3's MfmaUtil expression with GRBM_GUI_ACTIVE\n"
                "# summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back).  The kernels issue every product twice (bf16 hi + lo activation planes),\n"
                "# so the algorithmic share of the dense bf16 peak is half the busy share at best.\n"
                "kernel,dispatches,SQ_VALU_MFMA_BUSY_CYCLES,GRBM_GUI_ACTIVE,mfma_busy_pct,MFMA_MOPS_BF16_x512_flops\n")
        for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
            busy, act = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("GRBM_GUI_ACTIVE", 0.0)
            if busy <= 0:
                continue
            g.write(f"\"{k}\",{calls.get(k, 0)},{busy:.0f},{act:.0f},{100 * busy / (act / 8 * 1024):.1f},{v.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0) * 512:.3e}\n")
        if pb:
            g.write(f"# the un-profiled prefill-only run: {json.dumps(pb['prefill'])}\n")
    print("wrote", dst)

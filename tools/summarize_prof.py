"""Turn the raw rocprofv3 output of tools/profile_round.sh into the committed summaries under profiles/.
usage: python tools/summarize_prof.py TAG ROUND   (reads gpurun_out/{prof,pmc,bench}_TAG*, writes profiles/rROUND_*)"""
import csv
import glob
import json
import os
import re
import sys

tag, rnd = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("nvllm::", "")


# 1. kernel stats (rocprofv3 --kernel-trace --stats): copy the summary as it is
ks = glob.glob(os.path.join(root, "gpurun_out", f"prof_{tag}", "**", "*kernel_stats.csv"), recursive=True)
if ks:
    dst = os.path.join(out, f"r{rnd}_bench_decode_kernel_stats.csv")
    with open(ks[0]) as f, open(dst, "w") as g:
        g.write(f.read())
    print("wrote", dst)

# 2. FETCH_SIZE per kernel (separate --pmc pass)
cc = glob.glob(os.path.join(root, "gpurun_out", f"pmc_{tag}", "**", "*counter_collection.csv"), recursive=True)
if cc:
    acc = {}
    with open(cc[0]) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != "FETCH_SIZE":
                continue
            k = short(row["Kernel_Name"])
            a = acc.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    lines = [l for l in open(os.path.join(root, "gpurun_out", f"pmc_{tag}.log")).read().splitlines() if l.startswith('{"metric"')]
    bench = json.loads(lines[-1])
    cmd = "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --skip-tp-leg --profile-steps 0"
    dst = os.path.join(out, f"r{rnd}_pmc_fetch_size.csv")
    with open(dst, "w") as g:
        g.write(f"# {cmd} (MI355X)\n# FETCH_SIZE is in KB per dispatch; on gfx950 it reads 1/2 of a wide coalesced stream "
                "(MI355X_MICROARCH.md, HBM): corrected_MB = 2*KB/1024\n")
        g.write("kernel,dispatches,mean_FETCH_SIZE_KB,corrected_MB_per_launch\n")
        for k, (n, tot) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            g.write(f"\"{k}\",{n},{tot / n:.1f},{2 * tot / n / 1024:.2f}\n")
    print("wrote", dst)
    dom = bench["roofline"]["kernel"]
    n, tot = acc[dom]
    js = {"_comment": cmd + " on MI355X; FETCH_SIZE KB x 2 (gfx950 wide-stream correction, MI355X_MICROARCH.md HBM section) x 1024; "
          "algorithmic = roofline.bytes_per_launch of that same run. Full table: " + os.path.basename(dst),
          dom: {"fetch_bytes_per_launch": int(round(2 * tot / n * 1024)), "algorithmic_bytes_per_launch": int(bench["roofline"]["bytes_per_launch"])}}
    dst = os.path.join(out, f"r{rnd}_pmc_fetch_size.json")
    json.dump(js, open(dst, "w"), indent=1)
    print("wrote", dst, js[dom])

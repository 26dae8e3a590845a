#!/bin/bash
# Run on the GPU box (gpurun): the default bench line, a rocprofv3 kernel-trace summary of the same command, a separate
# FETCH_SIZE counter pass and the prefill-only MFMA pass.  Raw output -> gpurun_out/; tools/summarize_prof.py turns it into
# profiles/rNN_*.   usage: bash tools/profile_round.sh TAG
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-final}
cd /tmp && export TMPDIR=/tmp
# the driver's own command (round-end bench: --gpus 1 --steps 20 --warmup 5); the profiled passes run the SAME decode steps
# (warm-up 5 + timed 20 + the per-kernel HIP-event pass), so one bytes-per-launch figure holds for the line, the kernel
# stats and the counter pass
D="--gpus 1 --steps 20 --warmup 5"
timeout -k 10 700 python3 $R/bench.py $D > $R/gpurun_out/bench_$TAG.log 2>&1 || exit 1
tail -1 $R/gpurun_out/bench_$TAG.log | cut -c1-200
B="python3 $R/bench.py --no-cpu-baseline --skip-tp-leg --profile-meta --no-live-traffic"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o r -- $B $D > $R/gpurun_out/prof_$TAG.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG -o r -- $B $D > $R/gpurun_out/pmc_$TAG.log 2>&1 || exit 3
# prefill (MFMA side): kernel stats of a prefill-only run, then matrix-core counters in their own pass
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_prefill_$TAG -o r -- $B --prefill-only > $R/gpurun_out/prof_prefill_$TAG.log 2>&1 || exit 4
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_prefill_$TAG -o r -- $B --prefill-only --prefill-reps 1 > $R/gpurun_out/pmc_prefill_$TAG.log 2>&1 || echo "prefill pmc pass failed (counter names?)"
ls -R $R/gpurun_out/prof_$TAG $R/gpurun_out/pmc_$TAG $R/gpurun_out/pmc_prefill_$TAG 2>/dev/null | head -30

#!/bin/bash
# Run on the GPU box (gpurun): the default bench line, a rocprofv3 kernel-trace summary of the same command and a
# separate FETCH_SIZE counter pass.  Raw output -> gpurun_out/; tools/summarize_prof.py turns it into profiles/*.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-final}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 python3 $R/bench.py > $R/gpurun_out/bench_$TAG.log 2>&1 || exit 1
tail -1 $R/gpurun_out/bench_$TAG.log | cut -c1-200
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o r -- \
    python3 $R/bench.py --no-cpu-baseline --skip-tp-leg --profile-steps 0 > $R/gpurun_out/prof_$TAG.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG -o r -- \
    python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --skip-tp-leg --profile-steps 0 > $R/gpurun_out/pmc_$TAG.log 2>&1 || exit 3
ls -R $R/gpurun_out/prof_$TAG $R/gpurun_out/pmc_$TAG | head -20

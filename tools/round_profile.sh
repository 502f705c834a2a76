#!/bin/bash
# usage: tools/round_profile.sh <tag>: GPU tests, default bench, rocprof kernel stats and PMC traffic for profiles/
set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread 2>&1 | tail -3 || exit 1
bash tools/pmc_traffic.sh $1 || exit 1
mkdir -p profiles && cp gpurun_out/$1_pmc_traffic.json profiles/$1_pmc_traffic.json
bash tools/profile_step.sh $1 || exit 1
timeout -k 10 900 python bench.py > gpurun_out/$1_bench_default.json 2> gpurun_out/$1_bench_default.err || { tail -5 gpurun_out/$1_bench_default.err; exit 1; }
tail -1 gpurun_out/$1_bench_default.json

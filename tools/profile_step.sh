#!/bin/bash
# usage: tools/profile_step.sh <tag>  -> rocprofv3 kernel stats of a train-only bench run, copied to gpurun_out/<tag>_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/profs && mkdir -p gpurun_out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profs -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --eval-calls 0 > gpurun_out/$1_bench.json 2> gpurun_out/$1_bench.err
cp /tmp/profs/*/*kernel_stats.csv gpurun_out/$1_kernel_stats.csv
tail -1 gpurun_out/$1_bench.json | cut -c1-400

#!/bin/bash
# usage: tools/trace_eval_call.sh [B] [shot]: per-kernel totals of ONE few-shot prediction call (rocprofv3 kernel stats
# of tools/probe_eval_kernels.py, 20 calls) -> gpurun_out/eval_call_B<B>_<shot>shot.txt
B=${1:-4}; S=${2:-1}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof9
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof9 -- python tools/probe_eval_kernels.py $B $S > /tmp/evk.log 2>&1 || { tail -5 /tmp/evk.log; exit 1; }
F=$(find /tmp/prof9 -name '*kernel_stats.csv' | head -1)
python - "$F" > gpurun_out/eval_call_B${B}_${S}shot.txt <<EOF
import csv,sys
rows=list(csv.reader(open(sys.argv[1])))[1:]
tot=sum(float(r[2]) for r in rows)/20/1e3
print("total us per call %.1f, launches per call %.1f"%(tot, sum(int(r[1]) for r in rows)/20))
for r in rows[:40]:
    print("%-72s n/call %5.1f avg %7.1f us  per call %7.1f us"%(r[0][:72], int(r[1])/20, float(r[3])/1e3, float(r[2])/20/1e3))
EOF
cat gpurun_out/eval_call_B${B}_${S}shot.txt

#!/bin/bash
# usage: tools/trace_eval.sh B -> per-kernel totals of the prediction calls
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof6
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof6 -- python tools/probe_eval.py $1 > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/prof6/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
calls=46.0
tot=sum(float(r['TotalDurationNs']) for r in rows)/1e3/calls
print('total kernel us per call', tot)
for r in rows[:22]:
    print("%8.1f us/call  n/call %5.1f avg %7.1f us  %s" % (float(r['TotalDurationNs'])/1e3/calls, int(r['Calls'])/calls, float(r['AverageNs'])/1e3, r['Name'][:90]))
PY

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof8
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof8 -- python tools/probe_feature.py > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/prof8/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)/1e6
print('total kernel ms', tot)
for r in rows[:25]:
    print("%8.2f ms  n %5d avg %8.1f us  %s" % (float(r['TotalDurationNs'])/1e6, int(r['Calls']), float(r['AverageNs'])/1e3, r['Name'][:100]))
PY

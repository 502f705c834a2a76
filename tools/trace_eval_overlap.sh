#!/bin/bash
# usage: tools/trace_eval_overlap.sh [streams] -> GPU busy fraction and mean kernel concurrency of the pipelined evaluation
N=${1:-8}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof10
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof10 -- python tools/probe_eval_overlap.py $N > /tmp/evo.log 2>&1 || { tail -5 /tmp/evo.log; exit 1; }
tail -1 /tmp/evo.log
F=$(find /tmp/prof10 -name '*kernel_trace.csv' | head -1)
python - "$F" <<EOF
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
# the timed region: after the largest idle gap
last_end, cut = ev[0][1], 0
for i in range(1, len(ev)):                     # the LAST idle gap of more than 30 ms (the probe's sleep)
    if ev[i][0] - last_end > 30e6:
        cut = i
    last_end = max(last_end, ev[i][1])
ev = ev[cut:]
t0, t1 = ev[0][0], max(e[1] for e in ev)
busy, cur_s, cur_e = 0, ev[0][0], ev[0][1]
for s, e, _ in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _ in ev)
print('kernels %d  wall %.1f us  busy %.1f %%  sum of durations / wall = %.2f' % (len(ev), (t1 - t0) / 1e3, 100.0 * busy / (t1 - t0), tot / (t1 - t0)))
from collections import defaultdict
d = defaultdict(lambda: [0, 0])
for s, e, k in ev:
    d[k[:60]][0] += e - s; d[k[:60]][1] += 1
for k, (t, n) in sorted(d.items(), key=lambda kv: -kv[1][0])[:14]:
    print('%-60s n %5d  avg %7.1f us  share of wall %.2f' % (k, n, t / n / 1e3, t / (t1 - t0)))
EOF

#!/bin/bash
# usage: tools/trace_step.sh <tag> -> gpurun_out/<tag>_step_timeline.txt: every kernel of ONE training step (between two
# adam_kernel launches) with its duration and the idle gap in front of it (rocprofv3 kernel trace of tools/probe_host.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof8 && mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof8 -- python tools/probe_host.py > /tmp/step.log 2>&1 || { tail -5 /tmp/step.log; exit 1; }
tail -1 /tmp/step.log
python - "$1" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob("/tmp/prof8/*/*kernel_trace.csv")[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
a, b = ends[-3], ends[-2]
seg = rows[a + 1:b + 1]
busy = sum(e - s for s, e, _ in seg)
span = seg[-1][1] - rows[a][1]
with open('gpurun_out/%s_step_timeline.txt' % tag, 'w') as out:
    out.write("kernels %d  busy %.3f ms  span %.3f ms  idle %.3f ms\n" % (len(seg), busy / 1e6, span / 1e6, (span - busy) / 1e6))
    prev = rows[a][1]
    t0 = prev
    for s, e, n in seg:
        n = n.replace('void ', '').replace('cmvae::', '')
        out.write("%9.1f  gap %7.1f  dur %8.1f  %s\n" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, n[:110]))
        prev = e
    agg = collections.defaultdict(lambda: [0, 0.0])
    for s, e, n in seg:
        k = n.replace('void ', '').replace('cmvae::', '').split('(')[0][:70]
        agg[k][0] += 1; agg[k][1] += (e - s) / 1e3
    out.write("\nper kernel (one step):\n")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.write("%8.1f us  n %3d  %s\n" % (t, c, k))
print(open('gpurun_out/%s_step_timeline.txt' % tag).readline())
PY

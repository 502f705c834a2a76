#!/bin/bash
# usage: tools/trace_gaps.sh -> GPU busy vs idle time inside one training step (kernel trace of tools/probe_host.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof7
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof7 -- python tools/probe_host.py > /tmp/gaps.log 2>&1
tail -1 /tmp/gaps.log
python - <<PY
import csv,glob
f=glob.glob("/tmp/prof7/*/*kernel_trace.csv")[0]
rows=sorted(((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in csv.DictReader(open(f))))
# steps are delimited by adam_kernel
ends=[i for i,r in enumerate(rows) if "adam_kernel" in r[2]]
a,b=ends[-3],ends[-2]
seg=rows[a+1:b+1]
busy=sum(e-s for s,e,_ in seg)
span=seg[-1][1]-seg[0][0]
gaps=[(seg[i+1][0]-seg[i][1], seg[i][2][:50], seg[i+1][2][:50]) for i in range(len(seg)-1)]
print("kernels %d  busy %.2f ms  span %.2f ms  idle %.2f ms" % (len(seg), busy/1e6, span/1e6, (span-busy)/1e6))
import collections
pos=[g for g in gaps if g[0]>0]
print("gaps: n %d  median %.1f us  sum>20us %.2f ms" % (len(pos), sorted(g[0] for g in pos)[len(pos)//2]/1e3, sum(g[0] for g in pos if g[0]>20000)/1e6))
for g in sorted(gaps,reverse=True)[:12]:
    print("%8.1f us  after %-50s before %s" % (g[0]/1e3, g[1], g[2]))
PY

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof4
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof4 -- python tools/probe_em.py > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/prof4/*/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows=[r for r in rows if "cmvae" in r["Kernel_Name"]]
half=len(rows)//2
tot={}
for r in rows[half:]:
    k=r["Kernel_Name"].split("(")[0].replace("void cmvae::","").replace("cmvae::","")[:28]
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    tot.setdefault(k,[]).append(d)
for k,v in tot.items():
    print("%-30s n %3d total %9.1f us  max %8.1f  min %8.1f" % (k,len(v),sum(v),max(v),min(v)))
print("sum", sum(sum(v) for v in tot.values()))
PY

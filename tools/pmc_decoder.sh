#!/bin/bash
# usage: tools/pmc_decoder.sh "<counters>"   -> per-kernel averages of the counters over the decoder probe
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof5
timeout -s KILL 120 rocprofv3 --pmc $1 --output-format csv -d /tmp/prof5 -- python tools/probe_decoder.py > /tmp/pmc.log 2>&1
tail -3 /tmp/pmc.log
python - <<PY
import csv,glob,collections
fs=glob.glob("/tmp/prof5/*/*counter_collection.csv")
print(fs)
rows=list(csv.DictReader(open(fs[0])))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r["Kernel_Name"].split("(")[0].replace("void cmvae::","").replace("cmvae::","")[:24]
    if "conv_gemm" in k or "wgrad_kernel" in k or "dec_out" in k:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in agg.items():
    print(k, {c: "%.3g" % max(v) for c,v in d.items()})
PY

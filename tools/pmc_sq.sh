#!/bin/bash
# usage: tools/pmc_sq.sh <tag>: SQ counters (two rocprofv3 --pmc passes, 8 SQ counters each) of every kernel of one
# training step (tools/probe_layers.py) -> gpurun_out/<tag>_sq.txt: per kernel symbol, the largest launch's counters
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
P1="SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1)); rm -rf /tmp/sq_$i
  timeout -s KILL 200 rocprofv3 --pmc $P --output-format csv -d /tmp/sq_$i -- python tools/probe_layers.py > /tmp/sq_$i.log 2>&1 || { echo "pass $i failed"; tail -5 /tmp/sq_$i.log; exit 1; }
  echo "pass $i done"
done
python - "$1" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
best = {}
for i in (1, 2):
    f = glob.glob('/tmp/sq_%d/*/*counter_collection.csv' % i)[0]
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'].split('(')[0].replace('void ', '').replace('cmvae::', ''), r['Dispatch_Id'])
        per[k][r['Counter_Name']] = float(r['Counter_Value'])
    # largest launch of each kernel symbol by its first counter
    first = 'SQ_BUSY_CU_CYCLES' if i == 1 else 'SQ_WAVE_CYCLES'
    for (name, _), d in per.items():
        if name not in best or i not in best[name] or d.get(first, 0) > best[name][i].get(first, 0):
            best.setdefault(name, {})[i] = d
with open('gpurun_out/%s_sq.txt' % tag, 'w') as out:
    for name, d in sorted(best.items(), key=lambda kv: -kv[1].get(1, {}).get('SQ_BUSY_CU_CYCLES', 0))[:14]:
        out.write(name[:90] + '\n')
        for i in (1, 2):
            out.write('   ' + '  '.join('%s=%.4g' % (k.replace('SQ_', ''), v) for k, v in d.get(i, {}).items()) + '\n')
print(open('gpurun_out/%s_sq.txt' % tag).read())
PY

#!/bin/bash
# usage: tools/round_profile.sh <tag> [config]: PMC traffic + rocprof kernel stats + default bench line of one bench config
# -> gpurun_out/<tag>_<config>_{pmc_traffic.json,kernel_stats.csv,bench_profiled.json,bench_default.json}
set -o pipefail
CFG=${2:-o20}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
T=$1_$CFG
# --- PMC traffic (two passes, one counter each)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout -s KILL 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --eval-calls 2 --eval-streams 1 --graph-step off > /tmp/pmc_$c.log 2>&1 || { echo "pass $c failed"; tail -5 /tmp/pmc_$c.log; exit 1; }
  echo "pass $c done"
done
python - "$T" <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
out = collections.defaultdict(dict)
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob('/tmp/pmc_%s/*/*counter_collection.csv' % c)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != c:
            continue
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('cmvae::', '')
        agg[k].append(float(r['Counter_Value']))
    for k, v in agg.items():
        out[k][c + '_raw_mean'] = sum(v) / len(v)
        out[k]['launches'] = len(v)
        out[k][c + '_raw_max'] = max(v)
res = {'note': 'rocprofv3 --pmc, one counter per pass; raw counter values are in KB (1024 B); gfx950: FETCH_SIZE '
               'counts 128-B requests as 64 B, so read bytes = 2 * FETCH_SIZE * 1024 (MI355X_MICROARCH.md, HBM section)',
       'kernels': {}}
for k, d in out.items():
    if 'FETCH_SIZE_raw_mean' in d and 'WRITE_SIZE_raw_mean' in d:
        rd = 2.0 * d['FETCH_SIZE_raw_mean'] * 1024
        wr = d['WRITE_SIZE_raw_mean'] * 1024
        res['kernels'][k] = {'launches': d['launches'], 'read_bytes_per_launch': rd, 'write_bytes_per_launch': wr,
                             'hbm_bytes_per_launch': rd + wr,
                             'max_launch_read_bytes': 2.0 * d['FETCH_SIZE_raw_max'] * 1024,
                             'max_launch_write_bytes': d['WRITE_SIZE_raw_max'] * 1024}
json.dump(res, open('gpurun_out/%s_pmc_traffic.json' % tag, 'w'), indent=1)
for k, d in sorted(res['kernels'].items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches'])[:10]:
    print('%-44s n %4d  read %9.1f MB  write %9.1f MB  (max launch: %8.1f / %8.1f MB)' % (
        k[:44], d['launches'], d['read_bytes_per_launch'] / 1e6, d['write_bytes_per_launch'] / 1e6,
        d['max_launch_read_bytes'] / 1e6, d['max_launch_write_bytes'] / 1e6))
PY
# --- kernel stats
rm -rf /tmp/profs
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profs -- python bench.py --config $CFG --steps 20 --warmup 3 --no-cpu-baseline --eval-calls 4 --eval-streams 1 --graph-step off > gpurun_out/${T}_bench_profiled.json 2> gpurun_out/${T}_bench_profiled.err || { tail -5 gpurun_out/${T}_bench_profiled.err; exit 1; }
cp /tmp/profs/*/*kernel_stats.csv gpurun_out/${T}_kernel_stats.csv
# --- the default bench line (reads the traffic file committed under profiles/, so copy first)
mkdir -p profiles && cp gpurun_out/${T}_pmc_traffic.json profiles/${T}_pmc_traffic.json
timeout -k 10 900 python bench.py --config $CFG > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err || { tail -5 gpurun_out/${T}_bench_default.err; exit 1; }
tail -1 gpurun_out/${T}_bench_default.json | cut -c1-600

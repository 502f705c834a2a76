#!/bin/bash
# usage: tools/ab_side_stream.sh [bench args]: bench.py with the prior's EM on the second stream (shipped) and in place
# (model.em_side_stream = False), alternately on ONE box
set -o pipefail
for rep in 1 2; do
  for side in True False; do
    timeout -k 10 200 python -c "
import sys, runpy
import cmvae_amd.model as m
m._GMVAEBase.em_side_stream = $side
sys.argv = ['bench.py', '--no-cpu-baseline', '--eval-calls', '2', '--long-steps', '0'] + sys.argv[1:]
runpy.run_path('bench.py', run_name='__main__')" "$@" > /tmp/ab.log 2>&1 || { tail -5 /tmp/ab.log; exit 1; }
    python - "$side" <<'PY'
import json, sys
for l in open('/tmp/ab.log'):
    if l.startswith('{'):
        d = json.loads(l)
        print('em_side_stream=%-5s ms/step %.3f' % (sys.argv[1], d['ms_per_step']))
PY
  done
done

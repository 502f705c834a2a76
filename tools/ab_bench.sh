#!/bin/bash
# usage: tools/ab_bench.sh <variant lib> [bench args]: the shipped library and a variant build alternately on ONE box
# (boxes differ by several per cent, and so do runs after a hot kernel: only same-box A/B/A/B decides)
set -o pipefail
V=$1; shift
for rep in 1 2; do
  for lib in "" "$V"; do
    CMVAE_HIP_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --eval-calls 2 --long-steps 0 "$@" > /tmp/ab.log 2>&1 || { tail -3 /tmp/ab.log; exit 1; }
    python - "$lib" <<'PY'
import json, sys
for l in open('/tmp/ab.log'):
    if l.startswith('{'):
        d = json.loads(l)
        print('%-40s ms/step %.3f' % (sys.argv[1] or 'shipped', d['ms_per_step']))
PY
  done
done

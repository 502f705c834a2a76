#!/bin/bash
# Run GPU steps in order; a step that ends with a test failure (rc 1) lets the next one run, anything else
# (fault, abort, time limit) ends the script there.  usage: tools/gpu_steps.sh "cmd1" "cmd2" ...
for cmd in "$@"; do
    echo "== $cmd"
    bash -c "$cmd"
    rc=$?
    echo "== rc $rc"
    if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
done
exit 0

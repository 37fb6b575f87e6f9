#!/usr/bin/env bash
# GPU box: run the reference's example programs built against ungar_amd (oracle/_ref/*_example).
set -uo pipefail
mkdir -p gpurun_out /tmp/ex && cd /tmp/ex
for n in "$@"; do
  echo "== $n"; ( time timeout 1200 $GRAFT_REPO_ROOT/oracle/_ref/${n}_example > $GRAFT_REPO_ROOT/gpurun_out/example_$n.log 2> $GRAFT_REPO_ROOT/gpurun_out/example_$n.err ) 2>&1 | grep real
  echo "rc=$?"; head -3 $GRAFT_REPO_ROOT/gpurun_out/example_$n.log; tail -4 $GRAFT_REPO_ROOT/gpurun_out/example_$n.log; tail -5 $GRAFT_REPO_ROOT/gpurun_out/example_$n.err
done

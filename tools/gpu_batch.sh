#!/bin/bash
# Runs a list of GPU steps on the box; a step that times out or is killed ends the batch (no further GPU work is started).
# usage: tools/gpu_batch.sh <tag> "<timeout_s>|<name>|<command>" ...
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
for spec in "$@"; do
  IFS='|' read -r tmo name cmd <<< "$spec"
  echo "=== [$name] $cmd" | tee -a $out/batch.log
  start=$(date +%s)
  timeout -k 10 $tmo bash -c "$cmd" > $out/$name.log 2>&1
  rc=$?
  echo "=== [$name] rc=$rc in $(( $(date +%s) - start )) s" | tee -a $out/batch.log
  tail -n 6 $out/$name.log | tee -a $out/batch.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 134 ] || [ $rc -eq 139 ]; then
    echo "=== step $name timed out / was killed: stopping the batch" | tee -a $out/batch.log
    exit 1
  fi
done
exit 0

#!/bin/bash
# Runs __graft_entry__.smoke() the way the round-end driver does (block-buffered stdout into a file, `python3 -c`), N times.
# usage: tools/driver_smoke.sh <tag> <count>
tag=$1; n=${2:-3}
mkdir -p gpurun_out/smoke
for i in $(seq 1 $n); do
  PYTHONFAULTHANDLER=1 python3 -c 'import sys; sys.path.insert(0, "."); import __graft_entry__ as e
f = getattr(e, "smoke", None)
if f is None:
    print("__SMOKE_SKIP__ (no smoke() in __graft_entry__)"); sys.exit(0)
f(); print("__SMOKE_OK__")' > gpurun_out/smoke/${tag}_$i.log 2>&1
  echo "rc=$?" >> gpurun_out/smoke/${tag}_$i.log
  tail -n 3 gpurun_out/smoke/${tag}_$i.log
done

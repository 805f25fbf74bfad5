#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r03sab
for rep in 1 2; do for v in 00 10 01 11; do
  if [ $v = 11 ]; then unset PBSO_LIB; else export PBSO_LIB=$PWD/openpbso_amd/libpbso_v$v.so; fi
  for cfg in "c2 --objects 1 --modes 512" "c3 --objects 64 --modes 256 --scenario listener"; do
    set -- $cfg; name=$1; shift
    timeout 600 python bench.py --no-cpu-baseline --no-second-form --no-parity --steps 60 --warmup 3 "$@" > gpurun_out/r03sab/$name.$v.$rep.json 2>/dev/null
    python - gpurun_out/r03sab/$name.$v.$rep.json "$name jump=${v:0:1} prefetch=${v:1:1} rep$rep" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print("%-34s rt=%8.1f kernel=%.4f" % (sys.argv[2], d["realtime_x"], d["roofline"]["kernel_ms"]))
PY
  done
done; done

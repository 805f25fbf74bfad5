#!/bin/bash
# the audio stores of the block kernel: plain / nontemporal / write-through -- does the gap between two banks shrink?
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-second-form --no-strong-share "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['realtime_x']), 'x', round(d['ms_per_step'],4), 'ms kernel', round(d['roofline']['kernel_ms'],4), 'err', d.get('max_err'))"; }
for rep in 1 2; do
for v in plain nt1 nt2; do
[ $v = plain ] && unset PBSO_LIB || export PBSO_LIB=$PWD/openpbso_amd/csrc/libvariant_$v.so
echo "== $v: default $(run)"
echo "== $v: 128x512 $(run --objects 128)"
echo "== $v: 512x512 $(run --objects 512)"
done
done

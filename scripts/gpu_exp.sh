#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for e in 0 2 4 8 6 14; do
  for rep in 1 2; do
  PBSO_EXP=$e timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('EXP=$e kernel_ms=%.3f step=%.2f'%(d['roofline']['kernel_ms'], d['ms_per_step']))"
  done
done
PBSO_EXP=0 timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-qnorm 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('noqn kernel_ms=%.3f'%(d['roofline']['kernel_ms']))"
PBSO_EXP=14 timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-qnorm 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('noqn EXP=14 kernel_ms=%.3f'%(d['roofline']['kernel_ms']))"

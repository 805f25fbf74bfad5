#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for pr in auto 0 3; do for q in off sample; do
  env $( [ $pr = auto ] && echo "A=1" || echo "PBSO_K2_PRIO=$pr" ) timeout 600 python bench.py --no-cpu-baseline --no-parity --objects 8 --modes 4096 --scenario scraping --qnorm $q --steps 40 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); print('k2prio=$pr qnorm=$q', 'rt=%.1f ms/step=%.4f kernel=%.4f'%(d['realtime_x'],d['ms_per_step'],d['roofline']['kernel_ms']))"
done; done

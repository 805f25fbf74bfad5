#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for tw in 0 4 8; do for q in off sample; do
  if [ $tw = 0 ]; then unset PBSO_BLOCK_TEAM_WAVES; else export PBSO_BLOCK_TEAM_WAVES=$tw; fi
  timeout 600 python bench.py --no-cpu-baseline --no-parity --objects 8 --modes 4096 --scenario scraping --qnorm $q --steps 40 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); print('team waves $tw qnorm $q', 'rt=%.1f ms/step=%.4f kernel=%.4f W=%d'%(d['realtime_x'],d['ms_per_step'],d['roofline']['kernel_ms'],d['config']['waves_per_object']))"
done; done

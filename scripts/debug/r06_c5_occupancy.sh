#!/bin/bash
# configs[4] without qnorm rows, cut in time (bank_kernel=1): does the preparation overlap the bank when the bank leaves a wave slot per SIMD free?
# PBSO_TC_LDS_PAD pads the bank's workgroups (teams of four one-mode-per-lane waves, 48 KB: three per CU) so that fewer fit a CU.
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg --steps 40 --warmup 2 --buffers 86 --objects 8 --modes 4096 --scenario scraping "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['realtime_x'],1), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), d['parity']['pass'])"; }
echo "policy, qnorm off (five-role teams):          $(run --qnorm off)"
for pad in 0 16384 32768; do
  echo "cut in time, qnorm off, pad $pad:           $(PBSO_ENGINE_OPTS=bank_kernel=1 PBSO_TC_LDS_PAD=$pad run --qnorm off)"
done
echo "policy, qnorm rows (cut in time):             $(run)"
for pad in 16384 32768; do
  echo "qnorm rows, pad $pad:                       $(PBSO_TC_LDS_PAD=$pad run)"
done

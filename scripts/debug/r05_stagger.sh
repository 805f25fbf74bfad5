#!/bin/bash
# K1b, the teams of a CU started half a buffer apart (s_sleep 127 = 8 K cycles, once or twice, on rank bit 1 or 2): headline kernel time
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['realtime_x'],1), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), d['parity']['pass'])"; }
echo "product: $(run)"
for v in 1_1 1_2 2_1 2_2; do echo "stagger bit_n $v: $(PBSO_LIB=$GRAFT_REPO_ROOT/build/variants/libstag_$v.so run)"; done
echo "product: $(run)"

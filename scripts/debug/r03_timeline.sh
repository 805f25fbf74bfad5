#!/bin/bash
# PBSO_TIMELINE=1: device events and host submission times of every launch; prints the launches around the largest gap of each run
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/tl
ARGS="${ARGS:---objects 64 --modes 256 --scenario listener --steps 40 --warmup 2}"
for i in 1 2 3 4 5 6; do
  PBSO_TIMELINE=1 PBSO_TIMING_EVERY=1 python bench.py --no-cpu-baseline --no-second-form --time-every 1 $ARGS 2>gpurun_out/tl/$i.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('run: rt %.0f ms/step %.3f kernel %.3f' % (d['realtime_x'], d['ms_per_step'], d['roofline']['kernel_ms']))"
  python - gpurun_out/tl/$i.txt <<'PY'
import re, sys
rows = [l for l in open(sys.argv[1]) if "pbso timeline" in l]
ends = [float(re.search(r"bank starts ([\d.]+)", l).group(1)) for l in rows]
gaps = [(ends[k + 1] - ends[k], k) for k in range(34, len(ends) - 1)]      # (timed region only: after settle + warm-up)
g, k = max(gaps)
print("  largest bank-to-bank distance %.3f ms at launch %d:" % (g, k + 1))
for l in rows[k:k + 3]: print("   ", l.strip()[15:])
PY
done

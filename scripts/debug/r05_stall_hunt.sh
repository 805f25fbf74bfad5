#!/bin/bash
# which host-side stage of a step blocks for milliseconds in the outlier runs of the 128 x 512 x 860 share? (PBSO_TIMELINE=1, every launch timed)
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3 4 5 6 7 8 9 10; do
  PBSO_TIMELINE=1 python bench.py --no-cpu-baseline --no-second-form --no-parity --no-strong-share --no-one-second-leg --objects 128 --steps 16 --warmup 3 --time-every 1 > /tmp/sh.json 2> /tmp/sh.err
  ms=$(python -c "import json; print(round(json.loads(open('/tmp/sh.json').read().strip().splitlines()[-1])['ms_per_step'],3))")
  echo "run $i: ms_per_step $ms; $(python scripts/debug/r03_stalls.py /tmp/sh.err | tail -1)"
  python scripts/debug/r03_stalls.py /tmp/sh.err | grep "took" | head -8
done

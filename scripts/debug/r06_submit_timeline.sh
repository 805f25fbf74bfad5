#!/bin/bash
# where the 64 x 256 listener step's time goes with / without the second submitting thread: the engine's timeline (device events and
# host stamps of every launch) of the last steps of a short run
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06st
for st in 0 1; do
PBSO_TIMELINE=1 PBSO_TIMING_EVERY=1 python bench.py --no-cpu-baseline --no-second-form --no-strong-share --no-one-second-leg --steps 40 --warmup 5 --buffers 86 \
   --objects 64 --modes 256 --scenario listener --submit-thread $st > gpurun_out/r06st/timeline_c3_st$st.json 2> gpurun_out/r06st/timeline_c3_st$st.err
echo "== submit_thread=$st"; grep "pbso timeline" gpurun_out/r06st/timeline_c3_st$st.err | tail -12 | cut -c1-400
done

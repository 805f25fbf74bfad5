#!/bin/bash
# the real-time step under a sustained contact: profile rows + combine in one launch (and what it does to the device pipeline)
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_submit_thread.py tests/test_facade_cpp.py -q -x 2>&1 | tail -3
python scripts/latency.py 2>/dev/null | grep -E "sustained" | cut -c1-330

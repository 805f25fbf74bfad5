#!/bin/bash
# pbso_compute_transfer_batch (the HUD sphere: 10 242 positions of one 1024-mode object, 16 x 16-cell maps): the shared-geometry lookup
# against the per-(mode, 1024 positions) kernel (PBSO_FFAT_SHARED=0) -- wall time of the call (the 84 MB of weights cross PCIe either way)
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "transfer_batch or ffat_lookup_bit_exact" 2>&1 | tail -2
PBSO_FFAT_SHARED=0 timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "transfer_batch or ffat_lookup_bit_exact" 2>&1 | tail -2
cat > /tmp/tb.py <<PY
import sys, time
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import numpy as np
from openpbso_amd import Engine, synth
n_modes, n_pos = 1024, 10242
lam = synth.eigenvalues(n_modes, 3)
maps = synth.ffat_maps(lam, 3, dim=16)
rng = np.random.default_rng(1)
d = rng.standard_normal((n_pos, 3)); pos = 0.5 * d / np.linalg.norm(d, axis=1, keepdims=True)
with Engine() as eng:
    eng.add_object(lam, synth.RHO, synth.ALPHA, synth.BETA); eng.set_ffat_maps(0, maps); eng.finalize()
    ts = []
    for i in range(8):
        t0 = time.perf_counter(); ok, got = eng.compute_transfer_batch(0, pos, n_modes); ts.append(time.perf_counter() - t0)
    print("compute_transfer_batch 10242 x 1024: median %.2f ms, min %.2f ms; checksum %.17g" % (np.median(ts[2:]) * 1e3, min(ts) * 1e3, float(got.sum())))
PY
for v in 1 0; do echo "PBSO_FFAT_SHARED=$v: $(PBSO_FFAT_SHARED=$v python /tmp/tb.py 2>/dev/null | tail -1)"; done
export TMPDIR=/tmp
for v in 1 0; do rm -rf /tmp/tbp; (cd /tmp && PBSO_FFAT_SHARED=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tbp -- python3 /tmp/tb.py > /dev/null 2>&1); f=$(find /tmp/tbp -name "*kernel_stats.csv" | head -1); echo "PBSO_FFAT_SHARED=$v kernels:"; python3 -c "
import csv,sys
for r in csv.reader(open(sys.argv[1])):
    if 'ffat' in r[0]: print('   ', r[0].split('(')[0], 'calls', r[1], 'average ns', r[3])
" "$f"; done

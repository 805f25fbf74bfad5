#!/usr/bin/env python3
"""the scan kernel with NOTHING beside it: step, wait for the device, step ... (bench.py pipelines its steps, so the scan of
launch k + 1 runs beside the bank of launch k and is stretched -- the unlabelled second block of profiles/r04_scan_kernel_stages.txt).
usage (under rocprofv3 --kernel-trace --stats): r05_scan_isolated.py <objects> <modes>"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openpbso_amd import Engine, ForceMessage, synth, capi   # noqa: E402
n_obj, M, nb = int(sys.argv[1]), int(sys.argv[2]), 86
eng = Engine(qnorm=capi.QNORM_ALL, form=capi.FORM_BLOCK, time_chunks=11, scan_kernel=int(os.environ.get("PBSO_SCAN_KERNEL", "1")))
for i in range(n_obj):
    s = synth.seed_for(4, i)
    eng.add_object(synth.eigenvalues(M, s), synth.RHO, synth.ALPHA, synth.BETA, mode_shapes=synth.mode_shapes(M, s))
eng.finalize()
steps = 30
for i in range(n_obj):
    eng.set_use_transfer(i, False)
    s = synth.seed_for(4, i)
    hits, vns = synth.poisson_hits(steps * nb, s), synth.unit_normals(steps * nb, s)
    for b in np.nonzero(hits >= 0)[0]:
        eng.enqueue_force(i, ForceMessage(vid=int(hits[b]), vn=vns[b]), int(b))
for _ in range(steps):
    eng.step(nb)
    eng.sync()
print("time-chunked launches", eng.info()["total_time_chunk_launches"])

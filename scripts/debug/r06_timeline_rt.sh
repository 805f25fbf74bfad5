#!/bin/bash
# kernel timeline of the real-time step with a sustained contact on (one object, one buffer per step, a face hit per buffer): what the
# ~55 us of its device pipeline are, with the short launches fused (policy) and apart (fuse_short_launches = -1)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for fuse in 0 -1; do
rm -rf /tmp/tlrt; cat > /tmp/rt.py <<PY
import sys, time
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import numpy as np
from openpbso_amd import Engine, ForceMessage, synth, capi
n_modes = 512
eng = Engine(qnorm=capi.QNORM_ALL, fuse_short_launches=$fuse)
eng.add_object(synth.eigenvalues(n_modes, 6), synth.RHO, synth.ALPHA, synth.BETA, mode_shapes=synth.mode_shapes(n_modes, 6))
eng.finalize(); eng.set_use_transfer(0, False)
eng.enqueue_force(0, ForceMessage(forceType=capi.AUTOREGRESSIVE_FORCE, sustainedForceStart=True), 0)
vns = synth.unit_normals(200, 6)
for i in range(200):
    if i > 0: eng.enqueue_force(0, ForceMessage(vids=[0, 1, 2], coords=[0.2, 0.3, 0.5], vn=vns[i], forceType=capi.AUTOREGRESSIVE_FORCE), i)
    eng.step(1); eng.sync()
eng.close()
PY
(cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tlrt -- python3 /tmp/rt.py > /dev/null 2>&1)
echo "== fuse_short_launches = $fuse"
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("/tmp/tlrt/**/*kernel_trace.csv", recursive=True))[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:40]) for r in csv.DictReader(open(f))]
for g in glob.glob("/tmp/tlrt/**/*memory_copy_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")) for r in csv.DictReader(open(g))]
rows.sort()
banks = [i for i, r in enumerate(rows) if "iir_block" in r[2] or "iir_pipe" in r[2] or "iir_bank" in r[2]]
mid = banks[150]
lo = banks[149] + 1
t0 = rows[lo][0]
for s, e, name in rows[lo:mid + 2]:
    print(f"{(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f} us  (+{(e - s) / 1e3:6.1f})  {name}")
PY
done

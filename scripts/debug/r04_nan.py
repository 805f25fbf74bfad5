import numpy as np, sys
sys.path.insert(0, '.')
from openpbso_amd import capi
from tests.test_gpu_time_chunks import _every_kind_scene
from tests.scenarios import run_engine, run_oracle, rel_errors
nb = 16
objs, evs = _every_kind_scene(nb)
want = run_oracle(objs, evs, nb)
for cb in (1, 3, 16):
    for direct in (0, -1):
        got = run_engine(objs, evs, nb, time_chunks=cb, direct_hits=direct)
        a = got["audio"].reshape(len(objs), nb, 513)
        bad = ~np.isfinite(a).all(axis=2)
        err = np.abs(a - want["audio"].reshape(len(objs), nb, 513)).max(axis=2) / np.abs(want["audio"]).max(axis=1)[:, None]
        print("cb", cb, "direct", direct, "nan buffers per object:", [list(np.nonzero(r)[0]) for r in bad])
        print("   err per buffer obj3:", np.array2string(err[3], precision=1))
        print("   state finite:", [bool(np.isfinite(s[0]).all()) for s in got["state"]])

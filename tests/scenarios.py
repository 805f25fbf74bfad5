"""Scenario runner shared by the parity tests: plays ONE event script into
(a) the HIP engine through the C ABI (whole batch, stamped with not_before) and
(b) the fp64 CPU oracle buffer by buffer, the way the reference's GUI thread
would feed ModalSolver between step() calls.

An event is a dict: {"t": buffer, "obj": i, "kind": ..., ...} with kind in
  "force"     data= | vid=,vn= | vids=,coords=,vn= ; force_type, width, flags
  "arprm"     a, sigma, mu
  "listener"  pos
  "use_transfer" use
"""
import numpy as np

from openpbso_amd import Engine, ForceMessage, capi
from openpbso_amd import synth
from oracle import oracle_py as orc

B = 513


def force_ev(t, obj, data=None, vid=None, vids=None, coords=None, vn=None, force_type=0, width=0.0,
             start=False, end=False, clear=False):
    return dict(t=t, obj=obj, kind="force", data=data, vid=vid, vids=vids, coords=coords, vn=vn,
                force_type=force_type, width=width, start=start, end=end, clear=clear)


class ObjSpec:
    def __init__(self, lam, n_modes=None, shapes=None, maps=None, rho=synth.RHO, alpha=synth.ALPHA, beta=synth.BETA):
        self.lam, self.shapes, self.maps = lam, shapes, maps
        self.n_modes = len(lam) if n_modes is None else n_modes
        self.rho, self.alpha, self.beta = rho, alpha, beta


def run_engine(objs, events, n_buffers, split=None, **engine_kw):
    """returns dict(audio [n_obj][NB*513] f32, emitted, qnorm {(obj,buf): arr}, state, info)"""
    eng = Engine(**engine_kw)
    try:
        for o in objs:
            oid = eng.add_object(o.lam, o.rho, o.alpha, o.beta, o.n_modes, o.shapes)
            if o.maps is not None:
                eng.set_ffat_maps(oid, o.maps)
        eng.finalize()
        for ev in sorted(events, key=lambda e: e["t"]):
            k = ev["kind"]
            if k == "force":
                m = ForceMessage(data=ev["data"], forceType=ev["force_type"], gaussianWidth=ev["width"],
                                 sustainedForceStart=ev["start"], sustainedForceEnd=ev["end"],
                                 clearAllForces=ev["clear"], vid=ev["vid"], vids=ev["vids"],
                                 coords=ev["coords"], vn=ev["vn"])
                assert eng.enqueue_force(ev["obj"], m, ev["t"])
            elif k == "arprm":
                eng.enqueue_arprm(ev["obj"], ev["a"], ev["sigma"], ev["mu"], ev["t"])
            elif k == "listener":
                eng.compute_transfer(ev["obj"], ev["pos"], ev["t"])
            elif k == "use_transfer":
                eng.set_use_transfer(ev["obj"], ev["use"], ev["t"])
            else:
                raise ValueError(k)
        chunks = [n_buffers] if split is None else split
        assert sum(chunks) == n_buffers
        audio, emitted, qn = [], [], {}
        done = 0
        for nb in chunks:
            eng.step(nb)
            audio.append(eng.audio().copy())
            emitted.append(eng.emitted().copy())
            if eng.qnorm_mode != capi.QNORM_OFF:
                for oi in range(len(objs)):
                    for b in range(nb):
                        qn[(oi, done + b)] = eng.qnorm(oi, b).copy()
            done += nb
        state = [eng.state(i) for i in range(len(objs))]
        latest = [eng.latest_transfer(i) for i in range(len(objs))]
        return dict(audio=np.concatenate(audio, axis=1), emitted=np.concatenate(emitted, axis=1),
                    qnorm=qn, state=state, latest=latest, info=eng.info())
    finally:
        eng.close()


def _oracle_maps(maps):
    out = []
    for m in maps:
        dim = int(m["n_elements"][0][0])
        out.append(orc.uniform_cube(m["mode_id"], m["k"], m["center"], m["cell_size"], dim, m["psi"]))
    return out


def _oracle_one(oi, o, events, n_buffers):
    """one object through the oracle; returns (audio row, emitted row, {buffer: qnorm}, state, latest)"""
    audio = np.zeros(n_buffers * B)
    emitted = np.ones(n_buffers, dtype=bool)
    qn = {}
    s = orc.Solver(o.lam, o.rho, o.alpha, o.beta, n_modes=o.n_modes)
    if o.maps is not None:
        s.read_ffat_maps(_oracle_maps(o.maps))
    evs = sorted([e for e in events if e["obj"] == oi], key=lambda e: e["t"])
    # Model of the caller (GUI) thread for stamped scripts, the same on both sides of the comparison:
    # force messages go straight to the 1023-slot queue; the other calls of an object form a FIFO
    # in stamp order, and enqueueArprmMessageNoFail on a full 1-slot queue SPINS (modal_solver.h:
    # 382-393) -- it, and the calls behind it, wait until a step() has drained the slot.
    gui = []
    ei = 0
    for b in range(n_buffers):
        while ei < len(evs) and evs[ei]["t"] <= b:
            ev = evs[ei]
            ei += 1
            k = ev["kind"]
            if k == "force":
                n = o.n_modes
                if ev["data"] is not None:
                    data = np.asarray(ev["data"], dtype=np.float64)
                elif ev["vid"] is not None:
                    data = orc.modal_force_vertex(o.shapes, ev["vid"], ev["vn"], n)
                elif ev["vids"] is not None:
                    data = orc.modal_force_face(o.shapes, ev["vids"], ev["coords"], ev["vn"], n)
                else:
                    data = np.zeros(n)
                f = orc.make_force(ev["force_type"], ev["width"])
                assert s.enqueue_force(data, f, ev["start"], ev["end"], ev["clear"])
            else:
                gui.append(ev)
        while gui:
            ev = gui[0]
            k = ev["kind"]
            if k == "arprm":
                if not s.enqueue_arprm(ev["a"], ev["sigma"], ev["mu"]):
                    break                                   # spinning: nothing behind it happens yet
            elif k == "listener":
                s.compute_transfer(ev["pos"])
            elif k == "use_transfer":
                s.set_use_transfer(ev["use"])
            gui.pop(0)
        r = s.step()
        if r is None:
            emitted[b] = False
        else:
            audio[b * B:(b + 1) * B] = r[0]
            qn[b] = r[1].copy()
    out = (audio, emitted, qn, s.state(), s.latest_transfer())
    s.close()
    return out


def run_oracle(objs, events, n_buffers, only=None, threads=1):
    """same outputs in float64 from oracle/ (the reference's semantics).  `only`: object ids to run
    (rows of the result keep the order of `only`); threads > 1 runs objects side by side (they are
    independent, and the oracle calls release the GIL)."""
    ids = list(range(len(objs))) if only is None else list(only)
    if threads > 1 and len(ids) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=threads) as ex:
            res = list(ex.map(lambda oi: _oracle_one(oi, objs[oi], events, n_buffers), ids))
    else:
        res = [_oracle_one(oi, objs[oi], events, n_buffers) for oi in ids]
    audio = np.array([r[0] for r in res]).reshape(len(ids), n_buffers * B)
    emitted = np.array([r[1] for r in res]).reshape(len(ids), n_buffers)
    qn = {(k, b): v for k, r in enumerate(res) for b, v in r[2].items()}
    return dict(audio=audio, emitted=emitted, qnorm=qn, state=[r[3] for r in res], latest=[r[4] for r in res])


def rel_errors(got, want):
    """(max-abs / peak, relative L2) per object row."""
    got = np.asarray(got, dtype=np.float64)
    peak = np.abs(want).max(axis=-1)
    peak = np.where(peak == 0, 1.0, peak)
    maxerr = np.abs(got - want).max(axis=-1) / peak
    l2 = np.linalg.norm(got - want, axis=-1) / np.maximum(np.linalg.norm(want, axis=-1), 1e-300)
    return maxerr, l2

#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X modal sound engine.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over torch.distributed (backend nccl = RCCL).  Either the driver launches the
ranks (torch.distributed.run sets WORLD_SIZE) or this script starts them itself: with --gpus N > 1 and
no WORLD_SIZE in the environment the parent process -- before it has touched the GPU in any way --
runs `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child and exits with its
code.  PBSO_BENCH_BACKEND=gloo lets several ranks share one GPU (smoke test of the N > 1 path on a
1-GPU box; RCCL refuses two ranks on one device).

Workload (BASELINE.json configs[3]; at N = 1 all of it on ONE GPU): 1024 objects x 512 modes, a random
impulse stream per object (Poisson, ~20 PointForce hits/s, vertex hits projected onto the mode shapes on
the device), 86 buffers x 513 samples (~1 s of audio) per step.  Objects are independent, so they shard
across ranks with no data-path collective; the finished audio buffers of all ranks are all-gathered over
RCCL inside the timed region (--no-gather leaves that out).  The headline line is WEAK scaling (1024
objects per GPU); for N > 1 the same run also measures the configuration as written (1024 objects split
over the N ranks, balanced by the sum of modes) and reports it under "strong".

One "step" = one pbso_step(86): host bookkeeping of ModalSolver::step for every (object, buffer), upload
of the plan, projection, force combination, and the oscillator-bank kernel.  After the clock has stopped,
8 objects of the first timed step are checked against the fp64 oracle (the run's own output, not a
replay).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B = 513
SAMPLE_RATE = 44100
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F32_PEAK_TFLOPS = 157.3        # fp32 vector peak = dense fp32 MFMA peak (same guide; 64 FLOP/clk/SIMD either way)
BF16_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA peak (same guide)
FLOP_REF = 10                  # reference arithmetic per mode-sample incl. qnorm
# Minimum work of the block formulation per mode-sample: the output projection a_j Q + b_j D of every
# sample = 2 products = 4 flop on the f32 matrix pipe (or 3 bf16 x bf16 products = 12 bf16 flop in the split form), and
# the coarse recurrence x <- P x = 4 FMA per mode per 16 samples = 0.5 flop on the vector ALU.
FLOP_PROJ_F32, FLOP_PROJ_BF16, FLOP_COARSE = 4.0, 12.0, 0.5
# forced block path (dense force profiles): the state is stepped per sample (2 FMA + 1 add = 5 flop, + 2 with qnorm rows)
# and projected on the matrix pipe (4 flop)
FLOP_FORCED_STATE, FLOP_QNORM = 5.0, 2.0
# ... or, without qnorm rows (nothing needs the state of every sample), a block at a time: the increments F . T_n of a block's 16
# profile samples on the matrix pipe (2 products per mode-sample = 4 flop) + the coarse step with its two extra FMAs (6 FMA per
# mode and 16 samples = 0.75 flop); round 5: kernels_pipe.hip iir_pipe5_kernel, kernels_block.hip FTM
FLOP_FORCED_BLOCK = 4.0 + 0.75
XGMI_LINK_GBPS = 153.0         # one xGMI link (point to point; 7 per GPU): the figure the task statement and the guide quote
TOL_MAX, TOL_L2 = 5e-4, 1e-3   # stated fp32 tolerance vs the fp64 oracle
DTYPE_OF_FORM = {
    "block": "f32",
    "block_bf16": "f32 state, bf16x3 projection (operands split into two bf16 halves = 16 significant bits, three products, f32 accumulation)",
    "velocity": "f32", "direct": "f32",
}


def auto_settle(objects, modes, buffers):
    """Untimed steps in front of the warm-up: 40 s of audio (rounds 1 - 4), and -- round 5 -- at least 0.1 s of DEVICE time by the
    headline's rate (9.3 ms for 1024 x 512 x 860): a scene whose steps are a tenth of a millisecond was timed 10 ms after the device
    had idled for seconds, inside the shader clock's ramp (128 x 512 x 86: 0.156 ms per step with 40 settle steps, 0.145 with 400;
    scripts/debug/r05_ramp.sh).  Capped so that the parity check -- the oracle steps its eight sampled objects from buffer 0 -- stays
    within seconds."""
    base = max(2, round(40 * 86 / max(1, buffers)))
    est_ms = max(0.03, 9.3 * objects / 1024 * modes / 512 * buffers / 860)
    cap = max(base, int(6000 / (8 * max(1.0, modes / 512) * buffers / 86)))
    return int(min(cap, max(base, -(-100.0 // est_ms))))


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--settle", type=int, default=-1,
                    help="untimed steps run before the warm-up so that the shader clock has ramped (the first ~70 ms "
                         "after idle run 10 %% slower, profiles/r01_clock_ramp.txt) and the runtime's one-time work is behind "
                         "(one or two asynchronous uploads among an engine's first ~30 block the caller for 6 - 7 ms: "
                         "PBSO_TIMELINE=1, scripts/debug/r03_stalls.py); reported as settle_steps.  Default: 40 s of audio and at least "
                         "0.1 s of device time (auto_settle: 11 steps of 1024 x 512 x 860, 750 of 128 x 512 x 86)")
    ap.add_argument("--clock-ramp-ms", type=float, default=150.0,
                    help="milliseconds a scratch engine (256 x 512, no messages) is stepped right before every leg's settle steps, so that "
                         "short legs are not timed inside the shader clock's ramp from idle (scripts/debug/r05_ramp.sh); 0 = off")
    ap.add_argument("--objects", type=int, default=1024, help="objects per GPU (weak scaling); the total for the strong leg")
    ap.add_argument("--modes", type=int, default=512)
    ap.add_argument("--buffers", type=int, default=860,
                    help="audio buffers per step = per pbso_step call = per oscillator-bank launch (chunk_buffers follows).  860 = 10 s of "
                         "audio: the duration SURVEY.md 8(d) quotes throughput on (86 = 1 s is its parity size).  A launch's fixed "
                         "costs -- ramp, write drain, the hand-over from the preparation stream -- are per launch: rounds 1 - 3 and "
                         "the start of round 4 measured 86-buffer steps; the line's `steps_of_one_second` leg still does")
    ap.add_argument("--no-one-second-leg", action="store_true", help="skip the extra leg with 86-buffer steps (N = 1)")
    ap.add_argument("--form", choices=["block", "block_bf16", "velocity", "direct"], default="block",
                    help="block: block state-space form with the exact f32 MFMA projection (default: every product of the line is f32); "
                         "block_bf16: the same with the output projection as a split-bf16 MFMA product (mixed precision, reported as "
                         "such); velocity / direct: per-sample kernel (K1)")
    ap.add_argument("--plan-threads", type=int, default=0,
                    help="host planner threads (PBSO_PLAN_THREADS; helpers are pinned into the caller's core complex).  Default: one for "
                         "steps of up to 128 buffers (a one-second step's hits are planned in 0.07 ms: a second thread's hand-shake costs "
                         "more than it saves), four for longer ones: a ten-second step of 1024 objects takes ONE thread 3 ms to plan -- "
                         "overlapped with the device in steady state, but the timed region starts with an empty pipeline, and the first "
                         "step's planning is 0.15 ms per step of a 20-step region")
    ap.add_argument("--qnorm", choices=["sample", "closed", "off"], default="sample",
                    help="getQBufferNorm rows: on (the block form evaluates them in closed form), closed form, or off")
    ap.add_argument("--no-qnorm", action="store_true", help="same as --qnorm off")
    ap.add_argument("--modes-per-lane", type=int, default=0)
    ap.add_argument("--submit-thread", type=int, default=-1, choices=[-1, 0, 1],
                    help="1: the engine's second submitting thread (pbso_engine_desc::submit_thread): pbso_step records its stream calls and "
                         "returns, a worker makes them while the caller plans the next launch (legs without the device group only).  "
                         "-1 (default): on for steps of fewer than 256 buffers over at least 4096 (object, buffer) pairs -- where planning + "
                         "submitting a launch is comparable to the launch itself --, off for the long steps of the headline (nothing to gain: "
                         "8.91 against 8.90 ms) and for scenes of a few objects (nothing to plan)")
    ap.add_argument("--scenario", choices=["impulses", "scraping", "listener"], default="impulses",
                    help="impulses: Poisson PointForce hits (headline, configs[1]/[3]); scraping: sustained "
                         "AutoregressiveForce with one face hit per buffer (configs[4]); listener: impulses + FFAT maps "
                         "and a new listener position every buffer (configs[2])")
    ap.add_argument("--weak", action="store_true",
                    help="N > 1: make the weak-scaling run (--objects per GPU) the headline; default for N > 1 is the configuration as written "
                         "(--objects in total, BASELINE configs[3]) with the weak run as the side leg")
    ap.add_argument("--host-delivery", action="store_true",
                    help="one GPU: also time pbso_step_to_host (the oscillator bank writing straight into pinned host memory) in a loop of "
                         "its own behind the timed region.  Off by default: its launches are the headline kernel at the PCIe link's pace "
                         "(3.5 ms instead of 0.94), which would skew a profiler's per-kernel average over the default command")
    ap.add_argument("--no-strong-share", action="store_true",
                    help="one GPU: skip the strong-scaling proxy (the per-rank shares objects / 2, 4, 8 measured on this GPU)")
    ap.add_argument("--share-repeats", type=int, default=3, help="one GPU: runs of every strong-share leg (min / median / max reported)")
    ap.add_argument("--strong", action="store_true",
                    help="make the strong-scaling leg the headline: --objects is the TOTAL, split over the ranks")
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the nested strong-scaling measurement")
    ap.add_argument("--time-every", type=int, default=4, help="HIP-event pairs around every n-th oscillator-bank launch (PBSO_TIMING_EVERY)")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: leave the RCCL audio all-gather out of the timed region")
    ap.add_argument("--no-gather-cost", action="store_true", help="N > 1: skip the extra leg that times the same run without the all-gather")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of the first timed step")
    ap.add_argument("--no-second-form", "--no-f32-leg", dest="no_second_form", action="store_true",
                    help="one GPU, impulses: skip the second run of the same workload in the OTHER block form (headline f32 -> "
                         "mixed_precision_projection; headline block_bf16 -> exact_f32_projection), reported beside the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-objects", type=int, default=0, help="objects in the CPU baseline sample (0 = auto)")
    args = ap.parse_args(argv)
    args.settle_auto = args.settle < 0
    if args.settle < 0:
        args.settle = auto_settle(args.objects, args.modes, args.buffers)
    if args.plan_threads <= 0:
        args.plan_threads = 1 if args.buffers <= 128 else 4
    return args


def submit_thread_of(args):
    """--submit-thread for a leg of args.buffers buffers per step (-1: on for short steps)"""
    st = int(getattr(args, "submit_thread", -1))
    # (on where a step's planning is comparable to its launch -- thousands of (object, buffer) descriptors for a short launch; a scene of
    #  a few objects plans in a microsecond and only pays the hand-over to the worker: 1 x 512 x 86 0.030 -> 0.04 - 0.055 ms per step)
    return st if st >= 0 else (1 if (args.buffers < 256 and args.objects * args.buffers >= 4096) else 0)


# ----------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks ourselves, BEFORE anything in this process touches the GPU
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rank_launch_command(n, argv):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + list(argv)


def spawn_ranks(n, argv):
    """Runs the N ranks as a child job and returns its exit code (the JSON line is rank 0's stdout)."""
    env = dict(os.environ)
    env["PBSO_BENCH_SPAWNED"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(rank_launch_command(n, argv), env=env).returncode


# ----------------------------------------------------------------------------------------------------
def build_inputs(args, global_ids, total_buffers):
    """Deterministic inputs per GLOBAL object id: eigenvalues, mode shapes and the scenario's script (hit vertices and
    normals; scraping: one face hit per buffer; listener: FFAT maps and a listener position per buffer)."""
    from openpbso_amd import synth
    M = args.modes
    lam = np.empty((len(global_ids), M))
    shapes, scripts = [], []
    for i, gid in enumerate(global_ids):
        seed = synth.seed_for(4, gid)
        lam[i] = synth.eigenvalues(M, seed)
        shapes.append(synth.mode_shapes(M, seed))
        sc = {"hits": synth.poisson_hits(total_buffers, seed), "vns": synth.unit_normals(total_buffers, seed)}
        if args.scenario == "scraping":
            rng = np.random.default_rng(synth.seed_for(5, gid))
            bary = rng.random((total_buffers, 3))
            sc["bary"] = bary / bary.sum(axis=1, keepdims=True)
            sc["fids"] = rng.integers(0, synth.N_VERTS, (total_buffers, 3))
        elif args.scenario == "listener":
            sc["maps"] = synth.ffat_maps(lam[i], synth.seed_for(3, gid))
            sc["path"] = synth.listener_path(total_buffers) * (1.0 + 0.001 * i)
        scripts.append(sc)
    return lam, shapes, scripts


def pmc_counts(form, objects, args):
    """instruction counts per wave and buffer of the block kernel from the PMC passes kept under profiles/ (same configuration only)"""
    for name in ("r06_pmc_counts.json", "r05_pmc_counts.json", "r04_pmc_counts.json", "r03_pmc_counts.json", "r02_pmc_counts.json"):
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", name)))
            e = t["forms"][form]
            c = e["config"]
            qn = "off" if args.no_qnorm else args.qnorm
            if (c["objects_per_gpu"], c["modes"], c["buffers_per_step"], c["qnorm"], c["scenario"]) == (
                    objects, args.modes, args.buffers, qn, args.scenario):
                return dict(e, file="profiles/" + name)
        except Exception:
            pass
    return None


def measured_traffic(args, form, objects):
    """HBM bytes per launch from the PMC passes kept under profiles/ (same configuration only)."""
    for name in ("r06_pmc_traffic_%s.json" % form, "r05_pmc_traffic_%s.json" % form, "r04_pmc_traffic_%s.json" % form, "r03_pmc_traffic_%s.json" % form, "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", name)))
            c = t["config"]
            qn = "off" if args.no_qnorm else args.qnorm
            if (c["objects_per_gpu"], c["modes"], c["buffers_per_step"], c["qnorm"], c["form"]) == (
                    objects, args.modes, args.buffers, qn, form) and c.get("scenario", "impulses") == args.scenario:
                return t["traffic_bytes_per_launch"], name
        except Exception:
            pass
    return None, None


def host_cores():
    """threads this process may run on: the affinity mask, capped by a cgroup CPU quota if there is one"""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per))))
    except Exception:
        pass
    return n


def cpu_baseline(args, lam, shapes, scripts):
    """The fp64 oracle (a port: the reference itself cannot be built here) timed on this box's host cores on a
    bounded sample of the same workload: every thread builds and steps its own objects (first touch), one
    warm-up run, median of five; the parallel efficiency against one object on one thread is reported."""
    from oracle import oracle_py as orc
    from openpbso_amd import synth
    ncores = host_cores()
    try:
        lib = orc.lib(native=True)
    except Exception:
        lib = orc.lib()
    nb, M = min(args.buffers, 86), args.modes         # (a bounded sample: one second of audio per object whatever the GPU's step is)
    # sample size: about 15 s of single-thread work (one object-second costs ~15-40 ms), at least two objects per thread
    n_obj = args.cpu_objects or min(len(shapes), max(2 * ncores, 512))
    om = np.ascontiguousarray(lam[:n_obj])
    hit_data = np.zeros((n_obj, M))
    mask = np.zeros((n_obj, nb), dtype=np.uint8)
    for i in range(n_obj):
        hits, vns = scripts[i]["hits"], scripts[i]["vns"]
        mask[i] = (hits[:nb] >= 0).astype(np.uint8)
        first = int(np.argmax(hits[:nb] >= 0)) if mask[i].any() else 0
        # one spatial vector per object (the CPU leg times stepping, not projection)
        hit_data[i] = orc.modal_force_vertex(shapes[i], max(int(hits[first]), 0), vns[first])
    dp = orc._dp

    def run(n, threads, ftz):
        return lib.or_bench_run(n, M, nb, threads, dp(om), synth.RHO, synth.ALPHA, synth.BETA,
                                dp(hit_data), mask.tobytes(), None, ftz)

    def median_of(n, threads, ftz, reps=5):
        run(n, threads, ftz)                       # warm-up: thread pool, page faults, clocks
        return float(np.median([run(n, threads, ftz) for _ in range(reps)]))

    one = median_of(1, 1, 0, reps=3)
    secs = median_of(n_obj, ncores, 0)
    secs_ftz = median_of(n_obj, ncores, 1, reps=3)
    samples = n_obj * nb * B
    eff = (n_obj * one / min(ncores, n_obj)) / secs
    return {
        "value": samples / secs, "unit": "audio samples/s", "cores": ncores, "kind": "port",
        "realtime_x": (nb * B / SAMPLE_RATE) / secs,
        "value_flush_denormals": samples / secs_ftz,
        "single_thread_one_object": {"value": nb * B / one, "realtime_x": (nb * B / SAMPLE_RATE) / one},
        "parallel_efficiency": eff, "suspect": bool(eff < 0.5),
        "sample": f"{n_obj} objects x {M} modes x {nb} buffers (same generator/seeds as the GPU run; {n_obj * one:.1f} s of "
                  f"single-thread work), fp64 oracle (reference-literal loop incl. qnorm), OpenMP static shares over {ncores} "
                  f"threads (sched_getaffinity), every thread builds its own solvers; one warm-up + median of 5: {secs:.3f} s "
                  f"with the default FP environment, {secs_ftz:.3f} s with FTZ/DAZ; one object on one thread (the reference's "
                  f"threading model): {one:.3f} s; parallel efficiency {eff:.2f} (hardware threads, not cores, are counted)",
    }


def oracle_rows(args, lam, shapes, scripts, rows, n_steps):
    """fp64 oracle audio of step n_steps - 1 (0-based) for the local objects `rows`, stepped from the start with the
    scenario's own script (what measure() feeds the engine)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle_py as orc
    from openpbso_amd import synth
    nb = args.buffers

    def one(i):
        s = orc.Solver(lam[i], synth.RHO, synth.ALPHA, synth.BETA)
        sc = scripts[i]
        hits, vns = sc["hits"], sc["vns"]
        if args.scenario == "listener":
            maps = sc["maps"]
            s.read_ffat_maps([orc.uniform_cube(m["mode_id"], m["k"], m["center"], m["cell_size"], int(m["n_elements"][0][0]), m["psi"])
                              for m in maps])
        else:
            s.set_use_transfer(False)
        if args.scenario == "scraping":
            # tools/...:754-776: the dummy start message (data = 0) carries the one AutoregressiveForce of the contact
            assert s.enqueue_force(np.zeros(lam.shape[1]), orc.make_force(orc.AR), sustained_start=True)
        out = None
        for k in range(n_steps):
            bufs = []
            for b in range(k * nb, (k + 1) * nb):
                if args.scenario == "scraping":
                    if b >= 1:
                        s.enqueue_force(orc.modal_force_face(shapes[i], sc["fids"][b], sc["bary"][b], vns[b]), orc.make_force(orc.AR))
                else:
                    if args.scenario == "listener":
                        s.compute_transfer(sc["path"][b])
                    if hits[b] >= 0:
                        s.enqueue_force(orc.modal_force_vertex(shapes[i], int(hits[b]), vns[b]))
                snd = s.step()[0]
                if k == n_steps - 1:
                    bufs.append(snd.copy())
            out = bufs
        return np.concatenate(out)

    with ThreadPoolExecutor(max_workers=min(len(rows), max(1, host_cores()))) as ex:
        return np.array(list(ex.map(one, rows)))


# ----------------------------------------------------------------------------------------------------
def ramp_clock(ctx, ms=150.0):
    """The shader clock takes ~70 ms of work to ramp from idle (profiles/r01_clock_ramp.txt), and a leg's engine is built on the
    host for seconds while the device idles: a leg whose settle + warm-up steps are 10 ms of device time (the 128-object share at
    860 buffers: 4 + 2 steps of 1.3 ms) was TIMED inside the ramp -- 1.25 - 1.27 ms per step against 1.22 with 80 settle steps,
    the 512-object share bimodal (scripts/debug/r05_ramp.sh).  Settle steps of the leg's own script cost scripts and oracle time
    (the parity check steps the oracle from buffer 0), so short legs get the product's kernel on a SCRATCH engine first: 256 objects
    x 512 modes under the headline's hit rate (one second of hits, replayed), stepped for `ms` of wall time right before the leg's
    settle steps.  Untimed, like them."""
    sc = ramp_engine(ctx)
    eng, audio, (fo, fv, fn, ft0), done = sc
    t0 = time.perf_counter()
    n = 0
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(16):
            assert eng.enqueue_vertex_hits(fo, fv, fn, ft0 + 86 * (done + n)) == fo.size
            eng.step(86, into=audio.data_ptr())
            n += 1
        eng.sync()
    sc[3] = done + n
    took = (time.perf_counter() - t0) * 1e3
    ctx["_ramp_ms_per_step"] = took / max(1, n)          # (0.27 ms when the bank runs: 256 x 512 x 86)
    return took


def ramp_engine(ctx):
    """the scratch engine of ramp_clock, created BEFORE the first measured engine of the process (run_leg calls this in front of
    its own Engine(...)): an engine's streams shift where the runtime places the streams created after it (Engine::finalize: 58 ->
    112 us hand-over with one more stream in front), so every leg -- the headline, the shares, the one-second leg, the second form
    -- is built behind the same population of streams"""
    import torch
    from openpbso_amd import capi, synth
    from openpbso_amd.solver import Engine
    sc = ctx.get("_ramp")
    if sc is None:
        n, nb = 256, 86
        lam = synth.eigenvalues(512, synth.seed_for(4, 0))
        shapes = synth.mode_shapes(512, synth.seed_for(4, 0))
        eng = Engine(device=ctx["dev_index"], form=capi.FORM_BLOCK, qnorm=capi.QNORM_ALL, stream=ctx["stream"].cuda_stream, chunk_buffers=128)
        for i in range(n):
            eng.add_object(lam, synth.RHO, synth.ALPHA, synth.BETA, mode_shapes=shapes)
        eng.finalize()
        fo, fv, fn, ft = [], [], [], []
        for i in range(n):
            eng.set_use_transfer(i, False)
            hits = synth.poisson_hits(nb, synth.seed_for(4, i))
            hb = np.nonzero(hits >= 0)[0]
            fo.append(np.full(hb.size, i, dtype=np.int32))
            fv.append(hits[hb].astype(np.int32))
            fn.append(synth.unit_normals(nb, synth.seed_for(4, i))[hb])
            ft.append(hb.astype(np.int64))
        sc = ctx["_ramp"] = [eng, torch.empty(n, nb * 513, dtype=torch.float32, device=ctx["dev"]),
                             tuple(np.ascontiguousarray(np.concatenate(x)) for x in (fo, fv, fn, ft)), 0]
    return sc


def measure(args, ctx, global_ids, want_parity):
    """One engine, settle + warm-up + K timed steps.  Returns the rank-local numbers (elapsed is the max over ranks)."""
    import torch
    import torch.distributed as dist
    from openpbso_amd import Engine, ForceMessage, capi, synth
    from openpbso_amd.distributed import gather_audio

    dev, world, rank, backend = ctx["dev"], ctx["world"], ctx["rank"], ctx["backend"]
    n_obj = len(global_ids)
    if world > 1 and getattr(args, "settle_auto", False):
        # (several ranks: a leg settles by the device time of ITS shard -- the strong leg's steps are 1 / N as long as the weak leg's --;
        #  the largest shard decides, so that every rank runs the same number of steps and collectives)
        import copy
        args = copy.copy(args)
        args.settle = auto_settle(max(ctx.get("counts") or [n_obj]), args.modes, args.buffers)
    # (one more step's worth of script than is stepped: the producer side runs a step AHEAD of the solver, as the reference's
    #  GUI thread does -- the messages of step k + 1 are enqueued right after step k has been submitted, while the device is
    #  busy with it; every timed step still pays for exactly one feed inside the timed region)
    n_steps_all = args.settle + args.warmup + args.steps + 1
    total_buffers = n_steps_all * args.buffers
    lam, shapes, scripts = build_inputs(args, global_ids, total_buffers)
    stream = ctx["stream"].cuda_stream
    form_c = {"block": capi.FORM_BLOCK, "block_bf16": capi.FORM_BLOCK_BF16, "velocity": capi.FORM_VELOCITY, "direct": capi.FORM_DIRECT}[args.form]
    qnorm_c = {"sample": capi.QNORM_ALL, "closed": capi.QNORM_CLOSED, "off": capi.QNORM_OFF}["off" if args.no_qnorm else args.qnorm]
    # A leg with an RCCL collective runs through the C ABI's DEVICE GROUP (include/openpbso_amd.h: one engine per rank, the
    # gather called from C++ on its own stream, double-buffered); torch.distributed only starts the ranks, carries the
    # group's unique id and the barrier around the timed region.  Legs without a collective, and the gloo stand-in of the
    # one-GPU tests, use the engine directly.
    use_group = bool(ctx["use_dist"] and backend == "nccl" and (world > 1 or os.environ.get("PBSO_BENCH_GATHER_SELF") == "1")
                     and not args.no_gather and not ctx.get("leg_without_gather"))
    grp = None
    group_note = None
    if use_group:
        # (should the group not come up on some rank -- librccl missing, a communicator error -- every rank falls back to the
        #  torch.distributed collectives together: the ranks agree on it with one all-reduce, and the JSON line says so)
        ident = [None]
        if rank == 0 and world > 1:
            try:
                from openpbso_amd.group import unique_id
                ident[0] = unique_id()
            except Exception as ex:                      # (librccl missing: the broadcast below still has to happen on every rank)
                group_note = repr(ex)
        if world > 1:
            dist.broadcast_object_list(ident, src=0, device=ctx["coll_dev"])
        try:
            from openpbso_amd.group import Group
            if world > 1 and ident[0] is None:
                raise RuntimeError("no RCCL unique id from rank 0")
            counts_all = ctx.get("counts") or [n_obj]
            # (a one-rank group gathers nothing by itself; PBSO_BENCH_GATHER_SELF=1 asks for the communicator and the collectives
            #  all the same -- ncclCommInitRank, the in-place ncclAllGather, ncclSend / ncclRecv to itself, ncclAllReduce:
            #  PBSO_GROUP_RCCL_ALWAYS -- so that the RCCL calls of group.cpp EXECUTE on a one-GPU box)
            grp = Group([ctx["dev_index"]], world_size=world, first_rank=rank, unique_id=ident[0], form=form_c, qnorm=qnorm_c,
                        modes_per_lane=args.modes_per_lane, chunk_buffers=max(128, args.buffers),
                        transport=capi.GROUP_RCCL_ALWAYS if world == 1 else capi.GROUP_RCCL)
            grp.plan([args.modes] * int(sum(counts_all)))
            assert grp.span(rank) == (global_ids[0], global_ids[-1] + 1), (grp.span(rank), global_ids[0], global_ids[-1])
        except Exception as ex:
            group_note = group_note or repr(ex)
            if grp is not None:
                grp.close()
            grp = None
        ok = torch.tensor([1.0 if grp is not None else 0.0], device=ctx["coll_dev"])
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) < 1.0:
            if grp is not None:
                grp.close()
            grp, use_group = None, False
            group_note = group_note or "the device group failed on another rank"
    if use_group:
        for i, gid in enumerate(global_ids):
            grp.add_object(gid, lam[i], synth.RHO, synth.ALPHA, synth.BETA, mode_shapes=shapes[i])
        eng = grp.engine(rank)
    else:
        eng = Engine(device=ctx["dev_index"], form=form_c, qnorm=qnorm_c, modes_per_lane=args.modes_per_lane, stream=stream,
                     chunk_buffers=max(128, args.buffers), submit_thread=submit_thread_of(args))
        for i, gid in enumerate(global_ids):
            eng.add_object(lam[i], synth.RHO, synth.ALPHA, synth.BETA, mode_shapes=shapes[i])
    if args.scenario == "listener":
        for i, gid in enumerate(global_ids):
            eng.set_ffat_maps(i, scripts[i]["maps"])
    (grp or eng).finalize()
    n_hits = 0
    feed_obj, feed_vid, feed_vn, feed_t, feed_bary = [], [], [], [], []
    for i, gid in enumerate(global_ids):
        sc = scripts[i]
        hits, vns = sc["hits"], sc["vns"]
        if args.scenario == "scraping":
            # tools/...:754-776 + :1127-1160: dummy start message, then one GetModalForceFace per frame
            eng.set_use_transfer(i, False)
            assert eng.enqueue_force(i, ForceMessage(forceType=capi.AUTOREGRESSIVE_FORCE, sustainedForceStart=True), 0)
            feed_obj.append(np.full(total_buffers - 1, i, dtype=np.int32))
            feed_vid.append(sc["fids"][1:].astype(np.int32))
            feed_bary.append(sc["bary"][1:])
            feed_vn.append(vns[1:total_buffers])
            feed_t.append(np.arange(1, total_buffers, dtype=np.int64))
            n_hits += total_buffers - 1
            continue
        if args.scenario == "listener":
            # one computeTransfer per buffer (the camera moves every frame): the whole path in one call
            took = eng.compute_transfer_path(np.full(total_buffers, i, dtype=np.int32), sc["path"][:total_buffers],
                                             np.arange(total_buffers, dtype=np.int64))
            assert took.all()
        else:
            eng.set_use_transfer(i, False)             # no FFAT maps in this config: unit transfer
        hb = np.nonzero(hits >= 0)[0]
        feed_obj.append(np.full(hb.size, i, dtype=np.int32))
        feed_vid.append(hits[hb].astype(np.int32))
        feed_vn.append(vns[hb])
        feed_t.append(hb.astype(np.int64))
        n_hits += hb.size

    # the hit script is fed one step ahead with ONE pbso_enqueue_force_batch call per step (the
    # reference's force queue holds 1023 messages per object, modal_solver.h:105, so a long run
    # cannot be queued up front); message arrays are built here, outside the timed region
    feeds = [None] * n_steps_all
    if feed_obj:
        fo, fv, fn, ft = (np.concatenate(x) for x in (feed_obj, feed_vid, feed_vn, feed_t))
        fb = np.concatenate(feed_bary) if feed_bary else None
        step_of = ft // args.buffers
        # per step object by object, each object's stamps ascending: pbso_enqueue_force_batch then cuts the batch at its
        # threads' object ranges instead of walking it in every thread
        order = np.lexsort((ft, fo, step_of))
        fo, fv, fn, ft, step_of = fo[order], fv[order], fn[order], ft[order], step_of[order]
        fb = fb[order] if fb is not None else None
        bounds = np.searchsorted(step_of, np.arange(n_steps_all + 1))
        for k in range(n_steps_all):
            a, b = bounds[k], bounds[k + 1]
            if fb is None:
                # plain vertex hits: parallel arrays handed over as they are (pbso_enqueue_vertex_hits borrows them until the step)
                # (converted to the call's arguments here, outside the timed region: the arrays are the caller's either way)
                feeds[k] = ("hits", eng.prepare_vertex_hits(fo[a:b], fv[a:b], fn[a:b], ft[a:b]))
            else:
                feeds[k] = ("msgs",) + eng.hit_messages(fo[a:b], fv[a:b], fn[a:b], ft[a:b], coords=fb[a:b],
                                                        force_type=capi.AUTOREGRESSIVE_FORCE if args.scenario == "scraping" else capi.POINT_FORCE)

    nb = args.buffers
    # Gather: the finished buffers of all ranks are all-gathered over RCCL.  Audio and gather
    # targets are double-buffered and the collective is asynchronous on RCCL's own stream, so the gather of
    # step k runs beside the oscillator bank of step k+1; it is waited for only when its buffers are reused
    # (and before the clock stops).  Ragged shards (strong leg) are padded to the largest one.
    # (PBSO_BENCH_GATHER_SELF=1: a one-rank group gathers too -- the RCCL code path on a one-GPU box, tests/test_gpu_bench_ranks.py)
    do_gather = (ctx["use_dist"] and (world > 1 or os.environ.get("PBSO_BENCH_GATHER_SELF") == "1") and not args.no_gather
                 and not ctx.get("leg_without_gather"))
    counts = ctx.get("counts")                       # objects per rank in this leg
    cmax = max(counts) if counts else n_obj
    # leg "mix": the consumer wants ONE mixed stream: every rank sums its objects' buffers and the ranks
    # all-reduce that row (nb * 513 floats) instead of gathering every object's audio
    do_mix = bool(do_gather and ctx.get("leg_mix"))
    # leg "root": only rank 0 consumes the buffers: a gather to the root (send / receive) instead of the all-gather
    do_root = bool(do_gather and ctx.get("leg_root"))
    n_buf = 2 if do_gather else 1
    # All-gather IN PLACE: the engine writes this rank's buffers straight into its slice of the gather target (RCCL, like NCCL, skips
    # the local copy when the send buffer IS that slice: sendbuff == recvbuff + rank * count) -- 181 MB less traffic per step and
    # rank, and on a one-rank group nothing is left to do at all (the copy kernel used to fight the oscillator bank for CUs: 0.06 ms
    # of every step exposed).  The gather-to-root and mix legs keep separate audio buffers.
    in_place = bool(do_gather and not do_mix and not do_root and backend == "nccl")
    in_place = in_place and not use_group
    gathered = ([(torch.zeros if in_place else torch.empty)((world * cmax, nb * B), dtype=torch.float32, device=dev) for _ in range(n_buf)]
                if do_gather and not use_group and not do_mix and (not do_root or rank == 0) else None)
    audios = ([g[rank * cmax:(rank + 1) * cmax] for g in gathered] if in_place
              else [torch.zeros((cmax, nb * B), dtype=torch.float32, device=dev) for _ in range(n_buf)])
    mixes = [torch.zeros(nb * B, dtype=torch.float32, device=dev) for _ in range(n_buf)] if do_mix else None
    ones_obj = torch.ones(n_obj, dtype=torch.float32, device=dev) if do_mix else None
    pending = [None] * n_buf
    enqueue_s = [0.0]
    n_calls = [0]
    # parity: 8 rows of the FIRST TIMED step are copied aside on the engine's stream (1.4 MB device to device)
    prng = np.random.default_rng(0x9B50)
    rows = sorted(set([0, n_obj - 1] + prng.integers(0, n_obj, 6).tolist())) if n_obj > 1 else [0]
    rows_t = torch.tensor(rows, dtype=torch.long, device=dev)
    captured = [None]

    def feed(k):
        if feeds[k] is not None:
            te = time.perf_counter()
            if feeds[k][0] == "hits":
                taken, want_n = eng.enqueue_prepared_vertex_hits(feeds[k][1]), feeds[k][1][0]
            else:
                taken, want_n = eng.enqueue_force_batch(*feeds[k][1:]), feeds[k][1].size
            enqueue_s[0] += time.perf_counter() - te
            assert taken == want_n, "force queue overflow"

    feed(0)

    gather_mode = capi.GATHER_MIX if do_mix else (capi.GATHER_ROOT if do_root else capi.GATHER_ALL)

    deferred = (not use_group) and submit_thread_of(args) > 0

    def one_step(k, capture=False):
        if use_group:
            n_calls[0] += 1
            grp.step(nb)                               # (waits, on the device, for the collective that last read this target)
            feed(k + 1)
            if capture:
                captured[0] = torch.from_numpy(eng.audio_rows(rows))
            grp.gather(gather_mode)                    # asynchronous: runs beside the next step's oscillator bank
            return
        slot = n_calls[0] % n_buf
        n_calls[0] += 1
        if pending[slot] is not None:
            pending[slot].wait()
            pending[slot] = None
        eng.step(nb, into=audios[slot].data_ptr())
        feed(k + 1)                                    # the next step's messages, while the device runs this one
        if deferred and (capture or do_mix or do_root or do_gather):
            eng.flush()                                # (torch work behind the step on the engine's stream: its launches first)
        if capture:
            captured[0] = audios[slot].index_select(0, rows_t)
        if do_mix:
            torch.mv(audios[slot][:n_obj].t(), ones_obj, out=mixes[slot])          # sum over the rank's objects (a GEMV: reads the 181 MB once)
            if backend == "nccl":
                pending[slot] = dist.all_reduce(mixes[slot], async_op=True)
            else:
                m_host = mixes[slot].cpu()
                dist.all_reduce(m_host)
                mixes[slot].copy_(m_host)
        elif do_root:
            if backend == "nccl":
                pending[slot] = dist.gather(audios[slot], list(gathered[slot].chunk(world)) if rank == 0 else None, dst=0, async_op=True)
            else:
                host = audios[slot].cpu()
                parts = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
                dist.gather(host, parts, dst=0)
                if rank == 0:
                    gathered[slot].copy_(torch.cat(parts, dim=0))
        elif do_gather:
            if backend == "nccl":
                pending[slot] = dist.all_gather_into_tensor(gathered[slot], audios[slot], async_op=True)
            else:
                gathered[slot].copy_(gather_audio(audios[slot].cpu()))

    def drain():
        if use_group:
            grp.sync()
        elif deferred:
            eng.flush()                                # (the synchronize that follows must see every launch in its stream)
        for i, w in enumerate(pending):
            if w is not None:
                w.wait()
                pending[i] = None

    # (the interpreter's cyclic collector is kept out of the measured loops: with the run's inputs on the heap -- hit scripts, mode
    #  shapes, feeds -- a full collection takes 40 ms, and it used to fire inside a feed call around step 90 of every run: 128 x 512
    #  measured 0.57-1.0 ms per step for --steps >= 50 and 0.157 for --steps 40, scripts/debug/r04_steps.sh)
    import gc
    gc.collect()
    gc.disable()
    try:
        if args.clock_ramp_ms > 0 and not ctx.get("_ramp_failed"):
            try:
                ramp_clock(ctx, args.clock_ramp_ms)
            except Exception as ex:                   # (a warm-up aid must never take the measurement down with it)
                ctx["_ramp_failed"] = repr(ex)
                print("bench.py: clock ramp skipped: " + repr(ex), file=sys.stderr)
        for k in range(args.settle + args.warmup):
            one_step(k, capture=(k == 0 and want_parity))     # (loads the copy kernel's code object outside the timed region)
        drain()
        torch.cuda.synchronize()
        info0 = eng.info()
        if ctx["use_dist"]:
            dist.barrier()
        torch.cuda.synchronize()
        enqueue_s[0] = 0.0
        t0 = time.perf_counter()
        for k in range(args.steps):
            one_step(args.settle + args.warmup + k, capture=(k == 0 and want_parity))
        drain()
        torch.cuda.synchronize()
        if ctx["use_dist"]:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    finally:
        gc.enable()                                       # (also when a step raises: the caller may run further legs)
    if ctx["use_dist"]:
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctx["coll_dev"])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    info1 = eng.info()
    if use_group:
        # the gathered result holds this rank's rows where the engine wrote them, and finite values everywhere
        p_res, r_rows, r_row = grp.result_ptr(rank)
        assert p_res and r_row == nb * B and r_rows == (1 if do_mix else (world * cmax if (not do_root or rank == 0) else cmax))
        sample = grp.result(rank)
        assert np.isfinite(sample).all()
        if not do_mix:
            mine = sample[rank * cmax:rank * cmax + n_obj] if (not do_root or rank == 0) else sample[:n_obj]
            assert np.array_equal(mine[[0, n_obj - 1]], eng.audio_rows([0, n_obj - 1]))
    assert all(torch.isfinite(a).all() for a in audios)
    if do_mix:
        assert all(torch.isfinite(x).all() for x in mixes)
    if do_gather and not do_mix and backend == "nccl" and gathered is not None:
        last = (n_calls[0] - 1) % n_buf
        assert torch.equal(gathered[last][rank * cmax:rank * cmax + n_obj], audios[last][:n_obj])

    res = {
        "elapsed": elapsed, "n_local": n_obj, "n_hits": n_hits, "gather": bool(do_gather), "settle_steps": args.settle,
        # HIP events bracket every PBSO_TIMING_EVERY-th launch of the timed region (one launch per step here)
        "kernel_samples": info1["total_timed_launches"] - info0["total_timed_launches"],
        "kernel_ms": (info1["total_kernel_ms"] - info0["total_kernel_ms"]) / max(1, info1["total_timed_launches"] - info0["total_timed_launches"]),
        "device_ms": (info1["total_device_ms"] - info0["total_device_ms"]) / max(1, info1["total_timed_launches"] - info0["total_timed_launches"]),
        "plan_ms": (info1["total_host_plan_ms"] - info0["total_host_plan_ms"]) / args.steps,
        "submit_ms": (info1["total_host_submit_ms"] - info0["total_host_submit_ms"]) / args.steps,
        "enqueue_ms": enqueue_s[0] / args.steps * 1e3, "info": info1, "form_run": info1.get("recurrence_form"),
    }
    if res["kernel_samples"] <= 0:
        # (PBSO_TIMING_EVERY=0: no HIP events were recorded; the step time stands in, and says so)
        res["kernel_ms"] = elapsed / args.steps * 1e3
        res["kernel_ms_is_step_time"] = True
    if ctx.get("measure_d2h"):
        # what a HOST-side consumer of every object's buffers (the reference's PortAudio callback takes them from a host
        # queue, modal_solver.h:79-82, 359-363) would add: the step's audio, device to pinned host memory
        pinned = torch.empty((n_obj, nb * B), dtype=torch.float32, pin_memory=True)
        torch.cuda.synchronize()
        reps = []
        for _ in range(3):
            td = time.perf_counter()
            pinned.copy_(audios[0][:n_obj], non_blocking=True)
            torch.cuda.synchronize()
            reps.append(time.perf_counter() - td)
        res["d2h_ms"] = float(np.median(reps)) * 1e3
        res["d2h_bytes"] = n_obj * nb * B * 4
        k2 = max(4, min(10, args.steps))
        if args.host_delivery:
            # ... and MEASURED with the product's own delivery path: pbso_step_to_host into two pinned host buffers in turn (the
            # bank stores its samples straight into them over PCIe), a short timed loop of its own behind the timed region
            hb = [eng.host_buffer(nb), eng.host_buffer(nb)]
            for w in range(2):
                eng.step_to_host(nb, hb[w])
            eng.host_wait()
            td = time.perf_counter()
            for k in range(k2):
                eng.step_to_host(nb, hb[k % 2])
            eng.host_wait()
            res["to_host_ms_per_step"] = (time.perf_counter() - td) / k2 * 1e3
            assert np.isfinite(hb[0]).all() and np.isfinite(hb[1]).all()
        # ... and the one-stream consumer: the object mix on the device (pbso_mix_objects) behind every step
        mix_row = torch.zeros(nb * B, dtype=torch.float32, device=dev)
        for w in range(2):
            eng.step(nb, into=audios[0].data_ptr())
            eng.mix_objects(mix_row.data_ptr())
        eng.sync()
        k3 = max(4, min(40, args.steps))                 # (the pipeline's start-up -- one unhidden plan upload -- is 0.1 - 0.2 ms: amortised)
        td = time.perf_counter()
        for k in range(k3):
            eng.step(nb, into=audios[k % n_buf].data_ptr())
            eng.mix_objects(mix_row.data_ptr())
        eng.sync()
        torch.cuda.synchronize()
        res["mix_ms_per_step"] = (time.perf_counter() - td) / k3 * 1e3
        assert torch.isfinite(mix_row).all()
    if want_parity:
        tp = time.perf_counter()
        got = captured[0].cpu().numpy().astype(np.float64)
        want = oracle_rows(args, lam, shapes, scripts, rows, args.settle + args.warmup + 1)
        peak = np.abs(want).max(axis=1)
        mx = np.abs(got - want).max(axis=1) / peak
        l2 = np.linalg.norm(got - want, axis=1) / np.linalg.norm(want, axis=1)
        res["parity"] = {
            "parity_checked_objects": len(rows), "objects": [int(global_ids[r]) for r in rows],
            "step": "first timed step (buffers %d..%d of the run), this run's own output" % (
                (args.settle + args.warmup) * nb, (args.settle + args.warmup + 1) * nb - 1),
            "max_err": float(mx.max()), "rel_l2": float(l2.max()), "tol_max": TOL_MAX, "tol_l2": TOL_L2,
            "oracle": "oracle/pbso_oracle.c (fp64 restatement), stepped from buffer 0", "seconds": time.perf_counter() - tp,
            "pass": bool((mx <= TOL_MAX).all() and (l2 <= TOL_L2).all()),
        }
    res["_cpu_inputs"] = (lam, shapes, scripts)
    res["device_group"] = bool(use_group)
    res["collective_by"] = ("pbso_group (C ABI: RCCL called from C++" + ("; ONE rank: the collectives are issued on a one-rank communicator, "
                            "nothing crosses a link)" if world == 1 else ")")) if use_group else (
        ("torch.distributed" + (" (pbso_group not used: %s)" % group_note if group_note else "")) if do_gather else None)
    (grp or eng).close()
    return res


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # nothing above this line has initialised HIP (torch is not even imported yet)
        sys.exit(spawn_ranks(args.gpus, argv))

    # host planner threads (per-thread planning contexts merged in object order; the helpers are pinned into the
    # caller's core complex: unpinned they wander over the two sockets of these hosts and lose).  With N ranks on
    # one node every rank gets its share of the cores: max(1, cores // ranks - 1) helpers at most.
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    plan_threads = max(1, min(args.plan_threads, host_cores() // max(1, world_env) - 1))
    os.environ.setdefault("PBSO_PLAN_THREADS", str(plan_threads))
    os.environ.setdefault("PBSO_PLAN_PIN", "1" if int(os.environ["PBSO_PLAN_THREADS"]) > 1 else "0")
    # kernel durations for the roofline: HIP events around every 4th launch (around every launch they cost the stream
    # ~15 us per step; rocprofv3's average over ALL launches of the same command is in profiles/)
    os.environ.setdefault("PBSO_TIMING_EVERY", str(args.time_every if args.steps >= 4 * args.time_every else 1))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP engine has no CPU fallback")
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or leave WORLD_SIZE unset "
                         f"and let bench.py start the ranks)")
    # one rank per GPU; PBSO_BENCH_BACKEND=gloo lets several ranks share one GPU
    backend = os.environ.get("PBSO_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ      # under torch.distributed.run even for one rank
    rccl_ranks = None
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (gloo prints its connection report on fd 1: keep stdout for the ONE JSON line)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(backend)
            one = torch.ones(1, dtype=torch.float64, device=coll_dev)
            dist.all_reduce(one)                      # the first collective: proves every rank is in the group
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
        rccl_ranks = int(round(float(one.item())))
        assert rccl_ranks == dist.get_world_size() == world
    # a real (non-null) stream: handle 0 would make the engine create its own, and the RCCL gather
    # orders itself after the CURRENT torch stream
    run_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(run_stream)
    ctx = dict(dev=dev, dev_index=dev_index, world=world, rank=rank, backend=backend, use_dist=use_dist,
               coll_dev=coll_dev, stream=run_stream)

    from openpbso_amd.distributed import shard_by_modes
    # weak: every rank owns --objects objects (global ids rank * objects ...); strong: --objects in total,
    # contiguous blocks balanced by the sum of modes
    weak_ids = list(range(rank * args.objects, (rank + 1) * args.objects))
    spans = [shard_by_modes([args.modes] * args.objects, world, r) for r in range(world)]
    strong_ids = list(range(*spans[rank]))
    # N > 1: the headline is the configuration AS WRITTEN -- --objects in total, sharded (BASELINE configs[3]: "strong") -- and the
    # run with --objects per GPU is the side leg ("weak"); --weak swaps them
    order = ["weak", "strong"] if args.weak and not args.strong else ["strong", "weak"]
    if world == 1:
        order = ["weak"]                               # one rank: the two legs are the same run
    elif args.no_strong:
        order = order[:1]
    legs = {}
    for leg in order:
        ids = weak_ids if leg == "weak" else strong_ids
        ctx["counts"] = [args.objects] * world if leg == "weak" else [hi - lo for lo, hi in spans]
        ctx["measure_d2h"] = leg == order[0] and rank == 0
        legs[leg] = measure(args, ctx, ids, want_parity=(not args.no_parity and leg == order[0] and rank == 0))
        ctx["measure_d2h"] = False
    head = order[0]
    m = legs[head]
    # one GPU: the same workload once more in the OTHER block form, reported beside the headline
    second = second_form = None
    if world == 1 and args.form in ("block", "block_bf16") and not args.no_second_form and args.scenario == "impulses":
        import copy
        a2 = copy.copy(args)
        a2.form = second_form = "block_bf16" if args.form == "block" else "block"
        ctx["counts"] = [args.objects]
        second = measure(a2, ctx, weak_ids, want_parity=(not args.no_parity and rank == 0))
    # one GPU: the strong-scaling PROXY.  Objects are independent (modal_solver.h:100-126), so the per-rank compute of the
    # 2 / 4 / 8-GPU run of this configuration IS this GPU stepping objects / N of them: measured here, each with its own
    # oracle check, next to the efficiency it implies (this run's ms_per_step / N / the share's ms_per_step; the gather is extra)
    shares = []
    if world == 1 and not args.no_strong_share and args.scenario == "impulses" and args.objects >= 16:
        import copy
        for n_ranks in (2, 4, 8):
            if args.objects % n_ranks:
                continue
            a3 = copy.copy(args)
            a3.objects = args.objects // n_ranks
            # (as much device time before the clock starts as the headline's own settle + warm-up steps take, and at least 0.1 s:
            #  a share's steps are 1 / N as long)
            if args.settle_auto:
                a3.settle = max(auto_settle(a3.objects, a3.modes, a3.buffers), min(200, args.settle * n_ranks))
            ctx["counts"] = [a3.objects]
            # (three runs of every share, each a fresh engine: one hiccup -- a co-start collision, a host stall -- shows as such)
            runs = [measure(a3, ctx, list(range(a3.objects)), want_parity=(not args.no_parity and rank == 0 and rep == 0))
                    for rep in range(max(1, args.share_repeats))]
            shares.append((n_ranks, a3.objects, runs, a3.settle))
        ctx["counts"] = [args.objects]
    # one GPU: the same scene stepped ONE SECOND of audio at a time (86 buffers per pbso_step: the step size of rounds 1 - 3 and of
    # SURVEY 8(d)'s parity runs) -- what a launch's fixed costs take when they are paid every second of audio instead of every ten
    one_second = None
    if world == 1 and args.buffers > 86 and not args.no_one_second_leg and args.scenario == "impulses":
        import copy
        a4 = copy.copy(args)
        a4.buffers = 86
        a4.settle = max(args.settle, (auto_settle(a4.objects, a4.modes, 86) if args.settle_auto else 40) if args.steps >= 20 else 8)
        a4.steps = max(args.steps, 40) if args.steps >= 20 else args.steps
        ctx["counts"] = [args.objects]
        keep = {k: os.environ.get(k) for k in ("PBSO_PLAN_THREADS", "PBSO_PLAN_PIN")}
        os.environ["PBSO_PLAN_THREADS"], os.environ["PBSO_PLAN_PIN"] = "1", "0"      # (one planner thread, as in the lines it is compared with)
        try:
            one_second = (a4, measure(a4, ctx, weak_ids, want_parity=False))
        finally:
            for k, v in keep.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    # what the collective costs: the head leg once more with the all-gather left out, with a gather to rank 0 only
    # (send / receive) and with the reduce a consumer of ONE mixed stream needs (reported beside it, never as `value`)
    bare = mixed = rooted = None
    if m["gather"] and not args.no_gather_cost:
        cnt = [args.objects] * world if head == "weak" else [hi - lo for lo, hi in spans]
        ids = weak_ids if head == "weak" else strong_ids
        ctx["counts"] = cnt
        ctx["leg_without_gather"] = True
        bare = measure(args, ctx, ids, want_parity=False)
        ctx["leg_without_gather"] = False
        ctx["leg_mix"] = True
        mixed = measure(args, ctx, ids, want_parity=False)
        ctx["leg_mix"] = False
        ctx["leg_root"] = True
        rooted = measure(args, ctx, ids, want_parity=False)
        ctx["leg_root"] = False

    rc = 0
    if rank == 0:
        nb, M = args.buffers, args.modes

        def leg_numbers(leg, r):
            total_obj = world * args.objects if leg == "weak" else args.objects
            samples = total_obj * nb * B * args.steps
            return {"value": samples / r["elapsed"], "realtime_x": (nb * B * args.steps / SAMPLE_RATE) / r["elapsed"],
                    "ms_per_step": r["elapsed"] / args.steps * 1e3, "objects_total": total_obj,
                    "objects_rank0": r["n_local"], "kernel_ms_rank0": r["kernel_ms"], "gather": r["gather"]}

        def roofline_of(form, r):
            """roofline of the dominant kernel of one measured leg: HIP events on the launch stream, this rank.
            frac = (minimum work of the formulation that ran) / (kernel time): every term can be recomputed from this dict."""
            k_ms = r["kernel_ms"]
            ms = r["n_local"] * M * nb * B                       # mode-samples per launch
            info = r["info"]
            qn_on = not (args.no_qnorm or args.qnorm == "off")
            # (a block-form engine may run launches that are mostly dense-profile buffers on the per-sample kernel)
            block = r["form_run"] in (0, 3) and info["total_block_launches"] >= info["total_sample_launches"]
            bf16 = block and r["form_run"] == 3
            # a small scene whose launches are mostly dense-profile buffers runs on the pipeline kernel K1p (kernels_pipe.hip); other
            # small scenes run the block kernel cut along the time axis behind a scan of buffer-start states (K5, kernels_scan.hip)
            small = block and 2 * info.get("total_split_launches", 0) > info["total_block_launches"]
            dense = args.scenario == "scraping"
            small_kernel = "iir_pipe5_kernel" if (dense and not qn_on) else "iir_pipe_kernel"
            per_s = 1.0 / (k_ms * 1e-3)
            if block and not bf16 and not dense:
                work = {"f32_matrix_pipe": FLOP_PROJ_F32, "f32_vector_alu": FLOP_COARSE}
                flop, peak, bound = FLOP_PROJ_F32 + FLOP_COARSE, F32_PEAK_TFLOPS, "mfma"
                note = ("block state-space form, f32 projection: per mode-sample 4 flop on v_mfma_f32_16x16x4_f32 (a_j Q + b_j D) + 0.5 flop of "
                        "coarse recurrence on the vector ALU.  On gfx950 the f32-input MFMA and the f32 VALU share one 157.3 TFLOP/s "
                        "datapath: their cycles ADD (profiles/r03_mfma_valu_mix.txt: one MFMA + K v_fma_f32 takes 36 + 2.25 K cycles per "
                        "SIMD at two waves per SIMD, 32 alone; SQ_VALU_MFMA_COEXEC_CYCLES = 0), so min time = (4 + 0.5) x mode-samples / peak")
            elif block and bf16 and not dense:
                work = {"bf16_matrix_pipe": FLOP_PROJ_BF16, "f32_vector_alu": FLOP_COARSE}
                flop, peak, bound = FLOP_PROJ_BF16, BF16_PEAK_TFLOPS, "mfma"
                note = ("block state-space form, split-bf16 projection: per mode-sample 12 bf16 flop on v_mfma_f32_16x16x32_bf16 (three products of "
                        "bf16 halves) + 0.5 f32 flop of coarse recurrence on the vector ALU; the bf16 MFMA co-executes with the VALU, so min "
                        "time = max(12 x mode-samples / 2500 TFLOP/s, 0.5 x mode-samples / 157.3 TFLOP/s) = the matrix term; what the kernel "
                        "actually spends its time on is vector-ALU ISSUE (hi / lo split and packing of every block-start state: issue_utilisation)")
            elif block and dense and not qn_on:
                flop = FLOP_FORCED_BLOCK + FLOP_PROJ_F32
                work = {"f32_matrix_pipe": 4.0 + FLOP_PROJ_F32, "f32_vector_alu": flop - 4.0 - FLOP_PROJ_F32}
                peak, bound = F32_PEAK_TFLOPS, "mfma"
                note = ("dense force profile every buffer, no qnorm rows: the state moves a block of 16 samples at a time -- the increments "
                        "F . T_n on the f32 matrix pipe (4 flop per mode-sample), the coarse step x <- P x + g U_n on the vector ALU (0.75) -- "
                        "and is projected on the matrix pipe (4 flop): 8.75 flop per mode-sample on the one f32 datapath.  "
                        + ("Pipeline kernel, teams of five waves per 64 modes (kernels_pipe.hip, iir_pipe5_kernel): every increment is evaluated "
                           "once, the buffers are walked in order" if small else
                           "Cut in time (K5): the increments are evaluated TWICE, by dense_increment_kernel for the scan (not in kernel_ms) and "
                           "by the bank"))
            elif block and dense:
                flop = FLOP_FORCED_STATE + FLOP_PROJ_F32 + FLOP_QNORM
                work = {"f32_matrix_pipe": FLOP_PROJ_F32, "f32_vector_alu": flop - FLOP_PROJ_F32}
                peak, bound = F32_PEAK_TFLOPS, "valu"
                note = ("dense force profile every buffer, qnorm rows on: per mode-sample the state is stepped literally on the vector ALU "
                        "(2 FMA + 1 add + 1 FMA for the sum of q^2 = 7 flop) and projected on the f32 matrix pipe (4 flop).  "
                        + ("Cut in time (K5, round 5): dense_increment_kernel + the scan make the launch's chunks of buffers independent "
                           "(their time is in device_pipeline_ms, not in kernel_ms), the bank runs one mode per lane, three waves per SIMD"
                           if info.get("total_time_chunk_launches", 0) * 2 > info["total_block_launches"] else
                           "512 waves on 1024 SIMDs: the launch is bound by ONE wave's issue rate, not by the chip's peak"))
            else:
                flop, peak, bound = float(FLOP_REF), F32_PEAK_TFLOPS, "valu"
                work = {"f32_vector_alu": flop}
                note = "per-sample kernel K1: the reference's 10 flop per mode-sample on the fp32 vector ALU (157.3 TFLOP/s, also the dense fp32 MFMA peak)"
            tf = flop * ms * per_s * 1e-12
            min_ms = max((FLOP_PROJ_BF16 * ms / (BF16_PEAK_TFLOPS * 1e12), FLOP_COARSE * ms / (F32_PEAK_TFLOPS * 1e12))) * 1e3 if (block and bf16 and not dense) \
                else flop * ms / (peak * 1e12) * 1e3
            # algorithmic bytes of one launch (SURVEY 8(d) formula with NB_l = nb; M = modes per object); the block
            # form also reads its operand table once per launch (128 B per mode) and the coarse-step matrix (16 B)
            bytes_alg = r["n_local"] * (M * (12 + 8 + 8 + (4 + 4 + 4) * nb + (144 if block else 0)) + nb * B * (4 + 4))
            gbs = bytes_alg * per_s * 1e-9
            traffic, traffic_src = measured_traffic(args, form, r["n_local"])
            out = {
                "bound": bound, "kernel": small_kernel if small else ("iir_block_kernel" if block else "iir_bank_kernel"),
                "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
                "min_work": {"flop_per_mode_sample_by_pipe": work, "mode_samples_per_launch": ms, "min_kernel_ms": min_ms,
                             "kernel_ms": k_ms, "note": note},
                "kernel_ms": k_ms,
                "kernel_ms_source": ("no HIP events (PBSO_TIMING_EVERY=0): ms_per_step stands in for the kernel time" if r.get("kernel_ms_is_step_time") else
                                     "HIP events on the launch stream around every %s-th launch of the timed region: %d launches" % (
                                         os.environ["PBSO_TIMING_EVERY"], r["kernel_samples"])),
                "reference_equivalent": {"flop_per_mode_sample": FLOP_REF, "achieved": FLOP_REF * ms * per_s * 1e-12,
                                         "frac_of_f32_peak": FLOP_REF * ms * per_s * 1e-12 / F32_PEAK_TFLOPS,
                                         "note": "the reference's 10 flop per mode-sample at this kernel time: above 1 means the kernel does "
                                                 "not execute the reference's arithmetic (block reformulation), not that a roof was broken"},
                "hbm": {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_launch": bytes_alg},
                "traffic": traffic,
                "traffic_unit": f"HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/{traffic_src})" if traffic else None,
            }
            pc = pmc_counts(form, r["n_local"], args) if block else None
            if pc:
                # instruction-issue view, only for the configuration the PMC passes were taken on
                wb = r["n_local"] * M / (64.0 * info["modes_per_lane"]) * nb       # wave-buffers per launch
                out["issue_utilisation"] = {
                    "valu_per_wave_buffer": pc["valu_per_wave_buffer"], "mfma_per_wave_buffer": pc["mfma_per_wave_buffer"],
                    "valu_issue_slots_frac": pc["valu_per_wave_buffer"] * wb * 64 * 2 * per_s * 1e-12 / F32_PEAK_TFLOPS,
                    "source": pc["file"] + ": " + pc.get("source", ""),
                    "note": "every VALU instruction counted as 64 lanes x 2 flop against the 157.3 TFLOP/s f32 vector peak: issue-slot "
                            "utilisation, not a flop roofline (split / pack / head instructions count as much as FMAs)"}
            return out

        hn = leg_numbers(head, m)
        info = m["info"]
        roof = roofline_of(args.form, m)
        block_run = m["form_run"] in (0, 3) and info["total_block_launches"] >= info["total_sample_launches"]
        qn_txt = "off" if (args.no_qnorm or args.qnorm == "off") else (
            "rows on (closed form x0' G x0 in force-free block buffers, per-sample sums in dense-profile buffers)" if block_run else
            ("rows on (closed form)" if args.qnorm == "closed" else "rows on (per-sample sums)"))
        coll_txt = ("RCCL" if backend == "nccl" else backend + " (host transport standing in for RCCL)")
        host_ms = m["plan_ms"] + m["enqueue_ms"] + m["submit_ms"]
        out = {
            "metric": "audio samples/s & real-time x at N_obj x N_modes",
            "value": hn["value"], "unit": "audio samples/s", "realtime_x": hn["realtime_x"],
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_steps": m.get("settle_steps", args.settle),
            "clock_ramp_ms": 0.0 if ctx.get("_ramp_failed") else args.clock_ramp_ms,
            "ms_per_step": hn["ms_per_step"],
            "value_step_seconds": nb * B / SAMPLE_RATE,      # seconds of audio per step of `value` (rounds 1 - 3: 1.0; since round 4: 10.0 -- `steps_of_one_second` carries the comparable figure)
            "higher_is_better": True, "scaling": head, "vs_baseline": None,
            "dtype": DTYPE_OF_FORM[args.form], "data": "synthetic",
            "hbm_frac": roof["hbm"]["frac"],
            "config": {
                "workload": f"{m['n_local']} objects x {M} modes on rank 0 ({hn['objects_total']} in the job), " + {
                                "impulses": "Poisson impulse stream (~20 PointForce hits/s/object, on-device vertex projection), unit transfer, ",
                                "scraping": "sustained AutoregressiveForce scraping (one GetModalForceFace message per buffer, "
                                            "profiles generated on the device), unit transfer, ",
                                "listener": "Poisson impulse stream + FFAT maps (16x16 cube faces) with a new listener position every buffer, ",
                            }[args.scenario] + f"{nb} buffers x 513 samples = {nb * B / SAMPLE_RATE:.1f} s of audio per step "
                            f"(one pbso_step call, {-(-nb // max(128, nb))} oscillator-bank launch), "
                            f"qnorm {qn_txt}, {args.form} recurrence form"
                            + (f", {coll_txt} all-gather of the audio buffers inside the timed region" if m["gather"] else ""),
                "scenario": args.scenario, "objects_per_gpu": m["n_local"], "modes": M, "buffers_per_step": nb, "frames_per_buffer": B,
                "hits": m["n_hits"], "modes_per_lane": info["modes_per_lane"], "waves_per_object": info["waves_per_object"],
                "objects_total": hn["objects_total"],
                "time_chunked_launches": info.get("total_time_chunk_launches", 0), "bank_launches": info["total_block_launches"] + info["total_sample_launches"],
                "recurrence_form": args.form, "gather": m["gather"], "group_ranks": rccl_ranks, "rccl_ranks": rccl_ranks if backend == "nccl" else None,
                "backend": backend if use_dist else None,
                "collective_by": m.get("collective_by"),
                "submit_thread": 0 if m.get("device_group") else submit_thread_of(args),
                "host_planner_threads": int(os.environ["PBSO_PLAN_THREADS"]), "host_cores": host_cores(), "parallelism": f"object-sharded x{world}",
                "launched_by": "bench.py" if os.environ.get("PBSO_BENCH_SPAWNED") else ("torch.distributed.run" if use_dist else "single process"),
            },
            "roofline": roof,
            "timing": {"device_pipeline_ms": m["device_ms"], "host_plan_ms": m["plan_ms"], "host_enqueue_ms": m["enqueue_ms"],
                       "host_submit_ms": m["submit_ms"], "host_ms": host_ms, "host_bound": bool(host_ms > 0.8 * hn["ms_per_step"]),
                       "host_ms_over_kernel_ms": host_ms / m["kernel_ms"] if m["kernel_ms"] > 0 else None,
                       "note": "host_ms = the CALLING thread's work per step: feeding the messages + planning + packing / uploading / launching "
                               "(with submit_thread: recording the calls; a worker makes them); it overlaps the device.  host_bound: host_ms > "
                               "0.8 x ms_per_step -- the step is as long as the caller's own work.  (Rounds 1-5 compared planning + feeding "
                               "with the bank kernel's time and left the submit calls out.)"},
        }
        if "d2h_ms" in m:
            step_s = nb * B / SAMPLE_RATE
            out["host_delivered"] = {
                "d2h_ms_per_step": m["d2h_ms"], "bytes_per_step": m["d2h_bytes"],
                "realtime_x_if_copied_after_each_step": step_s / ((hn["ms_per_step"] + m["d2h_ms"]) * 1e-3),
                "realtime_x_if_copy_overlaps_compute": step_s / (max(hn["ms_per_step"], m["d2h_ms"]) * 1e-3),
                "to_host_ms_per_step_measured": m.get("to_host_ms_per_step"),
                "to_host_measured_by": "this run (--host-delivery)" if m.get("to_host_ms_per_step") else
                                       "python bench.py --host-delivery (profiles/r04_bench_host_delivery.json): not part of the default command",
                "realtime_x_overlapped_measured": step_s / (m["to_host_ms_per_step"] * 1e-3) if m.get("to_host_ms_per_step") else None,
                "overlapped_path": "pbso_step_to_host into pinned host memory: the oscillator bank stores its samples straight into the (device-"
                                   "mapped) host buffer over PCIe, no copy pass (PCIe-bound: bytes_per_step / to_host_ms_per_step_measured; a "
                                   "hipMemcpyAsync beside the next bank does not overlap on this stack: the copy runs as a blit kernel)",
                "mix_on_device": {"ms_per_step": m.get("mix_ms_per_step"),
                                  "realtime_x": step_s / (m["mix_ms_per_step"] * 1e-3) if m.get("mix_ms_per_step") else None,
                                  "frac_of_value": (hn["ms_per_step"] / m["mix_ms_per_step"]) if m.get("mix_ms_per_step") else None,
                                  "note": "a consumer of ONE mixed stream: pbso_mix_objects (the step's audio summed over the objects on the "
                                          "device, fixed order) behind every step; 176 KB instead of 181 MB leave the GPU.  A loop of its own behind the timed "
                                          "region (up to 40 steps; the objects ring on, no new hits are fed)"},
                "note": "`value` leaves every object's audio in HBM (SURVEY 8(b)/(e): the consumer is the gather / mix).  The reference's consumer is "
                        "host-side (a queue of SoundMessages, modal_solver.h:79-82, 359-363): delivering all objects' buffers to pinned host "
                        "memory (pbso_read_audio) costs d2h_ms_per_step on top -- measured here after the timed region, never part of `value`",
            }
        if "parity" in m:
            out["parity"] = m["parity"]
            out["parity_checked_objects"] = m["parity"]["parity_checked_objects"]
            out["max_err"] = m["parity"]["max_err"]
            if not m["parity"]["pass"]:
                rc = 3
        if bare is not None:
            bn = leg_numbers(head, bare)
            per_rank = max(ctx["counts"]) * nb * B * 4
            exposed = hn["ms_per_step"] - bn["ms_per_step"]
            # (a one-rank communicator -- PBSO_BENCH_GATHER_SELF -- receives nothing: no link, no bound)
            link_ms = per_rank * (world - 1) / (XGMI_LINK_GBPS * 1e9 * max(1, min(world - 1, 7))) * 1e3
            out["gather_cost"] = {
                "bytes_sent_per_rank": per_rank, "bytes_received_per_rank": per_rank * (world - 1),
                "ms_per_step_with_gather": hn["ms_per_step"], "ms_per_step_without_gather": bn["ms_per_step"],
                "value_without_gather": bn["value"], "realtime_x_without_gather": bn["realtime_x"],
                "exposed_ms": exposed,
                "inbound_GBps_if_gather_bound": per_rank * (world - 1) / (hn["ms_per_step"] * 1e-3) * 1e-9,
                "xgmi_inbound_peak_GBps": XGMI_LINK_GBPS * min(world - 1, 7),
                # the two numbers a SCALE reader needs side by side: how well the compute scales, and what the links
                # allow when every rank is handed every object's buffers (one link per peer, point to point; the all-gather of step k
                # runs beside the bank of step k + 1, so a step costs the larger of the two)
                "link_bound_ms": link_ms,
                "link_bound_ms_at_half_rate": 2.0 * link_ms,
                "expected_step_ms": max(bn["ms_per_step"], link_ms),
                "gather_bound": link_ms > bn["ms_per_step"],
                "note": "every rank produces bytes_sent_per_rank of audio per step and receives the other ranks' buffers; the "
                        "all-gather is asynchronous and double-buffered (it runs beside the next step's oscillator bank), so a "
                        "step costs max(compute, gather); at these sizes the links, not the kernels, set the step time "
                        "whenever bytes_received_per_rank / xgmi_inbound_peak exceeds ms_per_step_without_gather: link_bound_ms is that "
                        "quotient at the links' quoted rate (153 GB/s each; link_bound_ms_at_half_rate if that figure counts both "
                        "directions), expected_step_ms = max(compute, link bound); `mix` (one mixed row per rank, all-reduced) and "
                        "`gather_to_root` in this line are the consumers that do not need every buffer everywhere",
            }
        if second is not None:
            sn = leg_numbers(head, second)
            key = "mixed_precision_projection" if second_form == "block_bf16" else "exact_f32_projection"
            out[key] = {
                "value": sn["value"], "realtime_x": sn["realtime_x"], "ms_per_step": sn["ms_per_step"], "kernel_ms": second["kernel_ms"],
                "dtype": DTYPE_OF_FORM[second_form], "recurrence_form": second_form,
                "max_err": second.get("parity", {}).get("max_err"), "parity_pass": second.get("parity", {}).get("pass"),
                "roofline": roofline_of(second_form, second),
                "timing": {"host_plan_ms": second["plan_ms"], "host_enqueue_ms": second["enqueue_ms"]},
                "note": ("the same workload and engine with --form block_bf16: the output projection as three bf16 x bf16 products of split "
                         "operands (16 significant bits each, f32 accumulation) on v_mfma_f32_16x16x32_bf16 instead of the exact f32 MFMA "
                         "product; the state recurrence is f32 in both.  A mixed-precision variant: never the line's `value`, never `dtype` f32")
                        if second_form == "block_bf16" else
                        ("the same workload and engine with --form block: the output projection as an exact f32 MFMA product "
                         "(v_mfma_f32_16x16x4_f32); the state recurrence is f32 in both"),
            }
            if "parity" in second and not second["parity"]["pass"]:
                rc = 3
        if shares:
            rows_s = []
            for n_ranks, n_o, runs, settle_s in shares:
                lns = sorted((leg_numbers("strong", r_) for r_ in runs), key=lambda d: d["ms_per_step"])
                r = runs[0]                                   # (the run that carries the oracle check)
                ln = lns[len(lns) // 2]                       # the median run
                rows_s.append({"n_gpus": n_ranks, "objects": n_o, "ms_per_step": ln["ms_per_step"], "kernel_ms": float(np.median([r_["kernel_ms"] for r_ in runs])),
                               "realtime_x": ln["realtime_x"],
                               "implied_efficiency_at_N": hn["ms_per_step"] / n_ranks / ln["ms_per_step"],
                               "runs": len(runs), "settle_steps": settle_s, "ms_per_step_min_median_max": [lns[0]["ms_per_step"], ln["ms_per_step"], lns[-1]["ms_per_step"]],
                               "implied_efficiency_min_median_max": [hn["ms_per_step"] / n_ranks / lns[-1]["ms_per_step"],
                                                                     hn["ms_per_step"] / n_ranks / ln["ms_per_step"],
                                                                     hn["ms_per_step"] / n_ranks / lns[0]["ms_per_step"]],
                               "time_chunked_launches": r["info"].get("total_time_chunk_launches", 0),
                               "bank_launches": r["info"]["total_block_launches"] + r["info"]["total_sample_launches"],
                               "max_err": r.get("parity", {}).get("max_err"), "parity_pass": r.get("parity", {}).get("pass")})
                if "parity" in r and not r["parity"]["pass"]:
                    rc = 3
            out["strong_share"] = {
                "shares": rows_s,
                "note": "strong-scaling proxy measured on THIS GPU: objects are independent, so a rank of the N-GPU run of this configuration "
                        "steps objects / N of them -- this is that rank's compute, gather not included; implied_efficiency_at_N = "
                        "(this line's ms_per_step / N) / the share's ms_per_step.  Shares that leave SIMDs idle run the block kernel cut "
                        "along the time axis (K5: time_chunked_launches).  A share's steps are 1 / N as long as the headline's, so it takes "
                        "N times the settle steps (at least 0.1 s of device time: settle_steps) -- with the headline's four, the 128-object "
                        "share was timed inside the shader clock's ramp from idle (1.26 against 1.22 ms, profiles/r05_shares_inside_the_clock_ramp.txt)"}
        if one_second is not None:
            a4, r4 = one_second
            secs = 86 * B * a4.steps / SAMPLE_RATE
            out["steps_of_one_second"] = {
                "value": args.objects * 86 * B * a4.steps / r4["elapsed"], "realtime_x": secs / r4["elapsed"],
                "ms_per_step": r4["elapsed"] / a4.steps * 1e3, "buffers_per_step": 86, "steps": a4.steps, "kernel_ms": r4["kernel_ms"],
                "note": "the same engine and scene with 86 buffers (one second of audio) per pbso_step, the step size of the lines of "
                        "rounds 1 - 3: a launch's fixed costs (ramp, write drain, hand-over from the preparation stream: ~35 us) "
                        "are paid per second of audio instead of per ten"}
        if mixed is not None:
            out["mix"] = dict(leg_numbers(head, mixed), scaling=head, collective="all_reduce(sum) of one mixed row per rank",
                              bytes_per_rank=nb * B * 4,
                              note="the same leg when the consumer wants ONE mixed stream instead of every object's buffers (SURVEY 8(e)): "
                                   "each rank sums its objects' audio and the ranks all-reduce nb * 513 floats; reported beside the "
                                   "headline, never as `value`")
        if rooted is not None:
            out["gather_to_root"] = dict(leg_numbers(head, rooted), scaling=head, collective="gather to rank 0 (send / receive)",
                                         bytes_received_by_root=max(ctx["counts"]) * nb * B * 4 * (world - 1),
                                         note="the same leg when only rank 0 consumes the buffers (SURVEY 8(e): ncclSend / ncclRecv to a root): "
                                              "the root's inbound links carry what every rank's carry in the all-gather; the other ranks only send")
        for leg, r in legs.items():
            if leg != head:
                out[leg] = dict(leg_numbers(leg, r), scaling=leg,
                                note="--objects objects (the configuration as written) split over the ranks by the sum of modes"
                                if leg == "strong" else "--objects per GPU")
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(args, *m["_cpu_inputs"])
            except Exception as ex:   # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(out), flush=True)
        if rc:
            bad = [x["parity"] for x in (m, second or {}) if "parity" in x and not x["parity"]["pass"]]
            print("PARITY FAILED: " + json.dumps(bad), file=sys.stderr)
    if use_dist:
        dist.destroy_process_group()
    if rc:
        sys.exit(rc)


if __name__ == "__main__":
    main()

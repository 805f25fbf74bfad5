#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X modal sound engine.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[3], at N=1 on ONE GPU): 1024 objects x 512
modes, a random impulse stream per object (Poisson, ~20 PointForce hits/s,
vertex hits projected onto the mode shapes on the device), 86 buffers x 513
samples (~1 s of audio) per step.  Weak scaling: every rank owns 1024 objects
(objects are independent -- no data-path collective); with --gather the
finished audio buffers of all ranks are all-gathered over RCCL inside the
timed region, as a consumer of the whole mix would need.

One "step" = one pbso_step(86): host bookkeeping of ModalSolver::step for every
(object, buffer), upload of the plan, projection, force combination, and the
oscillator-bank kernel.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B = 513
SAMPLE_RATE = 44100
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_TFLOPS = 157.3       # fp32 vector peak (= fp32 MFMA dense peak), same guide
FLOP_PER_MODE_SAMPLE = 10      # reference arithmetic incl. qnorm


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--settle", type=int, default=30,
                    help="untimed steps run before the warm-up so that the shader clock has ramped (the first ~70 ms "
                         "after idle run 10 %% slower, profiles/r01_clock_ramp.txt); reported as settle_steps")
    ap.add_argument("--objects", type=int, default=1024, help="objects per GPU")
    ap.add_argument("--modes", type=int, default=512)
    ap.add_argument("--buffers", type=int, default=86, help="audio buffers per step")
    ap.add_argument("--form", choices=["block", "velocity", "direct"], default="block")
    ap.add_argument("--qnorm", choices=["sample", "closed", "off"], default="sample",
                    help="getQBufferNorm: per-sample accumulation (reference loop), closed form, or off")
    ap.add_argument("--no-qnorm", action="store_true", help="same as --qnorm off")
    ap.add_argument("--modes-per-lane", type=int, default=0)
    ap.add_argument("--scenario", choices=["impulses", "scraping", "listener"], default="impulses",
                    help="impulses: Poisson PointForce hits (headline, configs[1]/[3]); scraping: sustained "
                         "AutoregressiveForce with one face hit per buffer (configs[4]); listener: impulses + FFAT maps "
                         "and a new listener position every buffer (configs[2])")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --objects is the TOTAL, split evenly over the ranks (BASELINE configs[3] as written: "
                         "1024 objects over 8 GPUs); default is weak scaling, --objects per GPU")
    ap.add_argument("--gather", action="store_true", help="all-gather audio over RCCL inside the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-objects", type=int, default=0, help="objects in the CPU baseline sample (0 = auto)")
    return ap.parse_args()


def build_inputs(args, rank):
    """Deterministic per-rank inputs: eigenvalues, mode shapes, hit script."""
    from openpbso_amd import synth
    n_obj, M = args.objects, args.modes
    total_buffers = (args.steps + args.warmup + args.settle) * args.buffers
    lam = np.empty((n_obj, M))
    shapes = []
    scripts = []
    for i in range(n_obj):
        seed = synth.seed_for(4, rank * n_obj + i)
        lam[i] = synth.eigenvalues(M, seed)
        shapes.append(synth.mode_shapes(M, seed))
        hits = synth.poisson_hits(total_buffers, seed)
        vns = synth.unit_normals(total_buffers, seed)
        scripts.append((hits, vns))
    return lam, shapes, scripts


def measured_traffic(args):
    """HBM bytes per launch from the PMC passes kept under profiles/ (same config only)."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        c = t["config"]
        qn = "off" if args.no_qnorm else args.qnorm
        if (c["objects_per_gpu"], c["modes"], c["buffers_per_step"], c["qnorm"], c["form"]) == (
                args.objects, args.modes, args.buffers, qn, args.form):
            return t["traffic_bytes_per_launch"]
    except Exception:
        pass
    return None


def cpu_baseline(args, lam, shapes, scripts):
    """The fp64 oracle (a port: the reference itself cannot be built here) timed on
    this box's host cores on a bounded sample of the same workload."""
    from oracle import oracle_py as orc
    from openpbso_amd import synth
    ncores = os.cpu_count() or 1
    try:
        lib = orc.lib(native=True)
    except Exception:
        lib = orc.lib()
    n_obj = args.cpu_objects or min(args.objects, max(ncores, 8) * 2)
    nb, M = args.buffers, args.modes
    om = np.ascontiguousarray(lam[:n_obj])
    hit_data = np.zeros((n_obj, M))
    mask = np.zeros((n_obj, nb), dtype=np.uint8)
    for i in range(n_obj):
        hits, vns = scripts[i]
        mask[i] = (hits[:nb] >= 0).astype(np.uint8)
        first = int(np.argmax(hits[:nb] >= 0)) if mask[i].any() else 0
        # one spatial vector per object (the CPU leg times stepping, not projection)
        hit_data[i] = orc.modal_force_vertex(shapes[i], max(int(hits[first]), 0), vns[first])
    dp = orc._dp
    secs = lib.or_bench_run(n_obj, M, nb, ncores, dp(om), synth.RHO, synth.ALPHA, synth.BETA,
                            dp(hit_data), mask.tobytes(), None, 0)
    secs_ftz = lib.or_bench_run(n_obj, M, nb, ncores, dp(om), synth.RHO, synth.ALPHA, synth.BETA,
                                dp(hit_data), mask.tobytes(), None, 1)
    one = lib.or_bench_run(1, M, nb, 1, dp(om), synth.RHO, synth.ALPHA, synth.BETA,
                           dp(hit_data), mask.tobytes(), None, 0)
    samples = n_obj * nb * B
    return {
        "value": samples / secs, "unit": "audio samples/s", "cores": ncores, "kind": "port",
        "realtime_x": (nb * B / SAMPLE_RATE) / secs,
        "value_flush_denormals": samples / secs_ftz,
        "single_thread_one_object": {"value": nb * B / one, "realtime_x": (nb * B / SAMPLE_RATE) / one},
        "sample": f"{n_obj} objects x {M} modes x {nb} buffers (same generator/seeds as the GPU run), "
                  f"fp64 oracle (reference-literal loop incl. qnorm), OpenMP over objects on {ncores} threads: "
                  f"{secs:.2f} s with the default FP environment, {secs_ftz:.2f} s with FTZ/DAZ; "
                  f"one object on one thread (the reference's threading model): {one:.3f} s",
    }


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP engine has no CPU fallback")
    # one rank per GPU; PBSO_BENCH_BACKEND=gloo lets several ranks share one GPU (smoke test of the
    # multi-rank path on a 1-GPU box: RCCL refuses two ranks on one device)
    backend = os.environ.get("PBSO_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ      # under torch.distributed.run even for one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    if args.strong:
        if args.objects % world:
            raise SystemExit("--strong needs --objects divisible by the number of ranks")
        args.objects //= world
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")

    from openpbso_amd import Engine, ForceMessage, capi, synth
    from openpbso_amd.distributed import gather_audio

    lam, shapes, scripts = build_inputs(args, rank)
    # a real (non-null) stream: handle 0 would make the engine create its own, and the RCCL gather
    # orders itself after the CURRENT torch stream
    run_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(run_stream)
    stream = run_stream.cuda_stream
    eng = Engine(device=dev_index,
                 form={"block": capi.FORM_BLOCK, "velocity": capi.FORM_VELOCITY, "direct": capi.FORM_DIRECT}[args.form],
                 qnorm={"sample": capi.QNORM_ALL, "closed": capi.QNORM_CLOSED, "off": capi.QNORM_OFF}[
                     "off" if args.no_qnorm else args.qnorm],
                 modes_per_lane=args.modes_per_lane, stream=stream)
    for i in range(args.objects):
        eng.add_object(lam[i], synth.RHO, synth.ALPHA, synth.BETA, mode_shapes=shapes[i])
        if args.scenario == "listener":
            eng.set_ffat_maps(i, synth.ffat_maps(lam[i], synth.seed_for(3, rank * args.objects + i)))
    eng.finalize()
    n_hits = 0
    feed_obj, feed_vid, feed_vn, feed_t, feed_bary = [], [], [], [], []
    n_steps_all = args.settle + args.warmup + args.steps      # the clock-settle steps run the same script
    total_buffers = n_steps_all * args.buffers
    off = 0
    for i in range(args.objects):
        hits, vns = scripts[i]
        if args.scenario == "scraping":
            # tools/...:754-776 + :1127-1160: dummy start message, then one GetModalForceFace per frame
            eng.set_use_transfer(i, False)
            rng = np.random.default_rng(synth.seed_for(5, rank * args.objects + i))
            assert eng.enqueue_force(i, ForceMessage(forceType=capi.AUTOREGRESSIVE_FORCE, sustainedForceStart=True), off)
            bary = rng.random((total_buffers, 3))
            fids = rng.integers(0, synth.N_VERTS, (total_buffers, 3))
            feed_obj.append(np.full(total_buffers - 1, i, dtype=np.int32))
            feed_vid.append(fids[1:].astype(np.int32))
            feed_bary.append((bary / bary.sum(axis=1, keepdims=True))[1:])
            feed_vn.append(vns[1:total_buffers])
            feed_t.append(np.arange(1, total_buffers, dtype=np.int64))
            n_hits += total_buffers - 1
            continue
        if args.scenario == "listener":
            path = synth.listener_path(total_buffers) * (1.0 + 0.001 * i)
            for b in range(total_buffers):
                eng.compute_transfer(i, path[b], off + int(b))
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
        order = np.lexsort((fo, ft))                 # time-major: per object the stamps stay ascending
        fo, fv, fn, ft, step_of = fo[order], fv[order], fn[order], ft[order], step_of[order]
        fb = fb[order] if fb is not None else None
        bounds = np.searchsorted(step_of, np.arange(n_steps_all + 1))
        for k in range(n_steps_all):
            a, b = bounds[k], bounds[k + 1]
            feeds[k] = eng.hit_messages(fo[a:b], fv[a:b], fn[a:b], ft[a:b] + off, coords=None if fb is None else fb[a:b],
                                        force_type=capi.AUTOREGRESSIVE_FORCE if args.scenario == "scraping" else capi.POINT_FORCE)

    nb = args.buffers
    # --gather: the finished buffers of all ranks are all-gathered over RCCL.  Audio and
    # gather targets are double-buffered and the collective is asynchronous on RCCL's own stream, so the
    # gather of step k runs beside the oscillator bank of step k+1; it is waited for only when its
    # buffers are reused (and before the clock stops).
    do_gather = args.gather and use_dist
    n_buf = 2 if do_gather else 1
    audios = [torch.empty((args.objects, nb * B), dtype=torch.float32, device=dev) for _ in range(n_buf)]
    audio = audios[0]
    gathered = [torch.empty((world * args.objects, nb * B), dtype=torch.float32, device=dev) for _ in range(n_buf)] if do_gather else None
    pending = [None] * n_buf
    enqueue_s = [0.0]
    n_calls = [0]

    def one_step(k=-1):
        if k >= 0 and feeds[k] is not None:
            te = time.perf_counter()
            taken = eng.enqueue_force_batch(*feeds[k])
            enqueue_s[0] += time.perf_counter() - te
            assert taken == feeds[k][0].size, "force queue overflow"
        slot = n_calls[0] % n_buf
        n_calls[0] += 1
        if pending[slot] is not None:
            pending[slot].wait()
            pending[slot] = None
        eng.step(nb, into=audios[slot].data_ptr())
        if do_gather:
            if backend == "nccl":
                pending[slot] = dist.all_gather_into_tensor(gathered[slot], audios[slot], async_op=True)
            else:
                gathered[slot].copy_(gather_audio(audios[slot].cpu()))

    def drain():
        for i, w in enumerate(pending):
            if w is not None:
                w.wait()
                pending[i] = None

    for k in range(args.settle + args.warmup):
        one_step(k)
    drain()
    torch.cuda.synchronize()
    info0 = eng.info()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    enqueue_s[0] = 0.0
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step(args.settle + args.warmup + k)
    drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    info1 = eng.info()
    assert all(torch.isfinite(a).all() for a in audios)
    if do_gather and backend == "nccl":
        last = (n_calls[0] - 1) % n_buf
        assert torch.equal(gathered[last][rank * args.objects:(rank + 1) * args.objects], audios[last])

    if rank == 0:
        total_obj = world * args.objects
        samples = total_obj * nb * B * args.steps
        value = samples / elapsed
        rt = (nb * B * args.steps / SAMPLE_RATE) / elapsed
        # roofline of the dominant kernel (iir_bank): HIP events on the launch stream, this rank
        k_ms = (info1["total_kernel_ms"] - info0["total_kernel_ms"]) / args.steps
        d_ms = (info1["total_device_ms"] - info0["total_device_ms"]) / args.steps
        plan_ms = (info1["total_host_plan_ms"] - info0["total_host_plan_ms"]) / args.steps
        M = args.modes
        mode_samples = args.objects * M * nb * B
        flops = FLOP_PER_MODE_SAMPLE * mode_samples
        # algorithmic bytes of one launch (SURVEY 8(d) formula with NB_l = nb; M = modes per object)
        bytes_alg = args.objects * (M * (12 + 8 + 8 + (4 + 4 + 4) * nb) + nb * B * (4 + 4))
        tf = flops / (k_ms * 1e-3) * 1e-12
        gbs = bytes_alg / (k_ms * 1e-3) * 1e-9
        out = {
            "metric": "audio samples/s & real-time x at N_obj x N_modes",
            "value": value, "unit": "audio samples/s", "realtime_x": rt,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_steps": args.settle,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{args.objects} objects x {M} modes per GPU, " + {
                                "impulses": "Poisson impulse stream (~20 PointForce hits/s/object, on-device vertex projection), unit transfer, ",
                                "scraping": "sustained AutoregressiveForce scraping (one GetModalForceFace message per buffer, "
                                            "profiles generated on the device), unit transfer, ",
                                "listener": "Poisson impulse stream + FFAT maps (16x16 cube faces) with a new listener position every buffer, ",
                            }[args.scenario] + f"{nb} buffers x 513 samples per step, "
                            f"qnorm {'off' if args.no_qnorm else args.qnorm}, {args.form} recurrence form",
                "scenario": args.scenario, "objects_per_gpu": args.objects, "modes": M, "buffers_per_step": nb, "frames_per_buffer": B,
                "hits": n_hits, "modes_per_lane": info1["modes_per_lane"], "waves_per_object": info1["waves_per_object"],
                "gather": bool(do_gather), "parallelism": f"object-sharded x{world}",
            },
            "roofline": {
                "bound": "valu", "kernel": "iir_bank_kernel",
                "bound_note": "fp32 vector-ALU issue (no dense contraction on this path, so neither hbm nor mfma binds); the peak "
                              "used, 157.3 TFLOP/s, is also the dense fp32 MFMA peak of MI355X_MICROARCH.md, so frac is the "
                              "same number under either label; the hbm fraction is in roofline.hbm",
                "achieved": tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / VALU_PEAK_TFLOPS,
                "flop_per_mode_sample": FLOP_PER_MODE_SAMPLE, "kernel_ms": k_ms,
                "hbm": {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_launch": bytes_alg},
                "traffic": measured_traffic(args),
                "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_pmc_traffic.json)",
            },
            "timing": {"device_pipeline_ms": d_ms, "host_plan_ms": plan_ms, "host_enqueue_ms": enqueue_s[0] / args.steps * 1e3},
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(args, lam, shapes, scripts)
            except Exception as ex:   # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(out))
    eng.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py - LSQR iterations/s (and cell.obs/s) of the MI355X-native Tomofast-x hot path on a synthetic gravity inversion.

  python bench.py [--gpus N --steps K --warmup W] [--workload NAME]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one LSQR iteration (the loop body of lsqr_solve_sensit, src/inversion/lsqr_solver2.F90:163-290) over the
wavelet-compressed sensitivity matrix, everything resident in HBM.  Setup (not timed, reported separately as build
cell.obs/s): prism rows -> column weights -> wavelet -> threshold -> tiled matrix, all on the GPU.
Strong scaling: the same problem on N GPUs, column-partitioned like the reference's MPI decomposition, two RCCL
all-reduces per iteration issued by libtfx.so itself on its stream (tfx_comm_init_rccl).  torch.distributed is the CONTROL
channel only - a gloo group that starts the ranks, carries the 128-byte communicator id, the agreement flags of the start-up
ladder and the timing reduction - so exactly one RCCL user lives in the process; if the library's communicator can not be
set up on every rank, all ranks fall back to the torch.distributed hooks together and the line says so (`comm`).

One JSON line on rank 0.  `roofline` is about the dominant kernel (compressed SpMV or its adjoint, whichever is slower):
`achieved` = the bytes the kernel's algorithm streams per launch (the stored entry streams + the staged vectors)
over its HIP-event duration, `traffic` = the HBM bytes rocprofv3's counters saw for the same launch (profiles/), both
against the 8 TB/s spec; the reference-CSR-equivalent rate (8 B per non-zero) is reported next to it as
`csr_equivalent_GBs`, not as the fraction.  `cpu_baseline` is the REAL reference (oracle/_ref/tomofastx, compiled from
/root/reference by oracle/ref_build.sh) under mpiexec on all host cores of this box on the SURVEY-6 synthetic size, with
the headline-size figure a stated linear extrapolation in nnz."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[4] - the configuration the metric is quoted on ("10^7-cell / 10^5-obs ... D4")
    "hamersley_1e7": dict(nx=256, ny=256, nz=152, ox=316, oy=316, ctype=2, rate=0.02,
                          desc="synthetic gravity 256x256x152 cells (9.96e6), 316x316 obs (99856), D4 wavelet r=0.02"),
    # BASELINE.json configs[2]
    "haar_512": dict(nx=512, ny=512, nz=128, ox=256, oy=256, ctype=1, rate=0.01,
                     desc="synthetic gravity 512x512x128 cells, 256x256 obs, Haar r=0.01"),
    # BASELINE.json configs[1]: dense (uncompressed) sensitivity
    "dense_256": dict(nx=256, ny=256, nz=64, ox=64, oy=64, ctype=0, rate=1.0,
                      desc="synthetic gravity 256x256x64 cells, 64x64 obs, dense (uncompressed) sensitivity"),
    # BASELINE.json configs[3] at its stated size on ONE GPU: a gravity and a magnetic (TMI) kernel on 512x512x128 cells, 256x256 data
    # each, in one LSQR (joint_inverse_problem.F90:547-554, :712-739); 2 x 2.2e10 non-zeros = 2 x 124 GB resident
    "joint_512": dict(nx=512, ny=512, nz=128, ox=256, oy=256, ctype=1, rate=0.01, joint=True,
                      desc="joint gravity + magnetic (TMI): 512x512x128 cells, 2 x 256x256 obs, Haar r=0.01, two kernels in one LSQR"),
    "joint_small": dict(nx=96, ny=96, nz=32, ox=48, oy=48, ctype=1, rate=0.02, joint=True,
                        desc="joint gravity + magnetic (TMI): 96x96x32 cells, 2 x 48x48 obs, Haar r=0.02 (reduced)"),
    # reduced sizes for quick checks (NOT the headline; bench prints which one ran)
    "medium": dict(nx=128, ny=128, nz=64, ox=64, oy=64, ctype=2, rate=0.02,
                   desc="synthetic gravity 128x128x64 cells, 64x64 obs, D4 r=0.02 (reduced)"),
    "small": dict(nx=64, ny=64, nz=32, ox=32, oy=32, ctype=1, rate=0.1,
                  desc="synthetic gravity 64x64x32 cells, 32x32 obs, Haar r=0.1 (SURVEY 6 CPU-baseline size)"),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured copy ceiling)
# What ONE rank's share of a P-GPU run of the headline workload costs per LSQR iteration on one MI355X, measured by building exactly
# that share at full size on a single GPU (tools/one_rank_share.py, profiles/r05_one_rank_share.jsonl; ms, lowest - highest rank): the
# product kernels + vector work WITHOUT any peer latency - a measured 1 / 2 / 4 / 8-GPU curve is read against these lower bounds.
EXPECTED_MS_PER_STEP_BOUND = {"workload": "hamersley_1e7", "1": [36.3, 36.3], "2": [18.2, 18.3], "4": [9.3, 9.8], "8": [4.7, 4.8],
                              "source": "tools/one_rank_share.py (one rank's share built at full size on one GPU, final round-5 code)",
                              "excludes": "all-reduce latency over xGMI (2 per iteration: 0.8 MB + 8 B), clock differences between GPUs"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=os.environ.get("TFX_BENCH_WORKLOAD", "hamersley_1e7"))
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-large", action="store_true", help="also time the reference in the DRAM-resident regime (cpu_baseline_large: a 3.4 GB kernel "
                    "built on the GPU and re-loaded by the reference from SENSIT files; its reload alone takes the reference 1 - 2 minutes per run, so "
                    "the default run quotes the round-6 measurement recorded in profiles/r06_cpu_baseline_large_probes.json instead)")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket the matrix kernels with HIP events")
    ap.add_argument("--full-select", action="store_true", help="A/B: thresholds by the full radix select instead of the band select")
    ap.add_argument("--selftest", action="store_true", help="only the N-GPU self-test (tomofast-x_amd/distributed.py::comm_selftest): every "
                    "collective shape of the path once, step by step with a per-step timeout; prints its JSON verdict and exits")
    ap.add_argument("--selftest-timeout", type=float, default=float(os.environ.get("TFX_SELFTEST_TIMEOUT", "90")), help="seconds per self-test step (generous: a false time-out would move a healthy RCCL run to the host-staged hooks)")
    args = ap.parse_args()
    if os.environ.get("TFX_BENCH_WATCHDOG"):
        # diagnostics for a run that does not come back (tests set it): after that many seconds every thread's Python stack goes to
        # stderr - the run itself continues
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["TFX_BENCH_WATCHDOG"]), repeat=False, exit=False)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # plain `python bench.py --gpus N`: start the N ranks ourselves, exactly as the driver would
            return self_launch(args.gpus)
        args.gpus = world
    import torch
    tfx = importlib.import_module("tomofast-x_amd")
    dist = None
    backend = None
    force_comm = os.environ.get("TFX_BENCH_FORCE_COMM") == "1"      # world size 1 through the whole multi-rank start-up (tests)
    if world > 1 or force_comm:
        import torch.distributed as dist
        # The process group is the control channel (gloo): the data path's collectives are RCCL inside libtfx.so.
        # TFX_BENCH_SHARE_GPU=1: rehearsal of the multi-rank path on a single-GPU box (all ranks on GPU 0 -> the ladder's hook rung);
        # TFX_BENCH_BACKEND=nccl: torch's own NCCL group as the default group (a second RCCL communicator in the process).
        backend = os.environ.get("TFX_BENCH_BACKEND", "gloo")
        if os.environ.get("TFX_BENCH_SHARE_GPU") == "1":
            local_rank = 0
        if local_rank >= torch.cuda.device_count():
            sys.stderr.write("[bench] rank %d: LOCAL_RANK %d but %d GPUs visible - sharing GPU %d\n" %
                             (rank, local_rank, torch.cuda.device_count(), local_rank % max(1, torch.cuda.device_count())))
            local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):      # one node: the control channel needs no routable interface
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")            # (the container's hostname may not resolve)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def log(msg):
        if rank == 0:
            sys.stderr.write("[bench] %s\n" % msg)
            sys.stderr.flush()

    w = WORKLOADS[args.workload]
    nx, ny, nz = w["nx"], w["ny"], w["nz"]
    N = nx * ny * nz
    xs, ys, zs = tfx.synthetic.observations(nx, ny, w["ox"], w["oy"])
    D = xs.size
    ctx = tfx.Context(local_rank)
    if args.full_select:
        ctx.debug_set("band_select_min_cells", -1)
    if os.environ.get("TFX_FWD_GROUP"):          # tuning knob: row blocks sharing one staged x tile in the forward product
        ctx.debug_set("fwd_group", int(os.environ["TFX_FWD_GROUP"]))
    if os.environ.get("TFX_ITEMS_PER_CU"):       # tuning knob: work items per CU the tile list is cut into
        ctx.debug_set("items_per_cu", int(os.environ["TFX_ITEMS_PER_CU"]))
    # (TFX_ADJ_COPY=0/1/2 is read by the library itself: transposed copy of the tiles for the adjoint product - never / always / when it fits)
    info = ctx.device_info()
    log("device %s, %d CUs, %.0f GB; workload %s" % (info["name"], info["cus"], info["hbm_bytes"] / 1e9, w["desc"]))
    # ---- the memory plan of one rank, BEFORE anything large is allocated: what the build, the relayout, the transposed copy and the
    # solve will need per phase (tomofast-x_amd/distributed.py::memory_plan); a run that can not fit stops here with the numbers
    # instead of dying inside a hipMalloc on one rank while the others wait in a collective
    adj_copy_env = os.environ.get("TFX_ADJ_COPY", "2")
    plan = tfx.distributed.memory_plan(N, D, w["rate"], world, nkernels=2 if w.get("joint") else 1, dense=w["ctype"] == 0,
                                       adjoint_copy=adj_copy_env != "0", hbm_bytes=info["hbm_bytes"],
                                       exchange=os.environ.get("TFX_BUILD_MODE", "exchange") == "exchange")
    log("memory plan per rank (x%d): %s GB per phase, peak %.1f of %.1f GB%s" %
        (world, json.dumps(plan["phases_GB"]), plan["peak_GB"], plan["hbm_GB"],
         "" if plan["adjoint_copy_fits"] or w["ctype"] == 0 or adj_copy_env == "0" else
         " (WITHOUT the transposed copy: with it %.1f GB - the adjoint will run on the tiles of S)" % plan["peak_with_adjoint_copy_GB"]))
    if not plan["fits"]:
        sys.exit("bench.py: %s does not fit %d x %.0f GB (peak %.1f GB per rank even without the transposed copy): use more GPUs" %
                 (args.workload, world, plan["hbm_GB"], plan["peak_GB"]))
    if w.get("joint"):
        if world > 1:
            sys.exit("the joint workloads run on one GPU")
        ctx.set_grid(nx, ny, nz, *tfx.synthetic.grid(nx, ny, nz))
        return bench_joint(args, w, ctx, tfx, log, plan)
    # collectives: an RCCL communicator inside libtfx.so (nccl launch) - the library queues its reductions on its own stream;
    # the gloo rehearsal uses the torch.distributed hooks
    comm = tfx.distributed.setup_comm(ctx, rank, world, local_rank, log=log, force=force_comm)
    log("collectives: %s" % json.dumps(comm.report))
    # ---- N-GPU self-test (N > 1, or --selftest): every collective shape of the path once on a small known-answer input, each step under
    # a timeout and agreed between the ranks - a call that has never run with real peers fails HERE and says which one, not in the timed
    # region.  A failure on the RCCL rung moves all ranks to the torch.distributed hooks together (the line then says so).
    selftest = None
    if args.selftest or world > 1 or (force_comm and os.environ.get("TFX_BENCH_SELFTEST") == "1"):
        selftest = tfx.distributed.comm_selftest(ctx, comm, rank, world, local_rank, step_timeout=args.selftest_timeout, log=log)
        if not selftest["ok"] and comm.rccl and not args.selftest:
            why = next((s.get("why") for s in selftest["steps"] if s.get("ok") is False), "unknown")
            if os.environ.get("TFX_COMM") == "rccl":
                raise RuntimeError("TFX_COMM=rccl but the N-GPU self-test failed: %s" % why)
            log("self-test failed on the RCCL rung (%s): all ranks fall back to the torch.distributed hooks" % why)
            comm = tfx.distributed.fall_back_to_hooks(ctx, comm, rank, world, local_rank, why)
        comm.report["selftest"] = selftest
        if args.selftest:
            if rank == 0:
                print(json.dumps({"selftest": selftest, "n_gpus": world, "comm": {k: v for k, v in comm.report.items() if k != "selftest"},
                                  "expected_ms_per_step_bound": EXPECTED_MS_PER_STEP_BOUND}))
                sys.stdout.flush()
            if comm.rccl:
                ctx.comm_destroy()
            ctx.close()
            if dist is not None:
                dist.destroy_process_group()
            sys.exit(0 if selftest["ok"] else 1)
    ctx.set_grid(nx, ny, nz, *tfx.synthetic.grid(nx, ny, nz))
    cw = ctx.calculate_depth_weight(2.0, 0.0, 4.0e3)

    def barrier():
        comm.barrier()

    # ---- build (setup; reported as cell.obs/s)
    barrier()
    t0 = time.time()
    # N > 1: row-parallel build + point-to-point relayout (every row computed once); TFX_BUILD_MODE=redundant selects the
    # simpler scheme where every rank rebuilds all rows for its column range (also used for uncompressed kernels)
    use_exchange = world > 1 and w["ctype"] > 0 and os.environ.get("TFX_BUILD_MODE", "exchange") == "exchange"
    if use_exchange:
        # the relayout needs point-to-point transfers: try a ring of tiny messages first and agree on the outcome, so that a
        # backend without working send / recv degrades to the redundant build instead of failing the run
        ok = 1
        try:
            dev = torch.device("cuda", local_rank)
            probe = torch.full((4,), float(rank), dtype=torch.float32, device=dev)
            got = torch.empty(4, dtype=torch.float32, device=dev)
            comm.exchange([((rank + 1) % world, probe)], [((rank - 1) % world, got)])
            ok = int(bool((got == float((rank - 1) % world)).all().item()))
        except Exception as exc:      # noqa
            log("point-to-point probe failed (%s): falling back to the redundant build" % exc)
            ok = 0
        agreed = int(comm.allreduce_host(np.array([ok], np.int64))[0])
        if agreed != world:
            use_exchange = False
    if use_exchange:
        part = tfx.distributed.build_partitioned_exchange(ctx, rank, world, xs, ys, zs, cw, w["ctype"], w["rate"], device_index=local_rank,
                                                          comm=comm)
        build_mode = "row-parallel + relayout" + (" (ncclSend / ncclRecv)" if comm.rccl else "")
    else:
        part = tfx.distributed.build_partitioned(ctx, rank, world, xs, ys, zs, cw, w["ctype"], w["rate"], comm=comm)
        build_mode = "direct" if world == 1 else "redundant rows per column range"
    barrier()
    t_build_total = time.time() - t0
    # the transposed copy of the tiles for the adjoint (made when it fits) is timed by the library: reported on its own,
    # `build_s` stays the kernel build that earlier rounds reported
    t_copy = ctx.debug_set("adj_copy_build_ms") / 1e3 if w["ctype"] > 0 else 0.0
    t_copy = comm.max_over_ranks(t_copy) if world > 1 else t_copy
    t_build = max(t_build_total - t_copy, 1e-9)
    minfo = ctx.matrix_info()
    free_b, total_b = torch.cuda.mem_get_info(local_rank)
    used_after_build = total_b - free_b                   # (matrix + copy + grid + whatever the allocator keeps: compare with the plan's `solve`)
    nnz_total = part["nnz_total"]
    if os.environ.get("TFX_CHUNK_SPAN"):         # diagnostics: value-exponent span of the stored chunks (printed by the library)
        log("chunks within %s binades: %d per mille" % (os.environ["TFX_CHUNK_SPAN"], ctx.debug_set("chunk_exponent_span", int(os.environ["TFX_CHUNK_SPAN"]))))
    c0, c1 = part["col_range"]
    log("build %.1f s (%.3e cell.obs/s), nnz %d, compression error %.3e, rank-0 matrix %.2f GB, cols [%d,%d)" %
        (t_build, N * D / t_build, nnz_total, part["comp_error"], minfo["device_bytes"] / 1e9, c0, c1))

    # ---- right-hand side: data of the synthetic block model, d = S Wav(m_true / cw)   (model.F90:220-307)
    mtrue = tfx.synthetic.true_model(nx, ny, nz)
    xw = ctx.forward_wavelet(mtrue / cw, nx, ny, nz, w["ctype"]) if w["ctype"] > 0 else mtrue / cw
    d_obs = ctx.calc_data(xw[c0:c1], 1.0, None)
    alpha = 1e-7
    ncl = c1 - c0
    diag = [np.full(ncl, np.float32(alpha), np.float32)]
    rhs = [np.zeros(ncl)]

    # ---- size-independent property at full size: <S x, y> = <x, S^T y>
    rng = np.random.default_rng(1 + rank)
    xr = rng.standard_normal(ncl)
    yr = np.random.default_rng(99).standard_normal(D)
    lhs = float(np.dot(ctx.mult_vector(xr), yr))
    rhs_dot = float(np.dot(xr, ctx.trans_mult_vector(yr)))
    adj_err = abs(lhs - rhs_dot) / max(abs(lhs), abs(rhs_dot), 1e-300)
    log("adjoint identity rel. mismatch %.2e" % adj_err)

    # ---- LSQR: W warm-up iterations, then exactly K timed iterations
    ctx.lsqr_begin(d_obs, 1e-300, 0.0, 0.0, diag, rhs)        # rmin tiny: never stop early inside the timed region
    done, r = ctx.lsqr_iterate(args.warmup)
    assert done == args.warmup, "LSQR stopped during warm-up (%d of %d)" % (done, args.warmup)
    if not args.no_profile:
        ctx.profile_enable(True)
    # The K steps are timed REPEATS times back to back (same solve, K more iterations each time), every repeat bracketed by a barrier +
    # device synchronisation on both sides and reduced with MAX over the ranks; the line reports the MEDIAN repeat (`value`,
    # `ms_per_step`) and all of them (`ms_per_step_runs`): the boxes of the pool differ by +-5 % and one 0.75 s sample says nothing
    # about a 1 % change.
    REPEATS = 3
    runs, runs_local, runs_gpu = [], [], []
    for _ in range(REPEATS):
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.timer_start()
        done, r = ctx.lsqr_iterate(args.steps)
        ms_gpu_i = ctx.timer_stop_ms()
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        assert done == args.steps, "LSQR stopped early (%d of %d)" % (done, args.steps)
        runs_local.append(dt)
        runs_gpu.append(ms_gpu_i)
        runs.append(comm.max_over_ranks(dt) if world > 1 else dt)
    prof = [ctx.profile_get(0), ctx.profile_get(1), ctx.profile_get(2)] if not args.no_profile else [(0.0, 0), (0.0, 0), (0.0, 0)]
    ctx.profile_enable(False)
    x_final = ctx.lsqr_end()
    # ---- the residual LSQR reports against the residual of the augmented system recomputed from the products (outside the timed region):
    # r = |[b - S x ; -alpha x]| / |b| (lsqr_solver2.F90:163-290; tfx_calc_data all-reduces S x over the ranks)
    r_check = None
    try:
        sx = ctx.calc_data(x_final, 1.0, None)
        damp2 = float(np.sum((float(np.float32(alpha)) * x_final) ** 2))      # (the damping block is stored in fp32)
        if world > 1:
            damp2 = float(comm.allreduce_host(np.array([damp2]))[0])
        r_true = float(np.sqrt(np.sum((d_obs - sx) ** 2) + damp2) / np.linalg.norm(d_obs))
        r_check = {"r_from_products": r_true, "rel_err": abs(r - r_true) / r_true if r_true > 0 else None,
                   "iterations": args.warmup + REPEATS * args.steps}
        log("final r %.15e, recomputed from the products %.15e (relative difference %.1e)" % (r, r_true, r_check["rel_err"] or 0.0))
    except Exception as exc:      # noqa  (a diagnostic: it must not take the line down)
        log("residual check skipped: %r" % (exc,))
    mid = int(np.argsort(runs)[len(runs) // 2])
    t_steps, t_steps_local, ms_gpu = runs[mid], runs_local[mid], runs_gpu[mid]
    ms_per_step = 1e3 * t_steps / args.steps
    value = args.steps / t_steps

    # ---- roofline of the dominant kernel (rank 0's share of the matrix)
    nnz_loc = minfo["nnz"]
    names = ["k_spmv_fwd (compressed SpMV, b += S x)", "k_spmv_adj (compressed SpMtV, b += S^T x)"]
    if w["ctype"] > 0 and ctx.matrix_format().get("adjoint_copy"):
        names[1] = "k_spmv_fwd on the transposed copy of the tiles (compressed SpMtV, b += S^T x)"
    if w["ctype"] == 0:
        names = ["k_dense_fwd (dense fp32 block, b += S x)", "k_dense_adj (dense fp32 block, b += S^T x)"]
    roof = None
    # bytes per stored entry of the compressed layout: value + column stream + row-start bit (what tfx_matrix_info's stream sizes say;
    # DESIGN.md 3).  The kernel that runs the adjoint may stream a second, transposed copy of the same size.
    fmt = ctx.matrix_format() if w["ctype"] > 0 else {}
    bytes_per_entry = fmt.get("bytes_per_entry", 4.0)
    if prof[0][1] and prof[1][1]:
        avg = [prof[0][0] / prof[0][1], prof[1][0] / prof[1][1]]
        dom = 0 if avg[0] >= avg[1] else 1
        # Algorithmic bytes of one launch = what the kernel's algorithm has to stream: the stored entry streams
        # (compressed: 4 B value + 1.5 B column slot + 1 bit row start = 5.625 B per non-zero; dense block: 4 B per entry) plus the
        # vector on the streaming side once per row-block pass.  The reference's CSR moves 8 B per non-zero: that
        # rate is reported as csr_equivalent_GBs and is NOT the roofline fraction.
        if w["ctype"] == 0:
            alg_bytes = 4.0 * nnz_loc + 8.0 * (ncl + D)
            csr_b = 4.0
        else:
            alg_bytes = bytes_per_entry * nnz_loc + 8.0 * (ncl + D)
            csr_b = 8.0
        achieved = alg_bytes / (avg[dom] * 1e-3) / 1e9
        # the adjoint on the transposed copy IS k_spmv_fwd (second launch of an iteration): its PMC row is keyed "k_spmv_fwd_on_copy"
        adj_kernel = "k_spmv_fwd_on_copy" if fmt.get("adjoint_copy") else "k_spmv_adj"
        traffic, traffic_src = pmc_traffic(args.workload, "k_spmv_fwd" if dom == 0 else adj_kernel, nnz_loc, minfo["device_bytes"])
        roof = {"bound": "hbm", "kernel": names[dom], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": int(alg_bytes), "device_bytes_of_the_matrix": minfo["device_bytes"],
                "stored_bytes_per_entry": bytes_per_entry, "matrix_format": fmt,
                "avg_launch_ms": {"spmv_fwd": round(avg[0], 4), "spmv_adj": round(avg[1], 4)},
                # the same launch on the bytes that crossed HBM by PMC (>= the algorithmic bytes: tile padding, row markers, staging)
                "frac_on_traffic": None if traffic is None else round(traffic / (avg[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "traffic_over_algorithmic": None if traffic is None else round(traffic / alg_bytes, 4),
                "streaming_ceiling_GBs": 6290.0,     # MI355X_MICROARCH.md: measured float4 copy (read + write)
                # tools/read_bw_probe.hip on this part: a single contiguous read stream / the kernels' three streams without arithmetic
                "read_stream_ceiling_GBs": 7000.0, "three_stream_read_ceiling_GBs": 6700.0,
                # SURVEY 8d's own definition (the reference's CSR: 8 B per non-zero) next to the stored-bytes fraction; it can exceed 1
                # because this layout stores 5.625 B per non-zero - it is the reference-equivalent rate, never the roofline fraction
                "csr_equivalent_GBs": round(csr_b * nnz_loc / (avg[dom] * 1e-3) / 1e9, 1),
                "frac_csr_8B": round(csr_b * nnz_loc / (avg[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

    # ---- CPU baseline (rank 0, N = 1): the oracle on a bounded sample of the same workload
    cpu = None
    ref_cfg1 = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline_reference(tfx, int(nnz_total), N * D, log)
        if cpu is None:                     # no compiled reference / launcher on this box: the C port on one core
            cpu = cpu_baseline(tfx, w, args.cpu_seconds, cw, log)
        else:
            # The same reference in the DRAM-resident regime (a kernel the GPU built and wrote in the reference's file format): measured in
            # round 6 (tools/cpu_baseline_large_probe.py) and RECORDED - the reference needs 66 - 266 s to re-load such a kernel, per run
            try:
                rec = json.load(open(os.path.join(ROOT, "profiles", "r06_cpu_baseline_large_probes.json")))
                m0 = cpu.get("measured", {})
                cpu["large_recorded"] = {"source": "profiles/r06_cpu_baseline_large_probes.json (round 6, 256-core host of an MI355X box; NOT measured in this run: --cpu-large does)",
                                         "probes": rec["probes"],
                                         "this_run_small_leg": {"nnz": m0.get("nnz"), "ms_per_lsqr_iteration": m0.get("ms_per_lsqr_iteration")}}
            except Exception as e:      # noqa
                log("large_recorded not attached: %r" % (e,))
        if cpu is not None and cpu.get("kind") == "reference" and args.cpu_large:
            m = cpu.get("measured", {})
            large = cpu_baseline_large(tfx, int(nnz_total), m.get("ms_per_lsqr_iteration"), m.get("nnz"), log, device_index=local_rank)
            cpu["large"] = large
            if large and large.get("value_measured_large"):
                cpu["value_measured_large"] = large["value_measured_large"]
                cpu["value_measured_large_at"] = dict(large["at"], ranks=large["ranks"])
                cpu["value_from_the_large_point"] = large["headline_from_this_point"]["iterations_per_s"]
                cpu["large_measured_over_small_extrapolated"] = large["measured_over_extrapolated"]
        ref_cfg1 = reference_config1(log)

    # ---- per-rank facts (N > 1): product times, the event-timed all-reduces, matrix share, wall clock of the timed region
    mine = np.array([prof[0][0] / max(prof[0][1], 1), prof[1][0] / max(prof[1][1], 1), prof[2][0] / max(prof[2][1], 1), float(prof[2][1]),
                     float(nnz_loc), 1e3 * t_steps_local / args.steps, float(ms_gpu / args.steps)])
    per_rank = comm.allgather_host(mine) if world > 1 else [mine]
    stored_iter = None
    if w["ctype"] > 0:
        stored_iter = int(2 * bytes_per_entry * int(nnz_total) + 112 * N + 48 * (D + N))
    if rank == 0:
        out = {
            "metric": "LSQR iterations/s, synthetic gravity inversion (wavelet-compressed sensitivity kernel)",
            "value": round(value, 4), "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "ms_per_step_runs": [round(1e3 * t / args.steps, 4) for t in runs], "timed_repeats": REPEATS,
            "ms_per_step_spread": round(1e3 * (max(runs) - min(runs)) / args.steps, 4),
            "dtype": dtype_line(w, fmt), "data": "synthetic",
            "config": {"workload": args.workload + ": " + w["desc"], "cells": N, "obs": D, "nnz": int(nnz_total),
                       "compression": {0: "none", 1: "haar", 2: "d4"}[w["ctype"]], "rate": w["rate"],
                       "parallelism": "column-partitioned x%d" % world, "damping_alpha": alpha},
            "cell_obs_per_s_solve": round(N * D * value, 1),
            # the build as a default run pays it: wall clock from the first row to a matrix ready for both products, INCLUDING the transposed
            # copy the default adjoint runs on; the copy's share (timed by the library) and the remainder are secondary fields
            "cell_obs_per_s_build": round(N * D / t_build_total, 1), "build_s": round(t_build_total, 2), "build_mode": build_mode,
            "adjoint_copy_build_s": round(t_copy, 2), "build_without_adjoint_copy_s": round(t_build, 2),
            "cell_obs_per_s_build_without_adjoint_copy": round(N * D / t_build, 1),
            "build_threshold_batches": {"band_select": ctx.debug_set("band_batches"), "fell_back_to_full_select": ctx.debug_set("band_fallbacks")},
            "gpu_ms_per_step_hip_events": round(ms_gpu / args.steps, 4),
            # SURVEY 8d's formula on the REFERENCE's CSR (8 B per non-zero and pass): a CSR-equivalent figure like csr_equivalent_GBs, not
            # what this layout streams; `_stored` is (2 passes over the stored streams + the vector sweeps) and must stay below
            # 8 TB/s x ms_per_step x n_gpus
            "lsqr_bytes_per_iteration_csr_equivalent": 16 * int(nnz_total) + 112 * N + 48 * (D + N),
            "lsqr_bytes_per_iteration_stored": stored_iter,
            "lsqr_stored_GBs": None if stored_iter is None else round(stored_iter / (ms_per_step * 1e-3) / 1e9, 1),
            "comm": comm.report, "memory_plan": plan,
            "expected_ms_per_step_bound": EXPECTED_MS_PER_STEP_BOUND if args.workload == "hamersley_1e7" else None,
            "device_memory_used_after_build_GB": round(used_after_build / 1e9, 2),
            "per_rank": [{"rank": r, "spmv_fwd_ms": round(float(v[0]), 4), "spmv_adj_ms": round(float(v[1]), 4),
                          "allreduce_ms": round(float(v[2]), 4), "allreduces_timed": int(v[3]), "nnz": int(v[4]),
                          "ms_per_step_wall": round(float(v[5]), 4), "ms_per_step_hip_events": round(float(v[6]), 4)}
                         for r, v in enumerate(per_rank)],
            "adjoint_identity_rel_err": adj_err, "final_r": r, "final_r_check": r_check,
            "roofline": roof, "cpu_baseline": cpu, "reference_config1": ref_cfg1,
        }
        print(json.dumps(out))
        sys.stdout.flush()
    if comm.rccl:
        ctx.comm_destroy()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


def self_launch(nproc):
    """`python bench.py --gpus N` without a launcher: re-run this command line under torch.distributed.run (one rank per GPU,
    rendezvous on 127.0.0.1 at a free port) and pass its exit status on.  On a box with fewer than N GPUs the ranks share them
    (main() maps LOCAL_RANK onto the visible devices) and the start-up ladder ends on its hook rung - the line says so in `comm`."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("[bench] --gpus %d without a launcher: %s\n" % (nproc, " ".join(cmd)))
    sys.stderr.flush()
    sys.exit(subprocess.call(cmd))


def bench_joint(args, w, ctx, tfx, log, plan=None):
    """BASELINE configs[3]: two sensitivity kernels (gravity g_z and magnetic TMI) on one grid inside one LSQR - S = blockdiag(S_grav,
    S_mag), unknowns [m_grav ; m_mag], one damping block spanning both (joint_inverse_problem.F90:547-554, :712-739).  A step is one
    LSQR iteration of the joint system: two launches per product.  Oracle-free properties here (entry counts, adjoint identity,
    LSQR residual against the products); rows against the oracle: tests/test_gpu_parity.py::test_joint_two_kernels_*."""
    nx, ny, nz = w["nx"], w["ny"], w["nz"]
    N = nx * ny * nz
    xs, ys, zs = tfx.synthetic.observations(nx, ny, w["ox"], w["oy"])
    D = xs.size
    K = int(w["rate"] * N)
    field = np.array([-62.0, 11.0, 0.0, 57000.0])
    pws = (1.0, 3.0e-3)
    cws = [ctx.calculate_depth_weight(2.0, 0.0, 4.0e3), ctx.calculate_depth_weight(3.0, 0.0, 1.0)]
    rng = np.random.default_rng(3)
    kern, b, t_build = [], [], 0.0
    for i in range(2):
        ctx.select_problem(i)
        t0 = time.time()
        res = ctx.calculate_sensit(xs, ys, zs, cws[i], w["ctype"], w["rate"], problem_weight=pws[i], mag_field=field if i == 1 else None)
        dt = time.time() - t0
        t_build += dt
        assert 0.9999 * K * D <= res["nnz"] <= K * D, (res["nnz"], K * D)
        x, y = rng.standard_normal(N), rng.standard_normal(D)
        Sx, STy = ctx.mult_vector(x), ctx.trans_mult_vector(y)
        adj = abs(np.dot(Sx, y) - np.dot(x, STy)) / (np.linalg.norm(Sx) * np.linalg.norm(y))
        info, fmt = ctx.matrix_info(), ctx.matrix_format()
        kern.append({"problem": "gravity g_z" if i == 0 else "magnetic TMI", "nnz": int(res["nnz"]), "build_s": round(dt, 2),
                     "cell_obs_per_s_build": N * D / dt, "device_bytes": int(info["device_bytes"]), "adjoint_identity_rel_err": float(adj),
                     "bytes_per_entry": fmt["bytes_per_entry"], "adjoint_copy": fmt["adjoint_copy"]})
        log("kernel %d (%s): build %.1f s, nnz %d, %.1f GB, adjoint identity %.1e" % (i, kern[-1]["problem"], dt, res["nnz"], info["device_bytes"] / 1e9, adj))
        b.append(ctx.mult_vector(rng.standard_normal(N) * 1e-3))
    for i in range(2):                     # (the first kernel's automatic copy may have been given up for the second kernel's storage)
        ctx.select_problem(i)
        kern[i]["adjoint_copy"] = ctx.matrix_format()["adjoint_copy"]
    ctx.select_problem(0)
    assert ctx.system_dims() == (2 * D, 2 * N)
    alpha = np.concatenate([np.full(N, 1e-6, np.float32), np.full(N, 2e-6, np.float32)])
    rhs = np.concatenate(b)
    ctx.lsqr_begin(rhs, 1e-300, 0.0, 0.0, [alpha], [np.zeros(2 * N)])
    done, r = ctx.lsqr_iterate(args.warmup)
    assert done == args.warmup
    ctx.profile_enable(True)
    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.timer_start()
    done, r = ctx.lsqr_iterate(args.steps)
    ms_gpu = ctx.timer_stop_ms()
    torch.cuda.synchronize()
    t_steps = time.perf_counter() - t0
    assert done == args.steps
    prof = [ctx.profile_get(0), ctx.profile_get(1)]
    ctx.profile_enable(False)
    ctx.lsqr_end()
    per_it = [prof[0][0] / args.steps, prof[1][0] / args.steps]          # both kernels' launches of a product, per iteration
    dom = 0 if per_it[0] >= per_it[1] else 1
    nnz = sum(k["nnz"] for k in kern)
    alg_bytes = kern[0]["bytes_per_entry"] * nnz + 8.0 * 2 * (N + D)
    achieved = alg_bytes / (per_it[dom] * 1e-3) / 1e9
    value = args.steps / t_steps
    out = {"metric": "LSQR iterations/s, joint gravity + magnetic inversion (two wavelet-compressed sensitivity kernels in one system)",
           "value": round(value, 4), "unit": "iterations/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(1e3 * t_steps / args.steps, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": dtype_line(w, {"adjoint_copy": all(k["adjoint_copy"] for k in kern)}), "data": "synthetic",
           "config": {"workload": args.workload + ": " + w["desc"], "cells": N, "obs": [D, D], "nnz": nnz, "compression": "haar", "rate": w["rate"],
                      "parallelism": "both kernels on one GPU"},
           "cell_obs_per_s_solve": round(2.0 * N * D * value, 1), "cell_obs_per_s_build": round(2.0 * N * D / t_build, 1), "build_s": round(t_build, 2),
           "gpu_ms_per_step_hip_events": round(ms_gpu / args.steps, 4), "final_r": r, "kernels": kern,
           "roofline": {"bound": "hbm", "kernel": ["k_spmv_fwd", "k_spmv_adj"][dom] + " (the two kernels' launches of one product)",
                        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                        "algorithmic_bytes_per_launch": int(alg_bytes), "ms_per_iteration": {"spmv_fwd": round(per_it[0], 4), "spmv_adj": round(per_it[1], 4)}},
           "memory_plan": plan, "cpu_baseline": None}
    print(json.dumps(out))
    sys.stdout.flush()
    ctx.close()


def reference_config1(log):
    """BASELINE config 1 (parfiles/Parfile_mansf_slice.txt, 2x128x32 cells, 256 data, 60 x 100 LSQR iterations) end to end:
    the REAL reference (oracle/_ref/tomofastx, built from /root/reference by oracle/ref_build.sh; 1 MPI rank = 1 host core)
    next to this repo's Fortran host on the GPU (tomofast-x_amd/host/tomofastx_amd), same Parfile, same input files
    (written from tests/golden/mansf.npz).  Wall-clock of the whole program, I/O included.  None when a binary is missing."""
    import shutil
    import subprocess
    import tempfile
    ref = os.path.join(ROOT, "oracle", "_ref", "tomofastx")
    ours = os.path.join(ROOT, "tomofast-x_amd", "host", "tomofastx_amd")
    mpiexec = "/opt/conda/bin/mpiexec"
    if not (os.path.isfile(ref) and os.path.isfile(ours) and os.path.isfile(mpiexec)):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        import test_gpu_fortran_host as tf
        g = np.load(os.path.join(ROOT, "tests", "golden", "mansf.npz"))
        out = {"config": "parfiles/Parfile_mansf_slice.txt (BASELINE configs[0])", "cores": 1}
        for name, cmd in (("reference_s", [mpiexec, "-n", "1", ref, "-p", "Parfile.txt"]), ("gpu_host_s", [ours, "-p", "Parfile.txt"])):
            wd = tempfile.mkdtemp(prefix="tfx_cfg1_")
            try:
                tf.write_inputs(wd, g)
                t0 = time.time()
                p = subprocess.run(cmd, cwd=wd, capture_output=True, text=True, timeout=600)
                dt = time.time() - t0
                if p.returncode != 0 or "THE END." not in p.stdout:
                    log("config-1 run failed: %s" % " ".join(cmd))
                    return None
                out[name] = round(dt, 3)
                m = np.loadtxt(os.path.join(wd, "output", "mansf_slice", "model", "grav_final_model_full.txt"), skiprows=1)
                out[name.replace("_s", "_model_rel_l2_vs_golden")] = float(np.linalg.norm(m - g["model_final"]) / np.linalg.norm(g["model_final"]))
            finally:
                shutil.rmtree(wd, ignore_errors=True)
        log("config 1 end to end: reference %.2f s (1 core) vs GPU host %.2f s" % (out["reference_s"], out["gpu_host_s"]))
        return out
    except Exception as e:      # the baseline leg must never take the benchmark down
        log("reference_config1 skipped: %r" % (e,))
        return None


CONV_MAJOR = 3        # major iterations of the converged parity leg of cpu_baseline.reference_medium


def dtype_line(w, fmt):
    """The arithmetic of the path: fp32-stored matrix values, fp64 vectors; fp64 sums in both products when the adjoint runs on the
    transposed copy, 61-bit fixed-point column sums (exact integer accumulation per tile group) when it runs on the tiles of S."""
    if w["ctype"] == 0:
        return "f64 (fp32-stored dense block, fp64 vectors and accumulation)"
    if fmt.get("adjoint_copy"):
        return "f64 (fp32-stored matrix values, fp64 vectors; fp64 accumulation in both products: the adjoint is the forward kernel on the transposed copy)"
    return ("f64 (fp32-stored matrix values, fp64 vectors; forward product: fp64 accumulation; adjoint WITHOUT a transposed copy: 61-bit fixed-point "
            "column sums per tile group - every product rounded once to a 2^-60 grid of the group's bound, added exactly in 64-bit integers)")


def cpu_baseline_reference(tfx, nnz_headline, pairs_headline, log, nx=64, ny=64, nz=32, ox=32, oy=32, ctype=1, rate=0.1):
    """The compiled reference itself (oracle/_ref/tomofastx = /root/reference built by oracle/ref_build.sh; it travels to the
    GPU box as a binary) under `mpiexec -n <all host cores>` on the SURVEY-6 synthetic size (64x64x32 cells x 32x32 data,
    Haar r = 0.1, the same generator as the GPU workload): run A builds the kernel and does 1 x 1 LSQR iteration, run B
    re-loads the kernel from A's SENSIT files and does 1 x 101; build rate = N.D / (A - reload part), LSQR time per
    iteration = (B - B0) / 100 where B0 is B with 1 iteration.  The headline-size figure is a LINEAR EXTRAPOLATION in nnz
    (LSQR) resp. in cell.obs pairs (build) and is labelled so.  None when the binary or the launcher is missing."""
    import shutil
    import subprocess
    import tempfile
    ref = os.path.join(ROOT, "oracle", "_ref", "tomofastx")
    mpiexec = "/opt/conda/bin/mpiexec"
    if not (os.path.isfile(ref) and os.path.isfile(mpiexec)):
        return None
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    if cores >= 32:            # enough cores for the larger SURVEY-6 size (128x128x32 cells, nnz 2.7e7) inside the time budget
        nx, ny, rate = 128, 128, 0.05
    nd = ox * oy
    N = nx * ny * nz
    # The build is row-parallel and keeps scaling; the reference's LSQR all-reduces nlines doubles per iteration and its kernel
    # reload is rank-0-serial, so more ranks are not faster there: the build runs on up to 64 ranks (beyond that the 1024-row
    # build has < 16 rows per rank and start-up dominates), the LSQR leg is timed at 64 and at 16 ranks and the faster one counts.
    build_ranks = max(1, min(cores, 64, nd // 4))
    lsqr_ranks = max(1, min(cores, 16))
    lsqr_rank_counts = [lsqr_ranks]
    wd = tempfile.mkdtemp(prefix="tfx_refcpu_")
    try:
        def run(ranks, nminor, sensit_read, nmajor=1):
            tfx.synthetic.write_parfile_inputs(wd, nx, ny, nz, ox, oy, ctype, rate, nmajor=nmajor, nminor=nminor, sensit_read=sensit_read)
            t0 = time.time()
            p = subprocess.run([mpiexec, "-n", str(ranks), ref, "-p", "Parfile.txt"], cwd=wd, capture_output=True, text=True, timeout=900)
            dt = time.time() - t0
            if p.returncode != 0 or "THE END." not in p.stdout:
                raise RuntimeError("reference run failed: " + p.stdout[-400:] + p.stderr[-400:])
            return dt, p.stdout
        tA, outA = run(build_ranks, 1, 0)
        nnz = int(outA.split("nnz_total =")[1].split()[0])
        # what the build run spent outside the build: the same run re-loading the kernel instead of computing it
        t_reload_build_ranks, _ = run(build_ranks, 1, 1) if build_ranks != lsqr_ranks else (None, None)
        legs = {}
        for rk in lsqr_rank_counts:
            # (101 - 1) iterations as the difference of two runs: the 1-iteration run is repeated, its spread is the noise floor of
            # that difference.  (Round 3 also timed a leg at the build's 64 ranks: 92 ms per iteration against 6 - the reference
            # all-reduces every row per iteration - and 50 s of wall clock; dropped.)
            t0a, _ = run(rk, 1, 1)
            t0b, _ = run(rk, 1, 1)
            t1_, _ = run(rk, 101, 1)
            ref_timing_leg = collect_parfile_outputs(wd)      # the reference's 1 x 101-iteration run (the timing leg; NOT converged)
            # The parity leg: a CONVERGED inversion (CONV_MAJOR x 100 iterations on the same kernel files).  After one major iteration both
            # costs still sit near 1e-7 and last-bit differences of the sums are amplified by the Golub-Kahan recurrence; three major
            # iterations re-start LSQR from the updated residual and the solutions contract onto each other.
            run(rk, 100, 1, nmajor=CONV_MAJOR)
            ref_medium = collect_parfile_outputs(wd)
            ref_medium["timing_leg"] = ref_timing_leg
            if rk > 1:
                # the same inversion on half the ranks: the reference's OWN distance between two rank counts on this problem (the sums of
                # its products and norms are ordered differently) - the yardstick for the GPU host's distance in reference_medium
                run(max(1, rk // 2), 100, 1, nmajor=CONV_MAJOR)
                other = collect_parfile_outputs(wd)
                ref_medium["own_scatter"] = {
                    "ranks": [rk, max(1, rk // 2)],
                    "model_rel_l2": float(np.linalg.norm(other["model"] - ref_medium["model"]) / np.linalg.norm(ref_medium["model"])),
                    "data_cost": [ref_medium["data_cost"], other["data_cost"]],
                    "data_cost_ratio": other["data_cost"] / ref_medium["data_cost"]}
            base, noise = min(t0a, t0b), abs(t0a - t0b)
            diff = t1_ - base
            legs[rk] = {"reload_and_1_iteration_s": base, "reload_and_1_iteration_repeat_spread_s": noise, "reload_and_101_iterations_s": t1_,
                        "ms_per_lsqr_iteration": 1e3 * max(diff, 1e-9) / 100.0, "resolved": bool(diff > 3.0 * noise and diff > 0.02 * base)}
            if not legs[rk]["resolved"] and base < 6.0:
                # a noisy host (the box is shared): four times the iterations on the leg that is cheap to repeat
                t4_, _ = run(rk, 401, 1)
                diff4 = t4_ - base
                legs[rk].update({"reload_and_401_iterations_s": t4_, "ms_per_lsqr_iteration": 1e3 * max(diff4, 1e-9) / 400.0,
                                 "resolved": bool(diff4 > 3.0 * noise and diff4 > 0.02 * base)})
        if t_reload_build_ranks is None:
            t_reload_build_ranks = legs[lsqr_ranks]["reload_and_1_iteration_s"]
        usable = [rk for rk in legs if legs[rk]["resolved"]] or list(legs)
        best = min(usable, key=lambda rk: legs[rk]["ms_per_lsqr_iteration"])
        t_iter = legs[best]["ms_per_lsqr_iteration"] * 1e-3
        t_build = max(tA - t_reload_build_ranks, 1e-9)   # A = inputs + build + write + reload + 1 iteration
        out = {"value": 1.0 / (t_iter * nnz_headline / nnz), "unit": "iterations/s", "cores": best,      # the ranks `value` was measured on
               # `value` is an EXTRAPOLATION (linear in nnz) to the headline matrix; what was measured, at the size it was measured on:
               "value_is": "linear extrapolation in nnz of value_measured to the headline matrix (nnz %d)" % nnz_headline,
               "value_measured": 1.0 / t_iter, "value_measured_at": {"nnz": nnz, "cells": N, "obs": nd, "ranks": best},
               "build_cores": build_ranks, "host_cores": cores, "kind": "reference",
               "sample": "oracle/_ref/tomofastx (the compiled reference) under mpiexec on %dx%dx%d cells x %d data, Haar r = %g "
                         "(box: %d host cores): kernel build on %d ranks %.3e cell.obs/s; LSQR %.2f ms per iteration at nnz = %d on "
                         "%d ranks (%s; the "
                         "reference's per-iteration MPI_Allreduce of all rows does not scale further: 92 ms per iteration at 64 ranks in round 3); `value` is the LINEAR EXTRAPOLATION in nnz of that iteration time to the headline matrix "
                         "(the reference cannot hold / finish that size on a host)" %
                         (nx, ny, nz, nd, rate, cores, build_ranks, N * nd / t_build, 1e3 * t_iter, nnz, best, lsqr_rank_counts),
               "measured": {"cells": N, "obs": nd, "nnz": nnz, "ms_per_lsqr_iteration": 1e3 * t_iter, "lsqr_ranks": best,
                            "iterations_per_s": 1.0 / t_iter, "build_s": t_build, "build_ranks": build_ranks,
                            "build_cell_obs_per_s": N * nd / t_build, "build_and_1_iteration_wall_s": tA,
                            "reload_and_1_iteration_at_build_ranks_s": t_reload_build_ranks,
                            "lsqr_legs_by_ranks": {str(k): v for k, v in legs.items()}},
               "extrapolated_headline": {"ms_per_lsqr_iteration": 1e3 * t_iter * nnz_headline / nnz,
                                         "build_s": pairs_headline / (N * nd / t_build)}}
        log("cpu baseline (reference; box has %d cores): build %.3e cell.obs/s on %d ranks, %.2f ms / LSQR iteration at nnz %d on %d ranks" %
            (cores, N * nd / t_build, build_ranks, 1e3 * t_iter, nnz, best))
        out["reference_medium"] = reference_medium(tfx, wd, ref_medium, (nx, ny, nz, ox, oy, ctype, rate), log)
        return out
    except Exception as e:      # the baseline leg must never take the benchmark down
        log("cpu_baseline_reference skipped: %r" % (e,))
        return None
    finally:
        shutil.rmtree(wd, ignore_errors=True)


def timed_reference_run(cmd, cwd, timeout=900):
    """Runs the compiled reference; -> dict(wall_s, stdout).  (Stamping the arrival of its own log lines - "Entered subroutine
    lsqr_solve_sensit" ... "Finished lsqr solver" - does not time the solve: the Fortran runtime buffers unit 6 on a pipe and the whole log
    arrives at exit; measured here, round 6.  The solve is therefore timed as the difference of two runs.)"""
    import subprocess
    t0 = time.time()
    p = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, timeout=timeout)
    if p.returncode != 0 or "THE END." not in p.stdout:
        raise RuntimeError("reference run failed: " + p.stdout[-400:] + p.stderr[-400:])
    return dict(wall_s=time.time() - t0, stdout=p.stdout)


def cpu_baseline_large(tfx, nnz_headline, small_ms_per_iteration, small_nnz, log, device_index=0, rank_counts=(64,), iterations=51,
                       nx=128, ny=128, nz=32, ox=64, oy=32, ctype=1, rate=0.4, wall_budget_s=60.0):
    """The reference's LSQR in the DRAM-RESIDENT regime: the small leg's matrix is 13 MB per rank - cache-resident -
    and its iteration time is extrapolated 741 x to the headline.  Here the GPU builds a kernel of 128x128x32 cells x 128x128 data, Haar
    r = 0.05 (nnz 4.3e8 = 3.4 GB as the reference stores it, 27 - 54 MB per rank at 128 - 64 ranks: beyond the L2s, at or past the L3
    slices; 820 non-zeros per cell - the headline has 2000, the review's own suggestion of 256x256x64 cells x 64x64 data has 82 and was
    measured once: 443 ms per iteration at 64 ranks, its 117 s of start-up per run does not fit a default bench run), writes it as
    reference-format SENSIT files (sensit.readFromFiles = 1, problem_joint_gravmag.F90:172-202,
    sensitivity_gravmag.F90:648-883) - no 8.6e9-pair CPU build - and the compiled reference solves 1 x `iterations` on it under mpiexec
    at each rank count (difference of a 1 x `iterations` and a 1 x 1 run).  Reported next to the linear
    extrapolation of the small leg to THIS matrix: the ratio says how much the cache-resident point flatters the CPU.
    (The reference all-reduces its whole right-hand side - data rows AND the N damping rows - every iteration, lsqr_solver2.F90:214: its
    iteration time grows with the cell count as well as with nnz.)"""
    import shutil
    import tempfile
    ref = os.path.join(ROOT, "oracle", "_ref", "tomofastx")
    mpiexec = "/opt/conda/bin/mpiexec"
    if not (os.path.isfile(ref) and os.path.isfile(mpiexec)):
        return None
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    rank_counts = [r for r in rank_counts if r <= cores]
    if not rank_counts:
        return {"skipped": "needs >= %d host cores, this box has %d" % (min(rank_counts or [64]), cores)}
    t_leg = time.time()
    wd = tempfile.mkdtemp(prefix="tfx_refcpu_large_")
    try:
        N, D = nx * ny * nz, ox * oy
        ctx = tfx.Context(device_index)
        try:
            ctx.set_grid(nx, ny, nz, *tfx.synthetic.grid(nx, ny, nz))
            xs, ys, zs = tfx.synthetic.observations(nx, ny, ox, oy)
            cw = ctx.calculate_depth_weight(2.0, 0.0, 4.0e3)
            t0 = time.time()
            res = ctx.calculate_sensit(xs, ys, zs, cw, ctype, rate, want_hist=True)
            t_gpu_build = time.time() - t0
            csr = ctx.matrix_download_csr()
        finally:
            ctx.close()
        t0 = time.time()
        tfx.sensit_io.write_sensit(os.path.join(wd, "output", "synth", "SENSIT"), 1, csr, N, (nx, ny, nz), cw, ctype, res["comp_error"],
                                   depth_weighting_type=1, nnz_hist_total=res["nnz_hist"], nnz_total=res["nnz"])
        nnz = int(res["nnz"])
        del csr
        tfx.synthetic.write_parfile_inputs(wd, nx, ny, nz, ox, oy, ctype, rate, nmajor=1, nminor=iterations, sensit_read=1)
        t_files = time.time() - t0
        legs = {}
        for rk in rank_counts:
            if legs and time.time() - t_leg > wall_budget_s:
                log("cpu_baseline_large: %d-rank leg skipped (%.0f s of the %.0f s budget used)" % (rk, time.time() - t_leg, wall_budget_s))
                break
            cmd = [mpiexec, "-n", str(rk), ref, "-p", "Parfile.txt"]
            tfx.synthetic.write_parfile_text(wd, nx, ny, nz, D, ctype, rate, nmajor=1, nminor=1, sensit_read=1)
            run1 = timed_reference_run(cmd, wd)
            tfx.synthetic.write_parfile_text(wd, nx, ny, nz, D, ctype, rate, nmajor=1, nminor=iterations, sensit_read=1)
            run = timed_reference_run(cmd, wd)
            nnz_read = int(run["stdout"].split("nnz_total (of the read kernel)  =")[1].split()[0])
            if nnz_read != nnz:
                raise RuntimeError("the reference read %d non-zeros from files holding %d" % (nnz_read, nnz))
            import re
            m_it = re.findall(r"Finished lsqr solver, r =\s*\S+\s+iter =\s*(\d+)", run["stdout"])
            it_done = int(m_it[-1]) if m_it else iterations
            diff = run["wall_s"] - run1["wall_s"]
            legs[rk] = {"reload_and_1_iteration_s": round(run1["wall_s"], 2), "reload_and_%d_iterations_s" % iterations: round(run["wall_s"], 2),
                        "iterations": it_done, "ms_per_lsqr_iteration": 1e3 * max(diff, 1e-9) / max(it_done - 1, 1),
                        "resolved": bool(diff > 0.05 * run1["wall_s"]),
                        "matrix_MB_per_rank_as_stored_by_the_reference": round(8.0 * nnz / rk / 1e6, 1)}
            log("cpu baseline, DRAM-resident point: %d ranks, nnz %d: %.1f ms per LSQR iteration (runs of %.1f s and %.1f s)" %
                (rk, nnz, legs[rk]["ms_per_lsqr_iteration"], run1["wall_s"], run["wall_s"]))
        best = min(legs, key=lambda r: legs[r]["ms_per_lsqr_iteration"])
        ms = legs[best]["ms_per_lsqr_iteration"]
        lin = small_ms_per_iteration * nnz / small_nnz if small_ms_per_iteration and small_nnz else None
        return {"value_measured_large": 1e3 / ms, "unit": "iterations/s", "ranks": best, "host_cores": cores,
                "at": {"cells": N, "obs": D, "nnz": nnz, "compression": "haar", "rate": rate, "iterations": iterations,
                       "matrix_GB_as_stored_by_the_reference": round(8.0 * nnz / 1e9, 2)},
                "ms_per_lsqr_iteration": ms, "legs_by_ranks": {str(k): v for k, v in legs.items()},
                "kernel": "built on the GPU (%.2f s), written as reference-format SENSIT files (%.1f s incl. the Parfile inputs), read back by the "
                          "reference with sensit.readFromFiles = 1 - its entry count checked against the files'" % (t_gpu_build, t_files),
                "timing": "difference of a 1 x %d- and a 1 x 1-iteration run per rank count, both re-loading the kernel files" % iterations,
                # how the cache-resident small point extrapolates to THIS matrix, against what was measured on it
                "linear_extrapolation_of_the_small_leg_to_this_nnz_ms": lin,
                "measured_over_extrapolated": None if not lin else ms / lin,
                "headline_from_this_point": {"ms_per_lsqr_iteration_linear_in_nnz": ms * nnz_headline / nnz,
                                             "iterations_per_s": 1e3 / (ms * nnz_headline / nnz)},
                "leg_wall_s": round(time.time() - t_leg, 1)}
    except Exception as e:      # the baseline leg must never take the benchmark down
        log("cpu_baseline_large skipped: %r" % (e,))
        return {"skipped": repr(e)}
    finally:
        shutil.rmtree(wd, ignore_errors=True)


def collect_parfile_outputs(wd, out="output/synth"):
    """Final model, costs, nnz and the per-column nnz histogram a `tomofastx -p Parfile` run left in wd (reference formats:
    problem_joint_gravmag.F90:461-470 costs, sensitivity_gravmag.F90:360-392 SENSIT meta / nnz)."""
    o = {}
    t = open(os.path.join(wd, out, "model", "grav_final_model_full.txt")).read().split()
    o["model"] = np.array([float(v) for v in t[1:1 + int(t[0])]])
    txt = open(os.path.join(wd, out, "costs.txt")).read()
    if "clustering_cost_mag" in txt:
        # the reference: 20 column names, then list-directed records of 20 numbers wrapped over several lines; the last record is never
        # flushed completely (cut after 5 fields).  Column 1 = iteration, 2 = data cost of the gravity problem.
        toks = txt[txt.index("clustering_cost_mag") + len("clustering_cost_mag"):].split()
        last = (len(toks) - 1) // 20 * 20
        o["data_cost"] = float(toks[last + 1])
    else:
        # this repo's host: one line per record (iteration, then data_cost, model_cost, ADMM_cost, ADMM_weight per active problem)
        rows = [l.split() for l in txt.splitlines() if l.strip() and not l.lstrip().startswith("#")]
        o["data_cost"] = float(rows[-1][1])
    meta = open(os.path.join(wd, out, "SENSIT", "sensit_grav_meta.txt")).read().split()
    o["comp_error"], o["nnz_total"] = float(meta[8]), int(meta[11])
    z = open(os.path.join(wd, out, "SENSIT", "sensit_grav_nnz"), "rb").read()
    o["nnz_hist"] = np.frombuffer(z, ">i4", offset=4).astype(np.int64)
    return o


def reference_medium(tfx, wd, ref_out, cfg, log):
    """Mid-scale parity, live on the GPU box: the compiled reference's CONVERGED inversion of the cpu_baseline leg's problem (128x128x32
    cells x 1024 data, Haar r = 0.05 on a >= 32-core box; CONV_MAJOR x 100 LSQR iterations) against this repo's Parfile host on the GPU,
    same Parfile and inputs; reported next to the reference's own distance between two rank counts.  The 1 x 101-iteration run that
    times the reference's LSQR is compared too, as `timing_leg_1x101` (not converged: see DESIGN.md 4)."""
    import subprocess
    ours = os.path.join(ROOT, "tomofast-x_amd", "host", "tomofastx_amd")
    if not os.path.isfile(ours) or ref_out is None:
        return None

    def gpu_host(nmajor, nminor, sensit_read, env=None):
        tfx.synthetic.write_parfile_inputs(wd, nx, ny, nz, ox, oy, ctype, rate, nmajor=nmajor, nminor=nminor, sensit_read=sensit_read)
        t0 = time.time()
        p = subprocess.run([ours, "-p", "Parfile.txt"], cwd=wd, capture_output=True, text=True, timeout=600, env=env)
        if p.returncode != 0 or "THE END." not in p.stdout:
            raise RuntimeError("the GPU host failed: " + p.stdout[-300:] + p.stderr[-300:])
        return collect_parfile_outputs(wd), time.time() - t0

    def distance(got, ref):
        rm, gm = ref["model"], got["model"]
        return {"model_rel_l2": float(np.linalg.norm(gm - rm) / np.linalg.norm(rm)), "model_max_abs_diff": float(np.abs(gm - rm).max()),
                "data_cost": {"reference": ref["data_cost"], "gpu": got["data_cost"], "ratio_gpu_over_reference": got["data_cost"] / ref["data_cost"]}}

    try:
        nx, ny, nz, ox, oy, ctype, rate = cfg
        # (1) the GPU host on the REFERENCE'S OWN kernel: it reads the SENSIT files the reference's build left in the work directory
        # (sensit.readFromFiles = 1) - identical matrix bits, so what is left is the solver: the order of the sums in the two products and
        # the norms.  Run first: the host's own build (2) rewrites that folder.
        on_ref_kernel = None
        try:
            got1, _ = gpu_host(CONV_MAJOR, 100, 1, env=dict(os.environ, TFX_WRITE_SENSIT="0"))
            on_ref_kernel = distance(got1, ref_out)
        except Exception as e:      # noqa
            log("reference_medium: reload leg skipped: %r" % (e,))
        # (2) the GPU host building its own kernel: the converged leg, then the timing leg's 1 x 101 iterations
        got, dt = gpu_host(CONV_MAJOR, 100, 0)
        hist_same = float(np.mean(ref_out["nnz_hist"] == got["nnz_hist"]))
        out = {"config": "%dx%dx%d cells x %d data, %s r = %g, %d x 100 LSQR iterations (converged)" %
                         (nx, ny, nz, ox * oy, {1: "Haar", 2: "D4"}[ctype], rate, CONV_MAJOR),
               "gpu_host_wall_s": round(dt, 2), "model_max_abs": float(np.abs(ref_out["model"]).max())}
        out.update(distance(got, ref_out))
        out.update({"reference_own_scatter_between_rank_counts": ref_out.get("own_scatter"),
                    # the same inversion by the GPU host on the reference's own SENSIT files (identical matrix bits: the solver alone)
                    "gpu_host_on_the_reference_kernel": on_ref_kernel,
                    "nnz_total": {"reference": ref_out["nnz_total"], "gpu": got["nnz_total"]},
                    "compression_error": {"reference": ref_out["comp_error"], "gpu": got["comp_error"]},
                    "nnz_histogram": {"columns": int(ref_out["model"].size), "columns_with_identical_count": hist_same,
                                      "sum_abs_count_diff": int(np.abs(ref_out["nnz_hist"] - got["nnz_hist"]).sum())}})
        if ref_out.get("timing_leg") is not None:
            try:
                got_t, _ = gpu_host(1, 101, 0)
                out["timing_leg_1x101"] = dict(distance(got_t, ref_out["timing_leg"]),
                                               note="NOT converged (both costs ~1e-7): an unconverged Golub-Kahan recurrence amplifies last-bit "
                                                    "differences of the sums; kept because this run times the reference's LSQR")
            except Exception as e:      # noqa
                log("reference_medium: timing-leg comparison skipped: %r" % (e,))
        own = ref_out.get("own_scatter") or {}
        log("reference_medium (%d x 100, converged): model rel-L2 %.2e (on the reference's own kernel: %s; the reference's own %s-rank scatter: %s), "
            "data cost %.6e vs %.6e (ratio %.6f), nnz %d vs %d, identical column counts %.6f" %
            (CONV_MAJOR, out["model_rel_l2"], "%.2e" % on_ref_kernel["model_rel_l2"] if on_ref_kernel else "n/a", own.get("ranks"),
             "%.2e" % own["model_rel_l2"] if own else "n/a", got["data_cost"], ref_out["data_cost"], got["data_cost"] / ref_out["data_cost"],
             got["nnz_total"], ref_out["nnz_total"], hist_same))
        return out
    except Exception as e:      # never take the benchmark down
        log("reference_medium skipped: %r" % (e,))
        return None


def pmc_traffic(workload, kernel, nnz_loc, device_bytes):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC summary (profiles/rNN_pmc_summary.json:
    FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate --pmc passes) for this workload and matrix; None when there is none.
    PMC counters cannot be read from inside the benchmark process, so this is the value measured on the same command.  A summary
    is only accepted for the SAME matrix: workload, nnz and the device bytes of the matrix must all match the run's - a layout
    change without a re-profile then reports traffic = null instead of a stale number."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")), reverse=True):
        try:
            d = json.load(open(f))
            if (d.get("workload") == workload and int(d.get("nnz", -1)) == int(nnz_loc) and kernel in d.get("kernels", {})
                    and int(d.get("device_bytes_of_the_matrix", -1)) == int(device_bytes)):
                return d["kernels"][kernel]["traffic_bytes_per_launch"], os.path.relpath(f, ROOT)
        except Exception:
            continue
    return None, None


def cpu_baseline(tfx, w, budget_s, cw, log):
    """The CPU oracle (tests/oracle_lib.py -> oracle/libtfx_oracle.so, one core) on a bounded sample: it builds R rows of
    the same matrix and runs LSQR iterations on them; one iteration over all D rows costs D/R times that (LSQR cost is
    linear in nnz), so value = 1 / (t_iter_sample * D / R)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as orc
    nx, ny, nz = w["nx"], w["ny"], w["nz"]
    N = nx * ny * nz
    grid = tfx.synthetic.grid(nx, ny, nz)
    xs, ys, zs = tfx.synthetic.observations(nx, ny, w["ox"], w["oy"])
    D = xs.size
    K = int(w["rate"] * N) if w["ctype"] > 0 else N
    rows = []
    t0 = time.time()
    idx = np.linspace(0, D - 1, 64).astype(int)
    for r in idx:
        rows.append(orc.build_row_grav(grid, (nx, ny, nz), cw, (xs[r], ys[r], zs[r]), w["ctype"], K))
        if time.time() - t0 > 0.6 * budget_s:
            break
    t_build = time.time() - t0
    R = len(rows)
    rp = np.concatenate([[0], np.cumsum([c.size for c, _, _ in rows])]).astype(np.int64)
    cols = np.concatenate([c for c, _, _ in rows])
    vals = np.concatenate([v for _, v, _ in rows])
    S = (rp, cols, vals)
    b = np.random.default_rng(0).standard_normal(R + N)
    b[R:] = 0.0
    Cm = orc.diag_csr(np.full(N, np.float32(1e-7), np.float32))
    # whole sample iterations (matrix passes over R rows + the O(N) vector / damping work)
    t0 = time.time()
    _, it, _ = orc.lsqr(S, Cm, N, b, 2)
    t1 = (time.time() - t0) / max(it, 1)
    niter = max(2, min(40, int(0.25 * budget_s / max(t1, 1e-6))))
    t0 = time.time()
    _, it, _ = orc.lsqr(S, Cm, N, b, niter)
    t_iter = (time.time() - t0) / max(it, 1)
    # the two matrix passes alone (this is the part that scales with the number of rows)
    x = np.random.default_rng(1).standard_normal(N)
    y = np.random.default_rng(2).standard_normal(R)
    reps = max(2, min(40, int(0.15 * budget_s / max(t1, 1e-6))))
    t0 = time.time()
    for _ in range(reps):
        orc.spmv(rp, cols, vals, x)
        orc.spmtv(rp, cols, vals, y, N)
    t_mat = (time.time() - t0) / reps
    t_vec = max(t_iter - t_mat, 0.0)
    t_full = t_mat * D / R + t_vec
    val = 1.0 / t_full
    log("cpu baseline: %d sample rows built in %.1f s (%.3e cell.obs/s/core), %.4f s per sample iteration -> %.5f it/s" %
        (R, t_build, R * N / t_build, t_iter, val))
    return {"value": val, "unit": "iterations/s", "cores": 1, "kind": "port",
            "sample": "%d of %d rows built and iterated by the C oracle (oracle/tfx_oracle.c) on one core; the matrix part of "
                      "the measured iteration time is scaled by D/R (LSQR is linear in nnz)" % (R, D),
            "build_cell_obs_per_s_per_core": R * N / t_build, "sample_rows": R, "sample_iteration_s": t_iter,
            "sample_matrix_passes_s": t_mat}


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- streaming RNN-T hot path on MI355X, through the C ABI (libaprilasr.so).

Metric (BASELINE.json): real-time factor / concurrent sessions at RTF <= 0.1 for
aprilv0_en-us-sized weights, 16 kHz mono PCM16.  One "step" = every one of the B concurrent
sessions on a GPU is fed 100 ms (1600 samples, parec-style) in ONE batched call and processed
to completion (fbank -> 12-layer LSTM encoder -> joiner/greedy/decoder rounds -> callbacks).
  value        = audio seconds processed per wall second, whole job (all GPUs)
  rtf          = wall / audio per session  (every session advances together)
Weak scaling: B sessions per GPU; sessions never talk to each other; the only collective is
the RCCL broadcast of the packed weight blob at model load (rank 0 parses the .april file).

Launch: python bench.py --gpus 1 --steps K --warmup W
        python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
Data: synthetic (seeded LCG noise PCM per session; seeded random weights at aprilv0 dims --
there is no real model/audio offline).  Override with APRIL_MODEL=/path/model.april.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
FP16_MFMA_PEAK_TFLOPS = 2500.0     # dense fp16 MFMA peak (--precision f16 only)
HBM_PEAK_GBS = 8000.0              # spec; ~6300 measured achievable
TRAFFIC_JSON = "r06_gates_traffic.json"      # the committed PMC pass the roofline's `traffic` is taken from


def config5_leg(args, A, SM, torch, np):
    """BASELINE configs[4]: larger encoder (icefall lstm-transducer-stateless2-sized: 16 layers, d 768, cell 1536, ffn 3072),
    512 concurrent sessions on one GPU, fp16 MFMA path -- and the same model in fp32 beside it."""
    step_samples = 1600
    counts = np.zeros(6, np.uint64)
    lpath = os.environ.get("APRIL_MODEL_LARGE") or os.path.join(tempfile.gettempdir(), "bench_large_synth.april")
    if not os.path.exists(lpath):
        SM.write_model(lpath, SM.LARGE_DIMS)
    nb5, wu5, ts5 = 512, 6, 20
    config5 = {"sessions": nb5, "dims": "large (16 layers, d_model 768, cell 1536, ffn 3072; synthetic seeded weights)", "feed_ms": 100, "steps": ts5, "ingest": getattr(args, "ingest", "pipelined")}
    prev = os.environ.get("APRIL_PRECISION")
    precs = tuple(os.environ.get("APRIL_BENCH_C5_PRECS", "f16,f32").split(","))      # (measurement aid: "f16" alone halves an A/B run)
    for prec in precs:
        os.environ["APRIL_PRECISION"] = prec
        m5 = A.Model(lpath)
        s5 = [A.Session(m5, None, counters=counts) for _ in range(nb5)]
        g5 = A.SessionGroup(s5)
        pp = [SM.lcg_pcm16(step_samples * (wu5 + ts5), seed=12345 + 40_000_000 + i) for i in range(nb5)]
        g5.plan(pp, step_samples)
        pipelined = getattr(args, "ingest", "pipelined") != "lockstep"
        feed5 = g5.feed_planned_pipelined if pipelined else g5.feed_planned
        for s in range(wu5):
            feed5(s)
        g5.drain()
        torch.cuda.synchronize(); a = time.perf_counter()
        for s in range(wu5, wu5 + ts5):
            feed5(s)
        g5.drain()
        torch.cuda.synchronize(); b = time.perf_counter()
        ms = (b - a) / ts5 * 1e3
        leg = {"ms_per_step": round(ms, 3), "rtf": round(ms / 100.0, 5), "audio_s_per_s": round(nb5 * 0.1 / (ms * 1e-3), 1), "params": int(m5.dims.param_count)}
        # gates GEMM of this model by the engine's hipEvents: priced against the MFMA peak of the precision AND against HBM
        dd = m5.dims
        before = m5.stats()
        more = [SM.lcg_pcm16(step_samples * args.profile_steps, seed=12345 + 50_000_000 + i) for i in range(nb5)]
        m5.profile(True)
        g5.plan(more, step_samples)
        for s in range(args.profile_steps):
            g5.feed_planned(s)
        sp_ = m5.stats()
        m5.profile(False)
        launches = sp_.kernel_launches[0]
        # the gates clock (as in the headline's roofline): further feeds under graph replay, the gates kernels stamping their own start / end
        clocked = None
        more2 = [SM.lcg_pcm16(step_samples * (6 + args.profile_steps), seed=12345 + 60_000_000 + i) for i in range(nb5)]
        m5.profile(2)
        g5.plan(more2, step_samples)
        for s in range(6):                        # (plans with stamp slots are built and captured)
            feed5(s)
        g5.drain()
        m5.profile(2)                             # (the sums start from zero)
        for s in range(6, 6 + args.profile_steps):
            feed5(s)
        g5.drain()
        m5.profile(0)
        sg = m5.stats()
        if sg.gates_clock_launches:
            clocked = {"avg_launch_us": round(sg.gates_clock_ms / sg.gates_clock_launches * 1e3, 2), "rows_per_launch": round(sg.gates_clock_rows / sg.gates_clock_launches, 1),
                       "launches": int(sg.gates_clock_launches),
                       "by_problems_per_launch": {str(i + 1): {"launches": int(sg.gates_clock_launches_by_n[i]), "avg_launch_us": round(sg.gates_clock_ms_by_n[i] / sg.gates_clock_launches_by_n[i] * 1e3, 2)}
                                                  for i in range(4) if sg.gates_clock_launches_by_n[i]}}
        if launches:
            avg_ms = sp_.kernel_ms[0] / launches
            rows_per_launch = (sp_.chunks - before.chunks) * dd.n_layers / launches
            eager_us = avg_ms * 1e3
            if clocked:          # (priced on the gates clock; the eager dispatch-stamp figure is kept beside it)
                avg_ms = clocked["avg_launch_us"] * 1e-3
                rows_per_launch = clocked["rows_per_launch"]
            layers_per_launch = max(1.0, rows_per_launch / nb5)
            flops = 2.0 * rows_per_launch * (2 * dd.d_model) * (4 * dd.hidden)
            esz = 2 if dd.precision == 1 else 4
            wbytes = (2 * dd.d_model) * (4 * dd.hidden) * esz * layers_per_launch
            sbytes = rows_per_launch * (dd.d_model * esz * 2 + dd.hidden * 4 * 2 + dd.hidden * esz)      # x,h read; c read+write; u write
            tf = flops / (avg_ms * 1e-3) / 1e12
            peak = FP16_MFMA_PEAK_TFLOPS if dd.precision == 1 else FP32_MFMA_PEAK_TFLOPS
            gbs = (wbytes + sbytes) / (avg_ms * 1e-3) / 1e9
            leg["gates_gemm"] = {"avg_launch_us": round(avg_ms * 1e3, 2), "rows_per_launch": round(rows_per_launch, 1), "tflops": round(tf, 1),
                                 "frac_of_mfma_peak": round(tf / peak, 4), "mfma_peak_tflops": peak, "algorithmic_GBps": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4),
                                 "clock": "gates clock (graph replay, kernels stamp their own start / end)" if clocked else "eager launches, dispatch time stamps",
                                 "gates_clock": clocked, "eager_avg_launch_us": round(eager_us, 2),
                                 "class_ms": {k: round(sp_.kernel_ms[i], 3) for i, k in enumerate(["gates", "gemm_other", "row", "conv", "fbank", "dec_joint"])}}
        leg["replay_mismatch"] = int(m5.stats().replay_mismatch)
        config5[prec] = leg
        for s_ in s5:
            s_.close()
        m5.close()
    if prev is None:
        del os.environ["APRIL_PRECISION"]
    else:
        os.environ["APRIL_PRECISION"] = prev
    if "f32" in config5 and "f16" in config5:
        config5["f16_speedup_vs_f32"] = round(config5["f32"]["ms_per_step"] / config5["f16"]["ms_per_step"], 3)
    config5["bound"] = ("fp16: the gates / FFN-up GEMMs run on ping-pong tiles (GM_PP: 256 x 128 / 128 x 128 / 256 x 192, one workgroup per CU): a launch is whole rounds of "
                        "one tile time, of which the K loop (one MFMA-issuing wave per SIMD: 0.73 of the pipe) is ~60 %, the HBM start and the LSTM-cell epilogue the rest; the "
                        "N = d_model GEMMs (32 x 64 tiles) are 42 % of the layer chain; ~70 dependent launches per feed.  Neither the matrix pipe nor HBM binds; see DESIGN.md 3.5")
    return config5


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--sessions", type=int, default=int(os.environ.get("BENCH_SESSIONS", "256")), help="concurrent sessions per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=10)
    ap.add_argument("--steady-steps", type=int, default=200, help="length of the fixed steady-state series reported next to the driver's K steps (0 = skip)")
    ap.add_argument("--config5-only", action="store_true", help="measurement aid: run only the configs[4] leg and print its object")
    ap.add_argument("--no-config5", action="store_true", help="skip the BASELINE configs[4] leg (larger encoder, 512 sessions, fp16 MFMA path vs fp32)")
    ap.add_argument("--precision", choices=["f32", "f16"], default="f32",
                    help="f16 = opt-in fp16-operand mode (BASELINE configs[4]); the headline metric is quoted on f32")
    ap.add_argument("--pipeline-depth", type=int, default=2,
                    help="feeds per session the pipelined group feed keeps open (aprilx_feed_many_pipelined depth): 2 = one feed queued behind "
                         "the one on the GPU (default, `value`); deeper queues let the stepping thread find two feeds queued and step them as one "
                         "wavefront of ~5 chunks -- fewer, larger launches, one more feed of latency (measurement knob, named in the line)")
    ap.add_argument("--ingest", choices=["pipelined", "lockstep"], default="pipelined",
                    help="how the 100 ms feeds reach the library: pipelined = aprilx_feed_many_pipelined depth 2 (the call for feed k + 1 "
                         "returns when feed k is complete: the library prepares and launches a feed while the GPU works on the previous one, as "
                         "with continuously streaming clients); lockstep = aprilx_feed_many (one blocking call per feed).  The other mode is "
                         "measured too and reported next to the headline")
    ap.add_argument("--pre-roll", type=int, default=30,
                    help="untimed feeds BEFORE the W warm-up steps (set-up, like the model load): the launch chains of both feed shapes and "
                         "both flight parities are captured and the GPU's clocks have ramped by the time the warm-up starts (reported as pre_roll_steps)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="collective backend for the weight broadcast; gloo (host tensors, ranks may share a GPU) exists to "
                         "exercise the multi-process path on a one-GPU box")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    lanes = int(os.environ.get("APRIL_LANES", "1"))        # engines (stream + stepping thread) per GPU
    os.environ.setdefault("APRIL_MAX_SESSIONS", "4096")
    os.environ.setdefault("APRIL_MAX_BATCH", "8192")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import numpy as np
    import torch
    import torch.distributed as dist
    if args.backend == "gloo":                             # test mode: wrap the ranks onto the GPUs that exist
        local_rank %= max(1, torch.cuda.device_count())
    os.environ["APRIL_GPU_DEVICES"] = ",".join([str(local_rank)] * lanes)      # read by the library when it initialises
    import april_asr_amd as A
    from april_asr_amd import synth_model as SM

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if args.config5_only:
        print(json.dumps(config5_leg(args, A, SM, torch, np)), flush=True)
        return
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)      # RCCL over xGMI
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    cdev = dev if args.backend == "nccl" else torch.device("cpu")      # where collective tensors live
    if args.precision == "f16":
        os.environ["APRIL_PRECISION"] = "f16"       # read by the library when a model is created

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # ---------------- model: rank 0 parses the file, everyone else receives the packed blob over RCCL
    t_load0 = time.time()
    bcast_ms = None
    if rank == 0:
        path = os.environ.get("APRIL_MODEL")
        if not path:
            path = os.path.join(tempfile.gettempdir(), "bench_aprilv0_synth.april")
            if not os.path.exists(path):
                SM.write_model(path, SM.APRILV0_DIMS)
        model = A.Model(path)
    bcast_info = None
    rccl_fallback = False
    if world > 1 and args.backend == "nccl":
        # the one collective of the job, inside the library: rank 0's packed weights -> every other rank's GPU over
        # RCCL/xGMI (aprilx_model_broadcast).  torch.distributed only carries the 128-byte RCCL id.
        ids = [A.Model.broadcast_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        root = model if rank == 0 else None
        try:
            model = A.Model.broadcast(root, rank, world, ids[0])
            ok = 1
        except Exception as e:                      # the library reports RCCL failures as NULL -> exception, it does not abort
            print("rank %d: library broadcast failed (%s)" % (rank, e), file=sys.stderr, flush=True)
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            li = model.load_info()
            bcast_ms = li.broadcast_ms
            bcast_info = {"where": "libaprilasr.so (RCCL ncclBroadcast)", "bytes": int(li.broadcast_bytes), "ranks": int(li.ranks),
                          "comm_init_ms": round(li.comm_init_ms, 1)}
        else:
            # fallback so that a scaling run still produces numbers: the same blob through torch.distributed's RCCL communicator
            # into each rank's GPU memory, models built from the device copy (aprilx_model_from_blob).  LOUD: a top-level
            # "rccl_fallback": true in the line, a message on stderr, and a non-zero exit under APRIL_STRICT_RCCL=1.
            rccl_fallback = True
            print("bench.py rank %d: RCCL FALLBACK -- the library's own ncclBroadcast failed, weights go through torch.distributed" % rank, file=sys.stderr, flush=True)
            if os.environ.get("APRIL_STRICT_RCCL", "0") not in ("", "0"):
                sys.exit(3)
            if rank == 0:
                blob = torch.from_numpy(root.export_blob()).to(dev)
                size = torch.tensor([blob.numel()], dtype=torch.int64, device=dev)
            else:
                size = torch.zeros(1, dtype=torch.int64, device=dev)
            dist.broadcast(size, 0)
            if rank != 0:
                blob = torch.empty(int(size.item()), dtype=torch.uint8, device=dev)
            torch.cuda.synchronize(); t0 = time.time()
            dist.broadcast(blob, 0)
            torch.cuda.synchronize()
            bcast_ms = (time.time() - t0) * 1e3
            bcast_info = {"where": "torch.distributed nccl (fallback: the library's own RCCL broadcast failed)", "bytes": int(blob.numel()), "ranks": world}
            model = root if rank == 0 else A.Model.from_blob(None, device_ptr=blob.data_ptr(), size=blob.numel())
            del blob
    elif world > 1:
        # gloo test mode (ranks may share a GPU, no RCCL communicator possible): the same content as a host blob
        if rank == 0:
            blob = torch.from_numpy(model.export_blob())
            size = torch.tensor([blob.numel()], dtype=torch.int64)
        else:
            size = torch.zeros(1, dtype=torch.int64)
        dist.broadcast(size, 0)
        if rank != 0:
            blob = torch.empty(int(size.item()), dtype=torch.uint8)
        t0 = time.time()
        dist.broadcast(blob, 0)
        bcast_ms = (time.time() - t0) * 1e3
        bcast_info = {"where": "torch.distributed gloo (host blob, test mode)", "bytes": int(blob.numel()), "ranks": world}
        if rank != 0:
            model = A.Model.from_blob(blob.numpy())
        del blob
    load_s = time.time() - t_load0
    d = model.dims
    # every rank's view of the load (not only rank 0's): which RCCL it has mapped, how long its communicator set-up and its share of
    # the weight broadcast took, what the library says it used -- so that the first real multi-GPU line is diagnosable from the line
    try:
        li_r = model.load_info()
        rank_rec = {"rank": rank, "device": local_rank, "model_load_s": round(load_s, 2),
                    "weight_broadcast_ms": None if bcast_ms is None else round(float(bcast_ms), 2),
                    "comm_init_ms": round(float(li_r.comm_init_ms), 1), "library_broadcast_ms": round(float(li_r.broadcast_ms), 2),
                    "used_rccl": int(li_r.used_rccl), "library_ranks": int(li_r.ranks), "rccl_fallback": bool(rccl_fallback),
                    "rccl_libs_mapped": sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln}), "pid": os.getpid()}
    except Exception as e:                          # noqa: BLE001 -- diagnostics must not take the run down
        rank_rec = {"rank": rank, "error": repr(e)}
    per_rank = [rank_rec]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, rank_rec)

    # ---------------- sessions + synthetic audio
    B = args.sessions
    n_steps = args.warmup + args.steps
    step_samples = 1600
    counts = np.zeros(6, np.uint64)      # filled by the library's C handler (a Python callback per result would
                                         # cost more than the GPU step at thousands of sessions)

    def make_group(nsess, seed0):
        sess = [A.Session(model, None, counters=counts) for _ in range(nsess)]
        return sess, A.SessionGroup(sess)

    def pcm_for(nsess, nst, seed0):
        # one LCG stream per session: seed 12345 + global session id
        return [SM.lcg_pcm16(step_samples * nst, seed=12345 + seed0 + i) for i in range(nsess)]

    sess, grp = make_group(B, rank * B)
    pcm = pcm_for(B, n_steps, rank * B)
    if args.pre_roll > 0:                    # set-up: see --pre-roll
        pre = pcm_for(B, args.pre_roll, 50_000_000 + rank * B)
        grp.plan(pre, step_samples)
        for s in range(args.pre_roll):
            (grp.feed_planned(s) if args.ingest == "lockstep" else grp.feed_planned_pipelined(s, args.pipeline_depth))
        grp.drain()
        del pre

    step_wall = []                           # per-step wall time of the timed steps (p50/p99 latency of one 100 ms feed of all sessions)

    def run_steps(group, pcms, s0, s1, record=None, ingest=None, depth=None):
        """feeds s0 .. s1-1 of the plan; returns with every feed processed and every callback delivered"""
        ingest = ingest or args.ingest
        depth = depth or args.pipeline_depth
        if s0 == 0:
            group.plan(pcms, step_samples)       # pointer arrays built outside the timed region
        feed = group.feed_planned if ingest == "lockstep" else (lambda k: group.feed_planned_pipelined(k, depth))
        for s in range(s0, s1):
            if record is None:
                feed(s)
            else:
                a = time.perf_counter()
                feed(s)
                record.append(time.perf_counter() - a)
        if ingest != "lockstep":
            group.drain()

    run_steps(grp, pcm, 0, args.warmup)
    barrier()
    model.feed_latencies(reset=True)
    t0 = time.perf_counter()
    run_steps(grp, pcm, args.warmup, n_steps, step_wall)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    # hand-over -> delivery latency of every tick of the timed region, stamped inside the library (aprilx_model_feed_latency): in
    # pipelined mode the feed CALL returns as soon as the samples are queued, so its duration is not the latency of a feed
    feed_lat = model.feed_latencies(reset=True)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    st = model.stats()
    host_ms = [round(x, 2) for x in st.host_ms]
    # a longer, fixed-length series next to the driver's K steps (20 steps = 33 ms decide little; box-to-box spread
    # is +-5 %): the same sessions keep streaming, `steady_steps` further feeds, per-step wall times on rank 0, the mean as the
    # max over ranks.  Not `value`: that stays the driver's K steps.
    steady = None
    if args.steady_steps > 0:
        more_s = pcm_for(B, args.steady_steps, 30_000_000 + rank * B)
        sw = []
        run_steps(grp, more_s, 0, 4)                     # (plan + the first feeds outside the series)
        barrier()
        a = time.perf_counter()
        run_steps(grp, more_s, 4, args.steady_steps, sw)
        barrier()
        el = time.perf_counter() - a
        if world > 1:
            tt = torch.tensor([el], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        nst = args.steady_steps - 4
        steady = {"steps": nst, "ms_per_step": round(el / nst * 1e3, 3), "p50": round(float(np.percentile(sw, 50)) * 1e3, 3),
                  "p90": round(float(np.percentile(sw, 90)) * 1e3, 3), "p99": round(float(np.percentile(sw, 99)) * 1e3, 3),
                  "min": round(min(sw) * 1e3, 3), "rtf": round(el / (nst * 0.1), 5),
                  "what": "%d further 100 ms feeds of the same %d sessions per GPU after the timed region (max over ranks for the mean, rank 0 for the percentiles)" % (nst, B)}
        del more_s
        st = model.stats()
    # the other ingest mode on the same sessions, same number of steps (never `value`)
    other = "lockstep" if args.ingest == "pipelined" else "pipelined"
    more_o = pcm_for(B, args.steps + 4, 40_000_000 + rank * B)
    run_steps(grp, more_o, 0, 4, ingest=other)
    barrier()
    a = time.perf_counter()
    ow = []
    model.feed_latencies(reset=True)
    run_steps(grp, more_o, 4, args.steps + 4, ow, ingest=other)
    barrier()
    el_o = time.perf_counter() - a
    lat_o = model.feed_latencies(reset=True)
    if world > 1:
        tt = torch.tensor([el_o], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el_o = float(tt.item())
    other_ingest = {"ingest": other, "steps": args.steps, "ms_per_step": round(el_o / args.steps * 1e3, 3), "rtf": round(el_o / (args.steps * 0.1), 5),
                    "p50": round(float(np.percentile(ow, 50)) * 1e3, 3),
                    "feed_latency_ms": None if lat_o.size == 0 else {"p50": round(float(np.percentile(lat_o, 50)), 3), "p99": round(float(np.percentile(lat_o, 99)), 3), "n": int(lat_o.size)},
                    "what": "the same sessions, %d further feeds through %s" % (
                        args.steps, "aprilx_feed_many (one blocking call per 100 ms feed: nothing of the next feed can start before the previous one has been delivered)"
                        if other == "lockstep" else "aprilx_feed_many_pipelined depth 2")}
    del more_o
    # a deeper hand-over queue on the same sessions (never `value`): with depth 4 the stepping thread finds two feeds queued and steps
    # them as ONE wavefront of five chunks -- fewer, larger launches, one more feed of latency.  Reported so that the trade is visible.
    deeper = None
    if args.ingest == "pipelined" and args.pipeline_depth < 4:
        more_d = pcm_for(B, args.steps + 8, 50_000_000 + rank * B)
        run_steps(grp, more_d, 0, 8, depth=4)                   # (the five-chunk wavefront's launch chains are captured here)
        barrier()
        a = time.perf_counter()
        model.feed_latencies(reset=True)
        run_steps(grp, more_d, 8, args.steps + 8, depth=4)
        barrier()
        el_d = time.perf_counter() - a
        lat_d = model.feed_latencies(reset=True)
        if world > 1:
            tt = torch.tensor([el_d], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el_d = float(tt.item())
        deeper = {"depth": 4, "steps": args.steps, "ms_per_step": round(el_d / args.steps * 1e3, 3), "rtf": round(el_d / (args.steps * 0.1), 5),
                  "feed_latency_ms": None if lat_d.size == 0 else {"p50": round(float(np.percentile(lat_d, 50)), 3), "p99": round(float(np.percentile(lat_d, 99)), 3), "n": int(lat_d.size)},
                  "what": "the same sessions, %d further feeds through aprilx_feed_many_pipelined with depth 4 (up to three earlier feeds open): throughput against latency" % args.steps}
        del more_d
    st = model.stats()
    audio_per_session = args.steps * step_samples / 16000.0
    value = world * B * audio_per_session / elapsed
    rtf = elapsed / audio_per_session

    # ---------------- roofline of the dominant kernel (gates GEMM + fused LSTM cell), live hipEvent timing
    def gates_roofline(mdl, group, nsess, before, traffic_json):
        """profile_steps further feeds with the engine's per-class hipEvents on; returns (roofline dict, stats after)."""
        dd = mdl.dims
        more = pcm_for(nsess, args.profile_steps, 10_000_000)
        mdl.profile(True)
        run_steps(group, more, 0, args.profile_steps)
        sp_ = mdl.stats()
        mdl.profile(False)
        launches = sp_.kernel_launches[0]
        if not launches:
            return None, sp_
        avg_ms = sp_.kernel_ms[0] / launches
        rows = sp_.chunks - before.chunks        # session-chunks processed while profiling
        # rows per gates launch, averaged: every session-chunk passes the gates GEMM of each layer once; a launch covers one
        # layer of one chunk step, or the same launch of up to T layers when a feed's chunk steps run as a wavefront (z-batched)
        rows_per_launch = rows * dd.n_layers / launches
        eager = {"avg_launch_us": round(avg_ms * 1e3, 2), "rows_per_launch": round(rows_per_launch, 1), "launches": int(launches),
                 "what": "launches one by one (no graph replay), each gates kernel's own dispatch time stamps (hipExtLaunchKernel): the kernels run "
                         "slower than under replay (idle queue between launches: clocks and caches)"}
        # THE CLOCK OF `achieved` / `frac`: the gates clock -- further feeds under GRAPH REPLAY with the timed region's ingest, every flight on
        # one stream while the clock is on (with the three streams of the split feeds the front end and the search of the neighbouring flights
        # share the CUs with a gates launch and its span measures the contention: 73.7 us; one flight at a time leaves the GPU idle between
        # feeds and the kernels run at a lower clock: 48.5 us), the gates kernels stamping their own first start and last end
        # (s_memrealtime); rows per launch from the launch plans.
        # This is the quantity rocprofv3 --kernel-trace reports per kernel for the same invocation (profiles/r06_b256_pipelined_kernel_stats.csv).
        clocked = None
        warm2 = pcm_for(nsess, 6, 30_000_000)
        mdl.profile(2)
        run_steps(group, warm2, 0, 6)             # (the plans of both feed shapes are rebuilt with stamp slots and captured; a feed that waits behind a capture merges with the next one)
        more2 = pcm_for(nsess, args.profile_steps, 20_000_000)
        mdl.profile(2)                            # (on again = the sums start from zero, the plans and their slots stay)
        run_steps(group, more2, 0, args.profile_steps)      # (the timed region's ingest; the engine keeps every flight on ONE stream while the clock is on: no other stream's kernels beside the gates launches -- what rocprofv3's serialised trace sees too -- and no idle GPU between feeds)
        mdl.profile(0)
        sg = mdl.stats()
        if sg.gates_clock_launches:
            avg_ms = sg.gates_clock_ms / sg.gates_clock_launches
            rows_per_launch = sg.gates_clock_rows / sg.gates_clock_launches
            launches = sg.gates_clock_launches
            clocked = {"avg_launch_us": round(avg_ms * 1e3, 2), "rows_per_launch": round(rows_per_launch, 1), "launches": int(launches),
                       "by_problems_per_launch": {str(i + 1): {"launches": int(sg.gates_clock_launches_by_n[i]),
                                                               "avg_launch_us": round(sg.gates_clock_ms_by_n[i] / sg.gates_clock_launches_by_n[i] * 1e3, 2)}
                                                  for i in range(4) if sg.gates_clock_launches_by_n[i]},
                       "what": "1 / 2 / 3 problems per launch = the kernels gemm_f32_kernel / gemm_f32_zkernel / gemm_f32_zkernel_walk <4, 4, 1, ...> of a rocprofv3 kernel trace at 256 sessions"}
        layers_per_launch = max(1.0, rows_per_launch / nsess)      # a z-batched launch holds that many layers' weight matrices
        flops = 2.0 * rows_per_launch * (2 * dd.d_model) * (4 * dd.hidden)
        wbytes = (2 * dd.d_model) * (4 * dd.hidden) * 4 * layers_per_launch
        sbytes = rows_per_launch * (dd.d_model * 4 * 2 + dd.hidden * 4 * 3)      # x,h read; c read+write; u write
        tf = flops / (avg_ms * 1e-3) / 1e12
        mfma_peak = FP16_MFMA_PEAK_TFLOPS if dd.precision == 1 else FP32_MFMA_PEAK_TFLOPS
        if dd.precision == 1:
            wbytes /= 2                         # fp16 weight copies
        gbs = (wbytes + sbytes) / (avg_ms * 1e-3) / 1e9
        frac_mfma, frac_hbm = tf / mfma_peak, gbs / HBM_PEAK_GBS
        if frac_mfma >= frac_hbm:
            rl = {"bound": "mfma", "achieved": round(tf, 2), "peak": mfma_peak, "unit": "TFLOP/s", "frac": round(frac_mfma, 4), "traffic": None}
        else:
            rl = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(frac_hbm, 4), "traffic": None}
        # HBM traffic per launch from the committed PMC pass of the same workload (bench.py cannot collect PMC itself); per
        # launch it is proportional to the layers sharing the launch (weights) and to the rows (activations), i.e. to rows
        if traffic_json:
            try:
                tr = json.load(open(os.path.join(ROOT, "profiles", traffic_json)))
                if int(tr["sessions_per_gpu"]) == nsess and dd.precision == 0:
                    rl["traffic"] = int(tr["traffic_bytes_per_launch"] * rows_per_launch / float(tr["rows_per_launch"]))
                    rl["traffic_source"] = "profiles/%s (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, gfx950 x2 read correction; measured at %.1f rows per launch, scaled by rows)" % (traffic_json, float(tr["rows_per_launch"]))
                    rk = tr.get("rocprof_kernel_time")
                    if rk:      # the same kernels' execution time under rocprofv3 (committed summary of the default invocation): no event brackets, graph replay
                        rl["kernel_time_rocprof"] = {"weighted_avg_us_per_launch": rk["weighted_avg_us_per_launch"], "tflops": rk["weighted_tflops"],
                                                     "frac": rk["frac_of_157.3"], "source": rk["source"].split(" (")[0],
                                                     "by_problems_per_launch": rk.get("by_problems_per_launch"),
                                                     "what": "committed rocprofv3 kernel durations of the gates launches of the default invocation (profiles/, per problem count where the trace was split by grid size), for comparison; `achieved` / `frac` above are live, on the kernels' own clock"}
            except Exception:
                pass
        rl["algorithmic_bytes_per_launch"] = int(wbytes + sbytes)
        rl["clock"] = ("gates clock: %d further feeds under graph replay, pipelined ingest, every flight on one stream; every gates kernel stamps its first workgroup's start and its last "
                       "workgroup's end (s_memrealtime, 10 ns) -- the kernel duration rocprofv3 --kernel-trace (which serialises the queues) reports for the same "
                       "invocation" % args.profile_steps) if clocked else "eager launches with their dispatch time stamps (the gates clock saw no feed wavefront)"
        rl["gates_clock"] = clocked
        rl["eager_clock"] = eager
        rl.update({"kernel": "LSTM gates GEMM [rows,%d]x[%d,%d] + BasicNorm row scale + LSTM cell epilogue (rows = sessions x layers sharing the z-batched launch)" % (2 * dd.d_model, 2 * dd.d_model, 4 * dd.hidden),
                   "avg_launch_us": round(avg_ms * 1e3, 2), "rows_per_launch": round(rows_per_launch, 1),
                   "launches": int(launches), "alt_frac_hbm": round(frac_hbm, 4), "alt_frac_mfma": round(frac_mfma, 4),
                   "class_ms": {k: round(sp_.kernel_ms[i], 3) for i, k in enumerate(["gates", "gemm_other", "row", "conv", "fbank", "dec_joint"])}})
        return rl, sp_

    roofline = None
    if rank == 0:
        roofline, sp = gates_roofline(model, grp, B, st, TRAFFIC_JSON)
    for s in sess:
        s.close()

    # ---------------- what a client of the reference's 12 symbols alone gets (never `value`): the same number of sessions created with
    # APRIL_CONFIG_FLAG_ASYNC_NO_RT and fed one after the other with aas_feed_pcm16 from ONE client thread (100 ms each, as fast as the
    # calls return); the library's stepping thread gathers whatever sessions have audio queued into its ticks.  Two seconds of audio are
    # handed over per block (an asynchronous session queues at most three, src/audio_provider.c:31), then the client waits for the block
    # (aprilx_session_drain: the only call here that is not one of the 12 symbols -- the reference client would wait for its callbacks).
    ref_api = None
    if rank == 0 and world == 1 and not args.no_sweep:
        try:
            cnt_a = np.zeros(6, np.uint64)
            sa = [A.Session(model, None, asynchronous=True, no_rt=True, counters=cnt_a) for _ in range(B)]
            blocks, per_block = 3, 20
            pa = pcm_for(B, (blocks + 1) * per_block, 70_000_000)
            Lf = model._L
            hs = [s_._handle for s_ in sa]
            ptrs = [[int(p.ctypes.data) + 2 * step_samples * k for p in pa] for k in range((blocks + 1) * per_block)]

            # the client loop in C when a compiler is at hand (a Python loop of 256 ctypes calls per step costs ~0.5 ms per step by itself):
            # ten lines that call aas_feed_pcm16 through its address, built next to the run
            shim = None
            try:
                import ctypes as Cc
                sd = tempfile.mkdtemp(prefix="april_shim_")
                open(os.path.join(sd, "shim.c"), "w").write(
                    "#include <stddef.h>\n#include <stdint.h>\ntypedef void (*feed_fn)(void *, short *, size_t);\n"
                    "void feed_all(feed_fn feed, void **sessions, short **pcm, size_t n, size_t count) { for (size_t i = 0; i < n; ++i) feed(sessions[i], pcm[i], count); }\n")
                subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", os.path.join(sd, "shim.c"), "-o", os.path.join(sd, "shim.so")], timeout=60)
                shim = Cc.CDLL(os.path.join(sd, "shim.so"))
                shim.feed_all.argtypes = [Cc.c_void_p, Cc.POINTER(Cc.c_void_p), Cc.POINTER(Cc.c_void_p), Cc.c_size_t, Cc.c_size_t]
                shim.feed_all.restype = None
                feed_addr = Cc.cast(Lf.aas_feed_pcm16, Cc.c_void_p).value
                hs_c = (Cc.c_void_p * B)(*hs)
                ptrs_c = [(Cc.c_void_p * B)(*pk) for pk in ptrs]
            except Exception:
                shim = None

            def feed_block(b0):
                for k in range(b0 * per_block, (b0 + 1) * per_block):
                    if shim is not None:
                        shim.feed_all(feed_addr, hs_c, ptrs_c[k], B, step_samples)
                    else:
                        pk = ptrs[k]
                        for i in range(B):
                            Lf.aas_feed_pcm16(hs[i], pk[i], step_samples)
                for s_ in sa:
                    s_.drain()
            feed_block(0)                                   # (launch chains of the tick shapes are captured here)
            before_a = model.stats()
            a = time.perf_counter()
            for b0 in range(1, blocks + 1):
                feed_block(b0)
            el_a = time.perf_counter() - a
            after_a = model.stats()
            nst = blocks * per_block
            ref_api = {"sessions": B, "steps": nst, "ms_per_step": round(el_a / nst * 1e3, 3), "rtf": round(el_a / (nst * 0.1), 5),
                       "audio_s_per_s": round(B * nst * 0.1 / el_a, 1), "ticks": int(after_a.ticks - before_a.ticks),
                       "cant_keep_up": int(cnt_a[3]),
                       "client_loop": "C (gcc shim: one call per step that loops over aas_feed_pcm16)" if shim is not None else "Python (ctypes call per session and step)",
                       "what": "%d asynchronous sessions (APRIL_CONFIG_FLAG_ASYNC_NO_RT) fed by aas_feed_pcm16 alone from one client thread, 100 ms per call, "
                               "%d s of audio per session between waits; the library batches whatever is queued (`ticks` = GPU flights it took)" % (B, per_block // 10)}
            for s_ in sa:
                s_.close()
        except Exception as e:
            ref_api = {"error": repr(e)}

    # ---------------- concurrency sweep (outside the timed region): RTF at other batch sizes
    sweep = None
    if rank == 0 and world == 1 and not args.no_sweep:          # single-GPU runs only (the driver computes scaling from per-N values)
        sweep = {}
        sweep_runs = {}
        for nb in (1, 16, 64, 1024, 2048, 2304, 2432, 2560, 2816, 3072):
            ss, gg = make_group(nb, 0)
            wu, ts = 6, 20                                     # (both feed shapes -- 2 and 3 chunks -- are captured during warm-up)
            pp = pcm_for(nb, wu + ts, 20_000_000)
            run_steps(gg, pp, 0, wu)
            # three timed passes of 20 feeds on the same sessions (the same audio again: only the time matters here), the WORST one
            # is the point's value -- a capacity claim must not rest on one lucky 20-step run
            runs = []
            for rep in range(3 if nb >= 1024 else 1):
                if rep:
                    run_steps(gg, pp, 0, wu)                   # (re-plan from the start of the buffer; untimed)
                torch.cuda.synchronize(); a = time.perf_counter()
                run_steps(gg, pp, wu, wu + ts)
                torch.cuda.synchronize(); b = time.perf_counter()
                runs.append(round((b - a) / (ts * 0.1), 5))
            sweep[str(nb)] = max(runs)
            sweep_runs[str(nb)] = runs
            for s in ss:
                s.close()
        sweep[str(B)] = round(rtf, 5)

    # ---------------- BASELINE configs[1]: one session, 60 s of audio handed over in ONE feed (the reference's ./main use case):
    # long feeds take the layer-major schedule.  The first pass captures the launch chains, the second is timed.
    offline = None
    if rank == 0 and world == 1 and not args.no_sweep:
        secs = 60.0
        p60 = SM.lcg_pcm16(int(16000 * secs), seed=4321)
        for attempt in range(2):
            ss, gg = make_group(1, 0)
            st_a = model.stats()
            torch.cuda.synchronize(); a = time.perf_counter()
            gg.feed([p60])
            gg.flush()                            # RTF = feed + flush: the 28 flush chunks and the final callbacks are inside
            torch.cuda.synchronize(); b = time.perf_counter()
            st_b = model.stats()
            for s_ in ss:
                s_.close()
        nchunks = int(st_b.chunks - st_a.chunks)
        # per chunk the weights that cannot be shared across time are the recurrent half of the gates and the projection
        wbytes = d.n_layers * (d.d_model * 4 * d.hidden + d.hidden * d.d_model) * 4
        offline = {"audio_s": secs, "includes_flush": True, "wall_ms": round((b - a) * 1e3, 2), "rtf": round((b - a) / secs, 6), "chunks": nchunks,
                   "us_per_chunk": round((b - a) * 1e6 / max(1, nchunks), 2), "layer_major_chunks": int(st_b.lm_chunks - st_a.lm_chunks),
                   "streaming_100ms_rtf_same_session": sweep.get("1") if sweep else None,
                   "speedup_vs_streaming": round(sweep["1"] / ((b - a) / secs), 2) if sweep and sweep.get("1") else None,
                   "schedule": "layer-major wavefront: blocks of time steps, the same launch of all layers z-batched into one (Engine::run_lm_wavefront)",
                   "recurrent_weight_bytes_per_chunk": int(wbytes),
                   "frac_of_hbm_peak_if_restreamed": round(wbytes * nchunks / (b - a) / 1e9 / HBM_PEAK_GBS, 4),
                   "note": "recurrent weights (gate h-half + projection, 10 MB per layer = 120 MB for 12 layers) are re-read from HBM at every time step by two weight-stream launches (csrc/kernels_recur.hip); the streaming read rate of this GPU is 6.8 TB/s from 48 MB up (tools/bw_probe), the Infinity Cache adds nothing"}

    # ---------------- BASELINE configs[4]: larger encoder, 512 sessions, fp16 MFMA path (and the same model in fp32)
    config5 = None
    if rank == 0 and world == 1 and not args.no_config5:
        model.close()
        model = None
        # in a process of its own: the native library reports a failed device allocation or a missing kernel plan with abort(),
        # which an `except` cannot catch -- the headline line must not depend on this leg
        try:
            r5 = subprocess.run([sys.executable, os.path.abspath(__file__), "--config5-only", "--ingest", args.ingest, "--profile-steps", str(args.profile_steps)],
                                stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, text=True)
            lines5 = [ln for ln in r5.stdout.splitlines() if ln.startswith("{")]
            config5 = json.loads(lines5[-1]) if lines5 else {"error": "configs[4] leg exited with code %d: %s" % (r5.returncode, r5.stderr[-400:])}
        except Exception as e:
            config5 = {"error": repr(e)}

    # ---------------- CPU baseline: the oracle (plain-C port, 1 thread) on the host, bounded sample
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import orc_py as O          # checker/baseline only; never on the measured path
        path = os.environ.get("APRIL_MODEL") or os.path.join(tempfile.gettempdir(), "bench_aprilv0_synth.april")
        om = O.Model(path)
        osess = O.Session(om)
        sample_s = 12.0
        p = SM.lcg_pcm16(int(16000 * sample_s), seed=12345)
        a = time.perf_counter()
        for o in range(0, p.size, step_samples):
            osess.feed(p[o:o + step_samples])
        b = time.perf_counter()
        cpu = {"value": round(sample_s / (b - a), 4), "unit": "audio_seconds_per_second", "cores": 1, "kind": "port",
               "sample": "oracle/liborc.so (plain-C restatement, 1 thread, 1 session), %.0f s of 16 kHz LCG-noise PCM16 in 100 ms feeds, "
                         "%d chunks, no flush; ONNXRuntime (the reference's backend) is not installed here" % (sample_s, osess.chunks()),
               "rtf": round((b - a) / sample_s, 4), "host_cpus": os.cpu_count()}
        osess.close(); om.close()
        # the reference's own backend, when the host has it: ONNXRuntime CPU, intra = inter = 1, through
        # the same session driver; only meaningful with a model whose graphs ORT can run (APRIL_MODEL = a real export)
        from oracle import ort_leg as OL
        if OL.available():
            try:
                rs = OL.OrtSession(path)
                a = time.perf_counter()
                for o in range(0, p.size, step_samples):
                    rs.feed(p[o:o + step_samples])
                b = time.perf_counter()
                cpu["onnxruntime_cpu"] = {"value": round(sample_s / (b - a), 4), "unit": "audio_seconds_per_second", "cores": 1, "rtf": round((b - a) / sample_s, 4),
                                          "sample": "onnxruntime CPUExecutionProvider, 1 thread, 1 session, the same %.0f s of audio, %d chunks" % (sample_s, rs.chunks())}
                rs.close()
            except Exception as e:
                cpu["onnxruntime_cpu"] = {"error": repr(e)}
        else:
            cpu["onnxruntime_cpu"] = None       # not installed on this host: CPU baseline = the in-repo restatement only
        # the same port on many cores at once (one process, one session, one thread each -- how the reference would be
        # scaled on a CPU host: its ORT sessions run intra=inter=1, april_model.c:54-55)
        # one process per host CPU, bounded by memory: a worker holds the parsed model (~1.2 GB at aprilv0 size)
        try:
            avail_gb = [int(ln.split()[1]) for ln in open("/proc/meminfo") if ln.startswith("MemAvailable")][0] / 1048576.0
        except Exception:
            avail_gb = 64.0
        k = max(1, min(int(os.environ.get("BENCH_CPU_PROCS", str(os.cpu_count() or 1))), (os.cpu_count() or 1), int(avail_gb / 2.5)))
        start = os.path.join(tempfile.gettempdir(), "bench_cpu_start_%d" % os.getpid())
        if os.path.exists(start):
            os.remove(start)
        worker = os.path.join(ROOT, "oracle", "cpu_worker.py")
        wsecs = 6.0 if k <= 64 else 1.0           # (hundreds of workers share the host's DRAM bandwidth: 337 MB of weights per chunk each)
        procs = [subprocess.Popen([sys.executable, worker, path, str(wsecs), str(12345 + i), start], stdout=subprocess.PIPE, text=True) for i in range(k)]
        try:
            for pr in procs:
                assert pr.stdout.readline().startswith("ready")
            open(start, "w").close()
            times = [float(pr.stdout.readline().split()[1]) for pr in procs]
            cpu["all_cores"] = {"value": round(k * wsecs / max(times), 3), "unit": "audio_seconds_per_second", "cores": k, "host_cpus": os.cpu_count(),
                                "sample": "%d processes x (1 session, 1 thread, %.0f s of audio) started together; slowest %.2f s, fastest %.2f s" % (k, wsecs, max(times), min(times))}
        except Exception as e:                          # the single-core figure above stands on its own
            cpu["all_cores"] = {"error": repr(e)}
        finally:
            for pr in procs:
                pr.kill() if pr.poll() is None else None
            if os.path.exists(start):
                os.remove(start)

    if rank == 0:
        max_ok = None
        if sweep:
            ok = [int(k) for k, v in sweep.items() if v <= 0.1]
            max_ok = max(ok) if ok else 0
        out = {
            "metric": "audio_seconds_per_second (concurrent 16 kHz sessions x 1/RTF), aprilv0_en-us-sized streaming RNN-T",
            "value": round(value, 2), "unit": "audio_s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16" if d.precision == 1 else "f32", "data": "synthetic",
            "config": {"workload": "aprilv0_en-us dims (synthetic seeded weights), %d concurrent streaming sessions per GPU, "
                                   "100 ms PCM16 feeds via %s (BASELINE configs[2]; configs[3] at 8 GPUs)" % (
                                       B, "aprilx_feed_many_pipelined (depth 2)" if args.ingest == "pipelined" else "aprilx_feed_many (one blocking call per feed)"),
                       "sessions_per_gpu": B, "feed_ms": 100, "params": int(d.param_count), "parallelism": "sessions sharded, dp%d" % world},
            "rtf": round(rtf, 5), "sessions_total": world * B, "pre_roll_steps": args.pre_roll,
            "ingest": {"mode": args.ingest, "depth": args.pipeline_depth if args.ingest == "pipelined" else 1, "what": "aprilx_feed_many_pipelined, depth 2 by default: feed k + 1 is queued (samples copied) while feed k is on the GPU, every callback "
                                                     "of the K timed feeds is delivered inside the timed region (drain before the closing barrier)"
                       if args.ingest == "pipelined" else "aprilx_feed_many: one blocking call per feed"},
            "other_ingest": other_ingest, "reference_api_async": ref_api, "deeper_pipeline": deeper, "steady": steady, "config5_f16": config5,
            "rccl_fallback": rccl_fallback, "rccl_libs_mapped": sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln}),
            "per_rank": per_rank,
            # the duration of the feed CALL: in lockstep mode that is the latency of a feed (the call returns with every callback
            # delivered); in pipelined mode it is only the hand-over (the call returns when at most one earlier feed is still open)
            ("step_latency_ms" if args.ingest == "lockstep" else "handover_ms"): {
                "p50": round(float(np.percentile(step_wall, 50)) * 1e3, 3), "p99": round(float(np.percentile(step_wall, 99)) * 1e3, 3),
                "max": round(max(step_wall) * 1e3, 3), "series": [round(x * 1e3, 2) for x in step_wall[:200]],
                "what": ("wall time of one aprilx_feed_many call (100 ms of audio for every session, callbacks delivered), rank 0" if args.ingest == "lockstep" else
                         "wall time of one aprilx_feed_many_pipelined call, rank 0: the hand-over of 100 ms of audio for every session (returns once at most one earlier feed is open); NOT a latency")},
            "feed_latency_ms": None if feed_lat.size == 0 else {
                "p50": round(float(np.percentile(feed_lat, 50)), 3), "p90": round(float(np.percentile(feed_lat, 90)), 3), "p99": round(float(np.percentile(feed_lat, 99)), 3),
                "max": round(float(feed_lat.max()), 3), "n": int(feed_lat.size), "series": [round(float(x), 2) for x in feed_lat[:200]],
                "what": "hand-over -> delivery, stamped inside the library (aprilx_model_feed_latency), rank 0: from the feed call that queued the oldest audio a flight "
                        "served to the moment every callback of that flight had been delivered; with two flights in the air this includes the time queued behind the previous feed"},
            "max_sessions_per_gpu_rtf_le_0.1_tested": max_ok, "rtf_by_sessions_per_gpu": sweep,
            "rtf_sweep_runs": sweep_runs if sweep else None,      # every timed pass behind a sweep point (3 x 20 feeds from 1024 sessions up; the point is the worst)
            "callbacks": int(counts[0]), "tokens_in_callbacks": int(counts[5]), "model_load_s": round(load_s, 2), "weight_broadcast_ms": None if bcast_ms is None else round(bcast_ms, 2), "weight_broadcast": bcast_info,
            "engine_steps": int(st.steps), "host_phase_ms_total": host_ms, "max_batch_seen": int(st.max_batch_seen),
            "kernels_per_chunk_step": int(sp.kernels_per_step) if roofline else None,
            "offline_single_session_60s": offline, "flights": int(st.flights), "replay_mismatch": int(st.replay_mismatch),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if model is not None:
        model.close()
    if world > 1:
        barrier()                                  # rank 0 runs the roofline pass after the timed region; leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

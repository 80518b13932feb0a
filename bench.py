#!/usr/bin/env python3
"""bench.py -- training samples/sec of the DLS_VAE step (Yelp shape, bs=256 per GPU) on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = zero_grad + forward + backward + (gradient exchange) + clip/guards + Adam on one of NB (default 4) synthetic
batches that are already resident in HBM (two small device-to-device copies select the batch of a step).  The step is
replayed from HIP graphs: [zero_grad, forward, vocabulary backward] -> [rest of the backward] -> [guards + Adam].  With
more than one rank the all-reduce of the vocabulary projection's gradient (a third of the dense bytes) starts after
the first graph and travels over xGMI while the second one runs; everything else is exchanged after it.  Rank 0 prints ONE JSON line with the whole-job rate, the
roofline of the dominant kernel group (vocabulary projection + softmax + NLL, forward and backward; HIP-event
timed in an instrumented eager pass right after the timed region) and the CPU baseline (the oracle's composed
step on stock ATen fp32 kernels, rank 0 only, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BF16_DENSE_PEAK_TFLOPS = 2500.0      # MI355X bf16 MFMA, dense (MI355X_MICROARCH.md, chip-level parameters)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="yelp", choices=["yelp", "movie", "book", "mutual"],
                    help="BASELINE.json configs: yelp = configs[1]; movie = configs[2] (8k x 8k tables, decoder attention); "
                         "book = configs[3] (200k users, 100k items); mutual = configs[4] (--use_mutual + --do_pred)")
    ap.add_argument("--batches", type=int, default=4, help="distinct resident batches the steps cycle through")
    ap.add_argument("--force-split", type=int, default=0, help="run the N > 1 schedule at N = 1: 1 = backward cut behind the vocabulary projection (three graphs), "
                    "2 = also in front of the utterance encoder (four graphs)")
    ap.add_argument("--rehearse-dp", type=int, default=0, help="N = 1 only: join a ONE-rank RCCL group and run the N > 1 step "
                    "(cut backward graphs with the gradient collectives between them) on this GPU")
    ap.add_argument("--overlap", type=int, default=2, help="N > 1: 1 = start the vocabulary gradient's all-reduce before the rest "
                    "of the backward pass; 2 = also exchange everything but the encoder's gradients before the encoder's backward")
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--graph", type=int, default=1, help="replay fwd+bwd / optimizer from HIP graphs")
    ap.add_argument("--cpu-baseline", type=int, default=1)
    ap.add_argument("--cpu-steps", type=int, default=6, help="timed CPU-baseline steps (about 1.7 s each on 16 threads)")
    ap.add_argument("--dropout", type=float, default=0.3)
    ap.add_argument("--latent", default=None, choices=["gauss", "dls"],
                    help="latent stack: 'dls' = the reference's DLS pyramid (DlsRec/DlsPri/DLS_Auto), 'gauss' = flat")
    ap.add_argument("--attention", type=int, default=None, help="decoder attention over encoder states (configs[2])")
    ap.add_argument("--vocab-dense", type=int, default=0,
                    help="1 = run the vocabulary GEMMs over the padding rows too (no ignored-token compaction)")
    ap.add_argument("--profile-steps", type=int, default=5, help="instrumented eager steps for per-kernel times")
    ap.add_argument("--dw-beside", default="0.5,0.5",
                    help="N = 1: fractions of the vocabulary whose dW runs beside the decoder's / the encoder's backward recurrence, "
                         "half the chip each (ops.VOCAB_DW_BESIDE); 0 = one whole-chip launch in front of the decoder recurrence")
    ap.add_argument("--traffic", default="auto", choices=["auto", "measure", "file", "off"],
                    help="roofline.traffic: measure = two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over a short child run of "
                         "this script, before this process touches the GPU; file = the committed profiles/r03_pmc_traffic.json; "
                         "auto = measure when rocprofv3 is on PATH (N = 1, yelp config), else file")
    return ap.parse_args()


def cpu_baseline(cfg, batch, steps):
    """The oracle's composed step (oracle/ref_step.py) on the host cores: forward + backward + clip + Adam."""
    from oracle.ref_step import RefDLSVAE
    from pevae_amd.dls_vae import synthetic_batch
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    model = RefDLSVAE(cfg).use_aten_rnn(True)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=cfg.init_lr)
    feed = synthetic_batch(cfg, batch, 1234)
    times = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        opt.zero_grad()
        loss = model(feed)
        model.valid_loss(loss).backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.grad_clip, 2)
        opt.step()
        if i > 0:
            times.append(time.perf_counter() - t0)
    sec = sum(times) / len(times)
    return dict(value=batch / sec, unit="samples/s", cores=cores, kind="port",
                sample="%d full steps of the same workload (bs=%d, L=%d, V=%d) after 1 warm-up; %.2f s/step"
                       % (steps, batch, cfg.max_seq_len, cfg.vocab_size, sec))


VOCAB_KERNELS = ("vocab_fwd_dh_kernel", "vocab_fwd_combine_kernel", "vocab_dw_kernel", "vocab_dw_reduce_kernel")


def measure_traffic(args):
    """HBM bytes per step of the vocabulary group, measured NOW: one rocprofv3 --pmc pass per counter (the guide's
    prescription: separate passes, no trace domains beside --pmc) over a short eager child run of this script, the program
    itself behind `--`.  Per kernel the median over its launches of FETCH_SIZE x 2 (gfx950 counts 64-byte halves of its
    128-byte fetches once) + WRITE_SIZE, both in KiB; the child launches each of the four kernels once per step, on the
    whole chip (--dw-beside 0), as the roofline's instrumented pass does.  Runs before this
    process initialises the GPU.  -> (bytes per step, per-kernel dict) or (None, reason)."""
    import csv
    import glob
    import shutil
    import signal
    import statistics
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3")
    if prof is None:
        return None, "rocprofv3 is not on PATH"
    # never nest profilers: a --pmc child under a tracing parent is the combination the pool's guide warns about
    if any(k.startswith(("ROCPROF", "ROCPROFILER", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself under a profiler (ROCPROF* / LD_PRELOAD): no nested counter passes"
    child = ["python3", os.path.abspath(__file__), "--cpu-baseline", "0", "--graph", "0", "--steps", "3", "--warmup", "1",
             "--profile-steps", "1", "--traffic", "off", "--dw-beside", "0", "--batch", str(args.batch), "--dropout", str(args.dropout)]
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="pevae_pmc_", dir="/tmp")
        cmd = [prof, "--pmc", counter, "--output-format", "csv", "-d", out, "--"] + child
        try:
            p = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                                 stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = p.wait(timeout=180)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)
                p.wait()
                return None, "the %s pass did not finish in 180 s" % counter
            files = glob.glob(os.path.join(out, "**", "*_counter_collection.csv"), recursive=True)
            if rc != 0 or not files:
                return None, "the %s pass left no counter file (exit status %d)" % (counter, rc)
            with open(max(files, key=os.path.getmtime)) as fh:
                for r in csv.DictReader(fh):
                    if r["Counter_Name"] != counter:
                        continue
                    for k in VOCAB_KERNELS:
                        if (k + "(") in r["Kernel_Name"]:
                            vals.setdefault((k, counter), []).append(float(r["Counter_Value"]))
        finally:
            shutil.rmtree(out, ignore_errors=True)
    per, total = {}, 0
    for k in VOCAB_KERNELS:
        if (k, "FETCH_SIZE") not in vals or (k, "WRITE_SIZE") not in vals:
            return None, "no counter rows for %s" % k
        b = int((2 * statistics.median(vals[(k, "FETCH_SIZE")]) + statistics.median(vals[(k, "WRITE_SIZE")])) * 1024)
        per[k] = b
        total += b
    return total, per


def main():
    args = parse()
    from pevae_amd import dp
    measured = None
    if (args.traffic in ("auto", "measure") and args.gpus == 1 and "WORLD_SIZE" not in os.environ and args.config == "yelp"
            and not args.vocab_dense):
        measured = measure_traffic(args)
        if measured[0] is None:
            print("bench.py: roofline.traffic not measured in this run: %s" % measured[1], file=sys.stderr)
            if args.traffic == "measure":
                sys.exit(4)
    # --gpus N without torchrun: this process starts the N ranks itself (before it touches a GPU: device_count() does
    # not initialise one) and leaves with their status; rank 0's JSON line goes to the inherited stdout.  There is no
    # fall-back to fewer GPUs: a layout that cannot run exits non-zero with the reason.
    try:
        cmd = dp.self_launch_command(args.gpus, os.path.abspath(__file__), sys.argv[1:], os.environ,
                                     torch.cuda.device_count())
        if cmd is not None:
            import subprocess
            sys.exit(subprocess.call(cmd, env=dict(os.environ, OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "16"))))
        # one rank per GPU: the persistent recurrence kernels need the whole chip
        rank, world, local = dp.init_from_env(expect_world=args.gpus)
    except dp.LaunchError as err:
        print("bench.py: %s" % err, file=sys.stderr)
        sys.exit(2)
    from pevae_amd import ops
    from pevae_amd.dls_vae import DLSVAE, default_config, synthetic_batch
    from pevae_amd.optim import FlatAdam
    if args.vocab_dense:
        ops.SKIP_IGNORED_TOKENS = False
    ops.VOCAB_DW_BESIDE = None if args.dw_beside in ("", "0") else tuple(float(x) for x in args.dw_beside.split(","))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    # everything (warm-up, capture, replay, eager passes) runs on ONE non-default stream: autograd ties each node
    # to the stream it was created on, and the legacy default stream cannot take part in a capture
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)

    shape = {"yelp": dict(n_user=30000, n_item=20000), "movie": dict(n_user=8000, n_item=8000, use_attention=True,
                                                                       latent_stack="gauss"),
             "book": dict(n_user=200000, n_item=100000), "mutual": dict(use_mutual=True, do_pred=True)}[args.config]
    if args.latent is not None:
        shape["latent_stack"] = args.latent
    if args.attention is not None:
        shape["use_attention"] = bool(args.attention)
    cfg = default_config(**dict(dict(dropout=args.dropout, batch_size=args.batch, latent_stack="dls"), **shape))
    torch.manual_seed(0)                                    # identical initial weights on every rank
    model = DLSVAE(cfg).to(dev)
    model.train()
    # N > 1: the backward pass is cut after the vocabulary projection so that its gradient's all-reduce can start early
    # (north star item 5).  N = 1: nothing is exchanged, so the backward pass stays whole and the vocabulary projection's
    # weight gradient leaves the critical path instead (ops.defer_after_recurrence: it runs beside the latent stack).
    rehearse = bool(args.rehearse_dp) and world == 1
    if rehearse:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29571")
        torch.distributed.init_process_group("nccl", rank=0, world_size=1)
    split = ((world > 1 or rehearse) and bool(args.overlap)) or bool(args.force_split)
    three = split and (args.overlap >= 2 or args.force_split >= 2) and not cfg.use_attention
    model.split_backward = 2 if three else split
    # the flat gradient holds the SUM over ranks; 1/world is applied inside the guard / Adam kernels
    # Order of the flat buffers = order in which gradients become final in the backward pass, so that every stage of the
    # exchange is ONE contiguous stretch (one collective): vocabulary projection | decoder recurrence, latent stack, prior
    # path | utterance encoder and word table | the two tables exchanged as (ids, rows)
    g_first = [model.x_decoder.project.weight, model.x_decoder.project.bias]
    g_late = [model.x_embedding.weight] + list(model.x_encoder.parameters())
    g_tabs = [model.user_embedding.weight, model.item_embedding.weight]
    known = set(id(p) for p in g_first + g_late + g_tabs)
    g_mid = [p for p in model.parameters() if id(p) not in known]
    opt = FlatAdam(g_first + g_mid + g_late + g_tabs, lr=cfg.init_lr, max_norm=cfg.grad_clip, warmup_steps=cfg.warmup_steps,
                   grad_scale=1.0 / world)
    assert len(opt.params) == sum(1 for p in model.parameters() if p.requires_grad)
    dp.broadcast_parameters(opt.flat_p)
    # NB resident batches; the step reads `feed`, whose tensors are views into two static buffers (int64 / fp32)
    batches = [synthetic_batch(cfg, args.batch, 1234 + rank + 1000 * i, dev) for i in range(max(1, args.batches))]
    ikeys = [k for k, v in batches[0].items() if v.dtype == torch.int64]
    fkeys = [k for k, v in batches[0].items() if v.dtype != torch.int64]
    pack = lambda b, keys, dt: torch.cat([b[k].reshape(-1).to(dt) for k in keys])
    all_i = torch.stack([pack(b, ikeys, torch.int64) for b in batches])
    all_f = torch.stack([pack(b, fkeys, torch.float32) for b in batches])
    cur_i, cur_f = all_i[0].clone(), all_f[0].clone()
    feed, at = {}, 0
    for k in ikeys:
        n = batches[0][k].numel()
        feed[k] = cur_i[at:at + n].view(batches[0][k].shape)
        at += n
    at = 0
    for k in fkeys:
        n = batches[0][k].numel()
        feed[k] = cur_f[at:at + n].view(batches[0][k].shape)
        at += n

    def select(i):
        cur_i.copy_(all_i[i % len(batches)])
        cur_f.copy_(all_f[i % len(batches)])

    # user / item tables: only the looked-up rows carry gradient, so they are exchanged as (ids, rows), not all-reduced;
    # the vocabulary projection's gradient is final after the first backward kernels: its all-reduce starts early
    dec = model.x_decoder
    early = opt.spans(g_first) if args.overlap else []
    # three-stage schedule: once the latent stack and the prior path have run backward, only the utterance encoder's and
    # the word table's gradients are missing: everything else (and the two sparse tables) travels beside the encoder's backward
    mid = opt.spans(g_mid) if three else []
    reducer = dp.FlatGradReducer(opt.flat_g, world, sparse_tables=[opt.table_range(model.user_embedding.weight),
                                                                  opt.table_range(model.item_embedding.weight)],
                                 average=False, early_ranges=early, mid_ranges=mid, rehearse=rehearse)
    tables = sorted([(opt.table_range(model.user_embedding.weight)[0], "user_ids"),
                     (opt.table_range(model.item_embedding.weight)[0], "item_ids")])
    sparse_ids = [feed[k] for _, k in tables]
    n_params = sum(p.numel() for p in model.parameters())
    state = {}

    def part_a():                                   # zero_grad, forward, vocabulary backward (N = 1: the whole backward)
        opt.zero_grad(beside_forward=True)
        state["loss"] = loss = model(feed)
        if split:
            model.backward_vocab(loss)
        else:
            model.backward(0, loss)
        return loss

    # three stages on the device: the pack launches of the sparse exchange close the mid graph and the merge launches open
    # the optimizer graph, so the host's part of the exchange between the graphs is the collectives alone
    reducer.external_sparse = bool(three and (world > 1 or rehearse) and reducer.device_sparse(sparse_ids))

    def part_b():                                   # the rest of the backward pass (three stages: up to the encoder)
        if three:
            model.backward_mid(state["loss"])
            if reducer.external_sparse:
                reducer.pack_sparse(sparse_ids)
        elif split:
            model.backward_rest(state["loss"])

    def optimizer_step():
        if reducer.external_sparse:
            reducer.merge_sparse()
        opt.step()

    def part_c():                                   # three stages: the utterance encoder's backward
        if three:
            model.backward_encoder(state["loss"])

    def fwd_bwd():
        loss = part_a()
        part_b()
        part_c()
        return loss

    def exchange_mid():
        if three:
            reducer.start_mid(sparse_ids)

    def eager_step(i=0):
        select(i)
        loss = part_a()
        reducer.start()
        part_b()
        exchange_mid()
        part_c()
        reducer.finish(sparse_ids)
        optimizer_step()
        return loss

    # warm-up (eager, on a side stream as torch's graph-capture rules ask): allocates every workspace
    for i in range(max(3, args.warmup if not args.graph else 3)):
        eager_step(i)
    torch.cuda.synchronize()
    ops.gru_check_all()

    step = eager_step
    if args.graph:
        # no autograd graph from the warm-up may stay alive (its AccumulateGrad nodes are tied to the warm-up stream)
        # and capture runs on the SAME side stream as the warm-up
        import gc
        state.clear()
        gc.collect()
        # "thread_local": RCCL's watchdog thread polls the events of the warm-up's collectives; under the default
        # ("global") capture mode its hipEventQuery fails with "operation not permitted when stream is capturing" and
        # takes the process down (seen with --rehearse-dp 1).  All work is drained first for the same reason.
        torch.cuda.synchronize()
        CAPTURE_MODE = "thread_local"
        ga, gb, gc_, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # N = 1 without the cut schedule: nothing happens between the backward pass and the optimizer, so zero_grad + forward +
        # backward + guards + clip + Adam are ONE graph (as engine.GraphedStep replays them); a second graph for the optimizer
        # costs the host's launch latency between the two replays once per step
        one_graph = world == 1 and not split and not three and not rehearse and not reducer.external_sparse
        with torch.cuda.graph(ga, stream=side, capture_error_mode=CAPTURE_MODE):
            static_loss = part_a()
            if one_graph:
                optimizer_step()
        if split:
            with torch.cuda.graph(gb, pool=ga.pool(), stream=side, capture_error_mode=CAPTURE_MODE):
                part_b()
        if three:
            with torch.cuda.graph(gc_, pool=ga.pool(), stream=side, capture_error_mode=CAPTURE_MODE):
                part_c()
        if not one_graph:
            with torch.cuda.graph(g2, pool=ga.pool(), stream=side, capture_error_mode=CAPTURE_MODE):
                optimizer_step()
        ops.LAST_GRU_SYNC.clear()

        import collections
        cpu_t = collections.defaultdict(float)

        def step(i=0):
            # This runtime's launch of a multi-queue graph keeps the host inside replay() until most of the graph has run
            # (tools/graph_launch_cost.py), so a collective can only be issued once the graph before it is nearly done: the
            # exchange calls sit BETWEEN the graphs and are kept to one collective per stage (flat buffers ordered by stage)
            T = cpu_t
            t0 = time.perf_counter(); select(i); t1 = time.perf_counter(); T["select"] += t1 - t0
            ga.replay(); t2 = time.perf_counter(); T["ga"] += t2 - t1
            if split:
                reducer.start()
            t3 = time.perf_counter(); T["start"] += t3 - t2
            if split:
                gb.replay()
            t4 = time.perf_counter(); T["gb"] += t4 - t3
            if three:
                reducer.start_mid(sparse_ids)
            t5 = time.perf_counter(); T["start_mid"] += t5 - t4
            if three:
                gc_.replay()
            t6 = time.perf_counter(); T["gc"] += t6 - t5
            if not one_graph:
                reducer.finish(sparse_ids)
            t7 = time.perf_counter(); T["finish"] += t7 - t6
            if not one_graph:
                g2.replay()
            T["g2"] += time.perf_counter() - t7; T["n"] += 1
            return static_loss
    for i in range(args.warmup):
        step(i)
        if rank == 0 and os.environ.get("PEVAE_BENCH_DEBUG"):
            st = opt.state_dict_host()
            bad = [n for n, p in model.named_parameters() if not torch.isfinite(p.grad).all()]
            print("debug warm replay", i, "norm", st["last_norm"], "skip", st["skip"], "nll",
                  float(static_loss.vae_nll.detach()) if args.graph else None, "nonfinite grads:", bad[:6], file=sys.stderr)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()          # host barrier (gloo) between two device synchronisations
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    ops.gru_check_all()
    rccl_ranks, devices = 1, [dp.device_identity(local)]
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
        rccl_ranks = dp.rccl_rank_count(dev)          # what RCCL itself joined: must be N
        devices = [None] * world
        torch.distributed.all_gather_object(devices, dp.device_identity(local))
        if rccl_ranks != world:
            print("bench.py: RCCL joined %d ranks, %d were started" % (rccl_ranks, world), file=sys.stderr)
            sys.exit(3)
    if world > 1 and os.environ.get("PEVAE_BENCH_DEBUG"):
        # replicas must hold bit-identical parameters after any number of steps
        cs = opt.flat_p.double().sum().reshape(1)
        lo, hi = cs.clone(), cs.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        if rank == 0:
            print("debug: parameter checksum min/max over ranks", float(lo), float(hi), "identical:", float(lo) == float(hi),
                  file=sys.stderr)
    if rank == 0 and os.environ.get("PEVAE_BENCH_DEBUG"):
        print("debug: guard state", opt.state_dict_host(), "nll", float(loss.vae_nll.detach()), file=sys.stderr)
    ms = elapsed / args.steps * 1e3
    if args.graph and os.environ.get("PEVAE_BENCH_CPUTIME"):
        print("host us per call:", {k: round(v / cpu_t["n"] * 1e6, 1) for k, v in cpu_t.items() if k != "n"}, file=sys.stderr)
    value = args.batch * world * args.steps / elapsed
    final_nll = float(loss.vae_nll.detach())
    skipped = opt.state_dict_host()["skipped_total"]

    # ---- instrumented eager pass: HIP events on the launch stream around the dominant kernels ---------------
    roof, kern_ms = None, {}
    if rank == 0:
        select(0)
        ops.TIMERS = {}
        # the roofline prices the kernels on the whole chip: the instrumented pass launches the dW kernel once, on all CUs, in
        # front of the decoder recurrence (the timed steps above ran it in two half-chip launches beside the recurrences)
        beside, ops.VOCAB_DW_BESIDE = ops.VOCAB_DW_BESIDE, None
        for _ in range(args.profile_steps):
            fwd_bwd()
        ops.VOCAB_DW_BESIDE = beside
        torch.cuda.synchronize()
        for tag, evs in ops.TIMERS.items():
            per_step = len(evs) // args.profile_steps
            tot = sum(a.elapsed_time(b) for a, b in evs) / args.profile_steps
            kern_ms[tag] = dict(ms_per_step=round(tot, 4), launches_per_step=per_step)
        ops.TIMERS = None
        T, H, V = args.batch * cfg.max_seq_len, cfg.dec_cell_size, cfg.vocab_size
        # Only target rows that are not padding reach the vocabulary GEMMs (their NLL and gradients are exactly zero in
        # the reference too: NLLEntropy ignore_index, criterions.py), so only those rows are credited as work.
        select(0)
        torch.cuda.synchronize()
        T_live = int((feed["ans_ids"] != 0).sum()) if ops.SKIP_IGNORED_TOKENS else T     # the profiled pass runs batch 0
        alg_flop = 6.0 * T_live * H * V                  # fwd + dH + dW; the backward's logit recompute is NOT credited
        vocab_tags = [k for k in kern_ms if k.startswith("vocab_")]
        vocab_ms = sum(kern_ms[k]["ms_per_step"] for k in vocab_tags)
        ach = alg_flop / (vocab_ms * 1e-3) / 1e12
        # HBM bytes per step of the group: measured at the start of this run (measure_traffic), else the committed passes
        traffic, traffic_source, traffic_kernels = None, None, None
        if measured is not None and measured[0] is not None:
            traffic, traffic_kernels = measured
            traffic_source = ("measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (one pass each) over an "
                              "eager 5-step child run; per kernel median of FETCH_SIZE x2 + WRITE_SIZE (KiB), summed, per step")
        elif args.traffic != "off":
            pmc = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
            if os.path.exists(pmc):
                with open(pmc) as fh:
                    traffic = json.load(fh).get("vocab_group_hbm_bytes_per_step")
                traffic_source = "profiles/r03_pmc_traffic.json (FETCH_SIZE x2 + WRITE_SIZE, bytes per step; not measured in this run)"
        fused = "vocab_fused_fwd" in kern_ms
        fwd_tag = "vocab_fused_fwd" if fused else "vocab_fwd"
        # the forward entry point: logits + softmax + NLL (2 T H V); fused, it also carries the dH product (4 T H V)
        fwd_flop = (4.0 if fused else 2.0) * T_live * H * V
        roof = dict(bound="mfma", achieved=round(ach, 2), peak=BF16_DENSE_PEAK_TFLOPS, unit="TFLOP/s",
                    frac=round(ach / BF16_DENSE_PEAK_TFLOPS, 4), traffic=traffic,
                    traffic_source=traffic_source, traffic_per_kernel=traffic_kernels,
                    measured_on=("the instrumented eager pass (%d steps, HIP events on the launch stream), every kernel launched on "
                                 "the whole chip" % args.profile_steps) +
                                ("; the timed steps run vocab_dw_kernel as two half-vocabulary launches on half the CUs each, beside "
                                 "the decoder's and the encoder's backward recurrence" if (beside and not split and not three) else ""),
                    kernel=("vocab_fwd_dh_kernel + vocab_fwd_combine_kernel + vocab_dw_kernel + vocab_dw_reduce_kernel "
                            "(fused vocabulary projection + softmax + NLL forward, dH and dW; no [T,V] matrix in HBM)") if fused
                    else "gemm_nt_kernel<EpiSoftmax|EpiProb|EpiStore> (vocab projection+softmax+NLL fwd+bwd group)",
                    algorithmic_gflop_per_step=round(alg_flop / 1e9, 1), group_ms_per_step=round(vocab_ms, 4),
                    executed_gflop_per_step=round((8.0 if fused else 8.0) * T_live * H * V / 1e9, 1),
                    token_rows=T, token_rows_computed=T_live,
                    fwd_kernel_tflops=round(fwd_flop / (kern_ms[fwd_tag]["ms_per_step"] * 1e-3) / 1e12, 2)
                    if fwd_tag in kern_ms else None)

    if rank == 0:
        cpu = cpu_baseline(cfg, args.batch, args.cpu_steps) if (args.cpu_baseline and world == 1) else None
        out = {
            "metric": "training samples/sec (whole node) DLS_VAE Yelp-shape bs256", "value": round(value, 2),
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "DLS_VAE %s-shape bs=%d/GPU bf16 (configs[%d]): embed=100 dec_cell=512 L=50 "
                                   "V=20004 NU=%d NI=%d Z=64, dropout=%.1f, Adam lr=1e-4 clip=5, latent stack '%s'%s%s%s"
                                   % (args.config, args.batch, {"yelp": 1, "movie": 2, "book": 3, "mutual": 4}[args.config],
                                      cfg.n_user, cfg.n_item, args.dropout, cfg.latent_stack,
                                      ", decoder attention" if cfg.use_attention else "",
                                      ", DEM mutual-information stack (use_mutual)" if cfg.use_mutual else "",
                                      ", rating head (do_pred)" if cfg.do_pred else ""),
                       "resident_batches": len(batches), "skipped_steps": skipped,
                       "dp_overlap": bool(args.overlap) and world > 1,
                       "dp_exchange_stages": (3 if three else 2 if split else 1) if (world > 1 or rehearse) else 0,
                       "dp_rehearsal_one_rank": rehearse,
                       "global_batch": args.batch * world, "seq_len": cfg.max_seq_len, "params": n_params,
                       "parallelism": "dp%d" % world, "rccl_ranks": rccl_ranks, "devices": devices,
                       "hip_graph": bool(args.graph),
                       "vocab_dw_beside_recurrences": (list(ops.VOCAB_DW_BESIDE) if ops.VOCAB_DW_BESIDE and not split and not three
                                                       else None),
                       "padding_rows_skipped": bool(ops.SKIP_IGNORED_TOKENS),
                       "live_target_fraction": round(float(sum((b["ans_ids"] != 0).float().mean() for b in batches)) / len(batches), 4)},
            "roofline": roof, "cpu_baseline": cpu, "kernels_ms_per_step": kern_ms, "final_vae_nll": round(final_nll, 4),
        }
        print(json.dumps(out))
    if world > 1 or rehearse:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

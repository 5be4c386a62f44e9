#!/usr/bin/env python3
"""Forward-DP throughput bench (BASELINE.json metric: forward-DP cells/s and % of the
HBM roofline on LxL profile pairs; 1/2/4/8 GPUs).

One "step" = one pass of the hot path (profile prep + Forward fill + lpEnd) over one
batch of independent synthetic pair DPs, inputs already resident in HBM.  Workload at
every N: BASELINE.json configs[3] -- a batch of independent 2x2000-residue protein
profile pairs under WAG -- `--pairs` of them PER GPU (weak scaling: independent tree
nodes are farmed across ranks, no data-path collective; the only RCCL traffic is the
one-off broadcast of the rate-model constant block).

The line carries, beside the headline (the full, unbanded DP):
  * `banded_mode`: the same kind of pairs inside a band-20 GuideAlignmentEnvelope - the reference's default mode and
    north_star's own target ("the banded forward DP for 2x2000-residue profile pairs") - at 512 pairs and at a
    batch large enough to fill the chip, in-envelope cells counted;
  * for N > 1, `strong`: configs[3] as written - 512 pairs IN TOTAL dealt to the N ranks.

  python bench.py                      # 1 GPU, defaults finish in a few minutes
  python bench.py --gpus N             # starts N rank processes itself (torch.distributed.run), one per GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N      # the driver's form
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
BYTES_PER_CELL = 40          # 5 fp64 states written once per cell (reference forward.h:13-15,107)
MODES = ("trunc", "fast", "exact", "linear")
MODE_KEY = {"exact": "exact_mode", "fast": "fast_mode", "linear": "scaled_probability_mode",
            "trunc": "truncating_scaled_probability_mode"}
KERNELS = {("exact", False): "hx::k_fill_chain<0,...,ExactLse3>", ("fast", False): "hx::k_fill_chain<0,...,FastLse>",
           ("linear", False): "hx::k_fill_leaf_linear<W>", ("trunc", False): "hx::k_fill_leaf_linear<W,...,TRUNC>",
           ("exact", True): "hx::k_fill_band<exact>", ("fast", True): "hx::k_fill_band<fast>",
           ("linear", True): "hx::k_fill_band<scaled> / hx::k_fill_band2<false> (two pairs per wavefront above 1024 pairs)",
           ("trunc", True): "hx::k_fill_band<truncating scaled> / hx::k_fill_band2<true> (two pairs per wavefront above 1024 pairs)"}
ARITH = {"exact": "the reference's table log-sum-exp, cells bit-identical to the reference recursion",
         "fast": "LDS-table log-sum-exp with the reference's truncation (lpEnd within 1e-9 rel. of the reference's, best paths "
                 "identical to the reference's: 10000 of 10000 pairs, profiles/r03/trace_identity_sweep_seed11.json)",
         "trunc": "scaled-probability recursion with the reference's truncation: every pairwise sum of the reference's left-nested "
                  "log_sum_exp drops a term that is at most e^-10 of the other, as the reference's table does (src/logsumexp.h:45); "
                  "no table, log-probabilities at the store; lpEnd within 1e-9 rel. of the reference's, best paths identical to the "
                  "reference's: 10000 of 10000 pairs (profiles/r03/trace_identity_sweep_seed11.json)",
         "linear": "scaled-probability recursion (fp64 sums of probabilities with a per-cell exponent, log-probabilities at the "
                   "store; no truncation of small terms: lpEnd within 3e-5 rel., best paths differ at near-ties: 14 of 10000, "
                   "profiles/r03/trace_identity_sweep_seed11.json)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=512, help="independent pair DPs per GPU (weak scaling) or in total (strong)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --pairs pair DPs on every GPU.  strong: BASELINE configs[3] as written, --pairs pair DPs in "
                         "total dealt to the GPUs (with N > 1 the weak line carries a `strong` block anyway)")
    ap.add_argument("--len", type=int, default=2000, dest="length", help="residues per sequence")
    ap.add_argument("--model", default="wag")
    ap.add_argument("--tl", type=float, default=0.2)
    ap.add_argument("--tr", type=float, default=0.3)
    ap.add_argument("--cpu-pairs", type=int, default=10, help="pairs timed by the CPU baseline (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--band", type=int, default=-1,
                    help="guide-alignment band (maxDistanceFromGuide) of the HEADLINE workload; the guide is the pair's true "
                         "alignment.  -1 = full envelope (the headline configuration; the banded block is added to it)")
    ap.add_argument("--mode", choices=list(MODES), default="trunc",
                    help="arithmetic of the headline fill.  trunc (default) = scaled probabilities with the reference's truncation; "
                         "fast = LDS-table log-sum-exp with the reference's truncation; both have best paths identical to the "
                         "reference's on all 2000 pairs of tools/sweep_trace_identity.py.  exact = the reference's table bit for "
                         "bit.  linear = scaled probabilities without the truncation (best paths may differ at near-ties).  The "
                         "other policies are timed too")
    ap.add_argument("--single-mode", action="store_true", help="time only --mode; no banded block, no strong block")
    ap.add_argument("--banded-pairs", default="512,2560,4096", help="batch sizes of the banded block (band 20), comma separated; '' = none")
    ap.add_argument("--traffic", type=float, default=None,
                    help="HBM bytes per launch from a PMC run (default: profiles/traffic.json entry for this workload)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N rank processes (one per GPU) as fresh children - before
    anything in this process touches the GPU - and pass rank 0's line through."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))


def _cpu_fill_worker(spec):
    """CPU-baseline worker (all-cores variant, SURVEY 8d iii): rebuilds pair k of the batch from its seed and runs the
    plain-C restatement of the reference fill on it.  Runs in a spawned process (the parent has initialised the GPU)."""
    model_name, tl, tr, length, band, seed = spec
    from historian_amd import hostmodel, workload
    from oracle import c_oracle
    model = hostmodel.RateModel.load(os.path.join(ROOT, "tests", "golden", "models", model_name + ".json"))
    hmm = hostmodel.make_hmm(model, tl, tr)
    x, y, h, md = workload.leaf_pair(np.random.default_rng(seed), model, hmm, length, band=band)
    t0 = time.perf_counter()
    c_oracle.forward(x, y, h, md)
    return time.perf_counter() - t0


def main():
    args = parse()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: the launcher's rank count and --gpus must agree" % (args.gpus, world))
    strong = args.scaling == "strong"
    if strong and args.pairs < world:
        raise SystemExit("bench.py: --scaling strong needs at least one pair per rank (--pairs %d, %d ranks)" % (args.pairs, world))
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # HX_BENCH_REHEARSE=1: the multi-rank flow (dealing, broadcast, barriers, reductions, rank 0's line) on a box with ONE
    # GPU - every rank on device 0, collectives over gloo on host tensors.  For checking the N > 1 path, not for numbers.
    rehearse = bool(os.environ.get("HX_BENCH_REHEARSE"))
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    coll_dev = None if rehearse else dev      # where the collectives' tensors live (nccl: on the rank's GPU)

    from historian_amd import capi, farm, hostmodel, workload

    # ---- rate-model constant block: built on rank 0, broadcast over RCCL/xGMI ----------------
    model = hostmodel.RateModel.load(os.path.join(ROOT, "tests", "golden", "models", args.model + ".json"))
    block = farm.constant_block(model, args.tl, args.tr) if rank == 0 else None
    block = farm.broadcast_block(block, farm.block_len(model), rank, world, coll_dev)
    table, sub_l, sub_r = farm.split_block(model, block)

    capi.init(dev_index, table)
    hmm = hostmodel.make_hmm(model, args.tl, args.tr, sub_l, sub_r)
    stream = torch.cuda.current_stream().cuda_stream
    flags_of = {"exact": capi.HX_LSE_EXACT, "fast": capi.HX_LSE_FAST, "linear": capi.HX_LSE_LINEAR, "trunc": capi.HX_LSE_TRUNC}

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def build(n, first, band):
        """pairs `first` .. `first + n` of the global sequence of pairs (seed = 1000 + global index); with a
        band also each pair's in-envelope cells (reference src/forward.h:92-98: within the band, or at an edge)"""
        triples, env = [], []
        for k in range(n):
            rng = np.random.default_rng(farm.pair_seed(0, 0, first + k))
            triples.append(workload.leaf_pair(rng, model, hmm, args.length, band=band))
            if band >= 0:
                env.append(workload.in_envelope_cells(triples[-1][0].env_pos, triples[-1][1].env_pos, band))
            if rank == 0 and n > 1024 and (k + 1) % 1024 == 0:
                print("built %d of %d pairs" % (k + 1, n), file=sys.stderr, flush=True)
        return triples, env

    def timed(triples, mode, band, keep_traces=0):
        """W warm-up passes, then exactly K timed passes of the hot path between barrier + synchronize on both sides; the
        time is the MAX over ranks."""
        # banded batches are stored band-compressed (per strip only the swept step windows): thousands of pairs fit
        batch = capi.Batch(triples, flags_of[mode] | (capi.HX_BAND_COMPRESSED if band >= 0 else 0))
        n_cells = batch.total_cells()
        shared = batch.shared_wavefront_pairs()
        for _ in range(args.warmup):
            batch.forward(stream)
        barrier()
        t0 = time.perf_counter()
        k_ms = []
        for _ in range(args.steps):
            batch.forward(stream)
            k_ms.append(batch.kernel_ms(0))         # HIP events around the fill kernel, on its stream
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = farm.max_over_ranks(time.perf_counter() - t0, world, coll_dev)
        lp = batch.lp_end()
        traces = None
        if keep_traces:
            # outside the timed region: the device-side best-path traceback of every pair (hx_batch_best_trace); the first
            # few paths are compared with the CPU oracle's below
            tcells, tlen = batch.best_trace(raw=True)
            traces = [[tuple(int(v) for v in c) for c in tcells[k, :tlen[k]]] for k in range(min(keep_traces, len(triples)))]
        batch.close()
        assert np.all(np.isfinite(lp)) or os.environ.get("HX_BENCH_NOCHECK"), "non-finite Forward log-likelihood"
        return dict(dt=dt, kernel_ms=float(np.mean(k_ms)), lp_end=lp, cells=n_cells, traces=traces, shared_wavefront_pairs=shared)

    # ---- the headline workload ------------------------------------------------------------------------------------
    # strong scaling: BASELINE configs[3] as written - `--pairs` pairs IN TOTAL, dealt to the ranks; weak: per rank
    n_local, first = farm.deal(args.pairs, world, rank, strong)
    triples, env_cells_of = build(n_local, first, args.band)
    order = [args.mode] + ([m for m in MODES if m != args.mode] if not args.single_mode else [])
    runs = {m: timed(triples, m, args.band, keep_traces=args.cpu_pairs) for m in order}
    head = runs[args.mode]
    cells = sum(env_cells_of) if args.band >= 0 else head["cells"]      # the metric counts in-envelope cells
    gcells = farm.sum_over_ranks(cells, world, coll_dev) if strong else cells * world

    # ---- the strong-scaling block (N > 1 under a weak headline): configs[3] as written, 512 pairs in total ----------
    strong_block = None
    if world > 1 and not strong and not args.single_mode and args.band < 0:
        s_total = 512
        s_local, s_first = farm.deal(s_total, world, rank, True)
        # (rank 0's slice is a prefix of its weak batch; the other ranks build theirs)
        s_triples = triples[:s_local] if (s_first == first and s_local <= n_local) else build(s_local, s_first, -1)[0]
        r = timed(s_triples, args.mode, -1)
        s_cells = farm.sum_over_ranks(r["cells"], world, coll_dev)
        strong_block = {"workload": "BASELINE configs[3] as written: %d pairs in total dealt to the %d ranks" % (s_total, world),
                        "pairs_total": s_total, "pairs_per_gpu": s_local, "value": s_cells * args.steps / r["dt"], "unit": "cells/s",
                        "ms_per_step": r["dt"] / args.steps * 1e3, "scaling": "strong", "fill_mode": args.mode}

    # ---- the banded block: band 20, the reference's default mode and north_star's own target ------------------------
    banded_block = None
    if args.band < 0 and not args.single_mode and args.banded_pairs:
        banded_block = {"band": 20, "fill_mode": args.mode, "arithmetic": ARITH[args.mode], "kernel": KERNELS[(args.mode, True)],
                        "cells_counted": "in-envelope cells (band + envelope edges), band-compressed storage", "unit": "cells/s",
                        "scaling": "weak", "batches": []}
        sizes = sorted(int(v) for v in args.banded_pairs.split(","))
        # (weak: every rank its own pairs; a rank's smaller batch is a prefix of its larger one)
        have, have_env = build(sizes[-1], rank * sizes[-1], 20)
        for n_b in sizes:
            b_env = have_env[:n_b]
            r = timed(have[:n_b], args.mode, 20)
            b_cells = sum(b_env)
            banded_block["batches"].append({
                "pairs_per_gpu": n_b, "in_envelope_cells_per_gpu": b_cells, "value": b_cells * world * args.steps / r["dt"],
                "ms_per_step": r["dt"] / args.steps * 1e3, "kernel_ms": r["kernel_ms"],
                "pairs_sharing_a_wavefront": r["shared_wavefront_pairs"],
                "roofline_frac": b_cells * BYTES_PER_CELL / (r["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS})

    if rank == 0:
        dt, k_ms, lp_end = head["dt"], head["kernel_ms"], head["lp_end"]
        value = gcells * args.steps / dt
        traffic = args.traffic
        if traffic is None:
            try:        # measured with rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes), see DESIGN.md
                with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                    traffic = json.load(f).get("%s:%d:%d" % (args.mode, n_local, args.length)) if args.band < 0 else None
            except OSError:
                traffic = None
        achieved = cells * BYTES_PER_CELL / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "forward-DP cells/s", "value": value, "unit": "cells/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "batch of independent 2x%d-residue protein leaf-profile pairs, %s, t=%g/%g, %s Forward DP; %s" %
                                   (args.length, args.model.upper(), args.tl, args.tr,
                                    "full (unbanded)" if args.band < 0 else
                                    "band-%d (guide = the pair's true alignment; in-envelope cells counted; band-compressed storage)"
                                    % args.band, ARITH[args.mode]),
                       "pairs_per_gpu": n_local, "pairs_total": n_local * world if not strong else args.pairs,
                       "cells_per_gpu_per_step": cells,
                       "parallelism": "pairs farmed across %d rank(s); RCCL broadcast of model constants only" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": KERNELS[(args.mode, args.band >= 0)],
                         "kernel_ms": k_ms, "bytes_per_cell": BYTES_PER_CELL},
            "fill_mode": args.mode,
            "lp_end_pair0": float(lp_end[0]),
        }
        ref_lp = runs["exact"]["lp_end"] if "exact" in runs else None
        for m in order[1:]:
            r = runs[m]
            out[MODE_KEY[m]] = {
                "arithmetic": ARITH[m], "kernel": KERNELS[(m, args.band >= 0)],
                "value": gcells * args.steps / r["dt"], "unit": "cells/s", "ms_per_step": r["dt"] / args.steps * 1e3,
                "kernel_ms": r["kernel_ms"], "roofline_frac": cells * BYTES_PER_CELL / (r["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if ref_lp is not None:
            out["lp_end_max_rel_diff_to_exact"] = {m: float(np.max(np.abs(runs[m]["lp_end"] - ref_lp) / np.abs(ref_lp))) for m in order if m != "exact"}
        if banded_block is not None:
            out["banded_mode"] = banded_block
        if strong_block is not None:
            out["strong"] = strong_block
        if world == 1 and not args.no_cpu_baseline:
            from oracle import c_oracle           # the checker, timed as the CPU baseline ("port")
            c_oracle.load()
            n_cpu = max(1, min(args.cpu_pairs, n_local))
            from oracle import trace_oracle
            cpu_dt = 0.0
            t1 = time.perf_counter()
            cpu_cells = 0
            rel = 0.0
            same = {}
            for k in range(n_cpu):
                x, y, h, md = triples[k]
                r = c_oracle.forward(x, y, h, md)
                cpu_cells += env_cells_of[k] if args.band >= 0 else (x.n_states - 1) * (y.n_states - 1)
                rel = max(rel, abs(r["lp_end"] - lp_end[k]) / abs(r["lp_end"]))
                # traceback identity: the reference's bestTrace over the CPU matrix vs the device's paths
                cpu_dt += time.perf_counter() - t1
                path = trace_oracle.best_trace(x, y, h, md, r)
                for m in runs:
                    same[m] = same.get(m, 0) + (runs[m]["traces"][k] == path)
                t1 = time.perf_counter()
            out["cpu_baseline"] = {"value": cpu_cells / cpu_dt, "unit": "cells/s", "cores": 1, "kind": "port",
                                   "sample": "first %d pair(s) of the same batch, oracle/oracle_fill.c "
                                             "(dense-array restatement of the reference fill), 1 thread, %.1f s"
                                             % (n_cpu, cpu_dt)}
            # the same fill over the reference's own cell storage (a std::map per row, oracle_fill_map.cpp): the CPU
            # baseline with the reference's cost structure (variant ii), on two pairs
            n_map = min(2, n_cpu)
            t1 = time.perf_counter()
            for k in range(n_map):
                x, y, h, md = triples[k]
                c_oracle.forward_map(x, y, h, md)
            map_dt = time.perf_counter() - t1
            out["cpu_baseline"]["map_storage_value"] = (cpu_cells / n_cpu) * n_map / map_dt
            # variant iii: all host cores farming independent pairs (the reference itself is single-threaded)
            import multiprocessing as mp
            n_proc = max(1, min(os.cpu_count() or 1, 64))
            n_all = max(n_proc, 2 * n_proc if args.band >= 0 else n_proc)
            specs = [(args.model, args.tl, args.tr, args.length, args.band, farm.pair_seed(0, 0, first + (k % n_local))) for k in range(n_all)]
            with mp.get_context("spawn").Pool(n_proc) as pool:
                pool.map(_cpu_fill_worker, specs[:n_proc])          # (start-up: imports, library load)
                t1 = time.perf_counter()
                pool.map(_cpu_fill_worker, specs, chunksize=1)
                all_dt = time.perf_counter() - t1
            out["cpu_baseline"]["all_cores"] = {"value": (cpu_cells / n_cpu) * n_all / all_dt, "unit": "cells/s", "cores": n_proc,
                                                "sample": "%d pairs of the same workload over %d processes, %.1f s" % (n_all, n_proc, all_dt)}
            out["lp_end_max_rel_err_vs_cpu"] = rel
            out["best_trace_identical_to_cpu"] = {m: "%d of %d pairs" % (same.get(m, 0), n_cpu) for m in runs}
            assert rel <= 1e-4, "Forward log-likelihoods outside north_star's tolerance of the CPU path: %g" % rel
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

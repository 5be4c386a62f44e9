#!/usr/bin/env python3
"""Forward-DP throughput bench (BASELINE.json metric: forward-DP cells/s and % of the
HBM roofline on LxL profile pairs; 1/2/4/8 GPUs).

One "step" = one pass of the hot path (profile prep + Forward fill + lpEnd) over one
batch of independent synthetic pair DPs, inputs already resident in HBM.  Workload at
every N: BASELINE.json configs[3] -- a batch of independent 2x2000-residue protein
profile pairs under WAG -- `--pairs` of them PER GPU (weak scaling: independent tree
nodes are farmed across ranks, no data-path collective; the only RCCL traffic is the
one-off broadcast of the rate-model constant block).

  python bench.py                      # 1 GPU, defaults finish in a few minutes
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
BYTES_PER_CELL = 40          # 5 fp64 states written once per cell (reference forward.h:13-15,107)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=512, help="independent pair DPs per GPU (weak scaling) or in total (strong)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --pairs pair DPs on every GPU.  strong: BASELINE configs[3] as written, --pairs pair DPs in "
                         "total dealt to the GPUs")
    ap.add_argument("--len", type=int, default=2000, dest="length", help="residues per sequence")
    ap.add_argument("--model", default="wag")
    ap.add_argument("--tl", type=float, default=0.2)
    ap.add_argument("--tr", type=float, default=0.3)
    ap.add_argument("--cpu-pairs", type=int, default=10, help="pairs timed by the CPU baseline (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--band", type=int, default=-1,
                    help="guide-alignment band (maxDistanceFromGuide); the guide is the pair's true alignment. "
                         "-1 = full envelope (the headline configuration)")
    ap.add_argument("--mode", choices=["exact", "fast", "linear", "trunc"], default="fast",
                    help="arithmetic of the headline fill.  fast (default) = LDS-table log-sum-exp with the reference's truncation: "
                         "best paths identical to the reference's on all 2000 pairs of tools/sweep_trace_identity.py.  exact = the "
                         "reference's table bit for bit.  linear = scaled probabilities (HX_LSE_LINEAR): fastest, but without the "
                         "truncation 4 of those 2000 best paths differ, so it is reported as a secondary line.  The other "
                         "policies are timed too")
    ap.add_argument("--single-mode", action="store_true", help="time only --mode")
    ap.add_argument("--traffic", type=float, default=None,
                    help="HBM bytes per launch from a PMC run (default: profiles/traffic.json entry for this workload)")
    return ap.parse_args()


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
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    a, c = len(model.alphabet), model.components()
    block = farm.constant_block(model, args.tl, args.tr) if rank == 0 else None
    block = farm.broadcast_block(block, farm.block_len(model), rank, world, coll_dev)
    table, sub_l, sub_r = farm.split_block(model, block)

    capi.init(dev_index, table)
    hmm = hostmodel.make_hmm(model, args.tl, args.tr, sub_l, sub_r)
    pi = np.asarray(model.root[0], dtype=float)
    pi = pi / pi.sum()

    # strong scaling: BASELINE configs[3] as written - `--pairs` pairs IN TOTAL, dealt to the ranks; weak: per rank
    strong = args.scaling == "strong"
    n_local, first = farm.deal(args.pairs, world, rank, strong)
    triples = []
    env_cells = 0
    env_cells_of = []
    for k in range(n_local):
        rng = np.random.default_rng(farm.pair_seed(0, 0, first + k))       # seed = 1000 + global pair index
        triples.append(workload.leaf_pair(rng, model, hmm, args.length, band=args.band))
        if args.band >= 0:
            # in-envelope cells (reference src/forward.h:92-98): within the band, or at an edge
            n_in = workload.in_envelope_cells(triples[-1][0].env_pos, triples[-1][1].env_pos, args.band)
            env_cells_of.append(n_in)
            env_cells += n_in
        if rank == 0 and n_local > 1024 and (k + 1) % 1024 == 0:
            print("built %d of %d pairs" % (k + 1, n_local), file=sys.stderr, flush=True)
    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    traces = {}

    def run_mode(mode):
        """K timed passes of the hot path in one fill mode; returns (seconds, kernel ms list, lp_end, cells)."""
        # banded batches are stored band-compressed (per strip only the swept step windows): thousands of pairs fit
        batch = capi.Batch(triples, {"exact": capi.HX_LSE_EXACT, "fast": capi.HX_LSE_FAST, "linear": capi.HX_LSE_LINEAR, "trunc": capi.HX_LSE_TRUNC}[mode] |
                           (capi.HX_BAND_COMPRESSED if args.band >= 0 else 0))
        n_cells = batch.total_cells()
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
        dt = time.perf_counter() - t0
        dt = farm.max_over_ranks(dt, world, coll_dev)
        lp = batch.lp_end()
        # outside the timed region: the device-side best-path traceback of every pair (hx_batch_best_trace); the first
        # few paths are compared with the CPU oracle's below
        tcells, tlen = batch.best_trace(raw=True)
        n_keep = max(1, min(args.cpu_pairs, n_local))
        traces[mode] = [[tuple(int(v) for v in c) for c in tcells[k, :tlen[k]]] for k in range(n_keep)]
        batch.close()
        assert np.all(np.isfinite(lp)) or os.environ.get("HX_BENCH_NOCHECK"), "non-finite Forward log-likelihood"
        return dt, k_ms, lp, n_cells

    order = [args.mode] + ([m for m in ("trunc", "fast", "exact", "linear") if m != args.mode] if not args.single_mode else [])
    runs = {m: run_mode(m) for m in order}
    dt, kernel_ms, lp_end, cells = runs[args.mode]

    KERNELS = {("exact", False): "hx::k_fill_chain<0,...,ExactLse3>", ("fast", False): "hx::k_fill_chain<0,...,FastLse>",
               ("linear", False): "hx::k_fill_leaf_linear<W>", ("exact", True): "hx::k_fill_band<exact>",
               ("fast", True): "hx::k_fill_band<fast>", ("linear", True): "hx::k_fill_band<scaled>",
               ("trunc", False): "hx::k_fill_leaf_linear<W,...,TRUNC>", ("trunc", True): "hx::k_fill_band<truncating scaled>"}
    ARITH = {"exact": "the reference's table log-sum-exp, cells bit-identical to the reference recursion",
             "fast": "LDS-table log-sum-exp with the reference's truncation (lpEnd within 1e-9 rel. of the reference's, best paths "
                     "identical to the reference's: 2000 of 2000 pairs, profiles/r02/trace_identity_sweep.json)",
             "trunc": "scaled-probability recursion with the reference's truncation (every pairwise sum of the reference's left-nested "
                      "log_sum_exp drops a term that is at most e^-10 of the other, as the reference's table does; no table, "
                      "log-probabilities at the store)",
             "linear": "scaled-probability recursion (fp64 sums of probabilities with a per-cell exponent, log-probabilities at the "
                       "store; no truncation of small terms: lpEnd within 1e-5 rel., 4 of 2000 best paths differ from the reference's)"}
    if rank == 0:
        if args.band >= 0:
            cells = env_cells          # the metric counts in-envelope cells
        gcells = farm.sum_over_ranks(cells, world, coll_dev) if strong else cells * world
    elif strong:
        farm.sum_over_ranks(env_cells if args.band >= 0 else cells, world, coll_dev)
    if rank == 0:
        total_cells = gcells * args.steps
        value = total_cells / dt
        k_ms = float(np.mean(kernel_ms))
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
        ref_lp = runs["exact"][2] if "exact" in runs else None
        for m in order[1:]:
            dt_m, k_list, lp_m, _ = runs[m]
            k_m = float(np.mean(k_list))
            out[{"exact": "exact_mode", "fast": "fast_mode", "linear": "scaled_probability_mode", "trunc": "truncating_scaled_probability_mode"}[m]] = {
                "arithmetic": ARITH[m], "kernel": KERNELS[(m, args.band >= 0)],
                "value": gcells * args.steps / dt_m, "unit": "cells/s", "ms_per_step": dt_m / args.steps * 1e3, "kernel_ms": k_m,
                "roofline_frac": cells * BYTES_PER_CELL / (k_m * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if ref_lp is not None:
            out["lp_end_max_rel_diff_to_exact"] = {m: float(np.max(np.abs(runs[m][2] - ref_lp) / np.abs(ref_lp))) for m in order if m != "exact"}
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
                for m in traces:
                    same[m] = same.get(m, 0) + (traces[m][k] == path)
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
            out["best_trace_identical_to_cpu"] = {m: "%d of %d pairs" % (same.get(m, 0), n_cpu) for m in traces}
            assert rel <= 1e-4, "Forward log-likelihoods outside north_star's tolerance of the CPU path: %g" % rel
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

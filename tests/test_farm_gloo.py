"""N>1 path on CPU: two gloo ranks run the farm plumbing of bench.py -- constant-block
broadcast, pair dealing, max-over-ranks timing -- with the CPU oracle standing in for the
HIP fill (there is no GPU here; the GPU path itself is covered by the -m gpu tests)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, pairs, length, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from historian_amd import farm, hostmodel, workload
    from oracle import c_oracle
    model = hostmodel.RateModel.load(os.path.join(ROOT, "tests", "golden", "models", "jc.json"))
    block = farm.constant_block(model, .2, .3) if rank == 0 else None
    block = farm.broadcast_block(block, farm.block_len(model), rank, world)
    table, sub_l, sub_r = farm.split_block(model, block)
    hmm = hostmodel.make_hmm(model, .2, .3, sub_l, sub_r)
    pi = np.asarray(model.root[0])
    out = []
    for k in range(pairs):
        rng = np.random.default_rng(farm.pair_seed(rank, pairs, k))
        xs, ys = workload.synth_pair(rng, pi / pi.sum(), length)
        x, y = hostmodel.leaf_profile(xs, 4), hostmodel.leaf_profile(ys, 4)
        out.append(c_oracle.forward(x, y, hmm)["lp_end"])
    t = farm.max_over_ranks(1.0 + rank, world)
    # strong scaling (bench.py --scaling strong): 7 pair DPs in total dealt to the ranks, cell counts summed over ranks
    n_local, first = farm.deal(7, world, rank, True)
    total = farm.sum_over_ranks(n_local, world)
    q.put((rank, float(np.sum(table)), [float(np.sum(m)) for m in sub_l], out, t, (n_local, first, total)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_farm_matches_single_process():
    world, pairs, length = 2, 3, 40
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, pairs, length, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # every rank received rank 0's constants
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2]
    # max over ranks of the per-rank times
    assert res[0][4] == res[1][4] == 2.0
    # strong dealing: contiguous slices that cover the 7 pairs exactly once; the all-reduced count is the total
    assert [r[5] for r in res] == [(4, 0, 7), (3, 4, 7)]
    # union over ranks == what one process with world*pairs pairs computes
    sys.path.insert(0, ROOT)
    import bench
    from historian_amd import farm, hostmodel, workload
    from oracle import c_oracle
    model = hostmodel.RateModel.load(os.path.join(ROOT, "tests", "golden", "models", "jc.json"))
    hmm = hostmodel.make_hmm(model, .2, .3)
    pi = np.asarray(model.root[0])
    single = []
    for k in range(world * pairs):
        rng = np.random.default_rng(farm.pair_seed(0, world * pairs, k))
        xs, ys = workload.synth_pair(rng, pi / pi.sum(), length)
        single.append(c_oracle.forward(hostmodel.leaf_profile(xs, 4), hostmodel.leaf_profile(ys, 4), hmm)["lp_end"])
    assert res[0][3] + res[1][3] == single

"""Throughput of the counts-mode column kernel (hx_sumprod_columns): columns per second on a balanced tree with the
4-component protein mixture, the kernel's own duration (HIP events) beside the whole call (upload + kernel + download), and
the oracle's rate on a sample of the same columns (one core, numpy).

    python tools/sumprod_bench.py [columns] [leaves]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from historian_amd import capi, counts, hostmodel  # noqa: E402


def balanced(leaves, rng):
    parent, length, level = [], [], []
    for _ in range(leaves):
        level.append(len(parent))
        parent.append(-1)
        length.append(float(rng.uniform(.05, .4)))
    while len(level) > 1:
        nxt = []
        for k in range(0, len(level) - 1, 2):
            node = len(parent)
            parent.append(-1)
            length.append(float(rng.uniform(.05, .4)))
            parent[level[k]] = parent[level[k + 1]] = node
            nxt.append(node)
        if len(level) % 2:
            nxt.append(level[-1])
        level = nxt
    return parent, length


def main():
    n_cols = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    leaves = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    model = hostmodel.RateModel.load(os.path.join(ROOT, "tests", "golden", "models", "prot4.json"))
    rng = np.random.default_rng(3)
    parent, length = balanced(leaves, rng)
    n, a, c = len(parent), len(model.alphabet), model.components()
    # the shape of a reconstruction (`historian count -recon`): residues at the leaves, unknown residues ('*') at the
    # ancestors, so that every internal node carries full message vectors; no gaps
    tok = rng.integers(0, a, (n_cols, n)).astype(np.int8)
    tok[:, leaves:] = -1
    capi.init(0, hostmodel.lse_table())
    cc = counts.ColumnCounter(model, parent, length)
    cc.run(tok[:1000])
    t0 = time.perf_counter()
    out = cc.run(tok)
    wall = time.perf_counter() - t0
    ms = capi.sumprod_kernel_ms()
    # per component: G and D matrix-vector products on every branch, E and U on the branches above internal nodes,
    # the outer-product term D_k U_l on every branch
    flops = c * ((n - 1) * (4 * a * a + 2 * a * a) + (n - leaves - 1) * 4 * a * a)
    scratch = c * n * a * 8 * 4 + 8 * (4 * c * n + 2 * c * a)                    # E, G, U, D + scalars, written once
    line = dict(metric="sumprod_columns_per_s", columns=n_cols, nodes=n, components=c, alphabet=a, kernel_ms=ms,
                columns_per_s_kernel=n_cols / (ms * 1e-3), columns_per_s_call=n_cols / wall,
                gflops_kernel=flops * n_cols / (ms * 1e-3) / 1e9, flops_per_column=flops,
                scratch_bytes_per_column=scratch, scratch_write_gb_per_s=scratch * n_cols / (ms * 1e-3) / 1e9)
    try:
        from oracle import historian_oracle as ho
        from oracle import sumprod_oracle as so
        with open(os.path.join(ROOT, "tests", "golden", "models", "prot4.json")) as f:
            omodel = ho.RateModel(json.load(f))
        tree = so.Tree(parent, length, ["n%d" % k for k in range(n)])
        sp = so.SumProduct(omodel, tree)
        root = [np.zeros(a) for _ in range(c)]
        eig = [np.zeros((a, a), dtype=complex) for _ in range(c)]
        sample = 200
        t0 = time.perf_counter()
        for col in range(sample):
            sp.init_column({r: model.alphabet[tok[col, r]] if tok[col, r] >= 0 else "*" for r in range(n)})
            sp.fill_up()
            sp.fill_down()
            sp.accumulate_eigen_counts(root, eig)
        line["oracle_columns_per_s"] = sample / (time.perf_counter() - t0)
        line["oracle_sample"] = "%d columns, numpy, 1 core" % sample
    except ImportError:
        pass
    print(json.dumps(line))
    capi.shutdown()


if __name__ == "__main__":
    main()

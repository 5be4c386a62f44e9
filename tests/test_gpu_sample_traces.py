"""Sampled tracebacks on the device (hx_batch_sample_traces): the walks ForwardMatrix::sampleTrace
(reference src/forward.cpp:257-276) makes from a given state of the shared mt19937, found in the device-resident matrix from the
generator's canonical uniforms.  The checker is the oracle's sample_trace over the same matrix (exact mode: the matrix is the
oracle's bit for bit) with the same generator: same cells, walk after walk, and the same number of draws."""
import numpy as np
import pytest

from historian_amd import capi
from oracle import c_oracle
from oracle import historian_oracle as ho
from tests import helpers as H
from tests.test_gpu_traceback import ArrayForward

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _init():
    capi.init(0)


def uniforms(seed, n):
    g = ho.MT19937(seed)
    return [g.canonical() for _ in range(n)]


def oracle_walks(f, cells, lp_end, seed, n_walks):
    fm = ArrayForward(f, cells, lp_end)
    g = ho.MT19937(seed)
    walks, draws, used = [], [], 0
    for _ in range(n_walks):
        p = fm.sample_trace(g)
        used += len(p) - 1                      # one draw per step
        walks.append([tuple(c) for c in p])
        draws.append(used)
    return walks, draws


def check(f, seed=5489, n_walks=5, flags=capi.HX_LSE_EXACT):
    img = H.job_images(f)
    b = capi.Batch([img], flags)
    b.forward()
    want = c_oracle.forward(*img)
    cells = b.read_matrix(0)
    lp_end = float(b.lp_end()[0])
    if flags == capi.HX_LSE_EXACT:
        assert lp_end == want["lp_end"]
    walks, draws = oracle_walks(f, cells, lp_end, seed, n_walks)
    lay = b.layout(0)
    got, got_draws = b.sample_traces(0, n_walks, uniforms(seed, n_walks * (lay.n_rows + lay.n_cols + 4)))
    assert got_draws == draws
    assert got == walks
    b.close()
    return sum(len(w) for w in walks)


def test_leaf_pairs_and_seeds():
    aa = "arndcqeghilkmfpstwyv"
    for k, seed in enumerate((5489, 7, 20161005)):
        assert check(H.leaf_case(301 + k, 60 + 7 * k, 55 + 3 * k, tl=.1 + .1 * k, tr=.1), seed=seed) > 0
    check(H.leaf_case(303, 120, 70, alphabet=aa, jc=False, tl=.2, tr=.2), n_walks=4)
    check(H.leaf_case(2, 1, 1))
    check(H.leaf_case(3, 0, 5))
    check(H.leaf_case(5, 0, 0))


def test_general_profiles_with_null_states_and_high_in_degree():
    for f in (H.dag_case(31), H.dag_case(43, band=3), H.dag_case(67, n=9, band=2, keep_all=True), H.dag_case(81, n=40, samples=25)):
        check(f, n_walks=4)


def test_banded_pair_and_the_default_policy():
    f = H.leaf_case(304, 200, 210, band=8)
    check(f, n_walks=6)
    check(H.leaf_case(304, 200, 210, band=8), n_walks=6, flags=capi.HX_LSE_TRUNC)      # the walks are the oracle's through THIS matrix


def test_out_of_uniforms_is_reported():
    b = capi.Batch([H.job_images(H.leaf_case(9, 20, 20))], capi.HX_LSE_EXACT)
    b.forward()
    with pytest.raises(capi.HxError):
        b.sample_traces(0, 3, uniforms(1, 10))
    b.close()


# ---- through the C++ mirror (HX_DEVICE_SAMPLING=1): the reference's own fixtures, and no matrix crosses PCIe ----
import os
import re
import subprocess

from tests import recon_helpers as R
from tests import test_host_mirror as HM
from tests import test_oracle_testhist as TH

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_data", "")
HXRECON = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "historian_amd", "bin", "hxrecon")


@pytest.mark.parametrize("name", ["testforward.len2-4.n10.all.out", "testforward.len2-4.n10.hubs.out"])
def test_the_references_sampled_profile_fixtures_with_the_walks_made_on_the_device(name):
    # ten sampled traces from the default-seeded mt19937 (reference t/testforward.cpp): the profile they give is the file's
    assert HM.run(HM.GPU_CASES[name], {"HX_DEVICE_SAMPLING": "1"}) == open(G + name).read()


@pytest.mark.parametrize("name", sorted(n for n in TH.CASES if "profile_samples" in TH.CASES[n]["kw"]))
def test_testhist_in_sampling_mode_without_copying_a_matrix(tmp_path, name):
    # 43 sequences, 42 pair DPs, 100 sampled traces per node on the shared generator (reference Makefile:307-308): every
    # walk on the device, the generator advanced by what the walks used - the reference's file byte for byte, and the
    # timing line says that no matrix was read back
    case = TH.CASES[name]
    tree, seqs, guide = TH.load_case(case)
    kw = case["kw"]
    job = str(tmp_path / "job.txt")
    R.write_job(job, G + case["model"], tree, seqs, guide, str(tmp_path / "seqs.fa"), str(tmp_path / "guide.fa"),
                band=kw["max_distance_from_guide"], maxstates=0, seed=5489, samples=kw["profile_samples"])
    env = dict(os.environ, HX_FILL_MODE="exact", HX_DEVICE_SAMPLING="1", HX_TIMING="1")
    out = subprocess.run([HXRECON, job], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
    assert out.returncode == 0, out.stderr.decode()
    got = R.parse_hxrecon(out.stdout.decode())
    assert R.fasta_rows(tree, got["rows"]) == open(G + name.split()[0]).read()
    m = re.search(r"matrix D2H [0-9.]+ s in (\d+) reads", out.stderr.decode())
    assert m and int(m.group(1)) == 0, out.stderr.decode()[-2000:]

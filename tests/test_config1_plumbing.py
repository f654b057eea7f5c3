"""BASELINE config 1 -- "PBMC3k 512-cell trimmed SNN (k=5, dim=15) BQM 2-way partition on dwave-neal, CPU only
(plumbing, no GPU)" (SURVEY.md 8d): the n = 512 surrogate graph, the clustering_bqm model, the oracle's neal
restatement with num_reads = 64, num_sweeps = 1000 and an explicit beta range, the SampleSet surface the
reference reads.  CPU only; the GPU counterpart is tests/test_gpu_sampler.py::test_config1_on_the_gpu."""
import numpy as np
import pytest

from oracle import model_oracle as mo
from oracle import sa_oracle as so
from scrna_seq_qannealing_clustering_amd import graphs, models
from scrna_seq_qannealing_clustering_amd.sampleset import SampleSet


@pytest.fixture(scope="module")
def config1():
    nodes, eu, ev, w, truth = graphs.synthetic_snn(512, 5, 15, 15, 9, seed=0)
    G = graphs.EdgeListGraph(nodes, eu, ev, w)
    m = models.build_bqm_qubo(G, 0.05, k=8)
    return nodes, eu, ev, w, m


def test_graph_shape_matches_the_r_pipeline(config1):
    nodes, eu, ev, w, m = config1
    deg = np.bincount(np.concatenate([eu, ev]), minlength=512)
    assert deg.max() <= 15                                            # trim ord = 15
    assert set(np.round(w * 63).astype(int)) <= {7, 16, 27, 42, 63}  # s/(2k-s), k = 5: 1/9, 1/4, 3/7, 2/3, 1
    assert m.num_variables == 512 and m._dense is None


def test_p1_evaluation_parity_with_the_literal_model(config1):
    """P1: energies of arbitrary states from the array model == the literal dict restatement of
    BQM_clustering.py:29-47 (fp64, 1e-9 relative)."""
    nodes, eu, ev, w, m = config1
    edges = [(nodes[a], nodes[b], float(c)) for a, b, c in zip(eu, ev, w)]
    Q, gamma = mo.q_bqm(nodes, edges, 0.05, k=8)
    X = np.random.RandomState(1).randint(0, 2, size=(4, 512))
    for x, e in zip(X, m.energies(X)):
        assert e == pytest.approx(mo.qubo_energy(Q, dict(zip(nodes, x.tolist()))), rel=1e-9)
    assert m.info["gamma"] == pytest.approx(gamma, rel=1e-12)


def test_p3_neal_restatement_and_philox_chain_reach_the_same_optimum(config1):
    """P3 on the CPU: 64 reads x 1000 sweeps, explicit beta range.  Both chains of the oracle find a balanced
    cut with the same best energy (the neal restatement in fp64 Ising form, the Philox chain in fp32 QUBO form)."""
    nodes, eu, ev, w, m = config1
    betas = models.make_beta_schedule(1000, models.default_beta_range(m))
    Qs = m.dense_Qs()
    h, J, off = so.qubo_to_ising_dense(Qs)
    spins, en_ising, _ = so.sa_ising_neal_dense(h, J, 64, betas, seed=1234, threads=8)
    x_neal = ((spins + 1) // 2).astype(np.uint8)
    e_neal = m.energies(x_neal)
    assert np.allclose(e_neal, en_ising + off, rtol=1e-9, atol=1e-6)
    st, en, _ = so.sa_dense_philox(Qs.astype(np.float32), 64, betas, 1234)
    e_phil = m.energies(st)
    assert e_phil.min() == pytest.approx(e_neal.min(), rel=1e-3)
    best = st[int(np.argmin(e_phil))]
    assert 200 <= int(best.sum()) <= 312                               # a real 2-way split, not the trivial one
    # the SampleSet surface the reference reads (BQM_clustering.py:93-109,133-146)
    ss = SampleSet(st.astype(np.int8), e_phil, nodes, "BINARY")
    rec = ss.record
    assert np.all(np.diff(rec.energy) >= 0) and ss.first.energy == rec.energy[0]
    assert int(rec.num_occurrences.sum()) == 64
    lut = ss.first.sample
    S0 = [n for n in nodes if not lut[n]]
    S1 = [n for n in nodes if lut[n]]
    assert len(S0) + len(S1) == 512 and min(len(S0), len(S1)) > 5


def test_fp32_chain_decisions_differ_from_fp64_on_about_one_proposal_in_a_million():
    """The device chains compute in fp32 where neal computes in doubles (SURVEY.md 8d allows fp32 Q).  On bench.py's
    model the two predicates -- same state, same random word -- disagree on fewer than 5 proposals per million over a
    whole 200-sweep schedule (the fp32 field carries ~1e-7 relative rounding, the threshold polynomial 1e-7): energies
    are re-evaluated in fp64 anyway, so what fp32 costs is a slightly different, equally valid, random trajectory."""
    import ctypes as C
    import bench
    from oracle import sa_oracle as so
    m, Qs, betas, _, _ = bench.build_workload()
    betas = np.ascontiguousarray(betas[::5])
    rowptr = np.ascontiguousarray(m.rowptr, dtype=np.int32)
    col = np.ascontiguousarray(m.col, dtype=np.int32)
    val = np.ascontiguousarray(m.val, dtype=np.float32)
    lin = np.ascontiguousarray(m.lin, dtype=np.float32)
    counts = np.zeros(3, dtype=np.uint64)
    i32p, f32p = C.POINTER(C.c_int), C.POINTER(C.c_float)
    rc = so.lib().orc_csr_rank1_fp32_vs_fp64_decisions(
        rowptr.ctypes.data_as(i32p), col.ctypes.data_as(i32p), val.ctypes.data_as(f32p), lin.ctypes.data_as(f32p),
        C.c_float(float(np.float32(m.c_pair))), C.c_int(m.num_variables), C.c_int(16), C.c_uint32(0), C.c_int(len(betas)),
        betas.ctypes.data_as(C.POINTER(C.c_double)), C.c_uint64(1234), counts.ctypes.data_as(C.POINTER(C.c_uint64)))
    assert rc == 0 and int(counts[0]) == 16 * len(betas) * m.num_variables
    assert 0.2 < int(counts[2]) / int(counts[0]) < 0.5
    assert int(counts[1]) / int(counts[0]) < 5e-6

"""SampleSet look-alike: the members the reference reads.  CPU only."""
import numpy as np
import pytest

from scrna_seq_qannealing_clustering_amd.sampleset import SampleSet


def make():
    samples = np.array([[1, 0, 1], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 0, 1]])
    energies = np.array([-1.0, -3.0, -1.0, 2.0, -3.0])
    return SampleSet(samples, energies, ["a", "b", "c"], "BINARY", info={"k": 1})


def test_sorted_aggregated_histogram_mode():
    ss = make()
    assert len(ss) == 3
    assert ss.record.energy.tolist() == [-3.0, -1.0, 2.0]          # ascending, as BQM_clustering.py:133 assumes
    assert ss.record.num_occurrences.tolist() == [2, 2, 1]
    assert ss.record.sample[0].tolist() == [0, 0, 1]
    assert ss.record["energy"][0] == -3.0


def test_first_and_data_and_samples():
    ss = make()
    first = ss.first
    assert first.energy == -3.0 and first.num_occurrences == 2
    assert first.sample["c"] == 1 and not first.sample["a"]
    assert list(first.sample.values()) == [0, 0, 1]                 # variable order (plot_and_save.py:38)
    assert dict(first.sample) == {"a": 0, "b": 0, "c": 1}
    rows = list(ss.data(fields=["sample", "energy", "num_occurrences"]))
    assert [r.energy for r in rows] == [-3.0, -1.0, 2.0]
    s0, e0, o0 = rows[0]
    assert [k for k, v in s0.items() if v == 1] == ["c"] and o0 == 2
    assert len(ss.samples()[:2]) == 2                               # plot_and_save.py:106
    assert ss.samples()[1]["a"] == 1
    assert list(ss.data(fields=["energy"], reverse=True))[0].energy == 2.0
    assert ss.lowest().record.energy.tolist() == [-3.0]
    assert ss.info == {"k": 1} and ss.variables == ["a", "b", "c"] and ss.vartype == "BINARY"


def test_vartype_change_and_errors():
    ss = make()
    sp = ss.change_vartype("SPIN")
    assert sp.record.sample[0].tolist() == [-1, -1, 1]
    assert sp.change_vartype("BINARY").record.sample[0].tolist() == [0, 0, 1]
    with pytest.raises(ValueError):
        SampleSet(np.zeros((2, 3)), np.zeros(3), ["a", "b", "c"])
    with pytest.raises(ValueError):
        SampleSet(np.zeros((2, 3)), np.zeros(2), ["a", "b"])
    empty = SampleSet(np.zeros((0, 2)), np.zeros(0), ["a", "b"])
    with pytest.raises(ValueError):
        empty.first


def test_discrete_labels_kept():
    ss = SampleSet(np.array([[0, 2, 300]]), np.array([1.5]), [0, 1, 2], "DISCRETE")
    assert ss.first.sample[2] == 300


def test_row_aggregation_key_equals_numpy_unique():
    """The compact-key aggregation returns exactly np.unique(axis=0)'s rows, order and indices."""
    from scrna_seq_qannealing_clustering_amd.sampleset import _unique_rows
    rs = np.random.RandomState(0)
    cases = {
        "binary": rs.randint(0, 2, (300, 37)).astype(np.int8),
        "binary_with_duplicates": rs.randint(0, 2, (500, 5)).astype(np.int8),
        "spin": (2 * rs.randint(0, 2, (500, 6)) - 1).astype(np.int8),
        "labels": rs.randint(0, 5, (500, 4)).astype(np.int32),
        "large_labels": rs.randint(0, 1000, (500, 2)).astype(np.int32),
        "mixed_signs": rs.randint(-1, 2, (200, 3)).astype(np.int8),
        "no_columns": np.zeros((4, 0), dtype=np.int8),
        "all_minus_one": -np.ones((5, 3), dtype=np.int8),
    }
    for name, X in cases.items():
        want = np.unique(X, axis=0, return_index=True, return_inverse=True)
        got = _unique_rows(X)
        assert np.array_equal(want[0], got[0]), name
        assert np.array_equal(np.ravel(want[1]), np.ravel(got[1])), name
        assert np.array_equal(np.ravel(want[2]), np.ravel(got[2])), name

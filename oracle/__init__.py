"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this package.  Nothing under ``scrna_seq_qannealing_clustering_amd/`` imports it.

PARITY STATUS: *model side* (Q / DQM construction, energies, cut counts, exact optima) is pinned by
the known-answer values of SURVEY.md section 8c on the reference's bundled ``R/benchmarks/*.gexf``
graphs (fixtures under ``tests/golden/``).  *Sampler side* (the anneal) is "parity unpinned" against
real dwave-neal: neal is an un-vendored third-party dependency that cannot be installed offline and
the reference holds no sampler output at all; the chain restated here follows neal's published
algorithm (sa_oracle.c header).
"""

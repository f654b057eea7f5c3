"""MI355X-native simulated-annealing sampler for the graph-partition clustering models of
michal7kw/scRNA_seq_QAnnealing_Clustering (drop-in for its D-Wave / neal sampler calls).

Public surface:
    MI355XSampler            dimod-style sampler: sample_qubo / sample / sample_ising / sample_dqm
    SampleSet                dimod.SampleSet look-alike
    build_bqm_qubo, build_bqm2_qubo, build_bqm3_cut_qubo, build_dqm_potts   model builders
    BinaryQuadraticModel, DiscreteQuadraticModel     stand-ins for the dimod classes
    clustering_bqm, clustering_bqm_2, clustering_bqm_3, clustering_dqm  reference-shaped drivers
"""
from .bqm import BinaryQuadraticModel, DiscreteQuadraticModel
from .models import (PottsModel, QuboModel, add_size_window_penalty, build_bqm2_qubo,
                     build_bqm3_cut_qubo, build_bqm_qubo, build_dqm_potts, default_beta_range,
                     make_beta_schedule, qubo_dict_to_model)
from .sampleset import SampleSet

__all__ = [
    "MI355XSampler", "SampleSet", "QuboModel", "PottsModel", "BinaryQuadraticModel",
    "DiscreteQuadraticModel", "build_bqm_qubo", "build_bqm2_qubo", "build_bqm3_cut_qubo",
    "build_dqm_potts", "add_size_window_penalty", "default_beta_range", "make_beta_schedule",
    "qubo_dict_to_model",
]


def __getattr__(name):
    # the sampler (and everything that needs the HIP library) is imported lazily so that the pure
    # model layer can be used for inspection without a GPU
    if name == "MI355XSampler":
        from .sampler import MI355XSampler
        return MI355XSampler
    if name in ("clustering_bqm", "clustering_bqm_2", "clustering_bqm_3", "clustering_dqm"):
        from . import clustering
        return getattr(clustering, name)
    raise AttributeError(name)

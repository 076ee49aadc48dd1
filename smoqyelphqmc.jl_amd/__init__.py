"""MI355X-native CG / stochastic-trace hot path of SmoQyElPhQMC.jl.

Import as ``smoqyelphqmc_amd`` (alias module at the repository root).
"""
from . import lattice  # noqa: F401

"""MI355X-native CG / stochastic-trace hot path of SmoQyElPhQMC.jl.

Import as ``smoqyelphqmc_amd`` (alias module at the repository root).  The compute path is the
hand-written HIP library ``csrc/libsmoqy_hip.so`` behind the C ABI of ``include/smoqy_hip.h``;
nothing in this package computes on the CPU.
"""
from . import lattice, sharding  # noqa: F401
from .api import *  # noqa: F401,F403
from .api import I  # noqa: F401

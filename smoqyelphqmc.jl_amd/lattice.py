"""Host-side lattice geometry, checkerboard decomposition and synthetic model inputs.

The reference gets these objects from SmoQyDQMC / LatticeUtilities / Checkerboard.jl (none of
which are part of the hot path or present in /root/reference); here they only have to produce
the five plain arrays the hot path consumes: a colour-sorted neighbour table,
colour ranges, and the ``V`` / ``t`` / phonon-field arrays of a ``FermionPathIntegral``.

Geometry follows the reference driver scripts:
  * honeycomb:  tutorials/holstein_honeycomb.jl:147-185  (2 orbitals, bonds (1->2) with
    displacements [0,0], [-1,0], [0,-1], periodic L x L)
  * square:     examples/ossh_square.jl:113-161          (bonds +x, +y)
  * chain:      examples/bssh_chain.jl:112-140            (bond +1)
Site numbering is orbital-fastest, then x, then y (LatticeUtilities convention).  All ids
returned by this module are 1-based, like the Julia arrays they stand in for.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


def _site(orb, x, y, norb, L):
    return orb + norb * ((x % L) + L * (y % L))


def honeycomb_neighbor_table(L: int) -> np.ndarray:
    """(2, 3 L^2) int64, 1-based, bond-type major."""
    cols = []
    for dx, dy in ((0, 0), (-1, 0), (0, -1)):
        for y in range(L):
            for x in range(L):
                cols.append((_site(0, x, y, 2, L) + 1, _site(1, x + dx, y + dy, 2, L) + 1))
    return np.asfortranarray(np.array(cols, dtype=np.int64).T)


def square_neighbor_table(L: int) -> np.ndarray:
    cols = []
    for dx, dy in ((1, 0), (0, 1)):
        for y in range(L):
            for x in range(L):
                cols.append((_site(0, x, y, 1, L) + 1, _site(0, x + dx, y + dy, 1, L) + 1))
    return np.asfortranarray(np.array(cols, dtype=np.int64).T)


def chain_neighbor_table(L: int) -> np.ndarray:
    cols = [(x + 1, (x + 1) % L + 1) for x in range(L)]
    return np.asfortranarray(np.array(cols, dtype=np.int64).T)


def checkerboard_decomposition(neighbor_table: np.ndarray):
    """Greedy edge colouring standing in for Checkerboard.jl's ``checkerboard_decomposition!``
    (called at src/FermionDetMatrix.jl:96): every bond gets the smallest colour not yet used at
    either end, bonds are then stably sorted by colour.

    Returns ``(sorted_table, perm, colors)`` with ``perm`` the 1-based permutation
    (``sorted_table[:, h] == neighbor_table[:, perm[h]-1]``) and ``colors`` a ``(2, ncolors)``
    int64 array of 1-based inclusive ``[first, last]`` bond ranges.  The exact colouring of
    Checkerboard.jl is not reproducible here (source absent); the C ABI therefore takes the
    sorted table and colour ranges as *inputs* (SURVEY.md §7 hard part 2).
    """
    nt = np.asarray(neighbor_table, dtype=np.int64)
    Nh = nt.shape[1]
    if Nh == 0:
        return np.zeros((2, 0), dtype=np.int64, order="F"), np.zeros(0, dtype=np.int64), np.zeros((2, 0), dtype=np.int64, order="F")
    used: dict[int, set] = {}
    color = np.zeros(Nh, dtype=np.int64)
    for h in range(Nh):
        i, j = int(nt[0, h]), int(nt[1, h])
        if i == j:
            raise ValueError("self-loop bond")
        ui, uj = used.setdefault(i, set()), used.setdefault(j, set())
        c = 0
        while c in ui or c in uj:
            c += 1
        color[h] = c
        ui.add(c)
        uj.add(c)
    perm0 = np.argsort(color, kind="stable")
    sorted_nt = np.asfortranarray(nt[:, perm0])
    ncol = int(color.max()) + 1
    colors = np.zeros((2, ncol), dtype=np.int64, order="F")
    sc = color[perm0]
    for c in range(ncol):
        idx = np.nonzero(sc == c)[0]
        colors[0, c] = idx[0] + 1
        colors[1, c] = idx[-1] + 1
    return sorted_nt, (perm0 + 1).astype(np.int64), colors


@dataclass
class FermionPathIntegral:
    """The fields of SmoQyDQMC's ``FermionPathIntegral`` that the hot path reads
    (src/FermionDetMatrix.jl:72, 214): ``neighbor_table`` (2 x Nh, 1-based, model order),
    ``t`` (Nh x Ltau), ``V`` (N x Ltau), ``N``, ``beta``, ``dtau``, ``Ltau``."""

    neighbor_table: np.ndarray
    t: np.ndarray
    V: np.ndarray
    N: int
    beta: float
    dtau: float
    Ltau: int


@dataclass
class HolsteinParameters:
    """Subset of SmoQyDQMC's ``HolsteinParameters`` read by ``update_Λ!``
    (src/holstein_shift_matrix.jl:7-8)."""

    alpha: np.ndarray
    alpha3: np.ndarray
    coupling_to_phonon: np.ndarray  # 1-based
    coupling_to_site: np.ndarray  # 1-based
    ph_sym_form: np.ndarray  # per coupling (already expanded over unit cells)


@dataclass
class ElectronPhononParameters:
    """Subset of SmoQyDQMC's ``ElectronPhononParameters`` read by the hot path: ``x`` is the
    ``Nph x Ltau`` phonon field, ``dtau``, ``Ltau`` and the Holstein couplings."""

    x: np.ndarray
    dtau: float
    Ltau: int
    holstein: HolsteinParameters | None = None


@dataclass
class ForceCouplings:
    """Flattened couplings the force terms need (src/fermion_det_matrix_dervative.jl:189-289,
    src/holstein_shift_matrix.jl:156-201): what a shim extracts from SmoQyDQMC's
    ``holstein_parameters_up`` / ``ssh_parameters_up`` / ``phonon_parameters.M``.  All ids 1-based.
    ``s_bond[c]`` is the position of coupling c's hopping in the colour-sorted neighbour table."""

    x: np.ndarray            # (Nph, Ltau)
    dtau: float
    finite_mass: np.ndarray  # (Nph,) isfinite(M)
    h_alpha: np.ndarray
    h_alpha2: np.ndarray
    h_alpha3: np.ndarray
    h_alpha4: np.ndarray
    h_c2p: np.ndarray
    h_c2s: np.ndarray
    h_phsym: np.ndarray
    s_alpha: np.ndarray
    s_alpha2: np.ndarray
    s_alpha3: np.ndarray
    s_alpha4: np.ndarray
    s_c2p: np.ndarray        # (2, Nssh)
    s_bond: np.ndarray       # (Nssh,)
    # T = ComplexF64 (ssh_parameters.α::Vector{T}): imaginary parts of the SSH couplings, None for real couplings
    s_alpha_im: np.ndarray | None = None
    s_alpha2_im: np.ndarray | None = None
    s_alpha3_im: np.ndarray | None = None
    s_alpha4_im: np.ndarray | None = None


@dataclass
class SyntheticModel:
    name: str
    fpi: FermionPathIntegral
    elph: ElectronPhononParameters
    alpha: float
    mu: float
    kind: str  # "holstein" | "ossh" | "bssh"
    meta: dict = field(default_factory=dict)

    def force_couplings(self, perm) -> ForceCouplings:
        """Couplings of this model in the flattened form the force entry points take.  ``perm`` is the
        checkerboard permutation (sorted bond n is model hopping perm[n])."""
        x = self.elph.x
        z = np.zeros(0)
        zi = np.zeros(0, dtype=np.int64)
        inv = np.empty(len(perm), dtype=np.int64)
        inv[np.asarray(perm) - 1] = np.arange(1, len(perm) + 1)  # model hopping h -> sorted position n
        if self.kind == "holstein":
            hol = self.elph.holstein
            n = len(hol.alpha)
            return ForceCouplings(x, self.elph.dtau, np.ones(x.shape[0], dtype=np.int32), hol.alpha, np.zeros(n), hol.alpha3, np.zeros(n), hol.coupling_to_phonon, hol.coupling_to_site,
                                  np.asarray(hol.ph_sym_form, dtype=np.int32), z, z, z, z, np.zeros((2, 0), dtype=np.int64), zi)
        Nh = self.fpi.t.shape[0]
        a = np.full(Nh, self.alpha)
        zz = np.zeros(Nh)
        if self.kind == "bssh":
            # one finite-mass phonon per bond plus an infinite-mass partner pinned at 0 (how SmoQyDQMC
            # expresses a bond mode as a difference of two modes): rows 0..Nh-1 = x, last row = 0
            xe = np.asfortranarray(np.vstack([x, np.zeros((1, x.shape[1]))]))
            fm = np.ones(Nh + 1, dtype=np.int32)
            fm[Nh] = 0
            c2p = np.vstack([np.full(Nh, Nh + 1), np.arange(1, Nh + 1)]).astype(np.int64)
            return ForceCouplings(xe, self.elph.dtau, fm, z, z, z, z, zi, zi, np.zeros(0, dtype=np.int32), a, zz, zz, zz, c2p, inv[np.arange(Nh)])
        # optical SSH: x / y polarised mode on every site; +x bonds couple the x modes, +y bonds the y modes
        nt, N = self.fpi.neighbor_table, self.fpi.N
        off = np.where(np.arange(Nh) < Nh // 2, 0, N)
        c2p = np.vstack([nt[0] + off, nt[1] + off]).astype(np.int64)
        return ForceCouplings(x, self.elph.dtau, np.ones(x.shape[0], dtype=np.int32), z, z, z, z, zi, zi, np.zeros(0, dtype=np.int32), a, zz, zz, zz, c2p, inv[np.arange(Nh)])

    def bare_model(self):
        """(V0, t0): the path integral with the phonon contribution removed, i.e. what
        ``SmoQyDQMC.update!(fermion_path_integral, elph, x, -1)`` leaves (src/EFAPFFHMCUpdater.jl:148)."""
        return np.full(self.fpi.N, -self.mu), np.ones(self.fpi.t.shape[0])

    def refresh_from_x(self):
        """Recompute ``V`` / ``t`` of the path integral from the current phonon field ``x``
        (what SmoQyDQMC's ``update!(fermion_path_integral, …, x, ±1)`` does at
        src/EFAPFFHMCUpdater.jl:200-205)."""
        x = self.elph.x
        if self.kind == "holstein":
            self.fpi.V[...] = self.alpha * x - self.mu
        elif self.kind == "bssh":
            self.fpi.t[...] = 1.0 - self.alpha * x
        elif self.kind == "ossh":
            nt = self.fpi.neighbor_table
            N = self.fpi.N
            i = nt[0] - 1
            j = nt[1] - 1
            Nh = nt.shape[1]
            # bonds [0, Nh/2) are +x bonds coupled to the x-polarised modes (phonons 0..N-1),
            # bonds [Nh/2, Nh) are +y bonds coupled to the y-polarised modes (phonons N..2N-1)
            off = np.where(np.arange(Nh) < Nh // 2, 0, N)
            self.fpi.t[...] = 1.0 - self.alpha * (x[j + off] - x[i + off])


def _rng(seed: int):
    return np.random.Generator(np.random.PCG64(seed))


SEED0 = 20251004


def holstein_honeycomb(L: int, Ltau: int, dtau: float = 0.05, alpha: float = 1.0, mu: float = 0.0, walker: int = 0, smooth: bool = False) -> SyntheticModel:
    """Synthetic Holstein model on the honeycomb lattice (BASELINE.json configs 1, 2, 4).
    ``V[i,l] = alpha x[i,l] - mu``, ``t = 1``; ``x ~ N(0,1)`` iid from PCG64(SEED0 + walker)
    (``smooth=True`` low-pass filters x along tau, closer to an equilibrated HMC field)."""
    nt = honeycomb_neighbor_table(L)
    N, Nh = 2 * L * L, nt.shape[1]
    g = _rng(SEED0 + walker)
    x = g.standard_normal((N, Ltau))
    if smooth:
        x = _smooth_tau(x)
    x = np.asfortranarray(x)
    fpi = FermionPathIntegral(nt, np.asfortranarray(np.ones((Nh, Ltau))), np.asfortranarray(alpha * x - mu), N, dtau * Ltau, dtau, Ltau)
    hol = HolsteinParameters(np.full(N, alpha), np.zeros(N), np.arange(1, N + 1), np.arange(1, N + 1), np.ones(N, dtype=bool))
    return SyntheticModel(f"holstein_honeycomb_L{L}_Ltau{Ltau}", fpi, ElectronPhononParameters(x, dtau, Ltau, hol), alpha, mu, "holstein", {"L": L})


def ossh_square(L: int, Ltau: int, dtau: float = 0.05, alpha: float = 0.2, mu: float = 0.0, walker: int = 0, smooth: bool = False) -> SyntheticModel:
    """Synthetic optical-SSH model on the square lattice (BASELINE.json config 3): two phonon
    modes per site, ``t[h,l] = 1 - alpha (x_j - x_i)`` along the bond direction, ``V = -mu``."""
    nt = square_neighbor_table(L)
    N, Nh = L * L, nt.shape[1]
    g = _rng(SEED0 + walker)
    x = g.standard_normal((2 * N, Ltau))
    if smooth:
        x = _smooth_tau(x)
    x = np.asfortranarray(x)
    fpi = FermionPathIntegral(nt, np.asfortranarray(np.ones((Nh, Ltau))), np.asfortranarray(np.full((N, Ltau), -mu)), N, dtau * Ltau, dtau, Ltau)
    m = SyntheticModel(f"ossh_square_L{L}_Ltau{Ltau}", fpi, ElectronPhononParameters(x, dtau, Ltau, None), alpha, mu, "ossh", {"L": L})
    m.refresh_from_x()
    return m


def bssh_chain(L: int, Ltau: int, dtau: float = 0.05, alpha: float = 0.2, mu: float = 0.0, walker: int = 0, smooth: bool = False) -> SyntheticModel:
    """Synthetic bond-SSH chain (BASELINE.json config 5): one phonon per bond,
    ``t[h,l] = 1 - alpha x[h,l]``, ``V = -mu``."""
    nt = chain_neighbor_table(L)
    N, Nh = L, nt.shape[1]
    g = _rng(SEED0 + walker)
    x = g.standard_normal((Nh, Ltau))
    if smooth:
        x = _smooth_tau(x)
    x = np.asfortranarray(x)
    fpi = FermionPathIntegral(nt, np.asfortranarray(np.ones((Nh, Ltau))), np.asfortranarray(np.full((N, Ltau), -mu)), N, dtau * Ltau, dtau, Ltau)
    m = SyntheticModel(f"bssh_chain_L{L}_Ltau{Ltau}", fpi, ElectronPhononParameters(x, dtau, Ltau, None), alpha, mu, "bssh", {"L": L})
    m.refresh_from_x()
    return m


def _smooth_tau(x: np.ndarray, width: float = 4.0) -> np.ndarray:
    """Periodic Gaussian low-pass along tau, renormalised to unit variance."""
    Lt = x.shape[1]
    k = np.fft.fftfreq(Lt) * Lt
    filt = np.exp(-0.5 * (k / (Lt / (2 * np.pi * width))) ** 2)
    y = np.fft.ifft(np.fft.fft(x, axis=1) * filt[None, :], axis=1).real
    return y / y.std()


CONFIGS = {
    # BASELINE.json "configs", in order.  Couplings are the contract's (Ω = α = 1, the value the reference's own SSH tests
    # run at, test/test_example_ossh_square.jl:10, test/test_example_bssh_chain.jl:10).  With i.i.d. unit-variance phonon fields that makes
    # the SSH hoppings t = 1 - α Δx change sign from slice to slice: the τ-averaged preconditioner sees almost none of it and the solves
    # take 300-550 iterations ("KPM preconditioner stressed").  Rounds 1-2 ran the two SSH lattices at α = 0.2; those points stay available
    # under the *_alpha0p2 names below (and 0.2 stays the default of ossh_square() / bssh_chain(), which the small-lattice unit tests and the
    # committed fixtures call directly).
    "holstein_honeycomb_L4_Ltau40": lambda **kw: holstein_honeycomb(4, 40, **kw),
    "holstein_honeycomb_L8_Ltau80": lambda **kw: holstein_honeycomb(8, 80, **kw),
    "ossh_square_L12_Ltau100": lambda **kw: ossh_square(12, 100, alpha=1.0, **kw),
    "holstein_honeycomb_L16_Ltau128": lambda **kw: holstein_honeycomb(16, 128, **kw),
    "bssh_chain_L256_Ltau200": lambda **kw: bssh_chain(256, 200, alpha=1.0, **kw),
    # the weak-coupling points of rounds 1-2 (same lattices, α = 0.2: 11-15 iterations per solve)
    "ossh_square_L12_Ltau100_alpha0p2": lambda **kw: ossh_square(12, 100, alpha=0.2, **kw),
    "bssh_chain_L256_Ltau200_alpha0p2": lambda **kw: bssh_chain(256, 200, alpha=0.2, **kw),
}

"""HIP-graph capture of the per-iteration Fock build.

One SCF iteration of a 20-atom molecule is ~2 ms of GPU work issued as ~45 launches (the HIP kernels of
`libdqc_amd.so` plus the small (nao, nao) torch GEMMs of the orthogonaliser).  Issued eagerly, the launch cost is
comparable to the work; captured once into a hipGraph (`torch.cuda.CUDAGraph` is hipGraph on ROCm) the whole
`C_occ -> D = ao_orb2dm(C_occ, n) -> F = dm2scp(D)` chain replays with a single launch.  The C ABI entry points only
enqueue on the stream they are given (no synchronisation in the per-iteration calls), which is what makes them
capturable.

The graph owns static input/output buffers: `GraphedFock(engine)(orb)` copies `orb` in, replays, and returns the
static Fock-matrix tensor (valid until the next call).  Reference data flow: `scp2dm` -> `dm2scp`,
dqc/qccalc/hf.py:105-113, 182-201, dqc/qccalc/ks.py:176-187.
"""
import torch

from .utils.datastruct import SpinParam


class GraphedFock:
    def __init__(self, engine, warmup: int = 2, with_energy: bool = False):
        self.with_energy = with_energy
        if engine.polarized:
            raise NotImplementedError("GraphedFock covers the restricted engines; UHF/UKS run eagerly")
        self.engine = engine
        h = engine.hamilton
        n, norb = engine.shape[-1], engine.norb
        self.orb = torch.zeros((n, norb), dtype=engine.dtype, device=engine.device)
        self.orb[:norb, :norb] = torch.eye(norb, dtype=engine.dtype, device=engine.device)  # any orthonormal start
        self.weight = engine.orb_weight
        # warm-up on a side stream (allocator, lazy kernel attributes, the one-off occupation check), then capture
        s = torch.cuda.Stream(device=engine.device)
        s.wait_stream(torch.cuda.current_stream(engine.device))
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._body()
        torch.cuda.current_stream(engine.device).wait_stream(s)
        torch.cuda.synchronize(engine.device)
        getattr(engine.hamilton, "_tiles", None) if engine.hamilton.df is None else None  # (retires the tile-fill event before the capture)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.dm, self.fock, self.energy = self._body()
        h._jk_cache = None     # the memoised tensors belong to the graph's private pool
        h._jkpol_cache = None
        h._dm_factor = None
        h._energy_memo = None

    def _body(self):
        dm = self.engine.hamilton.ao_orb2dm(self.orb, self.weight)
        fock = self.engine.dm2scp(dm)
        # the total energy of this density rides along: its two-electron parts are by-products of the build just made
        # (the Hamiltonian's memo), the rest is tr(D h)
        return dm, fock, (self.engine.dm2energy(dm) if self.with_energy else None)

    def __call__(self, orb):
        """orb (nao, norb) occupied orbitals in the orthogonalised basis -> Fock matrix (static buffer)"""
        if orb is not self.orb:
            self.orb.copy_(orb)
        self.graph.replay()
        return self.fock

    def density_matrix(self):
        """the D of the last replay (static buffer)"""
        return self.dm


class GraphedSCFStep:
    """hipGraph of one whole SCF step  F_in -> D = n P(F_in) -> F_out = dm2scp(D)  with the occupied-space projector from
    GEMM-only purification (dqc_amd/purify.py) instead of an eigendecomposition: no rocSOLVER call, no host decision,
    one graph launch per iteration.  Needs uniform occupations per spin (closed shell, or high-spin unrestricted:
    F_in = stacked (F_u, F_d), D = SpinParam of the two projectors).

        step = GraphedSCFStep(engine);  fock, dm, err = step(f_in)     # static buffers, valid until the next call
    `err` (0-dim device tensor) is the idempotency + trace error of the projector(s); the caller checks it at the point
    where it synchronises anyway and falls back to the eigh path when purification did not converge."""

    def __init__(self, engine, warmup: int = 1, capture: bool = True):
        """capture=False: only the static members and `_body()` -- for a caller that captures the step inside a larger graph
        (dqc_amd/devscf.py: the whole SCF iteration)"""
        self.engine = engine
        self.pol = engine.polarized
        ws = [engine.orb_weight.u, engine.orb_weight.d] if self.pol else [engine.orb_weight]
        for w in ws:
            if w.numel() and not bool((w == w[0]).all()):
                raise NotImplementedError("purification needs uniform occupations")
        self.occ = [float(w[0]) if w.numel() else 0.0 for w in ws]
        self.nocc = [engine.norb.u, engine.norb.d] if self.pol else [engine.norb]
        self.weights = ws
        n = engine.shape[-1]
        gen = torch.Generator().manual_seed(20240229)
        self.omega = [torch.randn((n, r), dtype=engine.dtype, generator=gen).to(engine.device) for r in self.nocc]
        shape = (2, n, n) if self.pol else (n, n)
        self.f_in = torch.zeros(shape, dtype=engine.dtype, device=engine.device)
        idx = torch.arange(n, device=engine.device)
        self.f_in[..., idx, idx] = idx.to(engine.dtype)  # any matrix with a gap at n_occ
        self.graph = None
        if not capture:
            return
        s = torch.cuda.Stream(device=engine.device)
        s.wait_stream(torch.cuda.current_stream(engine.device))
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._body()
        torch.cuda.current_stream(engine.device).wait_stream(s)
        torch.cuda.synchronize(engine.device)
        getattr(engine.hamilton, "_tiles", None) if engine.hamilton.df is None else None  # (retires the tile-fill event before the capture)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.fock, self.dm, self.err = self._body()
        h = engine.hamilton
        h._jk_cache = None
        h._jkpol_cache = None
        h._dm_factor = None
        h._energy_memo = None

    def _dm_of_projector(self, p, s_):
        """occ * P as ao_orb2dm(Q, occ) with Q an orthonormal basis of range(P) (Cholesky QR of P . Omega, one small launch:
        dqc_orth_factor), so that the Hamiltonian knows the rank-n_occ factor of the matrix and the grid pass takes the
        factor-form density kernel; wide occupied spaces keep the anonymous matrix"""
        from . import lib
        r = self.nocc[s_]
        w = self.weights[s_]
        if not (0 < r <= 128 and lib.padded_norb(r) > 0 and self.engine.hamilton._lowrank_density):
            return p * self.occ[s_]
        y = p @ self.omega[s_]
        q = lib.orth_factor(y, y.transpose(-2, -1) @ y)
        dm = self.engine.hamilton.ao_orb2dm(q, w)
        # a failed factorisation (NaN / wrong range) must surface in the step's error so that the caller falls back to eigh
        self._factor_err = self._factor_err + (dm - p * self.occ[s_]).abs().max()
        return dm

    def _body(self):
        from .purify import projector_from_fock
        f = (self.f_in + self.f_in.transpose(-2, -1)) * 0.5
        self._factor_err = 0.0
        if not self.pol:
            p, err = projector_from_fock(f, self.nocc[0])
            dm = self._dm_of_projector(p, 0)
            return self.engine.dm2scp(dm), dm, err + self._factor_err
        dms, err = [], 0.0
        for s_ in range(2):
            if self.nocc[s_] == 0:  # no electron of this spin (H atom, ...)
                dms.append(torch.zeros_like(f[s_]))
                continue
            p, e = projector_from_fock(f[s_], self.nocc[s_])
            dms.append(self._dm_of_projector(p, s_))
            err = err + e
        dm = SpinParam(u=dms[0], d=dms[1])
        return self.engine.dm2scp(dm), dm, err + self._factor_err

    def __call__(self, f_in):
        if f_in is not self.f_in:
            self.f_in.copy_(f_in)
        if self.graph is None:  # (capture=False: the same step, launched eagerly -- direct SCF builds are not capturable)
            return self._body()
        self.graph.replay()
        return self.fock, self.dm, self.err

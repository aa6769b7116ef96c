"""HF / KS drivers with the reference's surface -- HF(mol).run().energy(), KS(mol, xc=...).run().energy(),
.aodm(), .dm2energy(dm) -- and the same engine data flow:

   _HFEngine / _KSEngine:  dm2scp (Fock build), scp2dm (diagonalise, occupy), dm2energy
        dqc/qccalc/hf.py:93-119, 166-247 ; dqc/qccalc/ks.py:110-130, 157-187
   SCF_QCCalc.run: dm0 = "1e" core guess, then the fixed point F = dm2scp(scp2dm(F))
        dqc/qccalc/scf_qccalc.py:84-116 (reference: xitorch Broyden-1, alpha=-0.5, maxiter=50)

Everything runs on the device; the self-consistent parameter is the Fock matrix in the orthogonalised basis,
as in the reference.  The fixed-point solver is Pulay DIIS on the commutator [F, D] (any convergent mixer
gives the same fixed point; only converged energies are compared).  Restricted closed-shell and
unrestricted (UHF/UKS, SpinParam densities, stacked Fock matrices) are implemented."""
import os
import warnings
from typing import Optional

import numpy as np
import torch

from .utils.datastruct import SpinParam
from .xc import get_xc


_DIIS_SCALE = os.environ.get("DQC_AMD_DIIS_SCALE", "1") != "0"


class _Engine:
    def __init__(self, system, xc=None, is_ks=False, restricted=None):
        self._system = system
        self.hamilton = system.get_hamiltonian()
        self.is_ks = is_ks
        self.xc = get_xc(xc) if is_ks else None
        # hf.py:49-53: polarised iff spin != 0 unless `restricted` says otherwise
        self.polarized = bool(system.spin != 0) if restricted is None else (not restricted)
        # spin != 0 with restricted=True: one set of orbitals with occupations [2, ..., 2, 1, ..., 1] (mol.py:421-443) -- the
        # reference's restricted open-shell treatment; non-uniform occupations take the eigh step (no purification)
        # hf.py:55-57 / ks.py:69-71: the grid is needed by KS always and by HF when the system carries an external
        # potential (vext is integrated on the grid inside build())
        if not system.requires_grid():
            self.hamilton.build()  # first: the ERI tile fill runs on a side stream underneath the grid setup that follows
        if is_ks or system.requires_grid():
            system.setup_grid()
            self.hamilton.setup_grid(system.get_grid(), self.xc)
        self.hamilton.build()
        self.orb_weight = system.get_orbweight(polarized=self.polarized)
        self.norb = SpinParam.apply_fcn(lambda w: int(w.shape[-1]), self.orb_weight)
        self.knvext = self.hamilton.get_kinnucl()
        self.shape = self.knvext.shape
        self.dtype, self.device = self.knvext.dtype, self.knvext.device
        self._enuc = torch.as_tensor(system.get_nuclei_energy()).to(device=self.device, dtype=self.dtype)  # on the device once
        # Mol(orthogonalize_basis=False): the Fock matrix lives in the raw AO basis and `diagonalize` is the generalised problem
        # F C = S C e (hf.py:227-247: lsymeig(A=fock, M=ovlp)).  Solved through S^-1/2: C = S^-1/2 U, U from eigh(S^-1/2 F S^-1/2)
        self.ovlp, self._sinvh = None, None
        if not getattr(self.hamilton, "orthogonalized", True):
            self.ovlp = self.hamilton.get_overlap().fullmatrix()
            ev, evec = torch.linalg.eigh(self.ovlp)
            self._sinvh = (evec * ev ** -0.5) @ evec.transpose(-2, -1)

    def _eigpairs(self, fock):
        """(eigenvalues ascending, eigenvectors) of the symmetrised Fock matrix; S-orthonormal vectors of F C = S C e when the
        basis is not orthogonal"""
        fock = (fock + fock.transpose(-2, -1)) * 0.5
        if self._sinvh is None:
            return torch.linalg.eigh(fock)
        e, u = torch.linalg.eigh(self._sinvh @ fock @ self._sinvh)
        return e, self._sinvh @ u

    def _eigvecs(self, fock):
        return self._eigpairs(fock)[1]

    def get_system(self):
        return self._system

    def _core_matrix(self):
        """the one-electron part of the Fock matrix as a contiguous tensor (handed to the Hamiltonian's fused build, which adds it in
        the launch that forms X^T (J - K / 2 + V) X)"""
        c = getattr(self, "_core_cache", None)
        if c is None:
            c = self._core_cache = self.knvext.fullmatrix().contiguous()
        return c

    # Fock build -- THE hot path (hf.py:182-201, ks.py:176-187)
    def dm2scp(self, dm):
        if self.polarized:  # scp = stacked (F_u, F_d)  (hf.py:93-103)
            if not isinstance(dm, SpinParam):
                dm = SpinParam(u=dm[0], d=dm[1])
            h = self.hamilton
            if not self.is_ks and h.df is None and hasattr(h, "get_elrep_exchange_pol") and dm.u.dim() == 2:
                # J[D_u + D_d], -K[2 D_u]/2, -K[2 D_d]/2 from ONE pass over the ERI tiles instead of three
                J, kx = h.get_elrep_exchange_pol(dm)
                core = self.knvext.fullmatrix() + J
                return torch.stack([core + kx.u, core + kx.d])
            if (self.is_ks and dm.u.dim() == 2 and h.df is None and hasattr(h, "get_elrep_plus_vxc_pol") and not getattr(h, "_direct", False)
                    and not getattr(h, "sharded", False) and getattr(h, "_tile_slice", None) is None):
                # J + Vxc_s with one batched AO -> orthogonal conversion, the Coulomb stream beside the grid pass (hamilton.py)
                return h.get_elrep_plus_vxc_pol(dm, core=self._core_matrix())
            core = self.knvext + h.get_elrep(dm.u + dm.d)
            v = h.get_vxc(dm) if self.is_ks else h.get_exchange(dm)
            return torch.stack([(core + v.u).fullmatrix(), (core + v.d).fullmatrix()])
        if self.is_ks and dm.dim() == 2 and hasattr(self.hamilton, "get_elrep_plus_vxc"):
            # J + Vxc with one AO -> orthogonal conversion (the operators' own sum, ks.py:176-187, converts each)
            return self.hamilton.get_elrep_plus_vxc(dm, core=self._core_matrix())
        if not self.is_ks and dm.dim() == 2 and self.hamilton.df is None and hasattr(self.hamilton, "get_elrep_plus_exchange"):
            # J - K / 2 with one AO -> orthogonal conversion (hf.py:198-199 converts each operator)
            return self.hamilton.get_elrep_plus_exchange(dm, core=self._core_matrix())
        elrep = self.hamilton.get_elrep(dm)
        if self.is_ks:
            fock = self.knvext + elrep + self.hamilton.get_vxc(dm)
        else:
            fock = self.knvext + elrep + self.hamilton.get_exchange(dm)
        return fock.fullmatrix()

    def scp2dm(self, scp):
        if self.polarized:
            out = []
            for f, w, n in ((scp[0], self.orb_weight.u, self.norb.u), (scp[1], self.orb_weight.d, self.norb.d)):
                out.append(self.hamilton.ao_orb2dm(self._eigvecs(f)[..., :n], w))
            return SpinParam(u=out[0], d=out[1])
        return self.hamilton.ao_orb2dm(self.scp2orb(scp), self.orb_weight)

    def scp2orb(self, scp):
        """occupied orbitals of a (restricted) Fock matrix: the `diagonalize` step of hf.py:227-247"""
        # generalised problem F C = S C e; S = identity in the orthogonalised basis
        return self._eigvecs(scp)[..., :self.norb]

    def scp2scp(self, scp):
        return self.dm2scp(self.scp2dm(scp))

    def dm2energy(self, dm):
        h = self.hamilton
        if self.polarized:  # hf.py:166-172 / ks.py:157-166 with dmtot = dm.u + dm.d
            tot = dm.u + dm.d
            e = h.get_e_hcore(tot) + h.get_e_elrep(tot) + (h.get_e_xc(dm) if self.is_ks else h.get_e_exchange(dm))
            return e + self._enuc
        e = h.get_e_hcore(dm) + h.get_e_elrep(dm)
        e = e + (h.get_e_xc(dm) if self.is_ks else h.get_e_exchange(dm))
        return e + self._enuc

    def energy_parts(self, dm):
        h = self.hamilton
        tot = SpinParam.sum(dm)  # hf.py:166-172: core and Coulomb terms see the total density
        p = {"e_core": float(h.get_e_hcore(tot)), "e_elrep": float(h.get_e_elrep(tot)),
             "e_nuc": float(self._system.get_nuclei_energy())}
        p["e_xc" if self.is_ks else "e_exch"] = float(h.get_e_xc(dm) if self.is_ks else h.get_e_exchange(dm))
        p["e_tot"] = sum(p.values())
        return p


class SCF_QCCalc:
    def __init__(self, engine):
        self._engine = engine
        self._has_run = False
        self.niter = 0
        self.converged = False   # max|[F, D]| < f_tol
        self.stalled = False     # stopped at the round-off floor of the Fock build, above f_tol (see run())
        self.scf_error = float("inf")  # max|[F, D]| of the returned iterate -- the achieved error, whatever the exit

    @property
    def accepted(self):
        """the run ended at a fixed point: either `converged` (f_tol met) or `stalled` at the round-off floor with
        `scf_error` < 100 f_tol; `scf_error` says what was achieved"""
        return self.converged or self.stalled

    def get_system(self):
        return self._engine.get_system()

    def run(self, dm0="1e", eigen_options=None, fwd_options=None, bck_options=None):
        """the SCF loop, driven synchronously: every host read of the generator below is a blocking device -> host copy.
        dqc_amd.batch.run_concurrent drives many of these generators at once, one stream per molecule."""
        # one molecule, core guess, purification step: the whole iteration replays as ONE hipGraph and the host only looks at two
        # doubles per iteration, one iteration late (dqc_amd/devscf.py); everything else takes the host-driven generator below
        from . import devscf
        opts = {"maxiter": 50, "f_tol": 1e-9, "history": 12}
        opts.update(fwd_options or {})
        if self._engine.device.type == "cuda" and devscf.eligible(self._engine, dm0, opts):
            loop = getattr(self, "_devloop", None)
            if loop is None or loop.H != int(opts["history"]):
                loop = self._devloop = devscf.DeviceLoop(self._engine, int(opts["history"]))
            if loop.run(self, opts):
                self.driver_used = "device"  # (which loop produced the result: diagnostics / tests)
                return self
        self.driver_used = "host"
        resume = getattr(self, "_resume_dm", None)  # the device loop's last good density (a projector failure mid-run)
        if resume is not None:
            dm0, self._resume_dm = resume, None
            self._resumed_after_failure = True
        gen = self._run_gen(dm0, fwd_options)
        # a Hamiltonian sharded over several GPUs (HamiltonMI355.shard_over) runs this loop on every rank: the scalars the
        # driver decides on are rank 0's, so that every rank takes the same branch and issues the same collectives
        sync = getattr(getattr(self._engine, "hamilton", None), "sync_scalars", lambda t: t)
        try:
            req = next(gen)
            while True:
                req = gen.send(sync(req).cpu().numpy())
        except StopIteration:
            pass
        return self

    def _run_gen(self, dm0="1e", fwd_options=None):
        """generator form of run(): yields the (small) device tensor it needs on the host -- ONE per SCF iteration -- and is
        resumed with that tensor's numpy copy; everything else is enqueued on the current stream without synchronising"""
        opts = {"maxiter": 50, "f_tol": 1e-9, "history": 12}
        opts.update(fwd_options or {})
        eng = self._engine
        if isinstance(dm0, str):
            if dm0 != "1e":
                raise RuntimeError("Unknown dm0: %s" % dm0)
            n = eng.shape[-1]
            z = torch.zeros((n, n), dtype=eng.dtype, device=eng.device)
            scp0 = eng.dm2scp(SpinParam(u=z, d=z) if eng.polarized else z)
            dm = eng.scp2dm(scp0)
        elif dm0 is None:
            raise RuntimeError("dm0 must be '1e' or a density matrix")
        else:
            dm = SpinParam.apply_fcn(lambda d: d.to(eng.device), dm0)
        if eng.polarized and not isinstance(dm, SpinParam):  # scf_qccalc.py:97-100
            dm = SpinParam(u=dm * 0.5, d=dm * 0.5)
        pol = eng.polarized
        fs, es = [], []
        fock = eng.dm2scp(dm)
        # restricted engines replay the Fock build as one hipGraph (dqc_amd/graph.py); "graph": False runs it eagerly
        # "diag": "purify" (default for closed shells) replaces eigh by GEMM-only purification inside the same graph
        # (dqc_amd/purify.py); "eigh" keeps the reference's diagonalise-and-occupy step (hf.py:105-113)
        graphed, purified = None, None
        # (direct SCF builds allocate stream-ordered scratch and upload pair tables per call: not captured)
        # (nor the builds of a Hamiltonian sharded over several GPUs: they hold collectives)
        # A direct-SCF engine still takes the purification step, launched eagerly: the 412 x 412 eigh of naphthalene / cc-pVTZ is
        # 6 ms of rocSOLVER launches per iteration against ~1 ms of GEMMs
        ham = getattr(eng, "hamilton", None)
        direct = bool(getattr(ham, "_direct", False))
        resumed_after_failure = bool(getattr(self, "_resumed_after_failure", False))
        self._resumed_after_failure = False
        skip_purify = bool(getattr(self, "_skip_purification", False))  # (the device loop wandered: eigh steps from the start)
        self._skip_purification = False
        if skip_purify:
            self.purification_dropped = True
        if opts.get("graph", os.environ.get("DQC_AMD_GRAPH", "1") != "0") and not getattr(ham, "sharded", False):
            from .graph import GraphedFock, GraphedSCFStep
            ws = [eng.orb_weight.u, eng.orb_weight.d] if pol else [eng.orb_weight]
            uniform = all((not w.numel()) or bool((w == w[0]).all()) for w in ws)
            if opts.get("diag", os.environ.get("DQC_AMD_DIAG", "purify")) == "purify" and uniform and getattr(eng, "ovlp", None) is None and not skip_purify:
                purified = GraphedSCFStep(eng, capture=not direct)
            elif not pol and not direct:
                graphed = GraphedFock(eng)
        perr = None
        fprev = None
        gram = np.zeros((0, 0))
        best_err, best_it = float("inf"), 0
        self.converged = self.stalled = False
        # iteration budget: `maxiter` steps -- and `maxiter` more from the restart, once, when the purification step is dropped (the
        # reference's diagonalise-and-occupy iteration gets the cap the caller set; the steps the purification wandered do not count)
        it, it_end = -1, int(opts["maxiter"])
        nfail = 1 if resumed_after_failure else 0   # projector failures so far (the device loop's hand-over counts)
        while it + 1 < it_end:
            it += 1
            self.niter = it + 1
            S = getattr(eng, "ovlp", None)  # overlap of a non-orthogonalised basis (None: identity)
            if pol:
                dms = torch.stack([dm.u, dm.d])
                err = fock @ dms - dms @ fock if S is None else fock @ dms @ S - S @ dms @ fock
            else:
                err = fock @ dm - dm @ fock if S is None else fock @ dm @ S - S @ dm @ fock  # [F, D] (S = 1) or F D S - S D F
            # ONE host read per iteration: max |[F, D]|, the projector error of the step just taken, and the new row of the
            # DIIS Gram matrix (this error vector against the stored ones) travel together
            ev = err.reshape(-1)
            hist = es[-(int(opts["history"]) - 1):] if int(opts["history"]) > 1 else []
            row = (torch.stack(hist + [ev]) * ev).sum(-1)
            zero = torch.zeros((), dtype=fock.dtype, device=fock.device)
            # the reference's own fixed-point residual max|F_out - F_in| (scp2scp(y) - y, scf_qccalc.py:109-113) rides along
            fres_t = (fock - fprev).abs().max() if fprev is not None else zero + float("inf")
            head = torch.stack([err.abs().max(), perr if perr is not None else zero, fres_t])
            host = yield torch.cat([head, row])
            emax, pe, fres, grow = float(host[0]), float(host[1]), float(host[2]), host[3:]
            self.fock_residual = fres
            if os.environ.get("DQC_AMD_SCF_TRACE"):
                print("scf it %2d  max|[F,D]| %.2e  max|F_out-F_in| %.2e" % (it, emax, fres), flush=True)
            if perr is not None and not pe < 1e-9:  # purification did not converge (vanishing gap): redo this step through eigh
                self.eigh_fallbacks = getattr(self, "eigh_fallbacks", 0) + 1
                nfail += 1
                dm = eng.scp2dm(fprev)
                fock = eng.dm2scp(dm)
                perr = None
                dmm = torch.stack([dm.u, dm.d]) if pol else dm
                err = fock @ dmm - dmm @ fock if S is None else fock @ dmm @ S - S @ dmm @ fock
                ev = err.reshape(-1)
                h2 = yield torch.cat([err.abs().max().reshape(1), (torch.stack(hist + [ev]) * ev).sum(-1)])
                emax, grow = float(h2[0]), h2[1:]
            if purified is not None and (not np.isfinite(emax) or (it - best_it >= 40 and emax > 1e-6) or nfail >= 3
                                         or (it + 1 >= it_end and emax > 1e-6)):
                # The purification step has no preferred basis inside a degenerate Fermi level (open p shells, ...): every step then
                # lands on another rotation of the degenerate orbitals, the iteration wanders for ever and the DIIS system eventually
                # blows up (UKS SCAN on the oxygen triplet: NaN after 88 steps, or -26 Ha).  The reference diagonalises (hf.py:105-113),
                # which fixes the orbitals: after 40 steps without progress -- or at the first non-finite error -- the loop drops the
                # purification and starts again from the core guess with eigh steps and a fresh history
                self.purification_dropped = True
                purified, graphed, perr, fprev = None, None, None, None
                fs, es, gram = [], [], np.zeros((0, 0))
                # (from the core guess again: continued from the best wandering iterate the eigh steps did not settle in 160 more
                # iterations on that system, from the core guess they converge in 23)
                n_ = eng.shape[-1]
                z_ = torch.zeros((n_, n_), dtype=eng.dtype, device=eng.device)
                dm = eng.scp2dm(eng.dm2scp(SpinParam(u=z_, d=z_) if pol else z_))
                fock = eng.dm2scp(dm)
                best_err, best_it = float("inf"), it
                it_end = it + 1 + int(opts["maxiter"])
                continue
            self.scf_error = emax  # max |[F, D]| of the last iterate
            # the commutator bottoms out at the round-off floor of the Fock build (fp64 atomics; ~1e-9 for ~200 AOs,
            # growing with the matrix size): an iterate that is within 100 f_tol and has not improved for 8 steps ends the
            # loop as `stalled` -- `converged` keeps meaning f_tol, `scf_error` reports what was achieved
            if emax < best_err * 0.9:
                best_err, best_it = emax, it
            # (the reference's fixed-point residual max|F_out - F_in|, kept in self.fock_residual, runs ~3x the commutator;
            # stopping on it as well -- 3e-9 or SURVEY.md 8d's 1e-8 -- saves 1-7 % of the iterations but costs a digit in the
            # non-variational energy components that the goldens pin to 1e-7: not done)
            if emax < opts["f_tol"]:
                self.converged = True
                break
            if emax < 100 * opts["f_tol"] and it - best_it >= 8:
                self.stalled = True
                warnings.warn("SCF stopped at the round-off floor of the Fock build: max|[F,D]| = %.2e (f_tol %.1e)"
                              % (emax, opts["f_tol"]))
                break
            fs.append(fock)
            es.append(ev)
            if len(fs) > opts["history"]:
                fs.pop(0)
                es.pop(0)
            m = len(fs)
            # the Gram matrix lives on the host and grows by the row just read (broadcast-multiply-reduce on the device:
            # the GEMM form E @ E.T hits a pathological rocBLAS path for the tall-skinny fp64 shape, 7 ms for 8 x 43264)
            keep = m - 1  # == len(hist): the stored vectors that survive
            gnew = np.zeros((m, m))
            if keep:
                gnew[:keep, :keep] = gram[-keep:, -keep:]
            gnew[keep, :] = grow
            gnew[:, keep] = grow
            gram = gnew
            if m > 1:
                B = np.zeros((m + 1, m + 1))
                # the Pulay coefficients do not change when the Gram block is scaled (only the multiplier does): normalised to
                # a unit largest diagonal, otherwise lstsq's rank cut (eps x largest singular value, set by the +-1 border)
                # discards the whole Gram block once the errors fall below ~1e-8 and the mix degrades to a plain average
                B[:m, :m] = gram / max(float(np.max(np.diag(gram))), 1e-300) if _DIIS_SCALE else gram
                B[m, :m] = -1
                B[:m, m] = -1
                rhs = np.zeros(m + 1)
                rhs[m] = -1
                # (m+1) x (m+1) Pulay system on the host with numpy: torch's CPU lstsq costs ~7 ms per call on a
                # 256-thread box (thread-pool wake-up), several times the whole Fock build
                try:
                    c = np.linalg.lstsq(B, rhs, rcond=None)[0][:m] if np.isfinite(B).all() else None
                except np.linalg.LinAlgError:  # (LAPACK's SVD gives up on an ill-scaled or non-finite Gram block)
                    c = None
                if c is None or not np.isfinite(c).all():
                    # a Pulay system that cannot be solved: forget the history and take the plain step from this Fock matrix (the next
                    # iterations rebuild the history; a non-finite commutator is caught at the top of the loop)
                    fs, es, gram = [fock], [ev], np.array([[float(grow[-1])]])
                    c = np.ones(1)
                    m = 1
                c = torch.as_tensor(c, dtype=fock.dtype).to(fock.device)
                fmix = (c.reshape((-1,) + (1,) * fock.dim()) * torch.stack(fs)).sum(0)
            else:
                fmix = fock
            fprev = fmix
            if purified is not None:
                f_out, d_out, perr = purified(fmix)
                if purified.graph is None:  # eager step (direct SCF): fresh tensors, and the Hamiltonian's caches stay keyed on them
                    fock, dm = f_out, d_out
                else:  # static buffers of the graph: copy out
                    fock, perr = f_out.clone(), perr.clone()
                    dm = SpinParam(u=d_out.u.clone(), d=d_out.d.clone()) if pol else d_out.clone()
            elif graphed is not None:
                fock = graphed(eng.scp2orb(fmix)).clone()
                dm = graphed.density_matrix().clone()
            else:
                dm = eng.scp2dm(fmix)
                fock = eng.dm2scp(dm)
        self._dm = dm
        self._fock = fock
        self._energy = None
        self._has_run = True
        if not self.accepted:  # the reference's xitorch solver emits a ConvergenceWarning here
            warnings.warn("SCF did not converge in %d iterations: max|[F,D]| = %.2e (f_tol %.1e); energy() and "
                          "nuclear_gradient() of this object refer to a non-stationary density"
                          % (self.niter, self.scf_error, opts["f_tol"]))

    def energy(self):
        assert self._has_run
        e = getattr(self, "_energy", None)  # the lockstep driver keeps dm2energy(dm) of the final Fock build (same call, same dm)
        if e is not None:
            return e
        return self._engine.dm2energy(self._dm)

    def aodm(self):
        assert self._has_run
        return self._dm

    def nuclear_gradient(self):
        """dE/dR (natm, 3) of the converged energy -- what the reference gets from torch.autograd.grad(energy, atompos)
        (test_hf.py:78-111, test_ks.py:114-137); restricted HF and LDA (dqc_amd/gradient.py)"""
        assert self._has_run
        if not self.accepted:  # the analytic gradient has no orbital-response terms: it is only valid at a fixed point
            warnings.warn("nuclear_gradient() of an unconverged SCF (max|[F,D]| = %.2e) is not the derivative of its energy"
                          % self.scf_error)
        from .gradient import nuclear_gradient
        return nuclear_gradient(self)

    def dm2energy(self, dm):
        return self._engine.dm2energy(dm)


class HF(SCF_QCCalc):
    def __init__(self, system, restricted: Optional[bool] = None, variational: bool = False):
        if variational:
            raise NotImplementedError("the variational solver is out of scope (SURVEY.md 2, row 3)")
        super().__init__(_Engine(system, None, is_ks=False, restricted=restricted))
        self._ctor_kwargs = {"restricted": restricted}  # to rebuild the same calculation on another geometry


class KS(SCF_QCCalc):
    def __init__(self, system, xc, restricted: Optional[bool] = None, variational: bool = False):
        if variational:
            raise NotImplementedError("the variational solver is out of scope (SURVEY.md 2, row 3)")
        super().__init__(_Engine(system, xc, is_ks=True, restricted=restricted))
        self._ctor_kwargs = {"xc": xc, "restricted": restricted}

"""Device-resident SCF loop of ONE molecule: every iteration is ONE hipGraph replay, the host never waits for the device.

The reference drives one self-consistency loop per molecule from the host (dqc/qccalc/scf_qccalc.py:84-116: `equilibrium` over
scp2scp = dm2scp(scp2dm(.)), engine steps hf.py:105-113 / ks.py:176-187).  Mirrored literally on a GPU -- dqc_amd/qccalc.py's
`_run_gen` -- every iteration ends in a blocking device -> host read (max|[F, D]| + the new DIIS Gram row), then the host solves
the Pulay system with numpy, uploads the coefficients, mixes, and only then launches the next step: measured on MI355X
(tools/gpu_one_molecule_iteration.py) 3.1 ms per iteration for a 20-atom cc-pVDZ molecule against 2.4 ms for the step's own
graph, 1.8 against 0.8 ms for benzene LDA and 1.2 against 0.6 ms for benzene RHF -- the device idles while the host thinks.

Here the whole iteration lives on the device and is captured ONCE:

    E = F D - D F, max|E|                      (the commutator of the entering pair)
    history rings, Gram row / column           index_copy_ with a DEVICE slot counter
    c = Pulay coefficients                     dqc_diis_solve_dev (valid slots read from the device counter)
    F_mix = sum_k c_k F_k
    P = projector(F_mix), Q = orth(P Omega)    GEMM-only purification (dqc_amd/purify.py), Cholesky QR
    D' = ao_orb2dm(Q, n), F' = dm2scp(D'), E_tot(D')   the Fock build (hcgto.py:204-269, 371-495)
    ring[k % 2] <- entering (F, D, E_tot);  pinned host <- (max|E|, projector error)      asynchronous copy

and the host loop is `replay(); look at the scalars of the iteration BEFORE the one just launched`.  The queue never drains: while
the device runs iteration k the host reads iteration k - 1's two doubles from pinned memory.  Convergence is therefore seen one
iteration late -- one speculative replay at the end -- and the converged pair is taken from the ring (the later iteration writes
the other slot).  Same numbers as `_run_gen`: same error vector, same history length, same minimum-norm Pulay solve, same
purification; the step from the core guess goes through eigh when its purification fails (degenerate bare-nucleus levels), a
later step that fails hands the run to `_run_gen` (vanishing gap: eigh needed), which also takes
everything it does not cover (a user dm0, non-uniform occupations, a raw AO basis, direct / sharded Hamiltonians).
"""
import os
import warnings

import torch

from . import lib
from .utils.datastruct import SpinParam


def eligible(engine, dm0, opts) -> bool:
    """can `DeviceLoop` run this calculation?  (otherwise the host-driven generator of qccalc.py does)"""
    if not isinstance(dm0, str) or dm0 != "1e":
        return False
    if not opts.get("graph", os.environ.get("DQC_AMD_GRAPH", "1") != "0"):
        return False
    if opts.get("driver", os.environ.get("DQC_AMD_SCF_DRIVER", "device")) != "device":
        return False
    if opts.get("diag", os.environ.get("DQC_AMD_DIAG", "purify")) != "purify":
        return False
    h = engine.hamilton
    if getattr(h, "_direct", False) or getattr(h, "sharded", False) or getattr(engine, "ovlp", None) is not None:
        return False
    if not (1 <= int(opts.get("history", 12)) <= 16):
        return False
    ws = [engine.orb_weight.u, engine.orb_weight.d] if engine.polarized else [engine.orb_weight]
    if not all((not w.numel()) or bool((w == w[0]).all()) for w in ws):
        return False
    if torch.cuda.is_current_stream_capturing():
        return False
    return True


class DeviceLoop:
    def __init__(self, engine, history: int = 12):
        from .graph import GraphedSCFStep
        self.eng = eng = engine
        self.pol = bool(eng.polarized)
        self.H = H = int(history)
        n, dev, dt = eng.shape[-1], eng.device, eng.dtype
        self.n, self.S = n, (2 if self.pol else 1)
        S = self.S
        self.step = GraphedSCFStep(eng, capture=False)
        # state of the loop (all static: the graph holds their addresses)
        self.fock = torch.zeros((S, n, n), dtype=dt, device=dev)
        self.dm = torch.zeros((S, n, n), dtype=dt, device=dev)
        self.etot = torch.zeros((), dtype=dt, device=dev)
        self.perr = torch.zeros((), dtype=dt, device=dev)          # projector error of the step that made (fock, dm)
        self.fh = torch.zeros((H, S * n * n), dtype=dt, device=dev)
        self.eh = torch.zeros((H, S * n * n), dtype=dt, device=dev)
        self.gram = torch.zeros((1, H, H), dtype=dt, device=dev)
        self.coef = torch.zeros((1, H), dtype=dt, device=dev)
        self.count = torch.zeros(1, dtype=torch.int64, device=dev)  # iterations done = vectors stored
        self.ring_f = torch.zeros((2, S, n, n), dtype=dt, device=dev)
        self.ring_d = torch.zeros((2, S, n, n), dtype=dt, device=dev)
        self.ring_e = torch.zeros(2, dtype=dt, device=dev)
        self.stats = torch.zeros((2, 2), dtype=dt, device=dev)      # per parity: max|[F, D]|, projector error of the entering pair
        self.host = torch.zeros((2, 2), dtype=dt).pin_memory()
        self.graph = None

    # ------------------------------------------------------------------ pieces
    def _build_from(self, fmix):
        """(fock, dm, etot, perr) <- one step from the mixed Fock matrix (purification + Fock build)"""
        st = self.step
        st.f_in.copy_(fmix if self.pol else fmix[0])
        f_out, d_out, perr = st._body()
        if self.pol:
            self.fock.copy_(f_out)
            self.dm[0].copy_(d_out.u)
            self.dm[1].copy_(d_out.d)
            dmx = d_out
        else:
            self.fock[0].copy_(f_out)
            self.dm[0].copy_(d_out)
            dmx = d_out
        if not self.pol:  # the two-electron parts are by-products of the build just made (the Hamiltonian's memo); an unrestricted
            self.etot.copy_(self.eng.dm2energy(dmx))  # engine evaluates dm2energy of the final densities once, after the loop
        self.perr.copy_(perr)

    def _iteration(self):
        H, S, n = self.H, self.S, self.n
        slot = self.count % H                       # (1,) int64 on the device
        par = self.count % 2
        a = torch.bmm(self.fock, self.dm)
        err = a - a.transpose(-2, -1)               # [F, D] per spin (both symmetric)
        ev = err.reshape(1, -1)
        self.eh.index_copy_(0, slot, ev)
        self.fh.index_copy_(0, slot, self.fock.reshape(1, -1))
        row = (self.eh * ev).sum(-1)                # scalar products with every stored error vector (unused slots hold zeros)
        self.gram[0].index_copy_(0, slot, row.reshape(1, H))
        self.gram[0].index_copy_(1, slot, row.reshape(H, 1))
        # the entering pair and its scalars go to the ring slot of this iteration's parity
        self.ring_f.index_copy_(0, par, self.fock.unsqueeze(0))
        self.ring_d.index_copy_(0, par, self.dm.unsqueeze(0))
        self.ring_e.index_copy_(0, par, self.etot.reshape(1))
        self.stats.index_copy_(0, par, torch.stack([err.abs().max(), self.perr]).reshape(1, 2))
        self.count += 1
        lib.diis_solve_dev(self.gram, self.count, self.coef)
        fmix = (self.coef[0].unsqueeze(-1) * self.fh).sum(0).reshape(S, n, n)
        self._build_from(fmix)
        self.host.copy_(self.stats, non_blocking=True)

    def _clear_memos(self):
        h = self.eng.hamilton
        h._jk_cache = None
        h._jkpol_cache = None
        h._dm_factor = None
        h._energy_memo = None

    def _capture(self):
        dev = self.eng.device
        # warm-up on a side stream (allocator, lazy kernel attributes), from a scratch copy of the state; then capture
        keep = [t.clone() for t in (self.fock, self.dm, self.etot, self.perr, self.fh, self.eh, self.gram, self.count)]
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            self._iteration()
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        for t, k in zip((self.fock, self.dm, self.etot, self.perr, self.fh, self.eh, self.gram, self.count), keep):
            t.copy_(k)
        getattr(self.eng.hamilton, "_tiles", None) if self.eng.hamilton.df is None else None  # (retires the tile-fill event)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._iteration()
        for t, k in zip((self.fock, self.dm, self.etot, self.perr, self.fh, self.eh, self.gram, self.count), keep):
            t.copy_(k)  # (a capture does not execute: belt and braces)
        self._clear_memos()

    # ------------------------------------------------------------------ the loop
    def run(self, qc, opts):
        """drive `qc` (an SCF_QCCalc) to convergence; returns True, or False when the caller has to take over with the
        host-driven loop (a purification that did not converge: the step needs eigh)"""
        eng, n, S, dev = self.eng, self.n, self.S, self.eng.device
        f_tol, maxiter = float(opts["f_tol"]), int(opts["maxiter"])
        if maxiter < 1:
            raise RuntimeError("maxiter must be >= 1")
        trace = bool(os.environ.get("DQC_AMD_SCF_TRACE"))
        qc._resume_dm = None
        # core guess (scf_qccalc.py:88-91): F0 = dm2scp(0), occupy its lowest orbitals -- the first step, eagerly
        z = torch.zeros((n, n), dtype=eng.dtype, device=dev)
        f0 = eng.dm2scp(SpinParam(u=z, d=z) if self.pol else z)
        self._build_from(f0.reshape(S, n, n))
        for t in (self.fh, self.eh, self.gram, self.count):
            t.zero_()
        if not float(self.perr) < 1e-9:   # (the one synchronisation before the loop)
            # the bare-nucleus Hamiltonian often has a degenerate or vanishing gap at the Fermi level (open pi shells, ...): this one
            # step through eigh, as the reference's (hf.py:137-150) -- later steps that fail hand the run to the host-driven loop
            qc.eigh_fallbacks = getattr(qc, "eigh_fallbacks", 0) + 1
            dm = eng.scp2dm(f0)
            fock = eng.dm2scp(dm)
            if self.pol:
                self.fock.copy_(fock)
                self.dm[0].copy_(dm.u)
                self.dm[1].copy_(dm.d)
            else:
                self.fock[0].copy_(fock)
                self.dm[0].copy_(dm)
                self.etot.copy_(eng.dm2energy(dm))
            self.perr.zero_()
        if self.graph is None:
            self._capture()
        evs = [torch.cuda.Event(), torch.cuda.Event()]
        best_err, best_it = float("inf"), 0
        qc.converged = qc.stalled = False
        done_at = None
        stream = torch.cuda.current_stream(dev)

        def look(j):
            """host-side bookkeeping of iteration j (its scalars are in pinned memory once evs[j % 2] has completed)"""
            nonlocal best_err, best_it
            evs[j % 2].synchronize()
            emax, pe = float(self.host[j % 2, 0]), float(self.host[j % 2, 1])
            qc.niter, qc.scf_error = j + 1, emax
            if trace:
                print("scf it %2d  max|[F,D]| %.2e  (device loop)" % (j, emax), flush=True)
            if not emax == emax or emax == float("inf"):  # a non-finite error: the host-driven loop starts over with eigh steps
                return "restart"
            if getattr(self, "_test_fail_at", None) == j:  # (test hook: a projector failure at iteration j)
                return "fallback"
            if not pe < 1e-9:  # the projector of this step failed (vanishing gap): the host-driven loop resumes from the last good density
                return "fallback"
            if emax < best_err * 0.9:
                best_err, best_it = emax, j
            if j - best_it >= 40 and emax > 1e-6:
                # the wander guard of the host-driven loop (qccalc.py): inside a degenerate Fermi level the purification step has no
                # preferred basis and the iteration never settles (UKS SCAN, oxygen triplet) -- hand over, eigh steps from the core guess
                return "restart"
            if emax < f_tol:
                qc.converged = True
                return "done"
            if emax < 100 * f_tol and j - best_it >= 8:
                qc.stalled = True
                warnings.warn("SCF stopped at the round-off floor of the Fock build: max|[F,D]| = %.2e (f_tol %.1e)" % (emax, f_tol))
                return "done"
            return None

        verdict = None
        for k in range(maxiter):
            self.graph.replay()
            evs[k % 2].record(stream)
            if k >= 1:
                verdict = look(k - 1)
                if verdict:
                    done_at = k - 1
                    break
        if verdict is None:
            verdict = look(maxiter - 1)
            done_at = maxiter - 1
            if verdict is None and qc.scf_error > 1e-6:
                # out of iterations far from a fixed point: the reference's diagonalise-and-occupy step (hf.py:105-113) gets its turn
                # before the run is reported as not converged (the degenerate-level wander needs 40 steps to be told from slow progress)
                verdict = "restart"
        torch.cuda.synchronize(dev)
        if verdict == "restart":
            # wandering or non-finite: nothing of this run is worth resuming from (a non-finite commutator means the entering pair of
            # iteration `done_at` is already non-finite) -- the host-driven loop starts at the core guess WITHOUT the purification step
            qc._resume_dm = None
            qc._skip_purification = True
            return False
        if verdict == "fallback":
            # a projector failure with a finite error: the entering density of the failing iteration is the last good iterate (every
            # earlier step passed its projector check) -- the host-driven loop resumes from it instead of the core guess
            p = done_at % 2
            if done_at >= 1:
                qc._resume_dm = SpinParam(u=self.ring_d[p, 0].clone(), d=self.ring_d[p, 1].clone()) if self.pol else self.ring_d[p, 0].clone()
            return False
        # the entering pair of iteration `done_at` (the iteration launched after it wrote the other ring slot)
        p = done_at % 2
        if self.pol:
            qc._dm = SpinParam(u=self.ring_d[p, 0].clone(), d=self.ring_d[p, 1].clone())
            qc._fock = self.ring_f[p].clone()
        else:
            qc._dm = self.ring_d[p, 0].clone()
            qc._fock = self.ring_f[p, 0].clone()
        qc._energy = None if self.pol else self.ring_e[p].clone()
        qc._has_run = True
        if not qc.accepted:
            warnings.warn("SCF did not converge in %d iterations: max|[F,D]| = %.2e (f_tol %.1e); energy() and "
                          "nuclear_gradient() of this object refer to a non-stationary density" % (qc.niter, qc.scf_error, f_tol))
        return True

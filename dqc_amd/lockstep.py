"""Lockstep SCF of a batch of same-size molecules: everything around the Fock build is ONE set of launches for the batch.

The reference runs one self-consistency loop per molecule (dqc/qccalc/scf_qccalc.py:84-116: `dm0 = "1e"` core guess, then
the fixed point F = dm2scp(scp2dm(F)), engine steps dqc/qccalc/hf.py:105-113 / ks.py:176-187).  A batch is M independent
problems; on the GPU the part of an iteration that is NOT the Fock build -- commutator, DIIS,
`diagonalize` + `ao_orb2dm` -- is small-matrix work that leaves 95 % of the chip idle and costs ~100 launches per molecule.
Here M molecules with the same (nao, n_occ) advance together:

    E = F D - D F, max|E|                    batched GEMM on the stacked (M, n, n) matrices
    Gram row, Pulay coefficients, F_mix      stacked history on the device, dqc_diis_solve (one block per molecule)
    P = projector(F_mix)                     dqc_purify_tc2_batched: one launch per TC2 step for ALL molecules
    Q = orth(P Omega)                        dqc_orth_factor_batched
    F[m], D[m] = Fock build of Q[m]          per molecule: the hipGraph of ao_orb2dm + dm2scp (dqc_amd/graph.py), dealt to
                                             a few HIP streams so that one molecule's kernel tails overlap the next one's heads

with ONE device -> host read per iteration for the whole batch (max|[F, D]| and the projector error of every molecule).
A molecule that has converged is frozen (its Fock builds stop; its slot in the stacked arrays idles) until the batch is
done.  Same numbers as the one-molecule driver (dqc_amd/qccalc.py) up to round-off: same error vector, same history
length, same least-squares Pulay solve, same purification.

Restricted closed-shell and unrestricted (stacked F_u, F_d; hf.py:93-103) engines with uniform occupations per spin channel in
an orthogonalised basis; `signature(qc)` says whether a calculation qualifies -- everything else (restricted open-shell,
fractional occupations, raw AO bases) keeps the one-molecule driver (`batch.run_lockstep` sorts that out)."""
import os
import warnings

import torch

from . import lib
from .purify import _TC2_ITERS
from .utils.datastruct import SpinParam


def _spin_channels(eng):
    """[(n_occ, occupation)] per spin channel (one entry for a restricted engine), or None when a channel's occupations are
    not uniform; an empty channel (the reference keeps one orbital of weight 0 there, mol.py:437-441) counts as n_occ = 0"""
    ws = [eng.orb_weight.u, eng.orb_weight.d] if eng.polarized else [eng.orb_weight]
    out = []
    for w in ws:
        if not w.numel() or bool((w == 0).all()):
            out.append((0, 0.0))
        elif bool((w == w[0]).all()):
            out.append((int(w.numel()), float(w[0])))
        else:
            return None
    return out


def signature(qc):
    """key under which calculations can share a lockstep batch, or None when the calculation needs the one-molecule driver"""
    eng = qc._engine
    if getattr(eng, "ovlp", None) is not None:
        return None
    # a Hamiltonian sharded over several GPUs issues collectives inside its Fock build: every rank has to take the same
    # branches, which only SCF_QCCalc.run guarantees (it broadcasts rank 0's per-iteration scalars, hamilton.sync_scalars)
    if getattr(getattr(eng, "hamilton", None), "sharded", False):
        return None
    ch = _spin_channels(eng)
    if ch is None:
        return None
    n = int(eng.shape[-1])
    if any(not (r <= 128 and r < n) for r, _ in ch) or all(r == 0 for r, _ in ch):
        return None
    if not eng.polarized:
        return (str(eng.device), n, ch[0][0], ch[0][1])
    return (str(eng.device), n, tuple(r for r, _ in ch), tuple(o for _, o in ch))


def projectors_from_focks(focks, nocc, iters=None, tol=1e-13):
    """focks (M, n, n) symmetric, orthonormal bases -> (P (M, n, n), err (M,)): the batched form of
    purify.projector_from_fock (Gershgorin bounds, TC2 with one launch per step for the whole batch, two McWeeny steps)"""
    if iters is None:
        iters = _TC2_ITERS
    M, n, _ = focks.shape
    dev, dt = focks.device, focks.dtype
    diag = torch.diagonal(focks, dim1=-2, dim2=-1)
    rad = focks.abs().sum(-1) - diag.abs()
    emin, emax = (diag - rad).amin(-1), (diag + rad).amax(-1)
    eye = torch.eye(n, dtype=dt, device=dev)
    x = (emax.reshape(M, 1, 1) * eye - focks) / (emax - emin).reshape(M, 1, 1)
    ld = (n + 15) // 16 * 16
    if ld != n:
        xp = torch.zeros((M, ld, ld), dtype=dt, device=dev)
        xp[:, :n, :n] = x
    else:
        xp = x.contiguous()
    tmp = torch.empty_like(xp)
    state = torch.empty((M, 2 * (iters + 2)), dtype=dt, device=dev)
    lib.purify_tc2_batched(xp, tmp, nocc, iters, tol, state)
    x = xp[:, :n, :n]
    for _ in range(2):
        x2 = torch.bmm(x, x)
        x = 3.0 * x2 - 2.0 * torch.bmm(x2, x)
    x = (x + x.transpose(-2, -1)) * 0.5
    tr = torch.diagonal(x, dim1=-2, dim2=-1).sum(-1)
    err = (torch.bmm(x, x) - x).abs().amax((-2, -1)) + (tr - nocc).abs()
    return x, err


class LockstepSCF:
    """M restricted closed-shell calculations (HF / KS objects, built, same `signature`) iterated together.
        LockstepSCF(qcs).run()      -> every qc has .energy(), .aodm(), .converged, .niter, .scf_error as after qc.run()"""

    def __init__(self, qcs, nstreams: int = 3, graph="auto"):
        sigs = {signature(q) for q in qcs}
        if len(sigs) != 1 or None in sigs:
            raise ValueError("LockstepSCF needs calculations of one (device, nao, n_occ per spin) signature with uniform occupations")
        self.qcs = list(qcs)
        self.engines = [q._engine for q in qcs]
        e0 = self.engines[0]
        self.device, self.dtype = e0.device, e0.dtype
        self._engine = e0  # (batch.run_concurrent reads the device from here)
        self.pol = bool(e0.polarized)
        self.channels = _spin_channels(e0)            # [(n_occ, occupation)] per spin
        self.S = len(self.channels)
        self.n = int(e0.shape[-1])
        self.r = self.channels[0][0]
        self.nstreams = max(1, min(nstreams, len(qcs)))
        self._graphs = None
        # the per-molecule Fock build replays as a hipGraph where it is launch-bound (small molecules: ~30 launches for
        # 0.1-0.4 ms of kernels); a 20-atom build (1.4 ms of kernels) is issued eagerly -- capturing 32 graphs would cost
        # more than the launches they save
        self.use_graph = ((self.n <= 160) if graph == "auto" else bool(graph)) and not self.pol  # (GraphedFock is restricted-only)
        if any(getattr(e.hamilton, "_direct", False) or getattr(e.hamilton, "sharded", False) for e in self.engines):
            self.use_graph = False  # direct SCF builds are not capturable (per-call scratch and table uploads)
        gen = torch.Generator().manual_seed(20240229)
        self.omega = [torch.randn((self.n, r), dtype=self.dtype, generator=gen).to(self.device) if r else None for r, _ in self.channels]
        self.eigh_fallbacks = 0

    # ------------------------------------------------------------------ pieces
    def _fock_builders(self):
        if self._graphs is None:
            from .graph import GraphedFock
            self._graphs = [GraphedFock(e, warmup=1, with_energy=True) for e in self.engines]
        return self._graphs

    def _occupied(self, fmix):
        """fmix (M, S, n, n) -> ([Q_s (M, n, r_s) orthonormal occupied-space bases per spin channel, None for an empty one],
        err (M,)) without an eigensolver"""
        qs, err = [], 0.0
        for s_, (r, _) in enumerate(self.channels):
            if r == 0:
                qs.append(None)
                continue
            p, e = projectors_from_focks(fmix[:, s_].contiguous(), r)
            y = torch.matmul(p, self.omega[s_])
            g = torch.bmm(y.transpose(-2, -1), y)
            q = lib.orth_factor_batched(y, g)
            # a failed factorisation (NaN / wrong range) must show in the error so that the molecule falls back to eigh
            err = err + e + (torch.bmm(q, q.transpose(-2, -1)) - p).abs().amax((-2, -1))
            qs.append(q)
        return qs, err

    def _eigh_orbitals(self, m, fmix_m, qs):
        """orbitals of molecule m from eigh (hf.py:227-247) into the stacked bases"""
        for s_, (r, _) in enumerate(self.channels):
            if r:
                qs[s_][m].copy_(self.engines[m]._eigvecs(fmix_m[s_])[..., :r])

    def occupied_orbitals(self, focks):
        """orthonormal occupied orbitals (M, n, n_occ) of the stacked Fock matrices (restricted batches: (M, n, n) in) -- the
        `diagonalize` step of hf.py:227-247 for the whole batch, eigensolver-free; a matrix whose purification fails (no gap) goes
        through eigh.  Synchronises."""
        f4 = focks if focks.dim() == 4 else focks.unsqueeze(1)
        qs, err = self._occupied(f4)
        for m in (~(err < 1e-9)).nonzero().reshape(-1).tolist():
            self._eigh_orbitals(m, f4[m], qs)
            self.eigh_fallbacks += 1
        return qs[0] if focks.dim() == 3 else qs

    def _build(self, q, active, fock, dm, etot, streams):
        """fock[m], dm[m], etot[m] <- Fock build (and total energy) of the orbitals q[m] for the active molecules, dealt to the
        side streams"""
        main = torch.cuda.current_stream(self.device)
        graphs = self._fock_builders() if self.use_graph else None
        ready = torch.cuda.Event()
        ready.record(main)
        for k, m in enumerate(active):
            s = streams[k % len(streams)]
            s.wait_event(ready)
            with torch.cuda.stream(s):
                if graphs is not None:
                    g = graphs[m]
                    g.orb.copy_(q[0][m])
                    g.graph.replay()
                    fock[m, 0].copy_(g.fock)
                    dm[m, 0].copy_(g.dm)
                    etot[m].copy_(g.energy)
                elif not self.pol:  # hf.py:105-113 (ao_orb2dm) + the Fock build, as the one-molecule driver issues them
                    e = self.engines[m]
                    d = e.hamilton.ao_orb2dm(q[0][m], e.orb_weight)
                    fock[m, 0].copy_(e.dm2scp(d))
                    dm[m, 0].copy_(d)
                    etot[m].copy_(e.dm2energy(d))  # by-products of the build just made + tr(D h): no second pass
                else:  # unrestricted (hf.py:93-103): D_u, D_d from their own orbitals, stacked (F_u, F_d) back
                    e = self.engines[m]
                    ws = (e.orb_weight.u, e.orb_weight.d)
                    ds = [e.hamilton.ao_orb2dm(q[s_][m], ws[s_]) if q[s_] is not None else torch.zeros_like(dm[m, s_])
                          for s_ in range(2)]
                    fock[m].copy_(e.dm2scp(SpinParam(u=ds[0], d=ds[1])))
                    dm[m, 0].copy_(ds[0])
                    dm[m, 1].copy_(ds[1])
        for s in streams[:min(len(streams), len(active))]:
            main.wait_stream(s)

    # ------------------------------------------------------------------ the loop
    def run(self, **kw):
        gen = self._run_gen(**kw)
        try:
            req = next(gen)
            while True:
                req = gen.send(req.cpu().numpy())
        except StopIteration:
            pass
        return self

    def _run_gen(self, dm0="1e", fwd_options=None):
        """generator with the protocol of SCF_QCCalc._run_gen: yields the small device tensor it needs on the host (one per
        iteration for the whole batch), is resumed with its numpy copy"""
        if dm0 != "1e":
            raise RuntimeError("LockstepSCF starts from the core guess dm0='1e' (scf_qccalc.py:88-91)")
        opts = {"maxiter": 50, "f_tol": 1e-9, "history": 12}
        opts.update(fwd_options or {})
        H = int(opts["history"])
        M, n, S, dev, dt = len(self.qcs), self.n, self.S, self.device, self.dtype
        main = torch.cuda.current_stream(dev)
        streams = [torch.cuda.Stream(device=dev) for _ in range(self.nstreams)]
        fock = torch.empty((M, S, n, n), dtype=dt, device=dev)   # S = 1 restricted, 2 unrestricted (stacked F_u, F_d: hf.py:93-103)
        dm = torch.empty((M, S, n, n), dtype=dt, device=dev)
        fh = torch.zeros((M, H, S * n * n), dtype=dt, device=dev)
        eh = torch.zeros((M, H, S * n * n), dtype=dt, device=dev)
        gram = torch.zeros((M, H, H), dtype=dt, device=dev)
        coef = torch.zeros((M, H), dtype=dt, device=dev)
        etot = torch.zeros(M, dtype=dt, device=dev)
        trace = bool(os.environ.get("DQC_AMD_SCF_TRACE"))
        for q in self.qcs:
            q.converged = q.stalled = False
            q.niter, q.scf_error = 0, float("inf")
            q.eigh_fallbacks = 0

        def fix_failed(qmat, fmix, perr_host, which):
            # purification did not converge (vanishing gap): that molecule's orbitals come from eigh (hf.py:227-247)
            for m in which:
                if not perr_host[m] < 1e-9:
                    self._eigh_orbitals(m, fmix[m], qmat)
                    self.qcs[m].eigh_fallbacks += 1
                    self.eigh_fallbacks += 1

        # core guess (scf_qccalc.py:88-91): F0 = dm2scp(0), occupy its lowest orbitals
        z = torch.zeros((n, n), dtype=dt, device=dev)
        f0 = torch.stack([e.dm2scp(SpinParam(u=z, d=z) if self.pol else z) for e in self.engines]).reshape(M, S, n, n)
        qmat, perr = self._occupied(f0)
        host = yield perr
        active = list(range(M))
        fix_failed(qmat, f0, host, active)
        self._build(qmat, active, fock, dm, etot, streams)

        best = [[float("inf"), 0] for _ in range(M)]
        restarts = [0] * M
        fmix = f0
        for it in range(int(opts["maxiter"])):
            a = torch.bmm(fock.reshape(M * S, n, n), dm.reshape(M * S, n, n))
            err = (a - a.transpose(-2, -1)).reshape(M, S, n, n)  # [F, D] per spin (both symmetric)
            emax_t = err.abs().amax((-3, -2, -1))
            slot = it % H
            ev = err.reshape(M, -1)
            eh[:, slot] = ev
            fh[:, slot] = fock.reshape(M, -1)
            row = (eh * ev.unsqueeze(1)).sum(-1)  # scalar products with every stored error vector (unused slots hold zeros)
            gram[:, slot, :] = row
            gram[:, :, slot] = row
            m_valid = min(it + 1, H)
            if m_valid > 1:
                lib.diis_solve(gram, m_valid, out=coef)
                fmix = (coef.unsqueeze(-1) * fh).sum(1).reshape(M, S, n, n)
            else:
                fmix = fock.clone()
            # the next projector is formed before the host has seen max|[F, D]| (speculatively: it is ~1 ms for the whole batch)
            qmat, perr = self._occupied(fmix)
            host = yield torch.cat([emax_t, perr])
            emax, pe = host[:M], host[M:]
            if trace:
                print("lockstep it %2d  max|[F,D]|: %s" % (it, " ".join("%.1e" % emax[m] for m in range(M))), flush=True)
            for m in list(active):
                qc = self.qcs[m]
                qc.niter, qc.scf_error = it + 1, float(emax[m])
                if emax[m] < best[m][0] * 0.9:
                    best[m] = [float(emax[m]), it]
                done = emax[m] < opts["f_tol"]
                stagnant = (not done) and emax[m] < 100 * opts["f_tol"]
                if stagnant and it - best[m][1] >= 5 and restarts[m] < 2:
                    # five steps without progress close to the tolerance (seen once in ~200 molecule runs of the C5 batch:
                    # 3e-9 at step 42): restart this molecule's Pulay subspace from the current iterate -- every slot holds
                    # the present (F, [F, D]) pair, a rank-one Gram block whose minimum-norm solution is F itself -- before
                    # calling it stalled
                    restarts[m] += 1
                    best[m][1] = it
                    eh[m, :] = eh[m, slot].clone()
                    fh[m, :] = fh[m, slot].clone()
                    gram[m, :, :] = gram[m, slot, slot].clone()
                stalled = stagnant and it - best[m][1] >= 8
                if done or stalled:
                    qc.converged, qc.stalled = bool(done), bool(stalled)
                    self._finish(qc, m, fock, dm, etot)
                    active.remove(m)
            if not active:
                break
            fix_failed(qmat, fmix, pe, active)
            if it + 1 < int(opts["maxiter"]):
                self._build(qmat, active, fock, dm, etot, streams)
        for m in active:  # maxiter exhausted
            qc = self.qcs[m]
            self._finish(qc, m, fock, dm, etot)
            warnings.warn("SCF did not converge in %d iterations: max|[F,D]| = %.2e (f_tol %.1e)"
                          % (qc.niter, qc.scf_error, opts["f_tol"]))

    def _finish(self, qc, m, fock, dm, etot):
        if self.pol:
            qc._dm = SpinParam(u=dm[m, 0].clone(), d=dm[m, 1].clone())
            qc._fock = fock[m].clone()
            qc._energy = None  # (energy() evaluates dm2energy of the stored densities)
        else:
            qc._dm = dm[m, 0].clone()
            qc._fock = fock[m, 0].clone()
            qc._energy = etot[m].clone()  # engine.dm2energy(dm) as evaluated with the Fock build of this very dm
        qc._has_run = True

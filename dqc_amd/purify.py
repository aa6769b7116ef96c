"""Eigensolver-free occupied-space projector: trace-correcting purification (TC2, Niklasson 2002) by GEMMs.

On MI355X the Fock build of a 20-atom molecule takes 1.9 ms while `torch.linalg.eigh` (rocSOLVER) of the 208 x 208
Fock matrix takes 4.7 ms: diagonalisation, not the integrals, bounds the SCF iteration.  The SCF step only needs the
density matrix D = n * P, P the projector onto the n_occ lowest eigenvectors of F (`diagonalize` + `ao_orb2dm`,
dqc/qccalc/hf.py:105-113, 227-247; hcgto.py:272-281).  P is obtained without eigenvectors:

    X0 = (e_max I - F) / (e_max - e_min)            Gershgorin bounds, spectrum mapped into [0, 1]
    X  <- X^2            if tr X > n_occ            (lowers the trace)
          2 X - X^2      otherwise                  (raises it)

converges quadratically to P (35-55 iterations for molecular spectra); once |X^2 - X|_max < tol the iterate is frozen
(continuing would let the trace test pick the error-doubling branch at round-off level), and two McWeeny steps
3 X^2 - 2 X^3 -- contracting at both 0 and 1 -- polish it.  Everything is device-side (`torch.where` on device
scalars, no host decision), so the whole map F -> D replays inside the Fock-build hipGraph.  The result equals the
eigh-based projector to round-off (tests compare both); non-convergence (a vanishing HOMO-LUMO gap) is reported through
the returned idempotency error and the caller falls back to eigh.
"""
import os

import torch

# launches per projector.  Early SCF iterations of the 20-atom molecules (HOMO-LUMO gaps of 0.02-0.03 Ha, Gershgorin bounds
# 4 x wider than the spectrum) need 45-55 iterations, converged ones ~40; with 52 launches 1-3 steps per SCF run fell back to
# eigh (4.5 ms each): 64 launches (frozen iterates cost 3 us each) make the run 8 % faster (tools/gpu_scf_time.py)
_TC2_ITERS = int(os.environ.get("DQC_AMD_TC2_ITERS", "64"))
# DQC_AMD_PURIFY=launch: one launch per TC2 iteration (the round-2 form, and what the lockstep batches use) instead of the persistent kernel
_PERSISTENT = os.environ.get("DQC_AMD_PURIFY", "persist") != "launch"


def projector_from_fock(fock: torch.Tensor, nocc: int, iters: int = None, tol: float = 1e-13, fused: bool = True):
    """fock (n, n) symmetric, orthonormal basis -> (P (n, n), idempotency error (0-dim device tensor)).
    fused=True runs the iterations in the HIP kernel of csrc/purify.hip (one launch each); fused=False is the same
    iteration written with torch ops (used by the CPU-side unit test and as the A/B reference)."""
    if iters is None:
        iters = _TC2_ITERS
    n = fock.shape[-1]
    if fused and fock.is_cuda and n <= 256 and _PERSISTENT and fock.dim() == 2:
        # everything below -- bounds, X0, the iterations, the McWeeny polish, the error -- in ONE persistent launch whose workers share
        # an XCD (csrc/purify.hip: projector_persist_kernel): ~0.15 instead of 0.45-0.58 ms and ~90 fewer nodes in the SCF-step graph.
        # Should the kernel give up (workers on several XCDs, a barrier time-out) the returned error is large and the caller falls
        # back, exactly as for a purification that did not converge
        from . import lib
        return lib.projector_tc2(fock, nocc, iters, tol)
    eye = torch.eye(n, dtype=fock.dtype, device=fock.device)
    diag = torch.diagonal(fock)
    rad = fock.abs().sum(-1) - diag.abs()
    emin, emax = (diag - rad).min(), (diag + rad).max()
    x = (emax * eye - fock) / (emax - emin)
    if fused and fock.is_cuda:
        from . import lib
        ld = (n + 15) // 16 * 16
        xp = torch.zeros((ld, ld), dtype=fock.dtype, device=fock.device)
        xp[:n, :n] = x
        tmp = torch.empty_like(xp)
        state = torch.empty(2 * (iters + 2), dtype=fock.dtype, device=fock.device)
        if ld <= 256 and _PERSISTENT:
            # ONE persistent launch whose workers share an XCD (csrc/purify.hip): 0.12-0.15 instead of 0.45-0.58 ms per step.
            # Should the kernel give up (workers on several XCDs, a barrier time-out) the error returned below is large and the
            # caller falls back, exactly as for a purification that did not converge
            ctl = torch.empty(4, dtype=torch.int32, device=fock.device)
            lib.purify_tc2_persist(xp, tmp, nocc, iters, tol, state, ctl)
        else:
            lib.purify_tc2(xp, tmp, nocc, iters, tol, state)
        x = xp[:n, :n]
    else:
        done = torch.zeros((), dtype=torch.bool, device=fock.device)
        for _ in range(iters):
            x2 = x @ x
            done = done | ((x2 - x).abs().max() < tol)
            cand = torch.where(torch.trace(x) > nocc, x2, 2.0 * x - x2)
            x = torch.where(done, x, cand)
    for _ in range(2):
        x2 = x @ x
        x = 3.0 * x2 - 2.0 * (x2 @ x)
    x = (x + x.transpose(-2, -1)) * 0.5
    err = ((x @ x) - x).abs().max() + (torch.trace(x) - nocc).abs()
    return x, err

import sys
sys.path.insert(0, "/root/repo")
import torch, dqc_amd
from dqc_amd import lib
from tests import molecules as M
for mol, basis in ((M.c5_molecule(0), "cc-pvdz"), (M.benzene(), "cc-pvdz"), (M.H2O, "cc-pvdz"), (M.naphthalene(), "cc-pvtz")):
    m = dqc_amd.Mol(mol, basis=basis, grid="sg2")
    eng = dqc_amd.KS(m, xc="gga_x_pbe+gga_c_pbe")._engine
    n = eng.shape[-1]
    f = eng.dm2scp(torch.zeros((n, n), dtype=torch.float64, device="cuda"))
    res = []
    for it in range(6):
        fock = (f + f.T) * 0.5
        eye = torch.eye(n, dtype=fock.dtype, device=fock.device)
        diag = torch.diagonal(fock)
        rad = fock.abs().sum(-1) - diag.abs()
        emin, emax = (diag - rad).min(), (diag + rad).max()
        x = (emax * eye - fock) / (emax - emin)
        ld = (n + 15) // 16 * 16
        xp = torch.zeros((ld, ld), dtype=fock.dtype, device=fock.device); xp[:n, :n] = x
        tmp = torch.empty_like(xp)
        iters = 60
        state = torch.empty(2 * (iters + 2), dtype=fock.dtype, device=fock.device)
        lib.purify_tc2(xp, tmp, eng.norb, iters, 1e-13, state)
        idem = state[iters + 2:].cpu().numpy()
        k = next((i for i, v in enumerate(idem[:iters]) if v < 1e-13), -1)
        ev = torch.linalg.eigvalsh(fock)
        res.append((k, float(ev[0]), float(ev[-1]), float(ev[eng.norb] - ev[eng.norb - 1]), float(emin), float(emax)))
        f = eng.dm2scp(eng.scp2dm(f))
    print(n, [r[0] for r in res], "spectrum %.1f..%.1f gap %.3f gersh %.1f..%.1f" % res[-1][1:])

"""screened direct SCF (dqc_direct_*): Schwarz bounds against the stored integrals, J / K against the unscreened pass and the
tile store for tau = 0 and tau > 0, share of the quartets launched, timings; `big`: the naphthalene dimer / cc-pVTZ SCF"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, dqc_amd
from dqc_amd import lib
from tests import molecules as M
dev = torch.device("cuda")
def ev(fn, k=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3
cases = [("H2O/cc-pvdz", M.H2O, "cc-pvdz"), ("CH4/cc-pvtz", M.CH4 if hasattr(M, "CH4") else M.H2O, "cc-pvtz"),
         ("C5", M.c5_molecule(0), "cc-pvdz")]
if len(sys.argv) > 1 and sys.argv[1] in ("c4", "big"):
    cases.append(("C4", M.naphthalene(), "cc-pvtz"))
for name, geo, basis in cases:
    h = dqc_amd.Mol(geo, basis=basis).get_hamiltonian()
    tab = h._tab
    n = tab.nao
    D = torch.as_tensor(M.seeded_dm_ao(n, 10, np.eye(n), 3), device=dev)
    t0 = time.perf_counter()
    ctx = lib.DirectContext(tab, dev)
    torch.cuda.synchronize()
    tc = time.perf_counter() - t0
    Jr, Kr = lib.jk_direct(tab, D, True)
    J0, K0 = ctx.jk(D, True, 0.0)
    line = "%s nao %d: create %.1f ms | tau 0 vs unscreened pass: J %.1e K %.1e" % (
        name, n, tc * 1e3, float((J0 - Jr).abs().max() / Jr.abs().max()), float((K0 - Kr).abs().max() / Kr.abs().max()))
    if n <= 60:  # the bounds against the dense tensor
        tiles = lib.eri_tiles(tab, dev)
        g = lib.eri_dense(tiles, n).cpu().numpy() if hasattr(lib, "eri_dense") else None
        if g is not None:
            q, sh = ctx.bounds()
            off = np.concatenate([[0], np.cumsum(2 * tab.bas[:, 1] + 1)])
            worst = 0.0
            for (a, b), qq in zip(sh, q):
                blk = g[off[a]:off[a + 1], off[b]:off[b + 1], off[a]:off[a + 1], off[b]:off[b + 1]]
                na, nb = blk.shape[0], blk.shape[1]
                dg = np.abs(np.einsum("abab->ab", blk)).max()
                worst = max(worst, abs(np.sqrt(dg) - qq) / max(qq, 1e-300))
            # Schwarz inequality over the whole tensor
            Qf = np.zeros((len(off) - 1, len(off) - 1))
            for (a, b), qq in zip(sh, q): Qf[a, b] = Qf[b, a] = qq
            viol = 0.0
            for a in range(len(off) - 1):
                for b in range(len(off) - 1):
                    blk = np.abs(g[off[a]:off[a + 1], off[b]:off[b + 1]]).reshape(off[a + 1] - off[a], off[b + 1] - off[b], -1)
                    # max over functions of shells c, d
                    for c in range(len(off) - 1):
                        for d in range(len(off) - 1):
                            m = np.abs(g[off[a]:off[a + 1], off[b]:off[b + 1], off[c]:off[c + 1], off[d]:off[d + 1]]).max()
                            viol = max(viol, m - Qf[a, b] * Qf[c, d])
            line += " | bounds vs dense tensor: rel %.1e, worst Schwarz violation %.1e" % (worst, viol)
    print(line, flush=True)
    for tau in (1e-13, 1e-11, 1e-9):
        J1, K1 = ctx.jk(D, True, tau)
        tot, lau, dmax = ctx.stats()
        t1 = ev(lambda: ctx.jk(D, True, tau))
        Jj, _ = ctx.jk(D, False, tau)
        totj, lauj, _ = ctx.stats()
        print("   tau %.0e: launched %.3f of %d quartets (J only %.3f), max|D| %.2f | abs err J %.1e K %.1e (J-only pass %.1e) | J+K %.2f ms" % (
            tau, lau / tot, tot, lauj / totj, dmax, float((J1 - Jr).abs().max()), float((K1 - Kr).abs().max()), float((Jj - Jr).abs().max()), t1), flush=True)
    t0_ = ev(lambda: ctx.jk(D, True, 0.0))
    told = ev(lambda: lib.jk_direct(tab, D, True))
    dD = D * 1e-5
    tdd = ev(lambda: ctx.jk(dD, True, 1e-13))
    totd, laud, _ = ctx.stats()
    print("   J+K: context tau 0 %.2f ms, dqc_jk_direct (tables per call) %.2f ms; density difference 1e-5 D at tau 1e-13: %.2f ms (%.3f launched)" % (
        t0_, told, tdd, laud / totd), flush=True)
    ctx.close()
if len(sys.argv) > 1 and sys.argv[1] == "big":
    zs, pos = M.naphthalene()
    pos = np.array(pos)
    zs2, pos2 = list(zs) + list(zs), np.concatenate([pos, pos + np.array([0.0, 0.0, 6.6])]).tolist()
    for tau in ("1e-13", "0"):
        os.environ["DQC_AMD_DIRECT_TAU"] = tau
        import importlib
        from dqc_amd import hamilton as H
        H.HamiltonMI355._DIRECT_TAU = float(tau)
        t0 = time.perf_counter()
        mol = dqc_amd.Mol((zs2, pos2), basis="cc-pvtz", grid="sg2")
        qc = dqc_amd.KS(mol, xc="gga_x_pbe+gga_c_pbe")
        h = mol.get_hamiltonian()
        torch.cuda.synchronize()
        ts = time.perf_counter() - t0
        t0 = time.perf_counter()
        qc.run(fwd_options={"maxiter": int(sys.argv[2]) if len(sys.argv) > 2 else 30})
        torch.cuda.synchronize()
        print("naphthalene dimer / cc-pVTZ (nao %d) tau %s: setup %.1f s, %d SCF iterations in %.1f s (%.2f s each), max|[F,D]| %.2e, E = %.10f, last pass launched %.3f" % (
            h._nao_ao, tau, ts, qc.niter, time.perf_counter() - t0, (time.perf_counter() - t0) / qc.niter, qc.scf_error, float(qc.energy()),
            h._direct_stats[1] / max(h._direct_stats[0], 1)), flush=True)

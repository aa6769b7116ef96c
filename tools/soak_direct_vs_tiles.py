"""screened direct path against the stored tiles on seeded random molecules (6-12 atoms of H, C, N, O in an elongated box so that
the Schwarz test has something to cut, random basis): J and K of a random density at tau = 1e-13 / 1e-11, J-only pass, parts,
and the RHF / RKS energy through HamiltonMI355 in direct mode (incremental builds) against the stored-tile run"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, dqc_amd
from dqc_amd import lib, hamilton as H
from oracle import basis as ob
dev = torch.device("cuda")
bad = 0
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 16)
for seed in range(lo, hi):
    rng = np.random.default_rng(5000 + seed)
    nat = int(rng.integers(6, 13))
    basis = ["3-21G", "cc-pvdz", "6-311++G**", "cc-pvtz"][int(rng.integers(0, 4))]
    elems = [1, 1, 6] if basis == "cc-pvtz" else [1, 1, 6, 7, 8]
    while True:
        zs = [int(z) for z in rng.choice(elems, nat)]
        if sum(zs) % 2 == 0 and max(zs) > 1:
            break
    box = np.array([2.0, 2.0, 9.0])
    pos = rng.uniform(-box, box, (nat, 3))
    for i in range(nat):
        for _ in range(200):
            if i == 0 or np.min(np.linalg.norm(pos[:i] - pos[i], axis=1)) > 1.7:
                break
            pos[i] = rng.uniform(-box, box, 3)
    geo = (zs, pos.tolist())
    try:
        t = ob.make_tables(geo, basis)
        tab = lib.Tables(t.atm, t.bas, t.env)
        n = tab.nao
        D = torch.as_tensor(rng.standard_normal((n, n)) * np.exp(-rng.uniform(0, 6, (n, 1))), device=dev)
        tiles = lib.eri_tiles(tab, dev)
        Jt, Kt = lib.jk(tiles, D, lib.jk_workspace(n, dev), True)
        ctx = lib.DirectContext(tab, dev)
        msg = []
        for tau in (1e-13, 1e-11):
            J, K = ctx.jk(D, True, tau)
            tot, lau, _ = ctx.stats()
            Jj, _ = ctx.jk(D, False, tau)
            parts = [ctx.jk(D, True, tau, (r, 3)) for r in range(3)]
            eJ, eK = float((J - Jt).abs().max()), float((K - Kt).abs().max())
            eJj = float((Jj - Jt).abs().max())
            eP = max(float((sum(p[0] for p in parts) - Jt).abs().max()), float((sum(p[1] for p in parts) - Kt).abs().max()))
            npair = tab.nbas * (tab.nbas + 1) // 2
            assert max(eJ, eK, eJj, eP) < tau * npair, (tau, eJ, eK, eJj, eP)
            msg.append("tau %.0e: launched %.2f, err %.1e" % (tau, lau / tot, max(eJ, eK, eJj, eP)))
        ctx.close(); del tiles
        xc = [None, "lda_x+lda_c_pw", "gga_x_pbe+gga_c_pbe"][int(rng.integers(0, 3))]
        es = []
        for mode in ("tiles", "direct"):
            os.environ["DQC_AMD_ERI"] = mode
            m = dqc_amd.Mol(geo, basis=basis, grid="sg2")
            q = (dqc_amd.KS(m, xc=xc) if xc else dqc_amd.HF(m)).run(fwd_options={"maxiter": 80})
            es.append((float(q.energy()), q.accepted, q.niter))
        os.environ.pop("DQC_AMD_ERI", None)
        ok = abs(es[0][0] - es[1][0]) < 2e-9 or not (es[0][1] and es[1][1])
        if not ok: raise AssertionError("energies %r" % (es,))
        print("seed %d ok: %d atoms %s nao %d xc %s | %s | dE %.1e (accepted %s/%s, %d/%d its)" % (
            seed, nat, basis, n, xc, "; ".join(msg), es[1][0] - es[0][0], es[0][1], es[1][1], es[0][2], es[1][2]), flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        os.environ.pop("DQC_AMD_ERI", None)
        print("seed %d FAILED (%s, %s): %r" % (seed, zs, basis, e), flush=True)
print("failures:", bad)

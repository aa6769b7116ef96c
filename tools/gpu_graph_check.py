"""GPU check: hipGraph replay of the Fock build == eager dm2scp, for changing orbitals; and timing"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dqc_amd
from dqc_amd.graph import GraphedFock
from tests import molecules as M

dev = torch.device("cuda:0")
for name, mol, xc in [("h2o-pbe", "O 0 0 0.2217; H 0 1.4309 -0.8867; H 0 -1.4309 -0.8867", "gga_x_pbe+gga_c_pbe"),
                      ("h2o-hf", "O 0 0 0.2217; H 0 1.4309 -0.8867; H 0 -1.4309 -0.8867", None),
                      ("c5-pbe", M.c5_molecule(0), "gga_x_pbe+gga_c_pbe")]:
    m = dqc_amd.Mol(mol, basis="cc-pvdz", grid="sg3", device=dev)
    eng = (dqc_amd.KS(m, xc=xc) if xc else dqc_amd.HF(m))._engine
    n = eng.shape[-1]
    g = GraphedFock(eng)
    f = eng.dm2scp(torch.zeros((n, n), dtype=torch.float64, device=dev))
    for it in range(3):
        orb = eng.scp2orb(f)
        f_eager = eng.dm2scp(eng.hamilton.ao_orb2dm(orb, eng.orb_weight))
        f_graph = g(orb).clone()
        err = float((f_eager - f_graph).abs().max())
        print(name, "iter", it, "max |F_eager - F_graph| = %.2e" % err, flush=True)
        assert err < 1e-10
        f = f_eager
    torch.cuda.synchronize()
    for label, fn in [("eager", lambda: eng.dm2scp(eng.hamilton.ao_orb2dm(orb, eng.orb_weight))), ("graph", lambda: g(orb))]:
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        print("   %s: %.3f ms per Fock build" % (label, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
print("GRAPH OK")

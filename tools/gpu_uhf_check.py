"""GPU check of the UHF/UKS path against the reference literals (test_hf.py:141-206, test_ks.py:296-345)."""
import sys, torch
sys.path.insert(0, ".")
import dqc_amd
for z,spin,ref in [(1,1,-4.96198609e-01),(3,1,-7.38151326e+00),(5,1,-2.43897617e+01),(8,2,-7.43936572e+01)]:
    mol=dqc_amd.Mol(([z],[[0,0,0.]]),basis="3-21G",spin=spin); qc=dqc_amd.HF(mol).run(); e=float(qc.energy()); print("UHF atom",z,e,ref,abs(e-ref)/abs(ref),qc.niter,qc.converged)
mol=dqc_amd.Mol(([7,8],[[-1.0,0,0],[1.0,0,0]]),basis="3-21G",spin=1); qc=dqc_amd.HF(mol).run(fwd_options={"maxiter":100}); e=float(qc.energy()); print("UHF NO",e,-1.28477807e+02,abs(e+128.477807)/128.477807,qc.niter,qc.converged)
mol=dqc_amd.Mol("H -0.5 0 0; H 0.5 0 0",basis="3-21G"); e=float(dqc_amd.HF(mol,restricted=False).run().energy()); print("UHF H2",e,-1.07195346)
for xc,ref in [("lda_x",-148.149998931489),("lda_x+lda_c_pw",-1.49259447e+02),("gga_x_pbe",-149.64097658035521),("gga_x_pbe+gga_c_pbe",None)]:
    mol=dqc_amd.Mol(([8,8],[[-1.0,0,0],[1.0,0,0]]),basis="6-311++G**",spin=2,grid=3); qc=dqc_amd.KS(mol,xc=xc).run(fwd_options={"maxiter":100}); e=float(qc.energy()); print("UKS O2",xc,e,ref,qc.niter,qc.converged)

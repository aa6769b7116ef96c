"""Nuclear gradient dE/dR of a converged restricted SCF energy.

The reference obtains it by `torch.autograd.grad(energy, atompos)` through its "ip" derivative integrals
(dqc/hamilton/intor/molintor.py:463-500, gtoeval.py:173-193) and the implicit-function backward of the SCF fixed point
(dqc/qccalc/scf_qccalc.py:63-67, 109-113; its tests: test_hf.py:78-111, test_ks.py:114-137).  At the fixed point the
same derivative is the Hellmann-Feynman + Pulay expression -- no response equations:

    dE/dR_A = sum D dh/dR_A - sum W dS/dR_A                 one-electron terms      -> dqc_int1e_grad
            + sum (d_A a b|c d) [2 D_ab D_cd - k D_ac D_bd]  two-electron term       -> dqc_eri_grad (k = 1 HF, 0 KS)
            + dE_xc/dR_A                                      LDA and GGA, INCLUDING the grid response
            + dE_nn/dR_A

The XC part differentiates the discretised functional E_xc = sum_g w_g(R) e(rho(r_g(R))) exactly, as the reference's
autograd does: (i) Becke-weight derivative (torch autograd through dqc_amd.grid's own weight code, e_g held fixed),
(ii) grid points riding on their parent atom: + sum_{g in A} w_g v_g grad rho(r_g), (iii) basis-function centres:
- 2 sum_g w_g v_g sum_{mu in A} grad phi_mu (D phi)_mu (LDA; the GGA forms with AO second derivatives are in
_xc_gga_gradient).  Meta-GGA and unrestricted gradients are not built yet; with density fitting the two-electron term is
_df_coulomb_gradient (dqc_df_grad).
"""
import torch

from . import lib
from .grid import get_predefined_grid


def nuclear_gradient(qc) -> torch.Tensor:
    """qc: a converged dqc_amd.HF / dqc_amd.KS (restricted).  Returns dE/dR, shape (natm, 3), Hartree / Bohr."""
    eng = qc._engine
    if eng.polarized:
        raise NotImplementedError("nuclear gradients of unrestricted calculations are not built yet")
    h = eng.hamilton
    if eng.is_ks and h.xcfamily not in (1, 2):
        raise NotImplementedError("nuclear gradients are built for HF, LDA and GGA functionals (not meta-GGA)")
    mol = eng.get_system()
    dev = h.device
    X = h._orthozer
    dm, fock = qc._dm, qc._fock
    fock = (fock + fock.transpose(-2, -1)) * 0.5
    eps, C = torch.linalg.eigh(fock)
    n = eng.norb
    Cocc = C[:, :n]
    w_orth = (Cocc * (eng.orb_weight * eps[:n]).unsqueeze(0)) @ Cocc.T      # energy-weighted density
    d_ao = X @ dm @ X.T
    w_ao = X @ w_orth @ X.T
    d_ao = (d_ao + d_ao.T) * 0.5
    T = lib.cart2sph_matrix(h._tab, dev)                                        # (nao, ncart)
    dcart = (T.T @ d_ao @ T).contiguous()
    wcart = (T.T @ w_ao @ T).contiguous()
    natm = len(mol.atomzs)
    grad = torch.zeros((natm, 3), dtype=torch.float64, device=dev)
    lib.int1e_grad(grad, dcart, wcart, h._tab, h._zs)
    if h.df is None:
        lib.eri_grad(grad, dcart, 0.0 if eng.is_ks else 1.0, h._tab)
    else:
        _df_coulomb_gradient(h, d_ao, grad)
    if eng.is_ks:
        grad = grad + (_xc_lda_gradient(eng, d_ao) if h.xcfamily == 1 else _xc_gga_gradient(eng, d_ao))
    return grad + _nuclei_gradient(mol).to(dev)


def _df_coulomb_gradient(h, d_ao, grad):
    """density-fitted J (DFMol.get_elrep, dfmol.py:60-79):  E_J = 1/2 t^T M^-1 t  ->  sum D c d(ij|k) - 1/2 c^T dM c"""
    df = h.df
    dev = h.device
    nao, _, naux = df.j3c.shape
    t = (df.j3c.reshape(nao * nao, naux) * d_ao.reshape(-1, 1)).sum(0)        # t_k = sum_ij D_ij (ij|k)
    c = df._inv_j2c @ t
    T = lib.cart2sph_matrix(df._tab, dev)                                       # all shells: orbital, then auxiliary
    nall = T.shape[0]
    dbig = torch.zeros((nall, nall), dtype=torch.float64, device=dev)
    dbig[:nao, :nao] = d_ao
    cbig = torch.zeros(nall, dtype=torch.float64, device=dev)
    cbig[nao:] = c
    dcart = (T.T @ dbig @ T).contiguous()
    ccart = (T.T @ cbig).contiguous()
    # the concatenated table lists every atom twice (orbital parent, auxiliary parent): fold the two halves
    gbig = torch.zeros((df._tab.natm, 3), dtype=torch.float64, device=dev)
    lib.df_grad(gbig, dcart, ccart, df._tab, df._orb_range, df._aux_range)
    natm = grad.shape[0]
    grad += gbig[:natm] + gbig[natm:]


def _nuclei_gradient(mol):
    z = mol.atomzs.to(torch.float64).cpu()
    pos = mol.atompos.to(torch.float64).cpu()
    d = pos.unsqueeze(1) - pos.unsqueeze(0)                 # R_A - R_B
    r = d.norm(dim=-1) + torch.eye(len(z), dtype=torch.float64)
    f = (z.unsqueeze(1) * z.unsqueeze(0)) / r ** 3
    f = f - torch.diag(torch.diag(f))
    return -(f.unsqueeze(-1) * d).sum(1)


def _xc_lda_gradient(eng, d_ao):
    h = eng.hamilton
    mol = eng.get_system()
    dev = h.device
    nao, ld = h._nao_ao, h._ld
    ao = h._ao if h._ao.dim() == 2 else h._ao[0]
    # AO gradients on the grid (the Hamiltonian of an LDA run only holds the values)
    aod = lib.eval_gto(h._tab, h.rgrid, 1)                                    # (4, ngrid, ld)
    dpad = lib.pad_matrix(d_ao, ld)
    rho, grho = lib.grid_density(aod, nao, dpad, True)
    edens, vrho, _ = lib.xc_eval(h.xc.terms, rho, None, want_e=True, want_v=True)
    wv = h.dvolume * vrho
    natm = len(mol.atomzs)
    # (ii) grid points ride on their parent atom
    owner = _grid_owner(mol, h.rgrid.shape[0], dev)
    g = torch.zeros((natm, 3), dtype=torch.float64, device=dev)
    g.index_add_(0, owner, (wv.unsqueeze(0) * grho).T.contiguous())
    # (iii) basis-function centres: -2 sum_g w v sum_{mu in A} dphi_mu (D phi)_mu
    m = (aod[0] @ dpad) * wv.unsqueeze(-1)                                    # (ngrid, ld)
    per_ao = torch.stack([(aod[d + 1] * m).sum(0) for d in range(3)], dim=-1)[:nao]   # (nao, 3)
    ao_atom = _ao_owner(h, dev)
    g.index_add_(0, ao_atom, -2.0 * per_ao)
    # (i) Becke-weight derivative, energy density held fixed
    pos = mol.atompos.to(dtype=torch.float64, device=dev).clone().requires_grad_(True)
    grid = get_predefined_grid(mol._grid_inp, mol.atomzs.tolist(), pos, dtype=torch.float64, device=dev)
    loss = (grid.get_dvolume() * edens.detach()).sum()
    g = g + torch.autograd.grad(loss, pos)[0]
    del ao
    return g


_HESS = ((4, 5, 6), (5, 7, 8), (6, 8, 9))  # component of d2/(d i d j) in the deriv-3 AO array


def _xc_gga_gradient(eng, d_ao):
    """GGA: with b = Phi D, c_i = d_i Phi D, u = 2 v_sigma grad rho (the `vgrad` of dqc_xc_eval), S_j = sum_i u_i d_i d_j Phi
        (ii)  grid points riding on atom A:  sum_{g in A} w [v_rho d_j rho + 2 sum_mu (b S_j + c_j (u . grad phi))]
        (iii) centres of the AOs on A:       -2 sum_g w sum_{mu in A} [d_j phi (v_rho b + u . c) + b S_j]
    plus (i) the Becke-weight derivative; (ii) + (iii) summed over atoms cancel identically."""
    h = eng.hamilton
    mol = eng.get_system()
    dev = h.device
    nao, ld = h._nao_ao, h._ld
    ao = lib.eval_gto(h._tab, h.rgrid, 3)                                      # (10, ngrid, ld)
    dpad = lib.pad_matrix(d_ao, ld)
    rho, grho = lib.grid_density(ao[:4], nao, dpad, True)
    edens, vrho, u = lib.xc_eval(h.xc.terms, rho, grho, want_e=True, want_v=True)
    w = h.dvolume
    natm = len(mol.atomzs)
    b = ao[0] @ dpad                                                           # (ngrid, ld)
    c = [ao[1 + i] @ dpad for i in range(3)]
    t1 = vrho.unsqueeze(-1) * b + sum(u[i].unsqueeze(-1) * c[i] for i in range(3))
    ugphi = sum(u[i].unsqueeze(-1) * ao[1 + i] for i in range(3))              # u . grad phi
    owner = _grid_owner(mol, h.rgrid.shape[0], dev)
    ao_atom = _ao_owner(h, dev)
    g = torch.zeros((natm, 3), dtype=torch.float64, device=dev)
    q = torch.empty((h.rgrid.shape[0], 3), dtype=torch.float64, device=dev)
    per_ao = torch.empty((nao, 3), dtype=torch.float64, device=dev)
    for j in range(3):
        s_j = sum(u[i].unsqueeze(-1) * ao[_HESS[i][j]] for i in range(3))      # (ngrid, ld)
        bs = b * s_j
        q[:, j] = w * (vrho * grho[j] + 2.0 * (bs.sum(1) + (c[j] * ugphi).sum(1)))
        per_ao[:, j] = ((ao[1 + j] * t1 + bs) * w.unsqueeze(-1)).sum(0)[:nao]
    g.index_add_(0, owner, q)
    g.index_add_(0, ao_atom, -2.0 * per_ao)
    pos = mol.atompos.to(dtype=torch.float64, device=dev).clone().requires_grad_(True)
    grid = get_predefined_grid(mol._grid_inp, mol.atomzs.tolist(), pos, dtype=torch.float64, device=dev)
    loss = (grid.get_dvolume() * edens.detach()).sum()
    return g + torch.autograd.grad(loss, pos)[0]


def _grid_owner(mol, ngrid, dev):
    """parent atom of every grid point: the atomic grids are concatenated atom by atom (dqc_amd.grid.get_grid)"""
    from .grid import get_predefined_grid as gpg
    sizes = []
    for z in mol.atomzs.tolist():
        one = gpg(mol._grid_inp, [z], torch.zeros((1, 3), dtype=torch.float64), dtype=torch.float64, device="cpu")
        sizes.append(one.get_rgrid().shape[0])
    assert sum(sizes) == ngrid
    return torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes)).to(dev)


def _ao_owner(h, dev):
    out = []
    for b in h._tab.bas:
        out.extend([int(b[0])] * (2 * int(b[1]) + 1))
    return torch.tensor(out, device=dev)

"""oracle/hamilton.py -- TEST INFRASTRUCTURE ONLY.

CPU (torch float64) restatement of the reference's Hamiltonian and SCF engines, with the
reference's own algorithm shape -- dense (nao^4) el_mat, the two einsum strings, separate
density / Vxc passes chunked at CHUNK_MEMORY -- used as the parity checker and as the
`cpu_baseline` leg of bench.py ("restatement of DQC's CPU algorithm, not an executed DQC").

  HamiltonCGTO.build / setup_grid          dqc/hamilton/hcgto.py:95-186
  get_elrep / get_exchange (returns -K/2)  dqc/hamilton/hcgto.py:204-241
  get_vxc, _dm2densinfo, _get_vxc_from_potinfo   hcgto.py:260-269, 371-495
  energies                                  hcgto.py:302-328
  OrbitalOrthogonalizer                     dqc/hamilton/orbconverter.py:67-116
  chunkify                                  dqc/utils/mem.py:6-38, config.CHUNK_MEMORY dqc/utils/config.py:10
  _HFEngine / _KSEngine dm2scp, scp2dm, dm2energy   dqc/qccalc/hf.py:93-247, ks.py:110-187
  SCF_QCCalc.run ("1e" guess, fixed point on the Fock matrix)   dqc/qccalc/scf_qccalc.py:84-116
  Mol.get_nuclei_energy, occupations        dqc/system/mol.py:252-260, 421-443
"""
import numpy as np
import torch

from . import natives, grid as ogrid, xc as oxc, basis as obasis

CHUNK_MEMORY = 16 * 1024 ** 2  # dqc/utils/config.py:10


def chunkify(a, dim, maxnumel):
    """dqc/utils/mem.py:6-38"""
    numel = a.numel()
    dimnumel = a.shape[dim]
    nondimnumel = numel // dimnumel
    if maxnumel < nondimnumel:
        raise RuntimeError("too small chunk")
    csize = min(maxnumel // nondimnumel, dimnumel)
    ioff = 0
    while ioff < dimnumel:
        iend = min(ioff + csize, dimnumel)
        yield a.narrow(dim, ioff, iend - ioff), ioff, iend
        ioff = iend


class Hamilton:
    """Restatement of HamiltonCGTO for isolated molecules (no DF, no efield/vext)."""

    def __init__(self, tables, orthozer=True, eri_mode="dense", df=None, efield=None):
        self.t = tables
        self.efield = efield  # tuple of flattened arrays (E,), (E, dE): hcgto.py:117-125
        self.eri_mode = eri_mode  # "dense": reference formulation; "s4": packed variant for big nao
        # df = (concatenated tables, orbital shell range, auxiliary shell range): density-fitted J (dqc/df/dfmol.py)
        self.df = df
        ovlp = torch.as_tensor(natives.int1e("ovlp", tables))
        if orthozer:
            ev, evec = torch.linalg.eigh(ovlp)
            acc = ev > 1e-6  # orbconverter.py:73
            self.X = evec[:, acc] * ev[acc] ** -0.5
        else:
            self.X = torch.eye(tables.nao, dtype=torch.float64)
        self.xc = None
        self.xcfamily = 1

    @property
    def nao(self):
        return self.X.shape[-1]

    # ---- orbconverter ----
    def convert2(self, m):
        return self.X.T @ m @ self.X

    def unconvert_dm(self, dm):
        return torch.einsum("kl,ik,jl->ij", dm, self.X, self.X)

    # ---- setups ----
    def build(self):
        t = self.t
        olp = torch.as_tensor(natives.int1e("ovlp", t))
        kin = torch.as_tensor(natives.int1e("kin", t))
        nuc = torch.as_tensor(natives.int1e("nuc", t, np.asarray(t.atomzs, dtype=np.float64)))  # fractional Z: molintor.py:105-112
        self.olp_mat = self.convert2(olp)
        if self.efield is not None:
            fac = 1.0
            for i, ef in enumerate(self.efield):
                fac *= i + 1
                mats = torch.as_tensor(natives.int1e("r0" * (i + 1), t))
                kin = kin + torch.einsum("dab,d->ab", mats, torch.as_tensor(np.asarray(ef, dtype=np.float64).reshape(-1))) / fac
        self.kinnucl_mat = self.convert2(kin + nuc)
        self.nucl_mat = self.convert2(nuc)
        if self.df is not None:  # DFMol.build, dfmol.py:24-58 (method "coulomb")
            tc, orb_range, aux_range = self.df
            self.j2c = torch.as_tensor(natives.int2c2e(tc, aux_range))            # (nxao, nxao)
            self.j3c = torch.as_tensor(natives.int3c2e(tc, orb_range, aux_range))  # (nao, nao, nxao)
            self.inv_j2c = torch.inverse(self.j2c)
            self.df_el_mat = torch.matmul(self.j3c, self.inv_j2c)                  # (nao, nao, nxao)
        elif self.eri_mode == "dense":
            el = torch.as_tensor(natives.int2e(t))
            X = self.X
            # orbconverter.convert4 (hcgto.py:132), done as four successive contractions
            el = torch.einsum("ijkl,im->mjkl", el, X)
            el = torch.einsum("mjkl,jn->mnkl", el, X)
            el = torch.einsum("mnkl,kp->mnpl", el, X)
            el = torch.einsum("mnpl,lq->mnpq", el, X)
            self.el_mat = el.contiguous()
        elif self.eri_mode == "s8":  # packed lower triangle of the s4 matrix (bases whose s4 matrix does not fit the host)
            self.el_s8 = natives.int2e_s8(t)
        else:
            self.el_s4 = torch.as_tensor(natives.int2e_s4(t))
        return self

    def setup_grid(self, rgrid, dvolume, xc):
        self.xc = xc
        self.xcfamily = 1 if xc is None else xc.family
        self.rgrid = rgrid
        self.dvolume = torch.as_tensor(dvolume)
        self.basis = torch.as_tensor(natives.eval_gto(self.t, rgrid, 0)).T.contiguous()  # (ngrid, nao)
        self.basis_dvolume = self.basis * self.dvolume.unsqueeze(-1)
        if self.xcfamily >= 2:
            self.grad_basis = torch.as_tensor(natives.eval_gto(self.t, rgrid, 1)).transpose(-2, -1).contiguous()
        if self.xcfamily == 4:  # hcgto.py:183-186
            self.lapl_basis = torch.as_tensor(natives.eval_gto(self.t, rgrid, 2)).T.contiguous()

    # ---- Fock components (all in the orthogonalised basis) ----
    def _pack_dm(self, dm_ao):
        n = dm_ao.shape[0]
        iu = torch.tril_indices(n, n)
        d = dm_ao + dm_ao.T
        d = d[iu[0], iu[1]]
        d[iu[0] == iu[1]] *= 0.5
        return d, iu

    def get_elrep(self, dm):
        if self.df is not None:  # DFMol.get_elrep, dfmol.py:60-79 (precomputed el_mat branch)
            dao = self.unconvert_dm(dm)
            df_coeffs = torch.einsum("ij,ijk->k", dao, self.df_el_mat)
            mat = torch.einsum("k,ijk->ij", df_coeffs, self.j3c)
            mat = (mat + mat.T) * 0.5
            return self.convert2(mat)
        if self.eri_mode == "dense":
            mat = torch.einsum("ij,ijkl->kl", dm, self.el_mat)
        else:  # packed-s4 variant (same numbers, different storage)
            dao = self.unconvert_dm(dm)
            d, iu = self._pack_dm(dao)
            jp = torch.as_tensor(natives.symv_s8(self.el_s8, d.numpy())) if self.eri_mode == "s8" else self.el_s4 @ d
            n = dao.shape[0]
            J = torch.zeros((n, n), dtype=dm.dtype)
            J[iu[0], iu[1]] = jp
            J = J + J.T - torch.diag(torch.diag(J))
            mat = self.convert2(J)
        return (mat + mat.T) * 0.5

    def get_exchange(self, dm):
        """returns -K/2 (hcgto.py:234)"""
        if self.df is not None:  # hcgto.py:229-230
            raise RuntimeError("Exact exchange cannot be computed with density fitting")
        assert self.eri_mode == "dense"
        mat = -0.5 * torch.einsum("il,ijkl->ijk", dm, self.el_mat).sum(dim=-3)
        return (mat + mat.T) * 0.5

    def dm2densinfo(self, dm):
        dmdmt = (dm + dm.T) * 0.5
        dmdmt = self.unconvert_dm(dmdmt)
        ngrid = self.basis.shape[0]
        dens = torch.empty(ngrid, dtype=torch.float64)
        gdens = torch.empty((3, ngrid), dtype=torch.float64) if self.xcfamily >= 2 else None
        maxnumel = CHUNK_MEMORY // 8
        for basis, ioff, iend in chunkify(self.basis, 0, maxnumel):
            dmao = basis @ dmdmt
            dens[ioff:iend] = torch.einsum("ri,ri->r", dmao, basis)
            if gdens is not None:
                for d in range(3):
                    gdens[d, ioff:iend] = torch.einsum("ri,ri->r", dmao, self.grad_basis[d, ioff:iend]) * 2
        return dens, gdens

    def dm2densinfo_mgga(self, dm):
        """value, grad, lapl, kin exactly as hcgto.py:420-438"""
        dmdmt = self.unconvert_dm((dm + dm.T) * 0.5)
        dens, gdens = self.dm2densinfo(dm)
        ngrid = self.basis.shape[0]
        lapl = torch.empty(ngrid, dtype=torch.float64)
        kin = torch.empty(ngrid, dtype=torch.float64)
        for basis, ioff, iend in chunkify(self.basis, 0, CHUNK_MEMORY // 8):
            dmao = basis @ dmdmt
            lapl_basis = torch.einsum("ri,ri->r", dmao, self.lapl_basis[ioff:iend])
            gg = 0
            for d in range(3):
                gb = self.grad_basis[d, ioff:iend]
                gg = gg + torch.einsum("ri,ri->r", gb @ dmdmt, gb)
            lapl[ioff:iend] = (lapl_basis + gg) * 2
            kin[ioff:iend] = gg * 0.5
        return dens, gdens, lapl, kin

    def vxc_from_potinfo_mgga(self, vrho, vgrad, vlapl, vkin):
        """hcgto.py:445-495 with the MGGA terms (:473-489)"""
        nao = self.basis.shape[-1]
        mat = torch.zeros((nao, nao), dtype=torch.float64)
        for basis, ioff, iend in chunkify(self.basis, 0, CHUNK_MEMORY // 8):
            vb = vrho[ioff:iend].unsqueeze(-1) * basis
            vg = vgrad[:, ioff:iend] * 2
            for d in range(3):
                vb += vg[d].unsqueeze(-1) * self.grad_basis[d, ioff:iend]
            vb += 2 * vlapl[ioff:iend].unsqueeze(-1) * self.lapl_basis[ioff:iend]
            mat += self.basis_dvolume[ioff:iend].T @ vb
            lk = (2 * vlapl[ioff:iend] + 0.5 * vkin[ioff:iend]) * self.dvolume[ioff:iend]
            for d in range(3):
                gb = self.grad_basis[d, ioff:iend]
                mat += gb.T @ (lk.unsqueeze(-1) * gb)
        mat = self.convert2(mat)
        return (mat + mat.T) * 0.5

    def vxc_from_potinfo(self, vrho, vgrad):
        nao = self.basis.shape[-1]
        mat = torch.zeros((nao, nao), dtype=torch.float64)
        maxnumel = CHUNK_MEMORY // 8
        for basis, ioff, iend in chunkify(self.basis, 0, maxnumel):
            vb = vrho[ioff:iend].unsqueeze(-1) * basis
            if vgrad is not None:
                vg = vgrad[:, ioff:iend] * 2
                for d in range(3):
                    vb += vg[d].unsqueeze(-1) * self.grad_basis[d, ioff:iend]
            mat += self.basis_dvolume[ioff:iend].T @ vb
        mat = self.convert2(mat)
        return (mat + mat.T) * 0.5

    def _mgga_eval(self, dm):
        dens, gdens, lapl, kin = self.dm2densinfo_mgga(dm)
        sig = torch.einsum("dr,dr->r", gdens, gdens)
        e, vr, vs, vt = self.xc.compute_mgga(dens.numpy(), sig.numpy(), kin.numpy())
        t = torch.as_tensor
        return t(e), t(vr), 2.0 * t(vs).unsqueeze(0) * gdens, torch.zeros_like(dens), t(vt)

    def get_vxc(self, dm):
        if self.xcfamily == 4:
            _, vr, vg, vl, vt = self._mgga_eval(dm)
            return self.vxc_from_potinfo_mgga(vr, vg, vl, vt)
        dens, gdens = self.dm2densinfo(dm)
        vr, vg = self.xc.get_vxc(dens.numpy(), None if gdens is None else gdens.numpy())
        return self.vxc_from_potinfo(torch.as_tensor(vr), None if vg is None else torch.as_tensor(vg))

    # ---- energies ----
    def get_e_hcore(self, dm):
        return torch.einsum("ij,ji->", self.kinnucl_mat, dm)

    def get_e_elrep(self, dm):
        return 0.5 * torch.einsum("ij,ji->", self.get_elrep(dm), dm)

    def get_e_exchange(self, dm):
        return 0.5 * torch.einsum("ij,ji->", self.get_exchange(dm), dm)

    def get_e_xc(self, dm):
        if self.xcfamily == 4:
            return torch.sum(self.dvolume * self._mgga_eval(dm)[0])
        dens, gdens = self.dm2densinfo(dm)
        e = self.xc.get_edensityxc(dens.numpy(), None if gdens is None else gdens.numpy())
        return torch.sum(self.dvolume * torch.as_tensor(e))

    def ao_orb2dm(self, orb, orb_weight):
        return orb @ (orb * orb_weight.unsqueeze(-2)).T


def nuclei_energy(zs, pos):
    """dqc/system/mol.py:252-260"""
    e = 0.0
    for i in range(len(zs)):
        for j in range(i):
            e += zs[i] * zs[j] / np.linalg.norm(pos[i] - pos[j])
    return float(e)


class Engine:
    """RHF (xc=None) / RKS engine: dm2scp, scp2dm, dm2energy as in hf.py / ks.py."""

    def __init__(self, tables, xc=None, grid="sg3", hf=None, eri_mode="dense", df=None, efield=None, spin=0):
        self.t = tables
        self.is_hf = (xc is None) if hf is None else hf
        self.xc = None if self.is_hf else (oxc.get_xc(xc) if isinstance(xc, str) else xc)
        self.h = Hamilton(tables, eri_mode=eri_mode, df=df, efield=efield).build()
        if not self.is_hf:
            rgrid, dvol = ogrid.get_predefined_grid(grid, tables.atomzs, tables.atompos)
            self.h.setup_grid(rgrid, dvol, self.xc)
        nel_f = float(np.sum(tables.atomzs))
        if abs(nel_f - round(nel_f)) > 1e-12 or isinstance(spin, float):
            # fractional mode (mol.py:402-443): n_dn = (n - spin) / 2 electrons per spin channel, the last orbital of each
            # channel partially occupied (safeops.occnumber: floor(a) ones, then a - floor(a))
            def occ(a, n=0):
                lo, hi = int(np.floor(a + 1e-12)), int(np.ceil(a - 1e-12))
                w = torch.zeros(max(hi, n), dtype=torch.float64)
                w[:lo] = 1.0
                if hi > lo:
                    w[hi - 1] = a - lo
                return w
            ndn_f = (nel_f - spin) * 0.5
            assert ndn_f >= 0
            wu = occ(ndn_f + spin)
            self.orb_weight = wu + occ(ndn_f, wu.numel())
            self.norb = wu.numel()
        else:
            nel = int(round(nel_f))
            assert (nel - spin) % 2 == 0, "spin inconsistent with the electron count"
            # restricted occupations [2, ..., 2, 1, ..., 1] (mol.py:421-443): closed shell for spin 0, the reference's
            # restricted open-shell treatment otherwise
            nup, ndn = (nel + spin) // 2, (nel - spin) // 2
            self.norb = nup
            self.orb_weight = torch.cat([torch.full((ndn,), 2.0, dtype=torch.float64), torch.ones(nup - ndn, dtype=torch.float64)])
        self.enuc = nuclei_energy(tables.atomzs, tables.atompos)

    def dm2scp(self, dm):
        F = self.h.kinnucl_mat + self.h.get_elrep(dm)
        if self.is_hf:
            return F + self.h.get_exchange(dm)
        return F + self.h.get_vxc(dm)

    def scp2dm(self, scp):
        fock = (scp + scp.T) * 0.5
        e, C = torch.linalg.eigh(fock)  # overlap is the identity in the orthogonalised basis
        return self.h.ao_orb2dm(C[:, :self.norb], self.orb_weight)

    def dm2energy(self, dm):
        e = self.h.get_e_hcore(dm) + self.h.get_e_elrep(dm)
        e = e + (self.h.get_e_exchange(dm) if self.is_hf else self.h.get_e_xc(dm))
        return float(e) + self.enuc

    def energy_parts(self, dm):
        p = {"e_core": float(self.h.get_e_hcore(dm)), "e_elrep": float(self.h.get_e_elrep(dm)),
             "e_nuc": self.enuc}
        if self.is_hf:
            p["e_exch"] = float(self.h.get_e_exchange(dm))
        else:
            p["e_xc"] = float(self.h.get_e_xc(dm))
        p["e_tot"] = sum(p.values())
        return p

    def run(self, maxiter=100, tol=1e-9, dm0=None):
        """SCF_QCCalc.run data flow: '1e' guess, then the fixed point F = dm2scp(scp2dm(F)).
        Mixer: Pulay DIIS on F_out - F_in (reference: Broyden-1; only the fixed point is compared)."""
        if dm0 is None:
            scp = self.dm2scp(torch.zeros((self.h.nao, self.h.nao), dtype=torch.float64))
            dm = self.scp2dm(scp)
        else:
            dm = dm0
        scp = self.dm2scp(dm)
        ys, rs = [], []
        self.niter = 0
        for it in range(maxiter):
            fy = self.dm2scp(self.scp2dm(scp))
            self.niter = it + 1
            r = fy - scp
            if r.abs().max() < tol:
                scp = fy
                break
            ys.append(fy.reshape(-1))
            rs.append(r.reshape(-1))
            if len(ys) > 10:
                ys.pop(0)
                rs.pop(0)
            n = len(ys)
            B = -torch.ones((n + 1, n + 1), dtype=torch.float64)
            B[n, n] = 0
            R = torch.stack(rs)
            B[:n, :n] = R @ R.T
            rhs = torch.zeros(n + 1, dtype=torch.float64)
            rhs[n] = -1
            c = torch.linalg.lstsq(B, rhs.unsqueeze(-1)).solution[:n, 0]
            scp = (c.unsqueeze(0) @ torch.stack(ys)).reshape(fy.shape)
        self.dm = self.scp2dm(scp)
        return self.dm2energy(self.dm)


def run_scf(moldesc, basis, xc=None, grid="sg3", auxbasis=None, efield=None, spin=0, **kw):
    t = obasis.make_tables(moldesc, basis)
    df = obasis.make_tables_df(moldesc, basis, auxbasis) if auxbasis is not None else None
    eng = Engine(t, xc=xc, grid=grid, df=df, efield=efield, spin=spin)
    e = eng.run(**kw)
    return e, eng


class EnginePol:
    """UHF / UKS engine (polarised branches of dqc/qccalc/hf.py:93-119,182-216 and ks.py:110-187):
    scp = stacked (F_u, F_d); J from the total density; exchange per spin = get_exchange(2 D_s) = -K[D_s]
    (hcgto.py:238-241); occupations from Mol (dqc/system/mol.py:421-443)."""

    def __init__(self, tables, spin, xc=None, grid="sg3", hf=None):
        self.t = tables
        self.is_hf = (xc is None) if hf is None else hf
        self.xc = None if self.is_hf else (oxc.get_xc(xc) if isinstance(xc, str) else xc)
        self.h = Hamilton(tables).build()
        if not self.is_hf:
            rgrid, dvol = ogrid.get_predefined_grid(grid, tables.atomzs, tables.atompos)
            self.h.setup_grid(rgrid, dvol, self.xc)
        nel = int(round(float(np.sum(tables.atomzs))))
        assert (nel - spin) % 2 == 0
        self.ndn = (nel - spin) // 2
        self.nup = self.ndn + spin
        self.enuc = nuclei_energy(tables.atomzs, tables.atompos)

    def _vxc(self, dmu, dmd):
        ru, gu = self.h.dm2densinfo(dmu)
        rd, gd = self.h.dm2densinfo(dmd)
        f = lambda a: None if a is None else a.numpy()  # noqa: E731
        e, vr, vg, _ = oxc.compute_pol(self.xc, ru.numpy(), rd.numpy(), f(gu), f(gd))
        t = lambda a: None if a is None else torch.as_tensor(a)  # noqa: E731
        vu = self.h.vxc_from_potinfo(torch.as_tensor(vr[0]), t(vg[0]))
        vd = self.h.vxc_from_potinfo(torch.as_tensor(vr[1]), t(vg[1]))
        exc = float(torch.sum(self.h.dvolume * torch.as_tensor(e)))
        return vu, vd, exc

    def dm2scp(self, dm):
        dmu, dmd = dm
        core = self.h.kinnucl_mat + self.h.get_elrep(dmu + dmd)
        if self.is_hf:
            return torch.stack([core + self.h.get_exchange(2 * dmu), core + self.h.get_exchange(2 * dmd)])
        vu, vd, _ = self._vxc(dmu, dmd)
        return torch.stack([core + vu, core + vd])

    def scp2dm(self, scp):
        out = []
        for f, nocc in ((scp[0], self.nup), (scp[1], self.ndn)):
            f = (f + f.T) * 0.5
            _, C = torch.linalg.eigh(f)
            out.append(C[:, :nocc] @ C[:, :nocc].T if nocc > 0 else torch.zeros_like(f))
        return tuple(out)

    def dm2energy(self, dm):
        dmu, dmd = dm
        tot = dmu + dmd
        e = float(self.h.get_e_hcore(tot) + self.h.get_e_elrep(tot))
        if self.is_hf:
            e += float(0.5 * torch.sum(self.h.get_exchange(2 * dmu) * dmu) + 0.5 * torch.sum(self.h.get_exchange(2 * dmd) * dmd))
        else:
            e += self._vxc(dmu, dmd)[2]
        return e + self.enuc

    def run(self, maxiter=200, tol=1e-9):
        n = self.h.nao
        z = torch.zeros((n, n), dtype=torch.float64)
        dm = self.scp2dm(self.dm2scp((z, z)))
        if self.nup == self.ndn:  # the reference halves a restricted guess (scf_qccalc.py:97-100)
            dm = ((dm[0] + dm[1]) * 0.5, (dm[0] + dm[1]) * 0.5)
        fs, es = [], []
        fock = self.dm2scp(dm)
        self.niter = 0
        for it in range(maxiter):
            self.niter = it + 1
            err = torch.stack([fock[s] @ dm[s] - dm[s] @ fock[s] for s in range(2)])
            if err.abs().max() < tol:
                break
            fs.append(fock)
            es.append(err.reshape(-1))
            if len(fs) > 8:
                fs.pop(0)
                es.pop(0)
            m = len(fs)
            if m > 1:
                E = torch.stack(es)
                B = torch.zeros((m + 1, m + 1), dtype=torch.float64)
                B[:m, :m] = E @ E.T
                B[m, :m] = -1
                B[:m, m] = -1
                rhs = torch.zeros(m + 1, dtype=torch.float64)
                rhs[m] = -1
                c = torch.linalg.lstsq(B, rhs.unsqueeze(-1)).solution[:m, 0]
                fmix = (c.reshape(-1, 1, 1, 1) * torch.stack(fs)).sum(0)
            else:
                fmix = fock
            dm = self.scp2dm(fmix)
            fock = self.dm2scp(dm)
        self.dm = dm
        return self.dm2energy(dm)


def run_scf_pol(moldesc, basis, spin, xc=None, grid="sg3", **kw):
    t = obasis.make_tables(moldesc, basis)
    eng = EnginePol(t, spin, xc=xc, grid=grid)
    return eng.run(**kw), eng


def nuclear_gradient_fd(moldesc, basis, xc=None, grid="sg3", h=1e-3, spin=None, **kw):
    """dE/dR by central finite differences of the (pinned) oracle SCF energy: by construction what the reference's
    autograd returns (its own gradient tests are gradcheck against finite differences, test_hf.py:82-111,
    test_ks.py:117-137).  Small molecules only: 6 natm SCF runs."""
    zs, pos = obasis.parse_moldesc(moldesc)
    pos = np.array(pos, dtype=np.float64)
    kw.setdefault("tol", 1e-11)
    g = np.zeros_like(pos)
    for a in range(len(zs)):
        for d in range(3):
            e = []
            for sgn in (+1, -1):
                p = pos.copy()
                p[a, d] += sgn * h
                if spin is None:
                    e.append(run_scf((list(zs), p.tolist()), basis, xc=xc, grid=grid, **kw)[0])
                else:  # unrestricted engines (hf.py:93-103, ks.py polarised branches)
                    e.append(run_scf_pol((list(zs), p.tolist()), basis, spin, xc=xc, grid=grid, **kw)[0])
            g[a, d] = (e[0] - e[1]) / (2 * h)
    return g

"""First-order electric properties of a converged SCF calculation.

The reference defines them as derivatives of the energy with respect to the electric field and its gradient, taken by
autograd through the SCF fixed point (dqc/api/properties.py:162-230, 439-485: dipole = -dE/dF + sum_A Z_A R_A, quadrupole =
-2 dE/dG + sum_A Z_A R_A R_A).  The field enters the core Hamiltonian as  sum_d r_d F_d + 1/2 sum_de r_d r_e G_de
(hcgto.py:117-125), the basis does not depend on it and the SCF energy is stationary in the density, so the derivatives are
the expectation values  dE/dF_d = Tr(D r_d),  dE/dG_de = 1/2 Tr(D r_d r_e)  (Hellmann-Feynman) -- no field needs to be
attached to the molecule and nothing is differentiated.  They double as the external (PySCF / CCCBDB literal) pin of the
multipole integrals r0 / r0r0 of the hot path's electric-field term.

The second half of the file (Hessian, vibrations, IR / Raman intensities, optimal_geometry, added late in round 2 by central
differences of the analytic gradient / dipole / polarisability) re-implements dqc/api/properties.py, which SURVEY.md 2 row 18
marks OUT OF SCOPE for this build: it is kept because its tests pin the analytic nuclear gradient (row f3) against
external literals, and is not developed further."""
import torch

from . import lib

# 1 atomic unit in other units (dqc/utils/units.py:14-17, 72-80)
_DEBYE = 2.541746473
_ANGSTROM = 5.29177210903e-11 / 1e-10
_DIPOLE_UNITS = {None: 1.0, "d": _DEBYE, "debye": _DEBYE}
_QUADRUPOLE_UNITS = {None: 1.0, "debye*angst": _DEBYE * _ANGSTROM}


def _unit(table, unit, what):
    key = unit.lower() if isinstance(unit, str) else unit
    if key not in table:
        raise ValueError("Unknown %s unit: %s (known: %s)" % (what, unit, [k for k in table if k]))
    return table[key]


def _total_ao_density(qc):
    h = qc.get_system().get_hamiltonian()
    dm = qc.aodm()
    if not isinstance(dm, torch.Tensor):  # SpinParam
        dm = dm.u + dm.d
    return h, h._unconvert_dm(dm)


def edipole(qc, unit="Debye"):
    """electric dipole moment (3,), pointing from negative to positive charge (properties.py:162-198)"""
    h, dao = _total_ao_density(qc)
    mol = qc.get_system()
    r = lib.int1e("r0", h._tab, h.device)  # (3, nao, nao)
    elec = -torch.einsum("dab,ba->d", r, dao)
    pos = mol.atompos.to(h.device)
    ion = torch.einsum("ad,a->d", pos, mol.atomzs.to(pos.dtype).to(h.device))
    return (elec + ion) * _unit(_DIPOLE_UNITS, unit, "dipole")


def equadrupole(qc, unit="Debye*Angst"):
    """electric quadrupole moment (3, 3): the raw second moment of the charge distribution (properties.py:200-230)"""
    h, dao = _total_ao_density(qc)
    mol = qc.get_system()
    rr = lib.int1e("r0r0", h._tab, h.device).reshape(3, 3, *dao.shape)
    elec = -torch.einsum("deab,ba->de", rr, dao)
    pos = mol.atompos.to(h.device)
    ion = torch.einsum("ad,ae,a->de", pos, pos, mol.atomzs.to(pos.dtype).to(h.device))
    return (elec + ion) * _unit(_QUADRUPOLE_UNITS, unit, "quadrupole")


def optimal_geometry(qc, length_unit=None, gtol=1e-5, maxiter=200):
    """atom positions (natoms, 3) that minimise the SCF energy (properties.py:321-341, 486-510).  The reference minimises by
    autograd gradients of the energy with respect to the positions; here every step is a fresh `Mol.make_copy(moldesc=...)` +
    SCF run and the ANALYTIC nuclear gradient of the converged density (`SCF_QCCalc.nuclear_gradient`), driven by BFGS."""
    import numpy as np
    from scipy.optimize import minimize
    system = qc.get_system()
    zs = system.atomzs
    x0 = system.atompos.detach().cpu().numpy().astype(np.float64)
    natm = x0.shape[0]
    cls = qc.__class__
    kwargs = dict(getattr(qc, "_ctor_kwargs", {}))

    def fun(x):
        pos = torch.as_tensor(x.reshape(natm, 3), dtype=torch.float64)
        q = cls(system.make_copy(moldesc=(zs, pos)), **kwargs).run()
        return float(q.energy()), q.nuclear_gradient().detach().cpu().numpy().reshape(-1)

    res = minimize(fun, x0.reshape(-1), jac=True, method="BFGS", options={"gtol": gtol, "maxiter": maxiter})
    pos = torch.as_tensor(res.x.reshape(natm, 3), dtype=torch.float64)
    if length_unit is not None:
        key = length_unit.lower()
        table = {"bohr": 1.0, "angst": _ANGSTROM, "angstrom": _ANGSTROM, "a": _ANGSTROM}
        if key not in table:
            raise ValueError("Unknown length unit: %s" % length_unit)
        pos = pos * table[key]
    return pos


# ---- second-order properties by central differences of FIRST-order analytic quantities -------------------------------------
# The reference differentiates twice through the SCF (properties.py:344-437).  Here the Hessian is the central difference of the
# analytic nuclear gradient and d(dipole)/dR the central difference of the dipole expectation value, over the same 6 natoms
# displaced SCF runs (milliseconds each on the GPU for the molecules these properties are asked for).
_TIME = 2.4188843265857e-17   # s per a.u. of time
_LIGHT_SPEED = 2.99792458e8   # m / s
_AMU = 5.485799090649e-4      # a.m.u. per a.u. of mass
_FREQ_UNITS = {None: 1.0, "cm-1": 1e-2 / _TIME / _LIGHT_SPEED, "cm^-1": 1e-2 / _TIME / _LIGHT_SPEED, "hz": 1.0 / _TIME,
               "thz": 1.0 / _TIME / 1e12}
_IR_UNITS = {None: 1.0, "(debye/angst)^2/amu": (_DEBYE / _ANGSTROM) ** 2 / _AMU,
             "km/mol": (_DEBYE / _ANGSTROM) ** 2 / _AMU * 42.256}


def _displaced(qc, step):
    """gradient (3N,) and dipole (3,) [a.u.] of the 6 N geometries displaced by +-step along every Cartesian coordinate"""
    system = qc.get_system()
    pos0 = system.atompos.detach().cpu().to(torch.float64)
    cls, kwargs = qc.__class__, dict(getattr(qc, "_ctor_kwargs", {}))
    out = {}
    for i in range(pos0.numel()):
        for sgn in (1.0, -1.0):
            pos = pos0.clone().reshape(-1)
            pos[i] += sgn * step
            q = cls(system.make_copy(moldesc=(system.atomzs, pos.reshape(pos0.shape))), **kwargs).run()
            out[(i, sgn)] = (q.nuclear_gradient().detach().cpu().reshape(-1), edipole(q, unit=None).cpu())
    return out


_FD_CACHE = []  # [weakref(qc), step, (Hessian, d dipole / dR)]: one set of displaced runs serves Hessian, vibrations and IR


def _second_order(qc, step):
    import weakref
    if not (_FD_CACHE and _FD_CACHE[0]() is qc and _FD_CACHE[1] == step):
        d = _displaced(qc, step)
        n = len(d) // 2
        hess = torch.stack([(d[(i, 1.0)][0] - d[(i, -1.0)][0]) / (2 * step) for i in range(n)])
        dmu = torch.stack([(d[(i, 1.0)][1] - d[(i, -1.0)][1]) / (2 * step) for i in range(n)], dim=-1)  # (3, 3N)
        _FD_CACHE[:] = [weakref.ref(qc), step, ((hess + hess.T) * 0.5, dmu)]
    return _FD_CACHE[2]


def hessian_pos(qc, unit=None, step=5e-3):
    """d2E / dR dR (3 natoms, 3 natoms) in Hartree / Bohr^2 (properties.py:21-41)"""
    if unit is not None:
        raise ValueError("hessian_pos: only atomic units are provided")
    return _second_order(qc, step)[0]


def _vibration_au(qc, step):
    hess, dmu = _second_order(qc, step)
    mass = qc.get_system().atommasses.cpu().repeat_interleave(3)
    isq = mass ** -0.5
    ev, u = torch.linalg.eigh(hess * isq.unsqueeze(0) * isq.unsqueeze(1))
    modes = u * isq.unsqueeze(1)  # M-orthonormal: modes^T M modes = 1, as xitorch.symeig(A, M) returns them
    freq = ev.abs().sqrt() * torch.sign(ev) / (2 * torch.pi)
    return torch.flip(freq, dims=(-1,)), torch.flip(modes, dims=(-1,)), dmu


def vibration(qc, freq_unit="cm^-1", length_unit=None, step=5e-3):
    """vibrational frequencies (largest first; imaginary ones negative) and normal modes (properties.py:43-71, 359-381)"""
    freq, modes, _ = _vibration_au(qc, step)
    if length_unit is not None:
        raise ValueError("vibration: normal modes are returned in atomic units only")
    return freq * _unit(_FREQ_UNITS, freq_unit, "frequency"), modes


def ir_spectrum(qc, freq_unit="cm^-1", ints_unit="(debye/angst)^2/amu", step=5e-3):
    """positive vibrational frequencies (largest first) and their IR intensities |d mu / d Q|^2 (properties.py:73-109, 383-404)"""
    freq, modes, dmu = _vibration_au(qc, step)
    keep = freq > 0
    freq, modes = freq[keep], modes[:, keep]
    dmu_dq = dmu @ modes
    ints = (dmu_dq * dmu_dq).sum(0)
    return freq * _unit(_FREQ_UNITS, freq_unit, "frequency"), ints * _unit(_IR_UNITS, ints_unit, "IR intensity")


_RAMAN_UNITS = {None: 1.0, "angst^4/amu": _ANGSTROM ** 4 / _AMU}


def _polarizability(qc_at, field_step):
    """static polarisability alpha = d mu / dF (3, 3) by central differences of the dipole expectation value in a small field"""
    system = qc_at.get_system()
    cls, kwargs = qc_at.__class__, dict(getattr(qc_at, "_ctor_kwargs", {}))
    cols = []
    for d in range(3):
        mus = []
        for sgn in (1.0, -1.0):
            f = torch.zeros(3, dtype=torch.float64)
            f[d] = sgn * field_step
            q = cls(system.make_copy(efield=f), **kwargs).run()
            mus.append(edipole(q, unit=None).cpu())
        cols.append((mus[0] - mus[1]) / (2 * field_step))
    return torch.stack(cols, dim=-1)  # alpha[i, d] = d mu_i / d F_d


def raman_spectrum(qc, freq_unit="cm^-1", ints_unit="angst^4/amu", step=5e-3, field_step=2e-3):
    """positive frequencies (largest first) and Raman activities 45 a'^2 + 7 g'^2 (properties.py:111-160, 406-437; eq. 2-4 of
    doi:10.1080/00268970701516412) from the derivative of the polarisability along the normal modes"""
    freq, modes, _ = _vibration_au(qc, step)
    keep = freq > 0
    freq, modes = freq[keep], modes[:, keep]
    system = qc.get_system()
    pos0 = system.atompos.detach().cpu().to(torch.float64)
    cls, kwargs = qc.__class__, dict(getattr(qc, "_ctor_kwargs", {}))
    dalpha = []
    for i in range(pos0.numel()):
        a = []
        for sgn in (1.0, -1.0):
            pos = pos0.clone().reshape(-1)
            pos[i] += sgn * step
            q = cls(system.make_copy(moldesc=(system.atomzs, pos.reshape(pos0.shape))), **kwargs).run()
            a.append(_polarizability(q, field_step))
        dalpha.append((a[0] - a[1]) / (2 * step))
    dalpha_dr = torch.stack(dalpha, dim=-1)        # (3, 3, 3N)
    dq = torch.matmul(dalpha_dr, modes)            # (3, 3, nmodes)
    alpha_p2 = (torch.einsum("iim->m", dq) / 3.0) ** 2
    gamma_p2 = 0.5 * ((dq[0, 0] - dq[1, 1]) ** 2 + (dq[0, 0] - dq[2, 2]) ** 2 + (dq[1, 1] - dq[2, 2]) ** 2
                      + 3 * (dq[0, 1] ** 2 + dq[0, 2] ** 2 + dq[1, 0] ** 2 + dq[1, 2] ** 2 + dq[2, 0] ** 2 + dq[2, 1] ** 2))
    ints = 45 * alpha_p2 + 7 * gamma_p2
    return freq * _unit(_FREQ_UNITS, freq_unit, "frequency"), ints * _unit(_RAMAN_UNITS, ints_unit, "Raman activity")

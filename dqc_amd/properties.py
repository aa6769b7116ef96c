"""First-order electric properties of a converged SCF calculation.

The reference defines them as derivatives of the energy with respect to the electric field and its gradient, taken by
autograd through the SCF fixed point (dqc/api/properties.py:162-230, 439-485: dipole = -dE/dF + sum_A Z_A R_A, quadrupole =
-2 dE/dG + sum_A Z_A R_A R_A).  The field enters the core Hamiltonian as  sum_d r_d F_d + 1/2 sum_de r_d r_e G_de
(hcgto.py:117-125), the basis does not depend on it and the SCF energy is stationary in the density, so the derivatives are
the expectation values  dE/dF_d = Tr(D r_d),  dE/dG_de = 1/2 Tr(D r_d r_e)  (Hellmann-Feynman) -- no field needs to be
attached to the molecule and nothing is differentiated.  Higher-order properties (Hessians, IR / Raman) are out of scope."""
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

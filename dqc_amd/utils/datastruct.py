"""Data structures of the Hamiltonian API, same names and meaning as the reference's
dqc/utils/datastruct.py (CGTOBasis :29-61, AtomCGTOBasis :63-67, SpinParam :78-137, ValGrad :139-184)."""
from dataclasses import dataclass
from math import gamma
from typing import Generic, List, Optional, TypeVar, Union

import torch

T = TypeVar("T")
ZType = Union[int, float, torch.Tensor]


def gaussian_int(n, alpha):
    """int_0^inf x^n exp(-alpha x^2) dx (reference dqc/utils/misc.py:53-56)"""
    n1 = (n + 1) * 0.5
    return gamma(n1) / (2 * alpha ** n1)


@dataclass
class CGTOBasis:
    angmom: int
    alphas: torch.Tensor  # (nbasis,)
    coeffs: torch.Tensor  # (nbasis,)
    normalized: bool = False

    def wfnormalize_(self):
        """radial normalisation + unit self-overlap of the contraction (datastruct.py:34-61)"""
        if self.normalized:
            return self
        l = self.angmom
        coeffs = self.coeffs / torch.sqrt(gaussian_int(2 * l + 2, 2 * self.alphas))
        ee = gaussian_int(2 * l + 2, self.alphas.unsqueeze(-1) + self.alphas.unsqueeze(-2))
        s1 = 1 / torch.sqrt(torch.einsum("a,ab,b", coeffs, ee, coeffs))
        self.coeffs = coeffs * s1
        self.normalized = True
        return self


@dataclass
class AtomCGTOBasis:
    atomz: ZType
    bases: List[CGTOBasis]
    pos: torch.Tensor  # (3,)


@dataclass
class DensityFitInfo:
    """dqc/utils/datastruct.py:73-76"""
    method: str
    auxbases: List[AtomCGTOBasis]


@dataclass
class SpinParam(Generic[T]):
    u: T
    d: T

    def sum(a):
        if isinstance(a, SpinParam):
            return a.u + a.d
        return a

    def reduce(a, fcn):
        if isinstance(a, SpinParam):
            return fcn(a.u, a.d)
        return a

    @staticmethod
    def apply_fcn(fcn, *a):
        if isinstance(a[0], SpinParam):
            return SpinParam(u=fcn(*(x.u for x in a)), d=fcn(*(x.d for x in a)))
        return fcn(*a)


@dataclass
class ValGrad:
    value: torch.Tensor                 # (*BD, ngrid)
    grad: Optional[torch.Tensor] = None  # (*BD, 3, ngrid)
    lapl: Optional[torch.Tensor] = None
    kin: Optional[torch.Tensor] = None

    # all four fields, None = absent = zero (datastruct.py:151-174 of the reference)
    def __add__(self, b):
        def add(x, y):
            return y if x is None else (x if y is None else x + y)
        return ValGrad(value=add(self.value, b.value), grad=add(self.grad, b.grad), lapl=add(self.lapl, b.lapl),
                       kin=add(self.kin, b.kin))

    def __mul__(self, f):
        def mul(x):
            return None if x is None else x * f
        return ValGrad(value=mul(self.value), grad=mul(self.grad), lapl=mul(self.lapl), kin=mul(self.kin))

    __rmul__ = __mul__

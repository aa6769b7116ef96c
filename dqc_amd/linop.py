"""Dense LinearOperator shim with the part of xitorch.LinearOperator's surface the reference's SCF engines
use on Hamiltonian results: LinearOperator.m(mat, is_hermitian), .fullmatrix(), '+', .shape/.dtype/.device
(call sites: dqc/hamilton/hcgto.py:211, dqc/qccalc/hf.py:77-87, 97, 184; ks.py:178-182).
When the real xitorch is importable its class is used instead, so results drop into dqc.qccalc unchanged."""
try:  # pragma: no cover - xitorch is absent in the build image
    from xitorch import LinearOperator  # type: ignore
except Exception:
    class LinearOperator:
        def __init__(self, mat, is_hermitian=False):
            self._mat = mat
            self.is_hermitian = bool(is_hermitian)

        @staticmethod
        def m(mat, is_hermitian=None):
            return LinearOperator(mat, bool(is_hermitian))

        @property
        def shape(self):
            return self._mat.shape

        @property
        def dtype(self):
            return self._mat.dtype

        @property
        def device(self):
            return self._mat.device

        def fullmatrix(self):
            return self._mat

        def mm(self, x):
            return self._mat @ x

        def mv(self, x):
            return (self._mat @ x.unsqueeze(-1)).squeeze(-1)

        def __add__(self, o):
            return LinearOperator(self._mat + o._mat, self.is_hermitian and o.is_hermitian)

        def __sub__(self, o):
            return LinearOperator(self._mat - o._mat, self.is_hermitian and o.is_hermitian)

        def __mul__(self, f):
            return LinearOperator(self._mat * f, self.is_hermitian)

        __rmul__ = __mul__

        def __neg__(self):
            return LinearOperator(-self._mat, self.is_hermitian)

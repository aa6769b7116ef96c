"""dqc_amd -- MI355X (gfx950) native SCF Fock-build engine behind DQC's Hamiltonian API.

Public names mirror the reference's top-level API (dqc/__init__.py:1-3)."""
from .utils.datastruct import CGTOBasis, AtomCGTOBasis, SpinParam, ValGrad, DensityFitInfo  # noqa: F401
from .basis import loadbasis, parse_moldesc  # noqa: F401
from .xc import get_xc, BaseXC, LibXC  # noqa: F401
from .hamilton import HamiltonMI355  # noqa: F401
from .system import Mol  # noqa: F401
from .qccalc import HF, KS  # noqa: F401
from .grid import get_grid, get_predefined_grid  # noqa: F401
from .properties import edipole, equadrupole, optimal_geometry, hessian_pos, vibration, ir_spectrum, raman_spectrum  # noqa: F401

__version__ = "0.1.0"

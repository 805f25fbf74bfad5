"""MI355X-native modal sound engine (openpbso hot path).

The product is the C-ABI shared library openpbso_amd/libopenpbso_amd.so
(include/openpbso_amd.h) built from openpbso_amd/csrc/ for gfx950.  This
package is the thin Python host side used by tests, bench.py and the headless
harness: ctypes bindings (capi) and a mirror of the reference's ModalSolver
interface (solver).  There is no CPU fallback: importing capi fails loudly if
the HIP library is missing.
"""
from . import capi  # noqa: F401
from .solver import Engine, ModalSolver, ForceMessage  # noqa: F401
from .solver import POINT_FORCE, GAUSSIAN_FORCE, AUTOREGRESSIVE_FORCE  # noqa: F401

FRAMES_PER_BUFFER = 513
SAMPLE_RATE = 44100

"""lpdnet_hip: Python face of liblpd_hip.so -- the gfx950 kernels of the LPD-Net hot path."""
from ._lib import LIB_PATH, LpdHipError, load  # noqa: F401
from . import ops  # noqa: F401

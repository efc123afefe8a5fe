"""MI355X-native doubly-stochastic DGP hot path behind the reference's Python surface
(`from doubly_stochastic_dgp.dgp import DGP`).  Compute lives in csrc/libdsdgp.so (HIP, gfx950)."""
from . import settings  # noqa: F401

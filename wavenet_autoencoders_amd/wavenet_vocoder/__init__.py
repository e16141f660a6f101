"""Drop-in replacement of the reference's ``wavenet_vocoder`` package (hot-path classes only)."""
from .wavenet import WaveNet, receptive_field_size  # noqa: F401
from . import modules  # noqa: F401
from .modules import ResidualConv1dGLU  # noqa: F401

__version__ = "0.2.0+mi355x"

"""wavenet_autoencoders_amd: MI355X-native hot path for WaveNet autoencoders (VQ-WAE / IN-WAE).

Host code in Python over PyTorch-ROCm (memory, streams, torch.distributed); all arithmetic in
hand-written gfx950 HIP kernels behind the C ABI of include/wae.h (libwae_hip.so)."""
from .packing import Geometry, ParamLayout  # noqa: F401

__version__ = "0.1.0"

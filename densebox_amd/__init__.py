"""densebox_amd -- MI355X-native DenseBox hot path.

Host-side mirror of the reference's Python surface (DenseBox.py); all compute
runs in hand-written HIP kernels behind the C ABI of libdensebox_hip.so.
"""
from .nets import DenseBox, DenseBoxLM, DenseBoxLMLOC  # noqa: F401

__all__ = ['DenseBox', 'DenseBoxLM', 'DenseBoxLMLOC']

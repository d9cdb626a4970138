"""gpuart_amd — MI355X-native device back end + host library for gpuart's per-pixel path tracer.

The product is native: `lib/libgpuart_hip.so` (hand-written HIP kernels for gfx950 behind the
C ABI of include/gpuart_hip.h) and `lib/libgpuart.so` (C++ Renderer/Scene API of the reference).
This Python package is plumbing only (ctypes bindings for tests and bench.py). There is no CPU
fallback: if the native libraries are missing, importing `gpuart_amd.binding` objects fails loudly.
"""
from . import synth_scenes  # noqa: F401

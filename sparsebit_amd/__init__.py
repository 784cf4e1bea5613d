"""sparsebit_amd -- MI355X-native fake-quantization hot path of Sparsebit.

Quantizer / Observer / unstructured-mask forward path as hand-written HIP
kernels for gfx950 behind a C ABI (include/sbq.h, libsbq.so), with a Python
host layer that mirrors the reference's plugin interface
(sparsebit/quantization/quantizers, observers; sparsebit/sparse/sparsers).
"""
__version__ = "0.1.0"

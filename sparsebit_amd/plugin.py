"""Install the MI355X path into an importable reference Sparsebit -- zero edits to it.

    import sparsebit, sparsebit_amd.plugin
    sparsebit_amd.plugin.install()

Later registrations overwrite the reference's map entries (quantizers/__init__.py:4-6,
observers/__init__.py:4-6, sparse/sparsers/__init__.py:4-6), so after install()
`QuantModel`, `QuantOpr.build_quantizer`, BN fusion and QDQ-ONNX export run
unmodified on top of the HIP kernels:
  * QUANTIZERS_MAP["uniform"], ["lsq"], ["lsq+"], ["pact"], ["dorefa"] -> sparsebit_amd.quantizers
  * OBSERVERS_MAP["minmax"], ["mse"], ["percentile"], ["moving_average"], ["aciq"] -> sparsebit_amd.observers
  * SPARSERS_MAP["l1norm"]                        -> sparsebit_amd.sparsers
  * quant_tensor.fake_quant_kernel                -> sparsebit_amd.fake_quant (for the
    reference quantizers that stay, e.g. PACT / DoReFa / LSQ+, which call STE.apply)
See INTEGRATION.md.
"""


def install(native_only=False):
    import sparsebit.quantization.quantizers as ref_q
    import sparsebit.quantization.observers as ref_o
    import sparsebit.quantization.quantizers.quant_tensor as ref_qt

    from . import fake_quant
    from . import observers as amd_o
    from . import quantizers as amd_q

    ref_qt.fake_quant_kernel = fake_quant
    installed = {"fake_quant_kernel": True, "quantizers": [], "observers": [], "sparsers": []}
    if native_only:
        return installed
    for name in ("uniform", "lsq", "lsq+", "pact", "dorefa"):
        ref_q.QUANTIZERS_MAP[name] = amd_q.QUANTIZERS_MAP[name]
        installed["quantizers"].append(name)
    for name in ("minmax", "mse", "percentile", "moving_average", "aciq"):
        ref_o.OBSERVERS_MAP[name] = amd_o.OBSERVERS_MAP[name]
        installed["observers"].append(name)
    try:
        import sparsebit.sparse.sparsers as ref_s

        from . import sparsers as amd_s

        ref_s.SPARSERS_MAP["l1norm"] = amd_s.SPARSERS_MAP["l1norm"]
        installed["sparsers"].append("l1norm")
    except ImportError:
        pass
    return installed

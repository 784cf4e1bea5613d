"""Install the MI355X path into an importable reference Sparsebit -- zero edits to it.

    import sparsebit_amd.plugin
    sparsebit_amd.plugin.preinstall()    # GPU boxes: answers the reference's import-time JIT build (see below)
    import sparsebit
    sparsebit_amd.plugin.install()

Later registrations overwrite the reference's map entries (quantizers/__init__.py:4-6,
observers/__init__.py:4-6, sparse/sparsers/__init__.py:4-6), so after install()
`QuantModel`, `QuantOpr.build_quantizer`, BN fusion, `CalibrationRunner` and QDQ-ONNX export run
unmodified on top of the HIP kernels:
  * QUANTIZERS_MAP["uniform"], ["lsq"], ["lsq+"], ["pact"], ["dorefa"] -> classes derived from
    (sparsebit_amd quantizer, the reference's `Quantizer` base)
  * OBSERVERS_MAP["minmax"], ["mse"], ["percentile"], ["moving_average"], ["aciq"] -> classes derived
    from (sparsebit_amd observer, the reference's `Observer` base)
  * SPARSERS_MAP["l1norm"]                        -> (sparsebit_amd sparser, reference `Sparser` base);
    `SparseModel.calc_params` keeps its per-layer loop (sparse/sparse_model.py:107-113) and finds the L1 thresholds of
    all unstructured layers already computed, by ONE grouped selection in front of it (`route_sparse_params`)
  * quant_tensor.fake_quant_kernel                -> sparsebit_amd.fake_quant (for the
    reference quantizers that stay, e.g. adaround / quadapter, which call STE.apply)

What makes this a drop-in rather than a registry swap (each point is exercised by
tests/test_plugin_reference.py against the real reference and by tests/test_gpu_plugin.py against a
reference-shaped harness on the GPU box):
  * the reference hands quantizers ITS enum members -- `TARGET = (QuantTarget.FEATURE,)`,
    `set_backend(Backend.VIRTUAL)` (modules/base.py:36-45, common.py:5-35) -- and
    sparsebit_amd.common's enums compare and hash by name, so `fake_quant_factory[backend]`,
    `qdesc.target == QuantTarget.FEATURE` etc. hold for either class;
  * `QuantModel.export_onnx` finds quantizers with `isinstance(m, Quantizer)` against the reference
    base class (quant_model.py:236,256; :284 also `Observer`); the installed classes ARE subclasses
    of those bases (every method the reference calls is defined on the sparsebit_amd side of the MRO,
    which comes first), so `enable_export_onnx()` is reached and the export branch stays on
    `torch.fake_quantize_*`;
  * `calibrate="device"`: `QuantModel.calc_qparams` (quant_model.py:191-199) is routed through
    `sparsebit_amd.calibration.DeviceCalibrator` -- one device-resident pass per batch instead of the
    fx walk with a host round trip per node per batch (tools/calibration.py:66-160).
See INTEGRATION.md.
"""

_QUANTIZERS = ("uniform", "lsq", "lsq+", "pact", "dorefa")
_OBSERVERS = ("minmax", "mse", "percentile", "moving_average", "aciq")


def _derive(amd_cls, ref_base):
    """type(amd_cls.__name__, (amd_cls, ref_base)): isinstance(obj, ref_base) holds, behaviour is amd_cls's."""
    if issubclass(amd_cls, ref_base):
        return amd_cls
    cls = type(amd_cls.__name__, (amd_cls, ref_base), {"__module__": amd_cls.__module__, "__doc__": amd_cls.__doc__,
                                                         "_sbq_impl": amd_cls})
    cls.__qualname__ = amd_cls.__qualname__
    return cls


def preinstall():
    """Call BEFORE `import sparsebit` on a box with a visible GPU.

    The reference compiles its CUDA extension while it is being imported (quant_tensor.py:7-22:
    `torch.utils.cpp_extension.load(name="fake_quant", sources=[export.cc, fake_quant_tensor.cu], ...)` as soon as
    `torch.cuda.is_available()`), into its own package directory.  On ROCm that call hipifies the CUDA sources and
    fails on `#include <cuda.h>` (common.cuh:9) -- measured on the MI355X box -- so the reference cannot even be
    imported there.  preinstall() answers exactly that one `load` call with the prebuilt HIP module
    (`sparsebit_amd.fake_quant`, the same four functions: export.cc:3-8): nothing is compiled at import time, the
    reference's sources stay untouched, and every other `load` call goes to torch unchanged.  Returns a function that
    restores torch's loader."""
    import torch.utils.cpp_extension as ext

    from . import fake_quant

    real = ext.load
    if getattr(real, "_sbq_wrapped", False):
        return lambda: None

    def load(name, *args, **kw):
        if name == "fake_quant":
            return fake_quant
        return real(name, *args, **kw)

    load._sbq_wrapped = True
    ext.load = load

    def restore():
        ext.load = real

    return restore


def install(native_only=False, calibrate=None):
    """native_only: swap only the native `fake_quant` module (reference classes stay).
    calibrate="device": also route QuantModel.calc_qparams through DeviceCalibrator (float-input protocol;
    the reference's asym=True mode computes the same qparams, see calibration.py)."""
    import sparsebit.quantization.quantizers as ref_q
    import sparsebit.quantization.observers as ref_o
    import sparsebit.quantization.quantizers.quant_tensor as ref_qt

    from . import fake_quant
    from . import observers as amd_o
    from . import quantizers as amd_q

    ref_qt.fake_quant_kernel = fake_quant
    installed = {"fake_quant_kernel": True, "quantizers": [], "observers": [], "sparsers": [], "calibrate": None}
    if native_only:
        return installed
    for name in _QUANTIZERS:
        ref_q.QUANTIZERS_MAP[name] = _derive(amd_q.QUANTIZERS_MAP[name], ref_q.Quantizer)
        installed["quantizers"].append(name)
    for name in _OBSERVERS:
        ref_o.OBSERVERS_MAP[name] = _derive(amd_o.OBSERVERS_MAP[name], ref_o.Observer)
        installed["observers"].append(name)
    # quantizers build their observer through sparsebit_amd's own registry: hand out the derived classes there
    # too, so that isinstance(q.observer, reference Observer) holds (quant_model.py:284)
    for name in _OBSERVERS:
        amd_o.OBSERVERS_MAP[name] = ref_o.OBSERVERS_MAP[name]
    try:
        import sparsebit.sparse.sparsers as ref_s

        from . import sparsers as amd_s

        ref_s.SPARSERS_MAP["l1norm"] = _derive(amd_s.SPARSERS_MAP["l1norm"], ref_s.Sparser)
        installed["sparsers"].append("l1norm")
        # SparseModel.calc_params (sparse/sparse_model.py:107-113) keeps its per-layer loop; the thresholds of all
        # unstructured layers are computed in ONE launch in front of it (round 6)
        import sparsebit.sparse.sparse_model as ref_sm

        route_sparse_params(ref_sm.SparseModel)
        installed["sparsers"].append("calc_params:model-wide")
    except ImportError:
        pass
    if calibrate == "device":
        _route_calibration()
        installed["calibrate"] = "device"
    elif calibrate is not None:
        raise ValueError("calibrate must be None or 'device', not {!r}".format(calibrate))
    return installed


def route_sparse_params(sparse_model_cls):
    """Wrap `sparse_model_cls.calc_params` (the reference's SparseModel, or a look-alike with the same loop): before
    the original walks the graph calling every SparseOpr's `calc_mask` (each of which asks its sparser for
    `calc_mask(weight)`: sparse/modules/conv.py:28-29), the L1 thresholds of all unstructured layers are computed by one
    grouped selection (`sparsers.l1norm.premasks`) and handed out inside the loop.  Masks are those of the per-layer
    calls, element for element; idempotent."""
    import torch

    from .sparsers import l1norm

    orig = sparse_model_cls.calc_params
    if getattr(orig, "_sbq_grouped", False):
        return

    def calc_params(self):
        pairs = []
        for m in self.model.modules():
            sp, w = getattr(m, "sparser", None), getattr(m, "weight", None)
            if sp is not None and isinstance(w, torch.Tensor):
                pairs.append((sp, w))
        with l1norm.premasks(pairs):
            return orig(self)

    calc_params._sbq_grouped = True
    sparse_model_cls.calc_params = calc_params


def uninstall_observer_aliases():
    """Undo the one change install() makes to sparsebit_amd's OWN registry (tests that install in-process)."""
    from . import observers as amd_o

    for name in _OBSERVERS:
        cls = amd_o.OBSERVERS_MAP[name]
        amd_o.OBSERVERS_MAP[name] = getattr(cls, "_sbq_impl", cls)


def _route_calibration():
    """QuantModel.prepare_calibration / calc_qparams (quant_model.py:181-199) on the DeviceCalibrator:
    prepare installs forward-pre hooks, the user's calibration forwards feed the observers on the device,
    calc_qparams finishes them.  Same call sequence as with the reference's CalibrationRunner."""
    import sparsebit.quantization.quant_model as ref_qm

    from .calibration import DeviceCalibrator

    def prepare_calibration(self):
        self.eval()
        self.calibration_runner = DeviceCalibrator(self.model)
        self.calibration_runner.prepare_calibration()

    def calc_qparams(self, asym=False, w_quant=False, a_quant=False):
        assert hasattr(self, "calibration_runner"), "run self.prepare_calibration first"
        self.calibration_runner.layerwise_calibration(self.device, asym, w_quant, a_quant)
        del self.calibration_runner

    ref_qm.QuantModel.prepare_calibration = prepare_calibration
    ref_qm.QuantModel.calc_qparams = calc_qparams

"""Observer plug-in point: `register_observer`, `OBSERVERS_MAP`, `build_observer(config, qdesc)` --
the names and behaviour of sparsebit/quantization/observers/__init__.py:1-15."""
from ..registry import Registry

OBSERVERS_MAP = Registry("observer", "TYPE")
register_observer = OBSERVERS_MAP.register

from .base import DataCache, Observer  # noqa: E402
from . import minmax, percentile, mse, moving_average, aciq  # noqa: E402,F401


def build_observer(config, qdesc):
    return OBSERVERS_MAP.resolve(config.OBSERVER.TYPE)(config, qdesc)

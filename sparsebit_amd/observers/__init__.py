"""Observer registry -- same contract as sparsebit/quantization/observers/__init__.py:1-15."""
OBSERVERS_MAP = {}


def register_observer(observer):
    OBSERVERS_MAP[observer.TYPE.lower()] = observer
    return observer


from .base import DataCache, Observer  # noqa: E402
from . import minmax, percentile, mse, moving_average, aciq  # noqa: E402,F401


def build_observer(config, qdesc):
    observer = OBSERVERS_MAP[config.OBSERVER.TYPE.lower()](config, qdesc)
    return observer

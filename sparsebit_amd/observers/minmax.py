"""minmax observer (mirrors sparsebit/quantization/observers/minmax.py:14-25)."""
from . import Observer as BaseObserver
from . import register_observer


@register_observer
class Observer(BaseObserver):
    TYPE = "minmax"

    def __init__(self, config, qdesc):
        super(Observer, self).__init__(config, qdesc)

    def calc_minmax(self):
        shards = self._shards()
        self.data_cache.reset()
        mn, mx = self._minmax_over_shards(shards)
        return self._store_minmax(mn, mx)

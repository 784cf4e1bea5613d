"""Common state of a sparser: what to prune (`type`), how (`strategy`), how much (`ratio`).
Interface of sparsebit/sparse/sparsers/base.py:6-26."""
from torch import nn


class Sparser(nn.Module):
    STRATEGY = "base"

    def __init__(self, config, opr=None):
        # (not super().__init__(): in a class derived for the reference -- plugin._derive -- the next __init__ in the MRO
        # is the reference base's, which wants (config, opr) and would set the same attributes)
        nn.Module.__init__(self)
        spec = config.SPARSER
        self.config, self.opr = config, opr
        self.type, self.strategy, self.ratio = spec.TYPE, spec.STRATEGY, spec.RATIO

    def set_ratio(self, ratio):
        self.ratio = ratio

    def calc_mask(self, x):
        """-> a mask shaped like x (1 keeps, 0 prunes); subclasses implement the strategy."""
        raise NotImplementedError(type(self).__name__)

    def __repr__(self):
        return ", ".join(str(v) for v in (self.type, self.strategy, self.ratio))

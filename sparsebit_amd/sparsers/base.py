"""Sparser base (mirrors sparsebit/sparse/sparsers/base.py:6-26)."""
from abc import ABC

from torch import nn


class Sparser(nn.Module, ABC):
    STRATEGY = "base"

    def __init__(self, config, opr=None):
        super(Sparser, self).__init__()
        self.config = config
        self.opr = opr
        self.type = config.SPARSER.TYPE
        self.strategy = config.SPARSER.STRATEGY
        self.ratio = config.SPARSER.RATIO

    def calc_mask(self, x):
        pass

    def set_ratio(self, ratio):
        self.ratio = ratio

    def __repr__(self):
        return "{}, {}, {}".format(self.type, self.strategy, self.ratio)

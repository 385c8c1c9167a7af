"""Mean functions that feed the hot path (reference: pyGPs/Core/mean.py -- Mean :45-138, Zero :279-294,
One :297-312, Const :315-336, Linear :339-382).  O(N D) host work: they produce the vector m and
the columns dm_i that Exact.evaluate hands to the device (Core/inf.py:358, 378-381)."""
import logging

import numpy as np


class Mean(object):
    def __init__(self):
        self.hyp = []
        self.para = []
        self.logger = logging.getLogger(__name__)

    def __repr__(self):
        return (str(type(self)) + ": to get the mean vector or mean derviatives use: \n"
                "model.meanfunc.getMean()\nmodel.meanfunc.getDerMatrix()")

    def getMean(self, x=None):
        raise NotImplementedError

    def getDerMatrix(self, x=None, der=None):
        raise NotImplementedError


class Zero(Mean):
    def __init__(self):
        self.hyp = []
        self.name = "0"

    def getMean(self, x=None):
        return np.zeros((x.shape[0], 1))

    def getDerMatrix(self, x=None, der=None):
        return np.zeros((x.shape[0], 1))


class One(Mean):
    def __init__(self):
        self.hyp = []
        self.name = "1"

    def getMean(self, x=None):
        return np.ones((x.shape[0], 1))

    def getDerMatrix(self, x=None, der=None):
        return np.zeros((x.shape[0], 1))


class Const(Mean):
    """hyp = [c]"""

    def __init__(self, c=5.):
        self.hyp = [c]

    def getMean(self, x=None):
        return self.hyp[0] * np.ones((x.shape[0], 1))

    def getDerMatrix(self, x=None, der=None):
        return np.ones((x.shape[0], 1)) if der == 0 else np.zeros((x.shape[0], 1))


class Linear(Mean):
    """hyp = alpha_list (one weight per input dimension)"""

    def __init__(self, D=None, alpha_list=None):
        self.hyp = [0.5 for _ in range(D)] if alpha_list is None else list(alpha_list)

    def getMean(self, x=None):
        n, D = x.shape
        return np.dot(x, np.reshape(np.array(self.hyp, dtype=float), (D, 1)))

    def getDerMatrix(self, x=None, der=None):
        n, D = x.shape
        if isinstance(der, int) and der < D:
            return np.reshape(x[:, der], (n, 1)).astype(float)
        return np.zeros((n, 1))

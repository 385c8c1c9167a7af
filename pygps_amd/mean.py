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

    # operator overloading (Core/mean.py:66-103)
    def __add__(self, other):
        return SumOfMean(self, other)

    def __mul__(self, other):
        if isinstance(other, (int, float)):
            return ScaleOfMean(self, other)
        if isinstance(other, Mean):
            return ProductOfMean(self, other)
        logging.getLogger(__name__).error("only numbers and Means are allowed for *")

    __rmul__ = __mul__

    def __pow__(self, number):
        if isinstance(number, int) and number > 0:
            return PowerOfMean(self, number)
        logging.getLogger(__name__).error("only non-zero integers are supported for **")


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


# ---- composites (Core/mean.py:140-276): O(N) host arithmetic on the children's vectors ------------------------
class _PairOfMean(Mean):
    def __init__(self, mean1, mean2):
        self.mean1, self.mean2 = mean1, mean2
        self.para = []

    @property
    def hyp(self):
        return list(self.mean1.hyp) + list(self.mean2.hyp)

    @hyp.setter
    def hyp(self, value):
        n1 = len(self.mean1.hyp)
        assert len(value) == n1 + len(self.mean2.hyp)
        self.mean1.hyp = list(value[:n1])
        self.mean2.hyp = list(value[n1:])


class SumOfMean(_PairOfMean):
    def getMean(self, x=None):
        return self.mean1.getMean(x) + self.mean2.getMean(x)

    def getDerMatrix(self, x=None, der=None):
        n1 = len(self.mean1.hyp)
        if der < n1:
            return self.mean1.getDerMatrix(x, der)
        if der < len(self.hyp):
            return self.mean2.getDerMatrix(x, der - n1)
        raise Exception("Error: der out of range for meanSum")


class ProductOfMean(_PairOfMean):
    def getMean(self, x=None):
        return self.mean1.getMean(x) * self.mean2.getMean(x)

    def getDerMatrix(self, x=None, der=None):
        n1 = len(self.mean1.hyp)
        if der < n1:
            return self.mean1.getDerMatrix(x, der) * self.mean2.getMean(x)
        if der < len(self.hyp):
            return self.mean2.getDerMatrix(x, der - n1) * self.mean1.getMean(x)
        raise Exception("Error: der out of range for meanProduct")


class _UnaryOfMean(Mean):
    def __init__(self, mean, first):
        self.mean = mean
        self.para = []
        self._first = [first]

    @property
    def hyp(self):
        return self._first + list(self.mean.hyp)

    @hyp.setter
    def hyp(self, value):
        assert len(value) == 1 + len(self.mean.hyp)
        self._first = [value[0]]
        self.mean.hyp = list(value[1:])


class ScaleOfMean(_UnaryOfMean):
    """c * m(x); hyp = [c] + m.hyp (the scale is a plain, not a log-space, hyper: Core/mean.py:229-241)."""

    def getMean(self, x=None):
        return self.hyp[0] * self.mean.getMean(x)

    def getDerMatrix(self, x=None, der=None):
        if der == 0:
            return self.mean.getMean(x)
        return self.hyp[0] * self.mean.getDerMatrix(x, der - 1)


class PowerOfMean(_UnaryOfMean):
    """m(x) ** max(|floor(d)|, 1); hyp = [d] + m.hyp (Core/mean.py:245-276)."""

    def _d(self):
        return max(np.abs(np.floor(self.hyp[0])), 1)

    def getMean(self, x=None):
        return self.mean.getMean(x) ** self._d()

    def getDerMatrix(self, x=None, der=None):
        d = self._d()
        a = self.mean.getMean(x)
        if der == 0:
            return a ** d * np.log(a)
        return d * a ** (d - 1) * self.mean.getDerMatrix(x, der - 1)

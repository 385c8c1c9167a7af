"""Random-restart configuration of the optimisers (reference: pyGPs/Optimization/conf.py:18-57)."""


class random_init_conf(object):
    """num_restarts / min_threshold stop rules and the per-hyper-parameter ranges the random initial
    points are drawn from (default (-5, 5) each; setters check the length)."""

    def __init__(self, mean, cov, lik):
        self.num_restarts = None
        self.min_threshold = None
        self.mean, self.cov, self.lik = mean, cov, lik
        self._ranges = {"mean": [(-5, 5) for _ in mean.hyp], "cov": [(-5, 5) for _ in cov.hyp],
                        "lik": [(-5, 5) for _ in lik.hyp]}

    def _set(self, which, owner, value, label):
        if len(value) != len(owner.hyp):
            raise Exception("The length of %sRange is not consistent with number of %s hyparameters" % (which, label))
        self._ranges[which] = value

    meanRange = property(lambda s: s._ranges["mean"], lambda s, v: s._set("mean", s.mean, v, "mean"))
    covRange = property(lambda s: s._ranges["cov"], lambda s, v: s._set("cov", s.cov, v, "covariance"))
    likRange = property(lambda s: s._ranges["lik"], lambda s, v: s._set("lik", s.lik, v, "liklihood"))

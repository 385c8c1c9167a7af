// Probit (cumulative Gaussian) likelihood, EP-mode moments -- reference: Core/lik.py Erf.evaluate :295-311,
// cumGauss :328-338, gauOverCumGauss :340-352, logphi :354-366.  Shared by the per-site device kernel and the
// host-side vector evaluations of EP (csrc/ep.hip).
#pragma once
#include <cmath>
#ifdef __HIPCC__
#define PGP_HD __host__ __device__
#else
#define PGP_HD
#endif

// log Phi(z): asymptotic expansion below -6.2, logistic blend on [-6.2, -5.5]
PGP_HD inline double erf_logphi(double z) {
    const double p = 0.5 * (1.0 + erf(z * 0.70710678118654752440));
    const double zmin = -6.2, zmax = -5.5;
    if (z > zmax) return log(p);
    const double asym = -0.5 * log(M_PI) - 0.5 * z * z - log(sqrt(0.5 * z * z + 2.0) - z * 0.70710678118654752440);
    if (z < zmin) return asym;
    const double lam = 1.0 / (1.0 + exp(25.0 * (0.5 - (z - zmin) / (zmax - zmin))));
    return (1.0 - lam) * asym + lam * log(p);
}

// N(f)/Phi(f) given p = "Phi(f)" (the caller passes exp(logphi)), tight upper bound below -6, blend on [-6,-5]
PGP_HD inline double erf_ratio(double f, double p) {
    const double naive = exp(-0.5 * f * f) * 0.39894228040143267794 / p;
    if (f > -5.0) return naive;
    const double bound = sqrt(0.25 * f * f + 1.0) - 0.5 * f;
    if (f < -6.0) return bound;
    const double lam = -5.0 - f;
    return (1.0 - lam) * naive + lam * bound;
}

// lZ, dlZ, d2lZ of  Z = int Phi(y f) N(f | mu, s2) df   (y in {+1,-1}; 0 counts as +1)
PGP_HD inline void erf_ep_moments(double y, double mu, double s2, double* lZ, double* dlZ, double* d2lZ) {
    const double ys = (y < 0.0) ? -1.0 : 1.0;
    const double den = sqrt(1.0 + s2);
    const double z = ys * mu / den;
    const double l = erf_logphi(z);
    *lZ = l;
    if (dlZ || d2lZ) {
        const double n_p = erf_ratio(z, exp(l));
        if (dlZ) *dlZ = ys * n_p / den;
        if (d2lZ) *d2lZ = -n_p * (z + n_p) / (1.0 + s2);
    }
}

// Entry points declared in include/pygps_amd.h whose device path is not built yet return -99.
#include "ctx.h"
extern "C" {
int pgp_ep_fit(pgp_ctx*, int, const double*, int, int, int, const double*, const double*, int, int, int, double*,
               double*, double*, double*, double*, double*, int*, pgp_factor**) { return -99; }
}

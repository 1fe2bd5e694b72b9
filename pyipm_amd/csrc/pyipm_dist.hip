// pyipm_dist.hip -- the distributed driver (per-panel schedule, exchanges, sweeps across ranks; include/pyipm_newton.h,
// the *_dist / comm_* / set_exchange* entries) as a translation unit of its own.  gfx950 only.
#include "driver.hpp"
using namespace pyipm;
using namespace pyipm::drv;
#pragma GCC visibility push(default)
#include "dist_impl.hpp"
#pragma GCC visibility pop

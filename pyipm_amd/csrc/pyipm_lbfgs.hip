// pyipm_lbfgs.hip -- the L-BFGS search direction (include/pyipm_lbfgs.h) as a translation unit of its own.  gfx950 only.
#include "driver.hpp"
using namespace pyipm;
using namespace pyipm::drv;
#include "kernels_lbfgs.hpp"
#pragma GCC visibility push(default)
#include "lbfgs_impl.hpp"
#pragma GCC visibility pop

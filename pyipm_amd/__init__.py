"""pyipm_amd — MI355X-native Newton-step core for pyipm-style interior-point solvers.

The package is deliberately thin: ``csrc/`` (hand-written HIP kernels + the C-ABI
in ``include/pyipm_newton.h``), ``newton.py`` (ctypes binding, PyTorch-ROCm
tensors as the device container), ``ipm.py`` (host ``IPM`` class mirroring
``/root/reference/pyipm.py:23,311-314,1567,1863``), ``qp.py`` (the same loop
device-resident for QPs), ``lbfgs.py`` (binding of ``include/pyipm_lbfgs.h``: the
limited-memory direction of ``lbfgs=m``), ``dist.py`` / ``batched.py`` (multi-GPU and batched
drivers), ``problems.py`` (example problems + synthetic QP generator).  Submodules are imported lazily so that the
pure-NumPy parts work in a GPU-less container; anything that needs the HIP
library fails loudly there — there is no CPU fallback.
"""
__version__ = "0.1.0"

__all__ = ["IPM", "NewtonCore", "LbfgsCore", "QPDeviceIPM", "problems"]


def __getattr__(name):
    if name == "IPM":
        from .ipm import IPM
        return IPM
    if name == "NewtonCore":
        from .newton import NewtonCore
        return NewtonCore
    if name == "LbfgsCore":
        from .lbfgs import LbfgsCore
        return LbfgsCore
    if name == "QPDeviceIPM":
        from .qp import QPDeviceIPM
        return QPDeviceIPM
    if name == "problems":
        from . import problems
        return problems
    raise AttributeError(name)

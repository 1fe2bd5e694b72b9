"""``python -m pyipm_amd K`` — run example problem K (1..10), the counterpart of ``python pyipm.py K``
(/root/reference/pyipm.py:1866-2133; BASELINE.json configs[0] is K=7).  Prints the reference's
transcript shape (README.md:101-122).  Needs a GPU: the Newton step runs on the HIP core."""
from __future__ import annotations

import sys

import numpy as np

from .ipm import IPM
from .problems import example_problem


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if not argv:
        raise SystemExit("usage: python -m pyipm_amd <problem 1..10> [seed]")
    k = int(argv[0])
    p = example_problem(k)
    rng = np.random.RandomState(int(argv[1])) if len(argv) > 1 else np.random
    x0 = rng.rand(6) if k == 6 else rng.randn(p["nvar"])
    if k == 6:
        x0 = x0 / np.sum(x0)
    ipm = IPM(x0=x0, f=p["f"], df=p["df"], d2f=p["d2f"], ce=p["ce"], dce=p["dce"], d2ce=p["d2ce"],
              ci=p["ci"], dci=p["dci"], d2ci=p["d2ci"], Ftol=1.0e-8, verbosity=1)
    x, s, lda, fval, kkt = ipm.solve()
    print('')
    print('Ground truth: x = {}'.format(' or '.join('[' + ', '.join(str(float(v)) for v in g) + ']' for g in p["ground_truth"])))
    print('Solver solution: x = [{}]'.format(', '.join(str(v) for v in x)))
    if p["nineq"]:
        print('Slack variables: s = [{}]'.format(', '.join(str(v) for v in s)))
    if p["neq"] or p["nineq"]:
        print('Lagrange multipliers: lda = [{}]'.format(', '.join(str(v) for v in lda)))
    print('f(x) = {}'.format(fval))
    print('Karush-Kuhn-Tucker conditions (up to a sign):\n{}'.format(kkt))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())

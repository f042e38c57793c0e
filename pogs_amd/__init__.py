"""pogs_amd -- MI355X-native POGS graph-form ADMM engine.

Drop-in for the hot path of the reference ``pogs`` Python package
(``python/pogs/__init__.py:27-51``): the same ``solve_*`` functions, served by
hand-written HIP kernels behind the reference's C ABI.
"""
from .graph import (  # noqa: F401
    Function,
    FunctionObj,
    FunctionVector,
    Ordering,
    Solver,
    _solve_graph_form,
    dist_unique_id,
    func_eval,
    proj_subgrad_eval,
    prox_eval,
    rand_uniform,
    solve_elastic_net,
    solve_huber,
    solve_lasso,
    solve_logistic,
    solve_nonneg_ls,
    solve_ridge,
    solve_svm,
)

from .cvxpy import pogs_solve  # noqa: E402,F401  (reference: python/pogs/__init__.py:29)

__version__ = "0.1.0"
__all__ = [
    "solve_lasso", "solve_ridge", "solve_elastic_net", "solve_logistic", "solve_svm", "solve_huber",
    "solve_nonneg_ls", "pogs_solve", "Function", "FunctionObj", "FunctionVector", "Ordering", "Solver",
]

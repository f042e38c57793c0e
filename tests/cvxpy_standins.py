"""Stand-ins for the handful of cvxpy node types the graph-form front end looks at (cvxpy is not
installed in the build image or on the GPU box).  Same class names, same attributes the
reference's python/pogs/cvxpy.py reads: `.args`, `.is_constant()`, `.value`, `.size`, `.p`,
objective `.NAME` / `.expr`, problem `.variables()` / `.constraints` / `.objective` / `.solve()`.
tests/golden/make_cvxpy_golden.py installs them as a fake `cvxpy` module to run the REFERENCE's
detector on the trees of `cases()`; tests/test_host_logic.py runs pogs_amd.cvxpy on the same trees."""
import types

import numpy as np


class _Node:
    args = ()

    def is_constant(self):
        return False


class Variable(_Node):
    def __init__(self, n):
        self.size = n
        self.value = None


class Constant(_Node):
    def __init__(self, value):
        self.value = value

    def is_constant(self):
        return True


class NegExpression(Constant):       # -Constant: constant, value already negated
    def __init__(self, c):
        super().__init__(-np.asarray(c.value))
        self.args = (c,)


class _Op(_Node):
    def __init__(self, *args):
        self.args = tuple(args)


class AddExpression(_Op):
    pass


class MulExpression(_Op):            # matrix product / constant * expression
    pass


class multiply(_Op):                 # noqa: N801  (element-wise product; cvxpy's scalar * expr)
    pass


class quad_over_lin(_Op):            # noqa: N801  (what cp.sum_squares builds)
    pass


class sum_squares(_Op):              # noqa: N801
    pass


class norm1(_Op):                    # noqa: N801
    pass


class Pnorm(_Op):
    def __init__(self, arg, p):
        super().__init__(arg)
        self.p = p


class NonNeg:
    def __init__(self, arg):
        self.args = (arg,)


class Equality:
    def __init__(self, a, b):
        self.args = (a, b)


class Minimize:
    NAME = "minimize"

    def __init__(self, expr):
        self.expr = expr


class Maximize(Minimize):
    NAME = "maximize"


class Problem:
    def __init__(self, objective, constraints=None, variables=None):
        self.objective = objective
        self.constraints = list(constraints or [])
        self._vars = list(variables or [])
        self._status = None
        self._value = None
        self.fallback_calls = []

    def variables(self):
        return self._vars

    def solve(self, **kw):           # the "default cvxpy solver": records the call
        self.fallback_calls.append(kw)
        self._status = "fallback"
        return float("nan")


def fake_cvxpy_module():
    """A module object laid out like the parts of cvxpy the reference touches."""
    cp = types.ModuleType("cvxpy")
    cp.atoms = types.SimpleNamespace(
        affine=types.SimpleNamespace(add_expr=types.SimpleNamespace(AddExpression=AddExpression),
                                     binary_operators=types.SimpleNamespace(MulExpression=MulExpression)))
    cp.constraints = types.SimpleNamespace(nonpos=types.SimpleNamespace(NonNeg=NonNeg))
    return cp


def cases(seed=5):
    """name -> Problem.  Data are seeded; every tree is built the way cvxpy would build it for the
    expression in the comment."""
    rng = np.random.default_rng(seed)
    m, n = 12, 5
    A, b = rng.standard_normal((m, n)), rng.standard_normal(m)
    out = {}

    def resid(x, with_b=True):       # A @ x - b
        ax = MulExpression(Constant(A), x)
        return AddExpression(ax, NegExpression(Constant(b))) if with_b else ax

    def prob(expr, x, cons=None, sense=Minimize):
        return Problem(sense(expr), cons, [x])

    x = Variable(n)   # sum_squares(A @ x - b) + 0.3 * norm1(x)
    out["lasso"] = prob(AddExpression(quad_over_lin(resid(x), Constant(1.0)), multiply(Constant(0.3), norm1(x))), x)
    x = Variable(n)   # 0.5 * sum_squares(A @ x - b) + 0.2 * norm(x, 1)
    out["lasso_half_pnorm"] = prob(AddExpression(multiply(Constant(0.5), sum_squares(resid(x))),
                                                 multiply(Constant(0.2), Pnorm(x, 1))), x)
    x = Variable(n)   # norm1(x) * 0.7 + sum_squares(A @ x)      (scale on the right, no offset)
    out["lasso_no_offset"] = prob(AddExpression(multiply(norm1(x), Constant(0.7)), sum_squares(resid(x, False))), x)
    x = Variable(n)   # sum_squares(x - b5) + 0.1 norm1(x)       (identity operator)
    out["lasso_identity"] = prob(AddExpression(sum_squares(AddExpression(x, NegExpression(Constant(b[:n])))),
                                               multiply(Constant(0.1), norm1(x))), x)
    x = Variable(n)   # 2 * sum_squares(A @ x - b) + 0.6 * sum_squares(x)
    out["ridge"] = prob(AddExpression(MulExpression(Constant(2.0), sum_squares(resid(x))),
                                      multiply(Constant(0.6), sum_squares(x))), x)
    x = Variable(n)   # sum_squares(A @ x - b), x >= 0
    out["nnls"] = prob(sum_squares(resid(x)), x, [NonNeg(x)])
    x = Variable(n)   # 0.5 * sum_squares(A @ x - b), x >= 0     (constant * expr as MulExpression)
    out["nnls_scaled"] = prob(MulExpression(Constant(0.5), sum_squares(resid(x))), x, [NonNeg(x)])
    # --- not graph form for this front end
    x = Variable(n)
    out["maximize"] = prob(AddExpression(sum_squares(resid(x)), multiply(Constant(0.3), norm1(x))), x, sense=Maximize)
    x, z = Variable(n), Variable(n)
    out["two_variables"] = Problem(Minimize(AddExpression(sum_squares(resid(x)), norm1(z))), [], [x, z])
    x = Variable(n)
    out["lasso_with_constraint"] = prob(AddExpression(sum_squares(resid(x)), norm1(x)), x, [Equality(x, Constant(0))])
    x = Variable(n)
    out["two_norms"] = prob(AddExpression(sum_squares(resid(x)), norm1(x), multiply(Constant(2.0), norm1(x))), x)
    x = Variable(n)
    out["plain_least_squares"] = prob(sum_squares(resid(x)), x)
    x = Variable(n)
    out["norm_of_residual"] = prob(AddExpression(sum_squares(resid(x)), norm1(resid(x))), x)
    x = Variable(n)   # sum_squares of a sum of two non-constant terms
    out["nonaffine_inside"] = prob(AddExpression(sum_squares(AddExpression(MulExpression(Constant(A), x), MulExpression(Constant(A), x))),
                                                 norm1(x)), x)
    return out

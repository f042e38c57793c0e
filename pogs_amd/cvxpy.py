"""CVXPY front end of the graph-form path: ``pogs_solve(problem)``.

Mirror of the reference's ``python/pogs/cvxpy.py`` (``pogs_solve`` :33-92; pattern detection
:95-375; dispatch :378-436): a CVXPY problem in ONE variable whose objective is one of

    lasso      s * sum_squares(A @ x - b) + t * norm1(x)              (:254-310)
    ridge      s * sum_squares(A @ x - b) + t * sum_squares(x)        (:313-359)
    nonneg LS  [c *] sum_squares(A @ x - b)   subject to  x >= 0      (:362-375)

is handed to ``solve_lasso`` / ``solve_ridge`` / ``solve_nonneg_ls`` (lambda = t / (2 s) resp.
t / s; the returned optimum is scaled back by 2 s), the variable's ``value`` and the problem's
status / value are filled in, and anything else falls through to ``problem.solve``.

The reference recognises nodes partly by ``isinstance`` against cvxpy classes and partly by class
name; cvxpy is not a dependency here (and is not installed in the build image), so this module
goes by class name throughout -- ``AddExpression``, ``MulExpression``, ``multiply``,
``sum_squares`` / ``quad_over_lin``, ``norm1`` / ``Pnorm`` with p = 1, ``NonNeg``, objective
``NAME == "minimize"`` -- which is also what makes it testable without cvxpy
(tests/test_host_logic.py builds the same trees from stand-in classes).

Reference behaviours kept on purpose (a caller switching libraries gets the same numbers):
  * non-negative least squares reports ``optval`` of 1/2 |Ax - b|^2 unscaled, whatever constant
    multiplies the objective (the reference sets no ``optval_scale`` there, :362-375 with :74-75);
  * a sparse constant ``A`` is densified (:184-185);
  * on a failed graph-form solve the problem is passed on to ``problem.solve`` (:85-92).
"""
import time

import numpy as np

from .graph import solve_lasso, solve_nonneg_ls, solve_ridge

_SUMSQ = ("sum_squares", "quad_over_lin")


def _kind(node):
    return type(node).__name__


def _is_const(node):
    fn = getattr(node, "is_constant", None)
    return bool(fn()) if callable(fn) else False


def _dense(value):
    toarray = getattr(value, "toarray", None)      # scipy.sparse constants are densified, as in the reference
    return np.asarray(toarray() if callable(toarray) else value, dtype=np.float64)


def _scaled(node):
    """(inner, factor) for `const * inner` in either of cvxpy's two product node types."""
    if _kind(node) in ("multiply", "MulExpression") and len(getattr(node, "args", ())) == 2:
        lhs, rhs = node.args
        if _is_const(lhs):
            return rhs, float(lhs.value)
        if _kind(node) == "multiply" and _is_const(rhs):
            return lhs, float(rhs.value)
    return node, 1.0


def _operator_of(node, x):
    """A for `A @ x` (constant A), the identity for x itself; None otherwise."""
    if node is x:
        return np.eye(int(x.size))
    args = getattr(node, "args", ())
    if len(args) == 2 and args[1] is x and _is_const(args[0]) and hasattr(args[0], "value"):
        return _dense(args[0].value)
    return None


def _affine_of(node, x):
    """(A, b) with node == A @ x - b; None when node is not of that shape."""
    if _kind(node) == "AddExpression":
        linear, offset = None, None
        for arg in node.args:
            if _is_const(arg):
                offset = arg.value                  # (like the reference: a later constant replaces an earlier one)
            elif linear is None:
                linear = arg
            else:
                return None
        A = _operator_of(linear, x) if linear is not None else None
        if A is None:
            return None
        b = np.zeros(A.shape[0]) if offset is None or np.size(offset) == 0 else -np.asarray(offset, np.float64).ravel()
        return A, b
    A = _operator_of(node, x)
    return None if A is None else (A, np.zeros(A.shape[0]))


def _is_norm1(node):
    return _kind(node) == "norm1" or (_kind(node) == "Pnorm" and getattr(node, "p", None) == 1)


def _constraint_class(constraints, x):
    if not constraints:
        return "free"
    for c in constraints:
        if _kind(c) == "NonNeg" and c.args[0] is x:
            return "nonneg"
    return "other"


def _match_lasso(obj, x, cons):
    if cons != "free" or _kind(obj) != "AddExpression":
        return None
    sq = l1 = None
    for term in obj.args:
        inner, w = _scaled(term)
        if _kind(inner) in _SUMSQ:
            if sq is not None:
                return None
            sq = (inner, w)
        elif _is_norm1(inner):
            if l1 is not None:
                return None
            l1 = (inner, w)
    if sq is None or l1 is None or l1[0].args[0] is not x:
        return None
    aff = _affine_of(sq[0].args[0], x)
    if aff is None:
        return None
    s, t = sq[1], l1[1]
    return "lasso", dict(A=aff[0], b=aff[1], lambd=float(t / (2 * s) if s != 0 else t), optval_scale=2.0 * s)


def _match_ridge(obj, x, cons):
    if cons != "free" or _kind(obj) != "AddExpression":
        return None
    data = reg = None
    for term in obj.args:
        inner, w = _scaled(term)
        if _kind(inner) not in _SUMSQ:
            continue
        if inner.args[0] is x:
            reg = (inner, w)
        elif _affine_of(inner.args[0], x) is not None:
            data = (inner, w)
    if data is None or reg is None:
        return None
    A, b = _affine_of(data[0].args[0], x)
    s, t = data[1], reg[1]
    return "ridge", dict(A=A, b=b, lambd=float(t / s if s != 0 else t), optval_scale=2.0 * s)


def _match_nonneg_ls(obj, x, cons):
    if cons != "nonneg":
        return None
    inner = obj
    if _kind(obj) == "MulExpression" and len(obj.args) == 2 and _is_const(obj.args[0]):
        inner = obj.args[1]
    if _kind(inner) not in _SUMSQ:
        return None
    aff = _affine_of(inner.args[0], x)
    return None if aff is None else ("nonneg_ls", dict(A=aff[0], b=aff[1]))


def detect_graph_form(problem):
    """("lasso" | "ridge" | "nonneg_ls", params) or None  (reference: _detect_graph_form, :95-120)."""
    objective = problem.objective
    if getattr(objective, "NAME", None) != "minimize":
        return None
    variables = problem.variables()
    if len(variables) != 1:
        return None
    x = variables[0]
    cons = _constraint_class(problem.constraints, x)
    for match in (_match_lasso, _match_ridge, _match_nonneg_ls):
        hit = match(objective.expr, x, cons)
        if hit is not None:
            return hit
    return None


def _run(kind, params, opts):
    common = dict(abs_tol=opts.get("abs_tol", 1e-4), rel_tol=opts.get("rel_tol", 1e-4), max_iter=opts.get("max_iter", 2500),
                  verbose=opts.get("verbose", 0), rho=opts.get("rho", 1.0))
    A = np.asarray(params["A"], np.float64)
    b = np.asarray(params["b"], np.float64).ravel()
    t0 = time.perf_counter()
    if kind == "lasso":
        out = solve_lasso(A, b, params["lambd"], **common)
    elif kind == "ridge":
        out = solve_ridge(A, b, params["lambd"], **common)
    else:
        out = solve_nonneg_ls(A, b, **common)
    out["solve_time"] = time.perf_counter() - t0
    return out


def pogs_solve(problem, verbose=False, **solver_opts):
    """Solves a CVXPY problem; the graph-form engine when the problem is a lasso, a ridge regression
    or a non-negative least squares in one variable, ``problem.solve`` otherwise.  Returns the
    optimal value (reference: pogs_solve, python/pogs/cvxpy.py:33-92)."""
    hit = detect_graph_form(problem)
    if hit is None:
        if verbose:
            print("POGS: No graph-form pattern detected, using default solver")
    else:
        kind, params = hit
        if verbose:
            print("POGS: Detected %s pattern, using fast graph-form solver" % kind)
        result = _run(kind, params, solver_opts)
        if result.get("status", 1) == 0:
            problem.variables()[0].value = result["x"]
            value = result["optval"] * params.get("optval_scale", 1.0)
            problem._status = "optimal"
            problem._value = value
            return value
        if verbose:
            print("POGS: Graph-form solver failed, falling back to default")
    return problem.solve(verbose=verbose, **solver_opts)

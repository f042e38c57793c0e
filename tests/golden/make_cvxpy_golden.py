"""Generates tests/golden/cvxpy_detection.json: what the REFERENCE's CVXPY front end
(/root/reference/python/pogs/cvxpy.py, `_detect_graph_form`) makes of the expression trees in
tests/cvxpy_standins.py:cases().  cvxpy itself is not installed, so the stand-in node classes are
installed as a fake `cvxpy` module (the reference only does isinstance / class-name / attribute
checks on the tree).  The reference's package __init__ would load its C library, so cvxpy.py is
loaded on its own with a stub `pogs.graph`.  Run in the build container:

    python tests/golden/make_cvxpy_golden.py"""
import importlib.util
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import numpy as np  # noqa: E402

import cvxpy_standins as S  # noqa: E402

REF = "/root/reference/python/pogs/cvxpy.py"


def main():
    sys.modules["cvxpy"] = S.fake_cvxpy_module()
    stub = types.ModuleType("pogs.graph")
    stub.solve_lasso = stub.solve_ridge = stub.solve_nonneg_ls = None
    sys.modules["pogs"] = types.ModuleType("pogs")
    sys.modules["pogs.graph"] = stub
    spec = importlib.util.spec_from_file_location("pogs.cvxpy", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    out = {}
    for name, problem in S.cases().items():
        det = ref._detect_graph_form(problem)
        if det is None:
            out[name] = None
        else:
            p = det["params"]
            out[name] = dict(type=det["type"], lambd=p.get("lambd"), optval_scale=p.get("optval_scale"),
                             A=np.asarray(p["A"]).tolist(), b=np.asarray(p["b"]).tolist())
    with open(os.path.join(HERE, "cvxpy_detection.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    for k, v in out.items():
        print("%-24s %s" % (k, None if v is None else (v["type"], v["lambd"], v["optval_scale"])))


if __name__ == "__main__":
    main()

"""Shared fixture helpers for the parity tests."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# name -> (proj_type, params) used by tests/golden/make_golden.py
SINGLE_MAPS = {
    "box01": ("box", {"lower": 0.0, "upper": 1.0}),
    "box_l0.05_u0.4": ("box", {"lower": 0.05, "upper": 0.4}),
    "simplex1": ("simplex", {"z": 1.0}),
    "simplex2.5": ("simplex", {"z": 2.5}),
    "cone_lower0": ("cone", {"lower": 0.0}),
    "cone_upper0.3": ("cone", {"upper": 0.3}),
}
NP_DT = {"f32": np.float32, "f64": np.float64}
# parity tolerances (relative to the magnitude of the compared vector)
RTOL = {"f32": 2e-4, "f64": 1e-9}


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def problem(z):
    return dict(m=int(z["m"]), n=int(z["n"]), colptr=z["colptr"], rowidx=z["rowidx"], a=z["a"], c=z["c"], b=z["b"])


def relerr(got, want):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    scale = max(1.0, float(np.max(np.abs(want)))) if want.size else 1.0
    return float(np.max(np.abs(got - want))) / scale if want.size else 0.0


def scala_5x5():
    """The 5x5 known-answer problem of the reference's tests/objectives/test_dualip_matching_simplex.py:10-99
    (values are data of the test; the matrix is its own transpose-by-construction layout: column j of the CSC
    matrix is row j of the dense "user x item" table)."""
    a = np.array(
        [
            [0.307766110869125, 0.483770735096186, 0.624996477039531, 0.669021712383255, 0.535811153938994],
            [0.257672501029447, 0.812402617651969, 0.882165518123657, 0.204612161964178, 0.710803845431656],
            [0.552322433330119, 0.370320537127554, 0.28035383997485, 0.357524853432551, 0.538348698290065],
            [0.0563831503968686, 0.546558595029637, 0.398487901547924, 0.359475114848465, 0.74897222686559],
            [0.468549283919856, 0.170262051047757, 0.76255108229816, 0.690290528349578, 0.420101450523362],
        ],
        dtype=np.float32,
    )
    # CSC of a.T : column j holds a[j, :] with rows 0..4
    colptr = np.arange(0, 26, 5, dtype=np.int64)
    rowidx = np.tile(np.arange(5, dtype=np.int64), 5)
    vals = a.reshape(-1)
    return dict(m=5, n=5, colptr=colptr, rowidx=rowidx, a=vals.copy(), c=-vals, b=np.full(5, 0.7, dtype=np.float32))


SCALA_GOLDEN = [(2, -3.6010155991401818), (16, -3.60842718733725), (23, -3.5080258013053136), (29, -3.4868496294227143)]

"""MPS reader (dualip_amd/utils/read_mps_data.py) -- CPU.  Pinned to what the reference's reader produced for the instance its
MIPLIB example ships (tests/golden/g6_miplib_v150.npz, written by make_golden_lp.py from examples/miplib_2017/read_mps_data.py),
plus hand-written files for the record kinds that instance does not use."""
import gzip
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from dualip_amd.utils.read_mps_data import read_mps_file  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def test_shipped_instance_parses_to_the_reference_readers_arrays():
    z = np.load(os.path.join(GOLD, "g6_miplib_v150.npz"))
    d = read_mps_file(os.path.join(GOLD, "v150d30-2hopcds.mps.gz"))
    assert len(d.b_vec) == int(z["m"]) == 7822 and len(d.C_vec) == int(z["n"]) == 150
    rows, cols = (np.array(v) for v in zip(*d.A_indices))
    assert np.array_equal(rows, z["coo_row"]) and np.array_equal(cols, z["coo_col"])  # same records in the same order
    assert np.array_equal(np.array(d.A_data), z["coo_val"])
    assert np.array_equal(np.array(d.C_vec), z["c"]) and np.array_equal(np.array(d.b_vec), z["b"])
    assert np.array_equal(np.array([b[0] for b in d.var_bounds]), z["lower"]) and np.array_equal(np.array([b[1] for b in d.var_bounds]), z["upper"])
    assert np.array_equal(np.array(d.equality_mask), z["equality_mask"])
    du = d.to_dualip_format(torch.float64)
    assert du.equality_mask is None and du.A.shape == (7822, 150) and du.A.is_sparse and du.A._nnz() == 103991
    assert list(du.projection_map) == ["bound_(0.0, 1.0)"]  # all binary
    e = du.projection_map["bound_(0.0, 1.0)"]
    assert e.proj_type == "box" and e.proj_params == {"lower": 0.0, "upper": 1.0} and list(e.indices) == list(range(150))
    dense = d.to_dualip_format(torch.float32, return_sparse=False)
    assert torch.equal(dense.A, du.A.to_dense().float())


SMALL = """* a comment
NAME          tiny
ROWS
 N  cost
 L  cap
 G  demand
 E  balance
 L  unused
COLUMNS
    x2        cost         2.0   cap          1.0
    x2        demand       1.0
    MARKER                 'MARKER'                 'INTORG'
    x10       cost        -1.0   balance      3.0
    x10       cap          4.0
    MARKER                 'MARKER'                 'INTEND'
    x1        demand       2.5   balance     -1.0
    y         cap          1.0
    z         balance      1.0
    w         cap          0.5
    v         demand       1.0
RHS
    RHS       cap         10.0   demand       3.0
    RHS       balance      1.5   cost         7.0
BOUNDS
 UP BND       x2           4.0
 LO BND       x2           1.0
 BV BND       x10
 FR BND       x1
 UP BND       y           -2.0
 MI BND       z
 FX BND       w            0.25
 PL BND       v
ENDATA
"""


def test_every_record_kind(tmp_path):
    p = tmp_path / "tiny.mps"
    p.write_text(SMALL)
    d = read_mps_file(str(p))
    assert d.column_names == ["v", "w", "x1", "x10", "x2", "y", "z"]  # string order: x10 before x2
    assert d.row_names == ["cap", "demand", "balance", "unused"]
    A = np.zeros((4, 7))
    for v, (i, j) in zip(d.A_data, d.A_indices):
        A[i, j] = v
    want = np.zeros((4, 7))
    want[0, [4, 3, 5, 1]] = [1.0, 4.0, 1.0, 0.5]           # cap (L): as written
    want[1, [4, 2, 0]] = [-1.0, -2.5, -1.0]                 # demand (G): negated
    want[2, [3, 2, 6]] = [3.0, -1.0, 1.0]                   # balance (E)
    assert np.array_equal(A, want)
    assert d.b_vec == [10.0, -3.0, 1.5, 0.0] and d.equality_mask == [False, False, True, False]
    assert d.C_vec == [0.0, 0.0, 0.0, -1.0, 2.0, 0.0, 0.0]
    inf = float("inf")
    assert d.var_bounds == [(0.0, inf), (0.25, 0.25), (-inf, inf), (0.0, 1.0), (1.0, 4.0), (-inf, -2.0), (-inf, inf)]
    # rows of A are grouped by constraint, constraints in file order
    assert [i for i, _ in d.A_indices] == sorted(i for i, _ in d.A_indices)
    du = d.to_dualip_format()
    assert du.equality_mask.tolist() == [False, False, True, False] and du.A.dtype == torch.float32
    assert du.projection_map["bound_(-inf, inf)"].indices == [2, 6]
    gz = tmp_path / "tiny.mps.gz"
    with gzip.open(gz, "wt") as fh:
        fh.write(SMALL)
    d2 = read_mps_file(str(gz))
    assert d2.A_data == d.A_data and d2.A_indices == d.A_indices and d2.var_bounds == d.var_bounds


def test_errors(tmp_path):
    def write(text):
        p = tmp_path / "bad.mps"
        p.write_text(text)
        return str(p)

    with pytest.raises(FileNotFoundError):
        read_mps_file(str(tmp_path / "missing.mps"))
    with pytest.raises(ValueError, match="Multiple objective rows"):
        read_mps_file(write("ROWS\n N a\n N b\nCOLUMNS\nENDATA\n"))
    with pytest.raises(ValueError, match="multiple inequality constraints"):
        read_mps_file(write("ROWS\n N a\n L r\n G r\nCOLUMNS\nENDATA\n"))
    with pytest.raises(ValueError, match="Malformed column line"):
        read_mps_file(write("ROWS\n N a\n L r\nCOLUMNS\n x r\nENDATA\n"))
    with pytest.raises(ValueError, match="RANGES"):
        read_mps_file(write("ROWS\n N a\n L r\nCOLUMNS\n x r 1\nRANGES\n RNG r 2\nENDATA\n"))
    with pytest.raises(ValueError, match="not a valid BoundType"):
        read_mps_file(write("ROWS\n N a\n L r\nCOLUMNS\n x r 1\nBOUNDS\n SC BND x 1\nENDATA\n"))

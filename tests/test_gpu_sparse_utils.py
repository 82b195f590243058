"""The reference's stand-alone CSC primitives (src/dualip/utils/sparse_utils.py) on the GPU: the cases of the reference's
tests/test_sparse_utils.py:10-250 re-expressed (same matrices, same expectations) plus kernel-form projection operators."""
import operator

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _csc(dense):
    return torch.tensor(dense, dtype=torch.float32).to_sparse_csc().to(DEV)


def _dense(t):
    return t.to_dense().cpu()


def test_stacking():
    from dualip_amd.utils.sparse_utils import hstack_csc, vstack_csc

    A, B = [[1.0, 0.0, 2.0], [0.0, 3.0, 0.0]], [[4.0, 5.0, 0.0], [0.0, 0.0, 6.0]]
    r = vstack_csc([_csc(A), _csc(B)])
    assert r.layout == torch.sparse_csc and torch.allclose(_dense(r), torch.vstack([torch.tensor(A), torch.tensor(B)]))
    A, B = [[1.0, 2.0], [3.0, 0.0]], [[0.0, 4.0, 5.0], [6.0, 0.0, 7.0]]
    r = hstack_csc([_csc(A), _csc(B)])
    assert r.layout == torch.sparse_csc and torch.allclose(_dense(r), torch.hstack([torch.tensor(A), torch.tensor(B)]))
    blocks = [torch.tensor(x) for x in ([[1.0, 2.0], [3.0, 4.0]], [[5.0, 6.0], [7.0, 8.0]], [[9.0, 10.0], [11.0, 12.0]], [[13.0, 14.0], [15.0, 16.0]])]
    sp = [b.to_sparse_csc().to(DEV) for b in blocks]
    r = vstack_csc([hstack_csc(sp[:2]), hstack_csc(sp[2:])])
    assert torch.allclose(_dense(r), torch.vstack([torch.hstack(blocks[:2]), torch.hstack(blocks[2:])]))


def test_left_and_right_multiply():
    from dualip_amd.utils.sparse_utils import left_multiply_sparse, right_multiply_sparse

    M = [[1.0, 0.0, 3.0], [0.0, 2.0, 0.0], [4.0, 0.0, 5.0]]
    v = torch.tensor([2.0, 3.0, 0.5])
    r = right_multiply_sparse(_csc(M), v.to(DEV))
    assert r.layout == torch.sparse_csc and torch.allclose(_dense(r), torch.tensor(M) @ torch.diag(v))
    r = left_multiply_sparse(v.to(DEV), _csc(M))
    assert r.layout == torch.sparse_csc and torch.allclose(_dense(r), torch.diag(v) @ torch.tensor(M))
    out = _csc(M)
    left_multiply_sparse(v.to(DEV), _csc(M), output_tensor=out)
    assert torch.allclose(_dense(out), torch.diag(v) @ torch.tensor(M))
    with pytest.raises(ValueError, match="CSC"):
        left_multiply_sparse(v.to(DEV), torch.tensor(M, device=DEV))


def test_elementwise_row_sums_dot():
    from dualip_amd.utils.sparse_utils import dot_product_csc, elementwise_csc, row_sums_csc

    M = [[1.0, 0.0, 3.0], [0.0, 2.0, 0.0], [4.0, 0.0, 5.0]]
    N = [[2.0, 0.0, 1.0], [0.0, 4.0, 0.0], [0.5, 0.0, 2.0]]
    A, B = _csc(M), _csc(N)
    tm, tn = torch.tensor(M), torch.tensor(N)
    for op, want in ((torch.add, tm + tn), (operator.sub, tm - tn), (torch.mul, tm * tn), (lambda x, y: x * 2 + y, tm * 2 + tn)):
        assert torch.allclose(_dense(elementwise_csc(A, B, op)), want)
    assert torch.allclose(_dense(elementwise_csc(A, B, torch.div)), torch.where(tn != 0, tm / torch.where(tn != 0, tn, torch.ones_like(tn)), torch.zeros_like(tm)))
    out = _csc(M)
    elementwise_csc(A, B, torch.mul, output_tensor=out)
    assert torch.allclose(_dense(out), tm * tn)
    with pytest.raises(ValueError, match="pattern"):
        elementwise_csc(A, _csc([[1.0, 1.0, 0.0], [0.0, 2.0, 0.0], [4.0, 0.0, 5.0]]), torch.add)
    assert torch.allclose(row_sums_csc(A).cpu(), tm.sum(1))
    assert float(dot_product_csc(A, B)) == pytest.approx(float((tm * tn).sum()))
    big = torch.rand(30000, 50).mul_(torch.rand(30000, 50) < 0.1).to_sparse_csc().to(DEV)  # more rows than the LDS plan holds
    assert torch.allclose(row_sums_csc(big).cpu(), big.to_dense().cpu().sum(1), atol=1e-5)


class TestApplyFToColumns:
    @staticmethod
    def _columnwise(M, F):
        res = M.clone()
        for j in range(M.shape[1]):
            nz = M[:, j] != 0
            if nz.any():
                res[:, j] = 0.0
                res[:, j][nz] = F(M[:, j][nz].unsqueeze(1)).squeeze(1)
        return res

    def test_callables(self):
        from dualip_amd.utils.sparse_utils import apply_F_to_columns

        M = torch.tensor([[1.0, 0.0, 3.0], [0.0, 2.0, 0.0], [4.0, 0.0, 5.0]])
        sp = M.to_sparse_csc().to(DEV)
        assert torch.allclose(_dense(apply_F_to_columns(sp, lambda x: x, [torch.arange(3)])), M)
        assert torch.allclose(_dense(apply_F_to_columns(sp, lambda x: 2 * x, [torch.arange(3)])), 2 * M)
        out = sp.clone()
        apply_F_to_columns(sp, lambda x: x * 3, [torch.arange(3)], output_tensor=out)
        assert torch.allclose(_dense(out), 3 * M)
        M5 = torch.tensor([[1.0, 0.0, 3.0, 0.0, 7.0], [0.0, 2.0, 0.0, 4.0, 0.0], [5.0, 0.0, 6.0, 0.0, 8.0]])
        sp5 = M5.to_sparse_csc().to(DEV)
        f = lambda x: x * 0.5  # noqa: E731
        assert torch.allclose(_dense(apply_F_to_columns(sp5, f, [torch.arange(5)])), M5 * 0.5)
        assert torch.allclose(_dense(apply_F_to_columns(sp5, f, [torch.tensor([0, 2, 4]), torch.tensor([1, 3])])), M5 * 0.5)
        Mv = torch.tensor([[1.0, 0.0, 3.0], [2.0, 0.0, 0.0], [3.0, 4.0, 0.0], [4.0, 0.0, 0.0]])
        assert torch.allclose(_dense(apply_F_to_columns(Mv.to_sparse_csc().to(DEV), lambda x: x**2, [torch.arange(3)])), self._columnwise(Mv, lambda x: x**2))
        M2 = torch.tensor([[1.0, 2.0], [3.0, 4.0]])
        assert torch.allclose(_dense(apply_F_to_columns(M2.to_sparse_csc().to(DEV), lambda x: -x, [torch.tensor([], dtype=torch.long), torch.arange(2)])), -M2)
        Mn = torch.tensor([[1.0, 0.0, -3.0], [0.0, -2.0, 0.0], [-4.0, 0.0, 5.0]])
        assert torch.allclose(_dense(apply_F_to_columns(Mn.to_sparse_csc().to(DEV), lambda x: x.clamp(min=0), [torch.arange(3)])), self._columnwise(Mn, lambda x: x.clamp(min=0)))

    def test_kernel_form_operators(self):
        """project(...) operators run as one launch per bucket and equal the operator applied column by column."""
        from dualip_amd.projections import project
        from dualip_amd.utils.sparse_utils import apply_F_to_columns

        g = torch.Generator().manual_seed(3)
        M = (torch.rand(40, 300, generator=g, dtype=torch.float64) * 2 - 0.5) * (torch.rand(40, 300, generator=g) < 0.2)
        sp = M.to_sparse_csc().to(DEV)
        for name, params in (("box", {"lower": 0.1, "upper": 0.6}), ("cone", {"lower": 0.2}), ("simplex", {"z": 1.0}), ("simplex_eq", {"z": 2.0})):
            op = project(name, **params)
            got = _dense(apply_F_to_columns(sp, op, [torch.arange(0, 150), torch.arange(150, 300)]))
            want = self._columnwise(M, lambda blk: op(blk.to(DEV)).cpu())
            assert torch.allclose(got, want, atol=1e-12), name
        half = _dense(apply_F_to_columns(sp, project("box", lower=0.0, upper=0.1), [torch.arange(0, 100)]))
        assert torch.equal(half[:, 100:], M[:, 100:])  # columns in no bucket keep their values

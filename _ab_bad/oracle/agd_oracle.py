"""oracle/agd_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement of the reference's host-side dual-ascent logic, in the reference's working precision:

  epilogue()            grad - b, dual objective, slacks          matching.py:25-34,164-178
  beta_seq()            Nesterov momentum sequence in float32     optimizers/agd.py:93-100
  StepSizer             Lipschitz-history step size               optimizers/agd_utils.py:12-89
  maximize()            accelerated projected gradient ascent     optimizers/agd.py:121-229 (+ _update_gamma :102-109)

Pinned against tests/golden/g2_*.npz / g4_beta_seq.npz (tests/test_oracle_golden.py).
"""
import math

import numpy as np


def epilogue(ax, obj0, sumsq, lam, b, gamma, dtype):
    """Finish ``calculate`` from the local partial (A x, c.x, sum x^2).

    reg = (gamma/2) * norm(x)**2 (matching.py:157); grad = A x - b; obj = c.x + reg + lam.grad (matching.py:31-33).
    Returns (grad, dual_obj, reg, dual_val_times_grad, max_pos_slack, sum_pos_slack) in ``dtype``.
    """
    T = np.dtype(dtype).type
    nrm = T(math.sqrt(sumsq))
    reg = T(T(gamma / 2) * T(nrm * nrm))
    grad = (np.asarray(ax, dtype=dtype) - np.asarray(b, dtype=dtype)).astype(dtype)
    dvtg = T(np.dot(np.asarray(lam, dtype=dtype), grad))
    obj = T(T(T(obj0) + reg) + dvtg)
    gmax = grad.max() if grad.size else T(0)
    max_pos = gmax if gmax > 0 else T(0)  # builtins.max(tensor, 0) (matching.py:168)
    sum_pos = T(np.maximum(grad, 0).sum())
    return grad, obj, reg, dvtg, max_pos, sum_pos


def beta_seq(max_iter: int) -> np.ndarray:
    """agd.py:93-100: t_i is rounded to float32 every time it is stored; sqrt is taken in double; the final
    division is a float32 / float32 operation."""
    t = np.zeros(max_iter + 2, dtype=np.float32)
    for i in range(1, max_iter + 2):
        sq = np.float32(t[i - 1] * t[i - 1])          # 0-dim fp32 tensor ** 2
        inner = np.float32(np.float32(4.0) * sq)       # 4 * tensor  -> fp32
        inner = np.float32(np.float32(1.0) + inner)    # 1 + tensor  -> fp32
        t[i] = np.float32((1.0 + math.sqrt(float(inner))) / 2.0)
    beta = np.zeros(max_iter, dtype=np.float32)
    for i in range(max_iter):
        beta[i] = np.float32(np.float32(np.float32(1.0) - t[i + 1]) / t[i + 2])
    return beta


class StepSizer:
    """agd_utils.py:65-89 with max_history_length = 15.  The reference recomputes every consecutive-pair
    Lipschitz estimate from the stored (gradient, dual) clones on each call; the values are a pure function of
    the stored vectors, so this keeps them as computed."""

    def __init__(self, dtype, max_history_length=15):
        self.T = np.dtype(dtype).type
        self.H = max_history_length
        self.grads = []
        self.duals = []

    def __call__(self, grad, dual, initial_step_size, max_step_size):
        if len(self.grads) == self.H:
            self.grads.pop(0)
            self.duals.pop(0)
        self.grads.append(grad.copy())
        self.duals.append(dual.copy())
        consts = []
        with np.errstate(divide="ignore", invalid="ignore"):
            for i in range(len(self.grads) - 1):
                num = self.T(np.linalg.norm(self.grads[i] - self.grads[i + 1]))
                den = self.T(np.linalg.norm(self.duals[i] - self.duals[i + 1]))
                consts.append(self.T(num / den))
        if not consts or len(consts) < self.H - 1:
            return float(initial_step_size)
        lmax = consts[0]
        for v in consts[1:]:  # builtins.max semantics: NaN is only "seen" in first position
            if v > lmax:
                lmax = v
        if math.isnan(float(lmax)) or math.isinf(float(lmax)):
            return float(initial_step_size)
        cand = 1.0 / float(lmax) if lmax != 0 else float(max_step_size)
        return min(cand, float(max_step_size))


def maximize(calc, lam0, max_iter, gamma, initial_step_size=1e-5, max_step_size=0.1, decay=None, eq_mask=None, dtype=np.float64):
    """agd.py:121-229.  ``calc(lam, gamma) -> (grad, dual_obj, extra)``; returns dict with logs and final state."""
    T = np.dtype(dtype).type
    beta = beta_seq(max_iter)
    x = np.asarray(lam0, dtype=dtype).copy()
    y = x.copy()
    sizer = StepSizer(dtype)
    obj_log, step_log = [], []
    last = None
    for i in range(1, max_iter + 1):
        grad, obj, extra = calc(x, gamma)
        last = (grad, obj, extra)
        obj_log.append(float(obj))
        step = sizer(np.asarray(grad, dtype=dtype), y, initial_step_size, max_step_size)
        step_log.append(step)
        y_new = (x + np.asarray(grad, dtype=dtype) * T(step)).astype(dtype)
        proj = np.maximum(y_new, T(0))
        y_new = np.where(eq_mask, y_new, proj) if eq_mask is not None else proj
        b = beta[i - 1]
        one_minus = np.float32(np.float32(1.0) - b)                      # 1.0 - fp32 0-dim tensor -> fp32
        x = (y_new * T(one_minus) + y * T(b)).astype(dtype)
        y = y_new
        if gamma is not None and decay is not None:
            if i % decay["decay_steps"] == 0:                             # agd.py:102-109
                gamma = gamma * decay["decay_factor"]
                max_step_size = step * decay["decay_factor"]
    return dict(dual_val=y, x_iter=x, dual_obj_log=np.array(obj_log), step_log=np.array(step_log), last=last, gamma=gamma)

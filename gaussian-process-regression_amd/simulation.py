"""The simulate_* harness around the hot path (reference R/simulation.R:80-139, :212-255, :316-336,
:338-349, :375-378).

What runs where: the test grid (`combine_all`), the kernel matrices, the factorisations, the predictions, the prior
draw of simulate_regression_gp (`multivariate_normal`) run on the MI355X through the C ABI.  The ground-truth function
`func`, the observation-noise closure and the index sampling are arbitrary user code / RNG draws and stay on the host,
as they are in R; so does the six-number `summary()` of the residuals (a sort of n* numbers).  The base-graphics /
ggplot layer of the reference is not reproduced: each function returns the reference's return value, a `Summary`
(dict: Min., 1st Qu., Median, Mean, 3rd Qu., Max.), with the data the plots are drawn from attached as `.data`.
"""
from __future__ import annotations

import ctypes as C
import math
import sys

import numpy as np

from . import _native as nat
from .covfunc import covariance_matrix
from .gpc import GPC
from .gpr import GPR
from .sampling import multivariate_normal

__all__ = ["combine_all", "iid_noise", "simulate_regression", "simulate_regression_gp", "simulate_classification", "Summary"]


class Summary(dict):
    """summary(abs(residual)): R's six numbers (quantile type 7); `.data` holds what the reference plots."""
    data = None

    @classmethod
    def of(cls, values, /, **data):
        v = np.asarray(values, dtype=np.float64)
        q = np.percentile(v, [0, 25, 50, 75, 100])                 # numpy's default "linear" == R's type 7
        s = cls({"Min.": q[0], "1st Qu.": q[1], "Median": q[2], "Mean": float(v.mean()), "3rd Qu.": q[3], "Max.": q[4]})
        s.data = data
        return s


def combine_all(lst, *, ctx=None):
    """combine_all(lst)  --  R/simulation.R:338-349: a length(lst) x prod(lengths) matrix whose columns are all
    combinations of the axis values, the last axis varying fastest.  Built on the device."""
    axes = [np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel()) for a in lst]
    if not axes or any(a.size == 0 for a in axes):
        raise ValueError("combine_all: a non-empty list of non-empty numeric vectors is required")
    d = len(axes)
    lengths = (C.c_int64 * d)(*[a.size for a in axes])
    vals = np.concatenate(axes)
    total = int(np.prod([a.size for a in axes], dtype=np.int64))
    out = np.empty((d, total), order="F")
    ctx = ctx or nat.default_context()
    nat.check(nat.lib().gprc_combine_all(ctx.handle, vals.ctypes.data, lengths, d, out.ctypes.data))
    return out


def iid_noise(distribution, *args, **kwargs):
    """iid_noise(distribution, ...)  --  R/simulation.R:375-378: returns function(X) distribution(ncol(X), ...).
    `distribution(n, ...)` is any callable returning n draws, e.g. lambda n, sd: rng.normal(0, sd, n)."""
    def noise(X):
        return distribution(np.asarray(X).shape[1], *args, **kwargs)
    return noise


def _limits(limits):
    lim = np.asarray(limits, dtype=np.float64)
    if lim.dtype.kind not in "fiu" or lim.size % 2:
        raise ValueError("is.numeric(limits), length(limits) %% 2 == 0 are not all TRUE")
    return lim if lim.ndim == 2 else lim.reshape(-1, 2)           # matrix(limits, ncol = 2, byrow = TRUE)


def _apply_cols(M, func):
    """apply(M, 2, func): func sees one column (a length-D vector) at a time."""
    return np.array([float(np.asarray(func(M[:, j] if M.shape[0] > 1 else M[0, j])).ravel()[0]) for j in range(M.shape[1])])


def _grid(lim, test_size, ctx=None):
    D = lim.shape[0]
    per = int(math.ceil(test_size ** (1.0 / D)))                 # seq.default: length.out <- ceiling(length.out), a bare ceiling
    return combine_all([np.linspace(lim[i, 0], lim[i, 1], per) for i in range(D)], ctx=ctx)


def _say(text):
    sys.stderr.write(text + "\n")


def simulate_regression(func, limits, training_points=None, training_size=10, observation_noise=lambda X: 0.0,
                        test_size=10000, show_pred=True, *, rng=None, **gpr_args):
    """simulate_regression(func, limits, training_points, training_size = 10L, observation_noise = function(x) 0,
    test_size = 10000L, show_pred = TRUE, ...)  --  R/simulation.R:80-139.  `...` goes to GPR$new (noise, k, cov_names).
    Returns summary(abs(residual)) on the equispaced test grid."""
    if not callable(func) or not callable(observation_noise):
        raise TypeError("is.function(func), is.function(observation_noise) are not all TRUE")
    lim = _limits(limits)
    if not (training_size > 0 and test_size > 0):
        raise ValueError("training_size > 0, test_size > 0 are not all TRUE")
    D = lim.shape[0]
    rng = rng if rng is not None else np.random.default_rng()
    if training_points is None:                                                              # :90-91
        training_points = np.vstack([rng.uniform(lim[i, 0], lim[i, 1], int(training_size)) for i in range(D)])
    else:
        training_points = np.asarray(training_points, dtype=np.float64).reshape(D, -1) if np.ndim(training_points) < 2 \
            else np.asarray(training_points, dtype=np.float64)
        if training_points.shape[0] != D:
            raise ValueError("nrow(limits) == nrow(training_points) is not TRUE")           # :93
        if not (np.all(training_points >= lim[:, :1]) and np.all(training_points <= lim[:, 1:])):
            raise ValueError("training_points must lie inside limits")                       # :94-95
    y = _apply_cols(training_points, func) + observation_noise(training_points)              # :97
    gaussian = GPR(training_points, np.asarray(y, dtype=np.float64).ravel(), **gpr_args)     # :98
    test_points = _grid(lim, test_size)                                                      # :101-102
    predictions = gaussian.predict(test_points, pointwise_var=True)                          # :103
    residual = predictions[:, 0] - _apply_cols(test_points, func)                            # :104
    _say("The mean absolute difference of predictions and ground truth in the considered limits is  %.15g" % np.mean(np.abs(residual)))
    x = np.arange(lim[0, 0], lim[0, 1] + 1e-12, 0.05)                                        # :115  seq(by = 0.05)
    if D == 1:
        plot_points = x.reshape(1, -1)
    else:                                                                                    # :120-122: first variable, others at mid-interval
        plot_points = np.vstack([x, np.repeat(lim[1:].mean(axis=1)[:, None], x.size, axis=1)])
    curve = gaussian.predict(plot_points, pointwise_var=True)
    return Summary.of(np.abs(residual), model=gaussian, test_points=test_points, predictions=predictions, residual=residual,
                      x=x, ground_truth=_apply_cols(plot_points, func), regression=curve[:, 0], variance=curve[:, 1])


def simulate_regression_gp(actual_cov, limits, observation_noise=lambda X: 0.0, test_size=300, training_size=10,
                           random_training=True, show_pred=False, *, rng=None, z=None, **gpr_args):
    """simulate_regression_gp(actual_cov, limits, observation_noise, test_size = 300L, training_size = 10L,
    random_training = TRUE, show_pred = FALSE, ...)  --  R/simulation.R:212-255: the ground truth is itself a draw from
    a Gaussian process with covariance function `actual_cov` (a cov_func closure) on the test grid."""
    if not callable(actual_cov) or not callable(observation_noise):
        raise TypeError("is.function(actual_cov), is.function(observation_noise) are not all TRUE")
    lim = _limits(limits)
    if not (test_size > 0 and 0 < training_size < test_size):
        raise ValueError("test_size > 0, training_size > 0, training_size < test_size are not all TRUE")
    if not isinstance(random_training, (bool, np.bool_)):
        raise TypeError("is.logical(random_training) is not TRUE")
    D = lim.shape[0]
    rng = rng if rng is not None else np.random.default_rng()
    testpoints = _grid(lim, test_size)                                                       # :223-224
    K = covariance_matrix(testpoints, testpoints, actual_cov)                                # :225
    f = multivariate_normal(1, np.zeros(K.shape[0]), K, rng=rng, z=z)[:, 0]                  # :226
    ncol = testpoints.shape[1]
    if random_training:
        training_set = rng.choice(ncol, int(training_size), replace=False)                   # sample.int  :229
    else:
        training_set = np.arange(1, int(training_size) + 1) * (ncol // int(training_size)) - 1   # :230 (1-based in R)
    X = testpoints[:, training_set]
    y = f[training_set] + observation_noise(X)                                               # :232
    regression_gp = GPR(X, np.asarray(y, dtype=np.float64).ravel(), **gpr_args)              # :233
    prediction = regression_gp.predict(testpoints)                                           # :236-242 (the 1-D plot uses the same grid)
    residual = prediction[:, 0] - f
    _say("The mean absolute difference of predictions and ground truth in the considered limits is  %.15g" % np.mean(np.abs(residual)))
    return Summary.of(np.abs(residual), model=regression_gp, testpoints=testpoints, f=f, training_set=training_set,
                      prediction=prediction, residual=residual, variance=prediction[:, 1])


def simulate_classification(func, limits, training_points=None, training_size=10, test_size=10000, *, rng=None, **gpc_args):
    """simulate_classification(func, limits, training_points, training_size = 10L, test_size = 10000L, ...)  --
    R/simulation.R:316-336.  `...` goes to GPC$new (k, epsilon).  Labels func(x) are -1 / +1."""
    if not callable(func):
        raise TypeError("is.function(func) is not TRUE")
    lim = _limits(limits)
    if not training_size > 0:
        raise ValueError("training_size > 0 is not TRUE")
    D = lim.shape[0]
    rng = rng if rng is not None else np.random.default_rng()
    if training_points is None:
        training_points = np.vstack([rng.uniform(lim[i, 0], lim[i, 1], int(training_size)) for i in range(D)])
    else:
        training_points = np.asarray(training_points, dtype=np.float64).reshape(D, -1) if np.ndim(training_points) < 2 \
            else np.asarray(training_points, dtype=np.float64)
        if training_points.shape[0] != D:
            raise ValueError("nrow(limits) == nrow(training_points) is not TRUE")
    y = _apply_cols(training_points, func)                                                   # :327
    gaussian = GPC(training_points, y, **gpc_args)                                           # :328
    test_points = _grid(lim, test_size)                                                      # :331-332
    predictions = gaussian.predict_class(test_points)                                        # GPC$plot(testpoints)$pred  :333-334
    residual = 2 * (predictions >= 0.5).astype(np.float64) - 1 - _apply_cols(test_points, func)   # :335
    _say("Proportion of misclassified test points. %.15g" % (np.mean(np.abs(residual)) / 2))
    return Summary.of(np.abs(residual), model=gaussian, test_points=test_points, predictions=predictions, residual=residual)

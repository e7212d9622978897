"""gprc on MI355X: the GP predict hot path of the R package `gprc`
(MoHawastaken/Gaussian-Process-Regression) behind the reference's own API surface.

Host mirror of the R6 / cov_func interface (reference R/GPRclass.R, R/GPCclass.R) over the C ABI of
include/gprc_native.h; all matrix arithmetic runs in hand-written HIP kernels for gfx950.  There is no
CPU fallback: without the built library / an MI355X the compute calls raise.
"""
from . import _native
from ._native import GprcError, NotPositiveDefinite, Context, default_context, device_count
from .covfunc import (cov_func, covariance_matrix, constant, linear, polynomial, sqrexp, gammaexp,
                      rationalquadratic, CovFunc)
from .gpr import (GPR, GPR_constant, GPR_linear, GPR_polynomial, GPR_sqrexp, GPR_gammaexp,
                  GPR_rationalquadratic)
from .gpc import GPC
from .fit import fit, dens, dens_deriv
from .sampling import multivariate_normal, expand_range, mvn_factor, sym_eigen
from .simulation import combine_all, iid_noise, simulate_regression, simulate_regression_gp, simulate_classification

__all__ = ["fit", "dens", "dens_deriv", "multivariate_normal", "expand_range", "mvn_factor", "sym_eigen", "combine_all", "iid_noise",
           "simulate_regression", "simulate_regression_gp", "simulate_classification", "GPR", "GPR_constant", "GPR_linear", "GPR_polynomial", "GPR_sqrexp", "GPR_gammaexp",
           "GPR_rationalquadratic", "GPC", "cov_func", "covariance_matrix", "constant", "linear", "polynomial",
           "sqrexp", "gammaexp", "rationalquadratic", "CovFunc", "GprcError", "NotPositiveDefinite", "Context",
           "default_context", "device_count"]

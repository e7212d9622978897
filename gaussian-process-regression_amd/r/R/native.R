# native.R -- R-side dispatch onto libgprc_native.so for the reference package `gprc`.
#
# NOT EXERCISED IN THIS REPOSITORY (no R toolchain in the build image).  It shows the few lines a
# maintainer adds next to R/GPRclass.R / R/GPCclass.R; everything else in the package stays as it is.
# Dispatch rule: a covariance function built by cov_func() from one of the six package kernels carries
# attr(k, "gprc_kernel") = list(id, params); with that tag and a visible MI355X the hot path runs
# natively, otherwise the original R expressions run unchanged (arbitrary user closures, no GPU).

.gprc_kernel_ids <- c(constant = 0L, linear = 1L, polynomial = 2L, sqrexp = 3L, gammaexp = 4L, rationalquadratic = 5L)
# parameter order of the native ABI (include/gprc_native.h)
.gprc_param_order <- list(constant = "c", linear = "sigma", polynomial = c("sigma", "p"), sqrexp = "l",
                          gammaexp = c("l", "gamma"), rationalquadratic = c("l", "alpha"))

gprc_native_available <- function() {
  isTRUE(getOption("gprc.backend", "auto") != "R") && .Call(gprc_R_device_count) > 0L
}

# replaces cov_func (R/GPRclass.R:424-427): same closure, plus the tag
cov_func <- function(func, ...) {
  force(func)
  args <- list(...)
  k <- function(x, y) func(x, y, ...)
  name <- names(Filter(function(f) identical(f, func), mget(names(.gprc_kernel_ids), envir = environment(cov_func))))
  if (length(name) == 1L) {
    ord <- .gprc_param_order[[name]]
    if (is.null(names(args))) names(args) <- ord[seq_along(args)]
    if (all(ord %in% names(args)))
      attr(k, "gprc_kernel") <- list(id = .gprc_kernel_ids[[name]], params = as.double(unlist(args[ord])))
  }
  k
}

# replaces covariance_matrix (R/GPRclass.R:355-357)
covariance_matrix <- function(A, B, covariance_function) {
  tag <- attr(covariance_function, "gprc_kernel")
  if (!is.null(tag) && gprc_native_available()) {
    storage.mode(A) <- "double"; storage.mode(B) <- "double"
    return(.Call(gprc_R_kernel_matrix, tag$id, tag$params, A, B))
  }
  outer(1:ncol(A), 1:ncol(B), function(i, j) covariance_function(A[, i, drop = F], B[, j, drop = F]))
}

# Multi-GPU: options(gprc.devices = 0:7) makes GPR$new / $predict(pointwise_var = TRUE) run the block-cyclic sweep over
# those GPUs from this one R process (gprc_mgpu_* of include/gprc_native.h: R cannot be forked per GPU);
# options(gprc.rccl = TRUE) selects the RCCL broadcast for the panel exchange (default: peer copies);
# options(gprc.exchange = "scatter_allgather" | "auto") the large-message form of that step (GPRC_MGPU_SCATTER_ALLGATHER /
# GPRC_MGPU_AUTO_EXCHANGE: timed against the rooted broadcast when the ranks are created, the faster kept).
.gprc_devices <- function() {
  dv <- getOption("gprc.devices", NULL)
  if (is.null(dv) || length(dv) < 2L) NULL else as.integer(dv)
}
.gprc_mgpu_flags <- function() {
  ex <- match.arg(getOption("gprc.exchange", "broadcast"), c("broadcast", "scatter_allgather", "auto"))
  (if (isTRUE(getOption("gprc.rccl", FALSE))) 1L else 0L) + c(broadcast = 0L, scatter_allgather = 4L, auto = 8L)[[ex]]
}

# body of GPR$initialize after the input checks (R/GPRclass.R:134-153), native branch
.gpr_initialize_native <- function(private, X, y, noise, k) {
  tag <- attr(k, "gprc_kernel")
  storage.mode(X) <- "double"
  dv <- .gprc_devices()
  res <- if (is.null(dv)) .Call(gprc_R_gpr_fit, tag$id, tag$params, X, as.double(y), as.double(noise))
         else .Call(gprc_R_mgpu_gpr_fit, dv, .gprc_mgpu_flags(), tag$id, tag$params, X,
                    as.double(y), as.double(noise))
  private$.multi <- !is.null(dv)
  if (res[[3]] > 1L) warning(sprintf("Noise got changed to %s to avoid errors in cholesky decomposition", res[[2]]))
  private$.handle <- res[[1]]
  private$.noise <- res[[2]]
  private$.alpha <- res[[4]]
  private$.logp <- matrix(res[[5]], 1, 1)
  private$.L <- NULL            # fetched by the `L` active binding on first read: .Call(gprc_R_model_L, handle)
}

# body of GPR$predict (R/GPRclass.R:160-169), native branch
.gpr_predict_native <- function(private, X_star, pointwise_var) {
  storage.mode(X_star) <- "double"
  if (isTRUE(private$.multi)) {
    if (!isTRUE(pointwise_var)) stop("gprc: the multi-GPU handle predicts pointwise variances; use one GPU for the full covariance")
    return(.Call(gprc_R_mgpu_gpr_predict, private$.handle, X_star))
  }
  .Call(gprc_R_gpr_predict, private$.handle, X_star, isTRUE(pointwise_var))
}

# body of GPC$initialize (R/GPCclass.R:73-103), native branch
.gpc_initialize_native <- function(private, X, y, k, epsilon) {
  tag <- attr(k, "gprc_kernel")
  storage.mode(X) <- "double"
  res <- .Call(gprc_R_gpc_fit, tag$id, tag$params, X, as.double(y), as.double(epsilon))
  message(sprintf("Convergence after %s iterations", res[[4]]))
  private$.handle <- res[[1]]
  private$.f_hat <- res[[2]]
  private$.logq <- res[[3]]
}

# GPC$predict_class (R/GPCclass.R:108-118), native branch: fs_bar / Vfs on the GPU, integrate() as before
.gpc_predict_class_native <- function(private, X_star) {
  if (!is.matrix(X_star)) dim(X_star) <- c(1, length(X_star))
  storage.mode(X_star) <- "double"
  lat <- .Call(gprc_R_gpc_predict_latent, private$.handle, X_star)
  sapply(seq_len(nrow(lat)), function(i) integrate(function(z)
    private$.sigmoid(z) * dnorm(z, mean = lat[i, 1], sd = lat[i, 2]), -Inf, Inf)$value)
}

# fit(): the two closures optim() is given (R/fit.R:117-139), native branch.  `id` is the kernel id of cov_dict[[cov]]$func
# (X is coerced ONCE, outside the closures: REAL() on an integer matrix would error or read garbage in the shim)
.dens_native <- function(id, X, y, noise) {
  storage.mode(X) <- "double"
  y <- as.double(y)
  function(v) .Call(gprc_R_log_marginal, id, as.double(v), X, y, as.double(noise))
}
.dens_deriv_native <- function(id, X, y) {
  storage.mode(X) <- "double"
  y <- as.double(y)
  function(v) .Call(gprc_R_fit_gradient, id, as.double(v), X, y)
}

# multivariate_normal (R/GPRclass.R:360-370), native branch: rnorm() stays in R, the factorisation and L %*% Z move
multivariate_normal <- function(n, mean, covariance, tol = 1e-6) {
  stopifnot(length(mean) == nrow(covariance))
  Z <- matrix(rnorm(n * length(mean), 0, 1), nrow = length(mean))
  if (!gprc_native_available()) return(drop(mean) + .mvn_factor_R(covariance, tol) %*% Z)   # the original lines :362-368
  storage.mode(covariance) <- "double"                                                       # keeps dim(); an integer matrix must not reach REAL()
  .Call(gprc_R_mvn_sample, as.double(mean), covariance, as.double(tol), Z)
}

# combine_all (R/simulation.R:338-349), native branch
combine_all <- function(lst) {
  if (!gprc_native_available()) return(.combine_all_R(lst))                                   # the original body
  .Call(gprc_R_combine_all, as.double(unlist(lst)), as.double(lengths(lst)))
}

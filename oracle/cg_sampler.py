"""Oracle restatement of the CG sampler (test infrastructure).

`cg_sample` follows ConjugateGradientSampler.sample
(reg_coef_sampler/cg_sampler.py:20-94) with precond_by='prior'
(cg_sampler.py:128-138) and precondition_linear_system (cg_sampler.py:96-113).

`scipy_style_cg` restates the third-party solver the reference calls at
cg_sampler.py:77-80: `scipy.sparse.linalg.cg` (SciPy is an unpinned dependency,
setup.py:66-69).  Published algorithm, SciPy 1.15.3
(`scipy/sparse/linalg/_isolve/iterative.py::cg`, M = identity):

    atol_eff = max(atol, rtol * ||b||)
    r = b - A x0  if x0 has a non-zero else b.copy()
    for k in range(maxiter):
        if ||r|| < atol_eff: return x, 0
        rho = r.r
        p = r + (rho / rho_prev) p    (k > 0)      |   p = r   (k == 0)
        q = A p ; alpha = rho / (p.q)
        x += alpha p ; r -= alpha q ; callback(x)
    return x, maxiter

The reference passes `tol = atol/||b||`; legacy `tol` is relative, so the
effective stop rule is ||r|| < atol (up to one rounding).  `use_scipy=True`
routes through the installed SciPy instead (used by tests to show the
restatement and the library agree, and by the CPU baseline so that its timing
reflects the primitives the reference really runs).
"""
import numpy as np


def prior_preconditioner(prior_prec_sqrt, coef_scaled_sd, n_unshrunk):
    """cg_sampler.py:128-138.  s = prior sd on shrunk coordinates and
    2 * (estimated posterior sd) on the unshrunk ones."""
    s = np.ones(len(prior_prec_sqrt))
    s[n_unshrunk:] = prior_prec_sqrt[n_unshrunk:] ** -1
    if n_unshrunk > 0:
        s[:n_unshrunk] = 2.0 * np.asarray(coef_scaled_sd)[:n_unshrunk]
    return s


def scipy_style_cg(matvec, b, x0, rtol, atol, maxiter, callback=None):
    b = np.asarray(b, dtype=np.float64)
    x = np.array(x0, dtype=np.float64, copy=True)
    bnrm2 = np.linalg.norm(b)
    atol_eff = max(float(atol), float(rtol) * float(bnrm2))
    if bnrm2 == 0:
        return b.copy(), 0
    r = b - matvec(x) if x.any() else b.copy()
    rho_prev, p = None, None
    for k in range(maxiter):
        if np.linalg.norm(r) < atol_eff:
            return x, 0
        rho = np.dot(r, r)
        if k > 0:
            p = r + (rho / rho_prev) * p
        else:
            p = r.copy()
        q = matvec(p)
        alpha = rho / np.dot(p, q)
        x += alpha * p
        r -= alpha * q
        rho_prev = rho
        if callback is not None:
            callback(x)
    return x, maxiter


def cg_sample(design, obs_prec, prior_prec_sqrt, z, coef_cg_init,
              coef_scaled_sd, n_unshrunk, randn_n, randn_P, maxiter, atol,
              use_scipy=False, return_details=False):
    """One draw from N(Sigma z, Sigma), Sigma^-1 = X~' Omega X~ + diag(phi^2).

    randn_n, randn_P are the two standard-normal vectors the reference draws
    from the global NumPy stream at cg_sampler.py:61-62 (n first, then P);
    passing them in makes the function a pure map that the HIP path can be
    compared with on identical inputs.
    """
    n, P = design.shape
    obs_prec = np.broadcast_to(np.asarray(obs_prec, dtype=np.float64), (n,))
    phi = np.asarray(prior_prec_sqrt, dtype=np.float64)
    s = prior_preconditioner(phi, coef_scaled_sd, n_unshrunk)
    d = (s * phi) ** 2                                   # cg_sampler.py:104

    def operator(x):                                     # cg_sampler.py:105-108
        return d * x + s * design.Tdot(obs_prec * design.dot(s * x))

    v = design.Tdot(obs_prec ** 0.5 * randn_n) + phi * randn_P   # :66-67
    b = s * (np.asarray(z, dtype=np.float64) + v)                # :68
    rtol = atol / np.linalg.norm(b)                              # :75
    x0 = np.asarray(coef_cg_init, dtype=np.float64) / s          # :76
    counter = {'n_iter': 0}

    def count(_):
        counter['n_iter'] += 1

    if use_scipy:
        import scipy.sparse.linalg as spla
        op = spla.LinearOperator((P, P), matvec=operator)
        x, info = spla.cg(op, b, x0=x0, maxiter=maxiter, rtol=rtol, atol=0.,
                          callback=count)
    else:
        x, info = scipy_style_cg(operator, b, x0, rtol, 0., maxiter, count)
    coef = s * x                                                 # :89
    cg_info = {'n_iter': counter['n_iter'], 'valid_input': info >= 0,
               'converged': info == 0}
    if return_details:
        cg_info.update(b=b, s=s, d=d, x_scaled=x)
    return coef, cg_info

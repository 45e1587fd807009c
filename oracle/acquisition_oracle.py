"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the MaxVar-family acquisition surfaces.

Only tests/ may import this; nothing under elfi_amd/ does (the product evaluates these surfaces on the device:
elfihip_gp_maxvar / elfihip_gp_expintvar, csrc/gp_predict.hip + csrc/special.hpp).

Follows the reference line by line, with the SciPy functions it calls:
    MaxVar.evaluate            elfi/methods/bo/acquisition.py:392-417
    MaxVar.evaluate_gradient   elfi/methods/bo/acquisition.py:419-463
    ExpIntVar.evaluate         elfi/methods/bo/acquisition.py:795-821  (the covariance of :800-808 is handed in)
PARITY: pinned -- tests/golden/maxvar.npz holds the outputs of the reference's own classes (oracle/make_golden_posterior.py)
and tests/test_maxvar.py checks this module against them.
"""
import numpy as np
import scipy.stats as ss


def maxvar_value(mean, var, sigma2_n, eps, prior_pdf):
    """acquisition.py:403-417; mean / var (S, 1) noiseless GP prediction, prior_pdf (S,)."""
    a = np.sqrt(sigma2_n) / np.sqrt(sigma2_n + 2. * var)
    scale = np.sqrt(sigma2_n + var)
    phi_skew = ss.skewnorm.cdf(eps, a, loc=mean, scale=scale)
    phi_norm = ss.norm.cdf(eps, loc=mean, scale=scale)
    var_p_a = phi_skew - phi_norm ** 2
    val_prior = np.asarray(prior_pdf).ravel()[:, np.newaxis]
    return val_prior ** 2 * var_p_a


def maxvar_gradient(mean, var, grad_mean, grad_var, sigma2_n, eps, prior_pdf, prior_grad_logpdf):
    """acquisition.py:434-463."""
    phi = ss.norm.cdf
    scale = np.sqrt(sigma2_n + var)
    a = (eps - mean) / scale
    b = np.sqrt(sigma2_n) / np.sqrt(sigma2_n + 2 * var)
    grad_a = (-1. / scale) * grad_mean - ((eps - mean) / (2. * (sigma2_n + var) ** (1.5))) * grad_var
    grad_b = (-np.sqrt(sigma2_n) / (sigma2_n + 2 * var) ** (1.5)) * grad_var
    _phi_a = phi(a)
    int_1 = _phi_a - _phi_a ** 2
    int_2 = phi(eps, loc=mean, scale=scale) - ss.skewnorm.cdf(eps, b, loc=mean, scale=scale)
    grad_int_1 = (1. - 2 * _phi_a) * (np.exp(-.5 * (a ** 2)) / np.sqrt(2. * np.pi)) * grad_a
    grad_int_2 = (1. / np.pi) * (((np.exp(-.5 * (a ** 2) * (1. + b ** 2))) / (1. + b ** 2)) * grad_b
                                 + (np.sqrt(np.pi / 2.) * np.exp(-.5 * (a ** 2)) * (1. - 2. * phi(a * b)) * grad_a))
    term_prior = np.asarray(prior_pdf).ravel()[:, np.newaxis]
    term_grad_prior = term_prior * np.asarray(prior_grad_logpdf)
    return 2. * term_prior * (int_1 - int_2) * term_grad_prior + term_prior ** 2 * (grad_int_1 - grad_int_2)


def expintvar_loss(cov_int, var_new, sigma2_n, eps, w_int, mean_int, var_int):
    """acquisition.py:809-819 with cov_int (S, M) the posterior covariance between candidates and integration points,
    var_new (S,) the candidates' noiseless variance, w_int (M,) = omegas_int * priors_int, mean_int / var_int (M,)."""
    var_new = np.asarray(var_new).reshape(-1, 1)
    mean_int, var_int = np.asarray(mean_int).reshape(1, -1), np.asarray(var_int).reshape(1, -1)
    delta_var_int = cov_int ** 2 / (sigma2_n + var_new)
    a = np.sqrt((sigma2_n + var_int - delta_var_int) / (sigma2_n + var_int + delta_var_int))
    phi_int = ss.norm.cdf(eps, loc=mean_int, scale=np.sqrt(sigma2_n + var_int))
    phi_skew_imp = ss.skewnorm.cdf(eps, a, loc=mean_int, scale=np.sqrt(sigma2_n + var_int))
    w = ((phi_int - phi_skew_imp) / 2)
    return 2 * np.sum(np.asarray(w_int).reshape(1, -1) * w, axis=1)

// ttm_bfgs.h - dense BFGS for the per-component problems of integrated-rectifier maps.
//
// The reference minimises every component of an integrated-rectifier map with scipy.optimize.minimize(method='BFGS',
// fun=objective_function, jac=objective_function_jacobian) (TM:3252-3257): SciPy's quasi-Newton loop on the inverse
// Hessian (H <- (I - rho s y')H(I - rho y s') + rho s s', gtol 1e-5 on the max-norm of the gradient, 200 n iterations)
// with a More-Thuente line search (MINPACK-2 dcsrch: c1 1e-4, c2 0.9, xtol 1e-14, steps in [1e-100, 1e100], first trial
// min(1, 2.02 (f_k - f_{k-1}) / g'p)) and, when that reports failure, the bracketing / zoom search of Nocedal & Wright
// (Alg. 3.5 / 3.6: cubic, then quadratic interpolation, then bisection; at most 10 + 10 trials).  This is that
// published method as a C++ host loop over an objective that returns value and gradient together (one device
// reduction per trial point instead of one Python round trip for f and one for g); the line-search kernel dcsrch /
// dcstep is shared with ttm_lbfgsb.h.  Same iterates up to the rounding of the matrix products
// (tests/test_native_bfgs.py compares them with SciPy's).
#pragma once

#include <math.h>

#include <vector>

#include "ttm_lbfgsb.h"

namespace ttm_opt {

struct BfgsOptions {
    double gtol = 1e-5;
    double c1 = 1e-4, c2 = 0.9;
    int maxiter = 0;             // 0: 200 n
};

struct BfgsResult {
    double f = 0.0;
    double gnorm = 0.0;
    int nit = 0;
    int nfev = 0;
    int status = 0;              // 0 converged, 1 iteration limit, 2 precision loss (no acceptable step), 3 NaN, -1 the objective failed
};

namespace detail {

// one trial point of a line search: phi(a) = f(x + a p), phi'(a) = grad f(x + a p) . p, evaluated together and kept
struct Ray {
    int n;
    const double *x, *p;
    const ObjectiveFn& fun;
    std::vector<double> xt, g;   // the last point evaluated and its gradient
    double a_last = NAN, f_last = 0.0, d_last = 0.0;
    int nfev = 0, rc = 0;
    Ray(int n_, const double* x_, const double* p_, const ObjectiveFn& fun_) : n(n_), x(x_), p(p_), fun(fun_), xt(n_), g(n_) {}
    bool eval(double a) {
        if (a == a_last) return rc == 0;
        for (int i = 0; i < n; ++i) xt[i] = x[i] + a * p[i];
        rc = fun(xt.data(), &f_last, g.data());
        ++nfev;
        double d = 0.0;
        for (int i = 0; i < n; ++i) d += g[i] * p[i];
        d_last = d;
        a_last = a;
        return rc == 0;
    }
};

inline bool cubicmin(double a, double fa, double fpa, double b, double fb, double c, double fc, double& xmin) {
    const double C = fpa, db = b - a, dc = c - a;
    const double denom = (db * dc) * (db * dc) * (db - dc);
    if (denom == 0.0 || !isfinite(denom)) return false;
    const double r0 = fb - fa - C * db, r1 = fc - fa - C * dc;
    double A = dc * dc * r0 - db * db * r1, B = -dc * dc * dc * r0 + db * db * db * r1;
    A /= denom;
    B /= denom;
    const double radical = B * B - 3 * A * C;
    if (!(radical >= 0.0) || A == 0.0) return false;
    xmin = a + (-B + sqrt(radical)) / (3 * A);
    return isfinite(xmin);
}

inline bool quadmin(double a, double fa, double fpa, double b, double fb, double& xmin) {
    const double db = b - a;
    if (db == 0.0) return false;
    const double B = (fb - fa - fpa * db) / (db * db);
    if (B == 0.0 || !isfinite(B)) return false;
    xmin = a - fpa / (2.0 * B);
    return isfinite(xmin);
}

// zoom phase (Nocedal & Wright Alg. 3.6); true with the accepted step in a_star (the ray holds its value and gradient)
inline bool zoom(Ray& ray, double a_lo, double a_hi, double phi_lo, double phi_hi, double derphi_lo, double phi0, double derphi0,
                 double c1, double c2, double& a_star) {
    const double delta1 = 0.2, delta2 = 0.1;
    double phi_rec = phi0, a_rec = 0.0;
    for (int i = 0; i <= 10; ++i) {
        const double dalpha = a_hi - a_lo;
        const double a = dalpha < 0 ? a_hi : a_lo, b = dalpha < 0 ? a_lo : a_hi;
        double a_j = 0.0;
        bool have = false;
        if (i > 0) {
            const double cchk = delta1 * dalpha;
            have = cubicmin(a_lo, phi_lo, derphi_lo, a_hi, phi_hi, a_rec, phi_rec, a_j) && !(a_j > b - cchk) && !(a_j < a + cchk);
        }
        if (!have) {
            const double qchk = delta2 * dalpha;
            have = quadmin(a_lo, phi_lo, derphi_lo, a_hi, phi_hi, a_j) && !(a_j > b - qchk) && !(a_j < a + qchk);
            if (!have) a_j = a_lo + 0.5 * dalpha;
        }
        if (!ray.eval(a_j)) return false;
        const double phi_aj = ray.f_last;
        if (phi_aj > phi0 + c1 * a_j * derphi0 || phi_aj >= phi_lo) {
            phi_rec = phi_hi; a_rec = a_hi;
            a_hi = a_j; phi_hi = phi_aj;
        } else {
            const double derphi_aj = ray.d_last;
            if (fabs(derphi_aj) <= -c2 * derphi0) { a_star = a_j; return true; }
            if (derphi_aj * (a_hi - a_lo) >= 0) {
                phi_rec = phi_hi; a_rec = a_hi;
                a_hi = a_lo; phi_hi = phi_lo;
            } else {
                phi_rec = phi_lo; a_rec = a_lo;
            }
            a_lo = a_j; phi_lo = phi_aj; derphi_lo = derphi_aj;
        }
    }
    return false;
}

// first trial step of both searches
inline double first_step(double phi0, double old_phi0, double derphi0) {
    double a1 = 1.0;
    if (derphi0 != 0) {
        a1 = fmin(1.0, 1.01 * 2 * (phi0 - old_phi0) / derphi0);
        if (a1 < 0 || a1 != a1) a1 = 1.0;
    }
    return a1;
}

// More-Thuente search on the ray; true with the step (value and gradient of the accepted point in the ray)
inline bool search_wolfe1(Ray& ray, double phi0, double old_phi0, double derphi0, double c1, double c2, double& stp) {
    stp = first_step(phi0, old_phi0, derphi0);
    LineSearch ls;
    double f = phi0, d = derphi0;
    for (int i = 0; i < 100; ++i) {
        dcsrch(ls, f, d, stp, c1, c2, 1e-14, 1e-100, 1e100);
        if (!isfinite(stp)) return false;
        if (ls.task != LineSearch::FG) break;
        if (!ray.eval(stp)) return false;
        f = ray.f_last;
        d = ray.d_last;
        if (i == 99) return false;                               // did not converge within 100 calls
    }
    return ls.task == LineSearch::CONVERGENCE;
}

// bracketing search (Nocedal & Wright Alg. 3.5).  0: failed, 1: step with its gradient in the ray, 2: step accepted at the
// trial limit (gradient to be evaluated by the caller)
inline int search_wolfe2(Ray& ray, double phi0, double old_phi0, double derphi0, double c1, double c2, double amax, double& stp) {
    double alpha0 = 0.0, alpha1 = fmin(first_step(phi0, old_phi0, derphi0), amax);
    if (!ray.eval(alpha1)) return 0;
    double phi_a1 = ray.f_last, phi_a0 = phi0, derphi_a0 = derphi0;
    for (int i = 0; i < 10; ++i) {
        if (alpha1 == 0 || alpha0 > amax) return 0;
        if (phi_a1 > phi0 + c1 * alpha1 * derphi0 || (phi_a1 >= phi_a0 && i > 0))
            return zoom(ray, alpha0, alpha1, phi_a0, phi_a1, derphi_a0, phi0, derphi0, c1, c2, stp) ? 1 : 0;
        if (!ray.eval(alpha1)) return 0;                         // (the same point: its gradient is already there)
        const double derphi_a1 = ray.d_last;
        if (fabs(derphi_a1) <= -c2 * derphi0) { stp = alpha1; return 1; }
        if (derphi_a1 >= 0) return zoom(ray, alpha1, alpha0, phi_a1, phi_a0, derphi_a1, phi0, derphi0, c1, c2, stp) ? 1 : 0;
        const double alpha2 = fmin(2 * alpha1, amax);
        alpha0 = alpha1;
        alpha1 = alpha2;
        phi_a0 = phi_a1;
        if (!ray.eval(alpha1)) return 0;
        phi_a1 = ray.f_last;
        derphi_a0 = derphi_a1;
    }
    stp = alpha1;
    return 2;
}

}  // namespace detail

inline BfgsResult bfgs_minimize(int n, double* x, const ObjectiveFn& fun, const BfgsOptions& opt = BfgsOptions()) {
    using namespace detail;
    BfgsResult res;
    const int maxiter = opt.maxiter > 0 ? opt.maxiter : 200 * n;
    std::vector<double> g(n), gn(n), p(n), s(n), y(n), H((size_t)n * n, 0.0), T((size_t)n * n), A((size_t)n * n);
    double fval = 0.0;
    if (fun(x, &fval, g.data())) { res.status = -1; return res; }
    res.nfev = 1;
    for (int i = 0; i < n; ++i) H[(size_t)i * n + i] = 1.0;
    double g2 = 0.0;
    for (int i = 0; i < n; ++i) g2 += g[i] * g[i];
    double old_old = fval + sqrt(g2) / 2;                        // first trial step ~ 1 in x
    auto maxnorm = [&](const std::vector<double>& v) { double m = 0.0; for (int i = 0; i < n; ++i) m = fmax(m, fabs(v[i])); for (int i = 0; i < n; ++i) if (v[i] != v[i]) m = NAN; return m; };
    double gnorm = maxnorm(g);
    int k = 0, warn = 0;
    while (gnorm > opt.gtol && k < maxiter) {
        for (int i = 0; i < n; ++i) {
            double acc = 0.0;
            for (int j = 0; j < n; ++j) acc += H[(size_t)i * n + j] * g[j];
            p[i] = -acc;
        }
        double derphi0 = 0.0;
        for (int i = 0; i < n; ++i) derphi0 += g[i] * p[i];
        Ray ray(n, x, p.data(), fun);
        double alpha = 0.0;
        bool ok = search_wolfe1(ray, fval, old_old, derphi0, opt.c1, opt.c2, alpha);
        if (ray.rc) { res.status = -1; res.nfev += ray.nfev; return res; }
        if (!ok) {
            const int r2 = search_wolfe2(ray, fval, old_old, derphi0, opt.c1, opt.c2, 1e100, alpha);
            if (ray.rc) { res.status = -1; res.nfev += ray.nfev; return res; }
            ok = r2 != 0;
        }
        if (!ok) { res.nfev += ray.nfev; warn = 2; break; }
        if (!ray.eval(alpha)) { res.status = -1; res.nfev += ray.nfev; return res; }     // (a no-op unless the trial limit accepted the step)
        res.nfev += ray.nfev;
        old_old = fval;
        fval = ray.f_last;
        double pn = 0.0, xn = 0.0;
        for (int i = 0; i < n; ++i) {
            s[i] = alpha * p[i];
            x[i] = x[i] + s[i];
            gn[i] = ray.g[i];
            y[i] = gn[i] - g[i];
            g[i] = gn[i];
            pn += p[i] * p[i];
            xn += x[i] * x[i];
        }
        ++k;
        gnorm = maxnorm(g);
        if (gnorm <= opt.gtol) break;
        if (alpha * sqrt(pn) <= 0.0 * (0.0 + sqrt(xn))) break;
        if (!isfinite(fval)) { warn = 2; break; }
        double rinv = 0.0;
        for (int i = 0; i < n; ++i) rinv += y[i] * s[i];
        const double rho = rinv == 0.0 ? 1000.0 : 1.0 / rinv;
        // H <- (I - rho s y') H (I - rho y s') + rho s s'
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) A[(size_t)i * n + j] = (i == j ? 1.0 : 0.0) - y[i] * s[j] * rho;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double acc = 0.0;
                for (int l = 0; l < n; ++l) acc += H[(size_t)i * n + l] * A[(size_t)l * n + j];
                T[(size_t)i * n + j] = acc;
            }
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) A[(size_t)i * n + j] = (i == j ? 1.0 : 0.0) - s[i] * y[j] * rho;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double acc = 0.0;
                for (int l = 0; l < n; ++l) acc += A[(size_t)i * n + l] * T[(size_t)l * n + j];
                H[(size_t)i * n + j] = acc + rho * s[i] * s[j];
            }
    }
    res.f = fval;
    res.gnorm = gnorm;
    res.nit = k;
    if (warn == 2) res.status = 2;
    else if (k >= maxiter) res.status = 1;
    else if (gnorm != gnorm || fval != fval) res.status = 3;
    else res.status = 0;
    for (int i = 0; i < n && res.status == 0; ++i) if (x[i] != x[i]) res.status = 3;
    return res;
}

}  // namespace ttm_opt

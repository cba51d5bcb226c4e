// ttm_lbfgsb.h - bound-constrained limited-memory BFGS for the per-component problems of optimize().
//
// The reference minimises every map component with scipy.optimize.minimize(method='L-BFGS-B') (TM:3108-3114: maxcor 10,
// ftol 2.22e-9, gtol 1e-5, bounds c >= 0 on all monotone coefficients but the constant's), i.e. with L-BFGS-B 3.0
// (Byrd, Lu, Nocedal, Zhu 1995; Zhu, Byrd, Lu, Nocedal 1997; Morales, Nocedal 2011).  Driving that from Python costs a
// host round trip per objective evaluation (~80 us against ~10 us of device work); this is the same algorithm as a C++
// host loop that calls the device reduction directly.
//
// It is a restatement of the published method, step for step - generalised Cauchy point along the projected
// steepest-descent path, subspace minimisation over the free variables with the projection / backtracking of version
// 3.0, More-Thuente line search (dcsrch / dcstep of MINPACK-2) with ftol 1e-3, gtol 0.9, xtol 0.1, the curvature test
// s'y > eps (-g'd) for accepting a pair, theta = y'y / s'y, both stopping tests - with ONE liberty: the problems here have
// at most a few dozen variables, so the limited-memory matrix B = theta I - W M W' is formed densely (the stored pairs
// applied to theta I as BFGS updates, which is the same matrix) and the reduced systems are solved by Cholesky
// factorisation instead of through the compact representation.  Same iterates up to rounding; the tests compare them
// with SciPy's (tests/test_native_lbfgsb.py).
#pragma once

#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <functional>
#include <vector>

namespace ttm_opt {

struct LbfgsbOptions {
    int maxcor = 10;
    double factr = 1e7;          // ftol / machine epsilon (SciPy: ftol = 2.2204460492503131e-09)
    double pgtol = 1e-5;
    int maxiter = 15000;
    int maxfun = 15000;
    int maxls = 20;
};

struct LbfgsbResult {
    double f = 0.0;
    double pgnorm = 0.0;
    int nit = 0;
    int nfev = 0;
    int status = 0;              // 0 converged (projected gradient), 1 converged (relative reduction), 2 iteration / evaluation
                                 // limit, 3 abnormal termination in the line search, -1 the objective failed
};

// f and gradient at x; returns non-zero on failure
typedef std::function<int(const double* x, double* f, double* g)> ObjectiveFn;

namespace detail {

const double kEps = 2.220446049250313e-16;

// ---- More-Thuente line search (MINPACK-2 dcsrch / dcstep) ------------------------------------------------------------
struct LineSearch {
    bool brackt = false;
    int stage = 1;
    double ginit = 0, gtest = 0, gx = 0, gy = 0, finit = 0, fx = 0, fy = 0, stx = 0, sty = 0, stmin = 0, stmax = 0, width = 0, width1 = 0;
    enum Task { START, FG, CONVERGENCE, WARNING, ERROR } task = START;
};

inline void dcstep(double& stx, double& fx, double& dx, double& sty, double& fy, double& dy, double& stp, double fp, double dp,
                   bool& brackt, double stpmin, double stpmax) {
    const double sgnd = dp * (dx / fabs(dx));
    double stpf;
    if (fp > fx) {                                   // case 1: higher function value: the minimum is bracketed
        const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
        const double s = std::max(fabs(theta), std::max(fabs(dx), fabs(dp)));
        double gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
        if (stp < stx) gamma = -gamma;
        const double p = (gamma - dx) + theta, q = ((gamma - dx) + gamma) + dp, r = p / q;
        const double stpc = stx + r * (stp - stx);
        const double stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx);
        stpf = fabs(stpc - stx) < fabs(stpq - stx) ? stpc : stpc + (stpq - stpc) / 2.0;
        brackt = true;
    } else if (sgnd < 0.0) {                         // case 2: derivatives of opposite sign: bracketed
        const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
        const double s = std::max(fabs(theta), std::max(fabs(dx), fabs(dp)));
        double gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
        if (stp > stx) gamma = -gamma;
        const double p = (gamma - dp) + theta, q = ((gamma - dp) + gamma) + dx, r = p / q;
        const double stpc = stp + r * (stx - stp);
        const double stpq = stp + (dp / (dp - dx)) * (stx - stp);
        stpf = fabs(stpc - stp) > fabs(stpq - stp) ? stpc : stpq;
        brackt = true;
    } else if (fabs(dp) < fabs(dx)) {                // case 3: same sign, the derivative decreases in magnitude
        const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
        const double s = std::max(fabs(theta), std::max(fabs(dx), fabs(dp)));
        double gamma = s * sqrt(std::max(0.0, (theta / s) * (theta / s) - (dx / s) * (dp / s)));
        if (stp > stx) gamma = -gamma;
        const double p = (gamma - dp) + theta, q = (gamma + (dx - dp)) + gamma, r = p / q;
        double stpc;
        if (r < 0.0 && gamma != 0.0) stpc = stp + r * (stx - stp);
        else if (stp > stx) stpc = stpmax;
        else stpc = stpmin;
        const double stpq = stp + (dp / (dp - dx)) * (stx - stp);
        if (brackt) {
            stpf = fabs(stpc - stp) < fabs(stpq - stp) ? stpc : stpq;
            if (stp > stx) stpf = std::min(stp + 0.66 * (sty - stp), stpf);
            else stpf = std::max(stp + 0.66 * (sty - stp), stpf);
        } else {
            stpf = fabs(stpc - stp) > fabs(stpq - stp) ? stpc : stpq;
            stpf = std::min(stpmax, stpf);
            stpf = std::max(stpmin, stpf);
        }
    } else {                                         // case 4: same sign, the derivative does not decrease
        if (brackt) {
            const double theta = 3.0 * (fp - fy) / (sty - stp) + dy + dp;
            const double s = std::max(fabs(theta), std::max(fabs(dy), fabs(dp)));
            double gamma = s * sqrt((theta / s) * (theta / s) - (dy / s) * (dp / s));
            if (stp > sty) gamma = -gamma;
            const double p = (gamma - dp) + theta, q = ((gamma - dp) + gamma) + dy, r = p / q;
            stpf = stp + r * (sty - stp);
        } else if (stp > stx) {
            stpf = stpmax;
        } else {
            stpf = stpmin;
        }
    }
    if (fp > fx) {
        sty = stp; fy = fp; dy = dp;
    } else {
        if (sgnd < 0.0) { sty = stx; fy = fx; dy = dx; }
        stx = stp; fx = fp; dx = dp;
    }
    stp = stpf;
}

inline void dcsrch(LineSearch& ls, double f, double g, double& stp, double ftol, double gtol, double xtol, double stpmin, double stpmax) {
    const double xtrapl = 1.1, xtrapu = 4.0;
    if (ls.task == LineSearch::START) {
        if (stp < stpmin || stp > stpmax || g >= 0.0 || stpmax < stpmin) { ls.task = LineSearch::ERROR; return; }
        ls.brackt = false;
        ls.stage = 1;
        ls.finit = f; ls.ginit = g; ls.gtest = ftol * g;
        ls.width = stpmax - stpmin; ls.width1 = ls.width / 0.5;
        ls.stx = 0.0; ls.fx = f; ls.gx = g;
        ls.sty = 0.0; ls.fy = f; ls.gy = g;
        ls.stmin = 0.0; ls.stmax = stp + xtrapu * stp;
        ls.task = LineSearch::FG;
        return;
    }
    const double ftest = ls.finit + stp * ls.gtest;
    if (ls.stage == 1 && f <= ftest && g >= 0.0) ls.stage = 2;
    if (ls.brackt && (stp <= ls.stmin || stp >= ls.stmax)) ls.task = LineSearch::WARNING;
    if (ls.brackt && ls.stmax - ls.stmin <= xtol * ls.stmax) ls.task = LineSearch::WARNING;
    if (stp == stpmax && f <= ftest && g <= ls.gtest) ls.task = LineSearch::WARNING;
    if (stp == stpmin && (f > ftest || g >= ls.gtest)) ls.task = LineSearch::WARNING;
    if (f <= ftest && fabs(g) <= gtol * (-ls.ginit)) ls.task = LineSearch::CONVERGENCE;
    if (ls.task == LineSearch::WARNING || ls.task == LineSearch::CONVERGENCE) return;
    if (ls.stage == 1 && f <= ls.fx && f > ftest) {
        const double fm = f - stp * ls.gtest;
        double fxm = ls.fx - ls.stx * ls.gtest, fym = ls.fy - ls.sty * ls.gtest;
        const double gm = g - ls.gtest;
        double gxm = ls.gx - ls.gtest, gym = ls.gy - ls.gtest;
        dcstep(ls.stx, fxm, gxm, ls.sty, fym, gym, stp, fm, gm, ls.brackt, ls.stmin, ls.stmax);
        ls.fx = fxm + ls.stx * ls.gtest; ls.fy = fym + ls.sty * ls.gtest;
        ls.gx = gxm + ls.gtest; ls.gy = gym + ls.gtest;
    } else {
        dcstep(ls.stx, ls.fx, ls.gx, ls.sty, ls.fy, ls.gy, stp, f, g, ls.brackt, ls.stmin, ls.stmax);
    }
    if (ls.brackt) {
        if (fabs(ls.sty - ls.stx) >= 0.66 * ls.width1) stp = ls.stx + 0.5 * (ls.sty - ls.stx);
        ls.width1 = ls.width;
        ls.width = fabs(ls.sty - ls.stx);
    }
    if (ls.brackt) {
        ls.stmin = std::min(ls.stx, ls.sty);
        ls.stmax = std::max(ls.stx, ls.sty);
    } else {
        ls.stmin = stp + xtrapl * (stp - ls.stx);
        ls.stmax = stp + xtrapu * (stp - ls.stx);
    }
    stp = std::max(stp, stpmin);
    stp = std::min(stp, stpmax);
    if ((ls.brackt && (stp <= ls.stmin || stp >= ls.stmax)) || (ls.brackt && ls.stmax - ls.stmin <= xtol * ls.stmax)) stp = ls.stx;
    ls.task = LineSearch::FG;
}

// Cholesky solve of the k x k system A y = b (A symmetric positive definite, row-major, overwritten); false if not PD
inline bool chol_solve(std::vector<double>& A, int k, std::vector<double>& b) {
    for (int j = 0; j < k; ++j) {
        double s = A[j * k + j];
        for (int p = 0; p < j; ++p) s -= A[j * k + p] * A[j * k + p];
        if (!(s > 0.0)) return false;
        const double ljj = sqrt(s);
        A[j * k + j] = ljj;
        for (int i = j + 1; i < k; ++i) {
            double t = A[i * k + j];
            for (int p = 0; p < j; ++p) t -= A[i * k + p] * A[j * k + p];
            A[i * k + j] = t / ljj;
        }
    }
    for (int i = 0; i < k; ++i) {
        double t = b[i];
        for (int p = 0; p < i; ++p) t -= A[i * k + p] * b[p];
        b[i] = t / A[i * k + i];
    }
    for (int i = k - 1; i >= 0; --i) {
        double t = b[i];
        for (int p = i + 1; p < k; ++p) t -= A[p * k + i] * b[p];
        b[i] = t / A[i * k + i];
    }
    return true;
}

}  // namespace detail

// nbd[i]: 0 unbounded, 1 lower bound only, 2 both, 3 upper bound only (the convention of L-BFGS-B)
inline LbfgsbResult lbfgsb_minimize(int n, double* x, const double* l, const double* u, const int* nbd, const ObjectiveFn& fun,
                                    const LbfgsbOptions& opt = LbfgsbOptions()) {
    using namespace detail;
    LbfgsbResult res;
    const int m = opt.maxcor;
    std::vector<double> g(n), xold(n), gold(n), d(n), z(n), xcp(n), B(n * n), tbrk(n), r(n);
    std::vector<std::vector<double>> S, Y;            // stored pairs, oldest first
    std::vector<int> iwhere(n, 0), order(n);
    double theta = 1.0, f = 0.0;

    bool cnstnd = false, boxed = true;
    for (int i = 0; i < n; ++i) {                    // `active`: project the start, classify the variables
        if (nbd[i] > 0) {
            if (nbd[i] <= 2 && x[i] <= l[i]) x[i] = l[i];
            else if (nbd[i] >= 2 && x[i] >= u[i]) x[i] = u[i];
        }
        if (nbd[i] != 2) boxed = false;
        if (nbd[i] == 0) iwhere[i] = -1;
        else {
            cnstnd = true;
            iwhere[i] = (nbd[i] == 2 && u[i] - l[i] <= 0.0) ? 3 : 0;
        }
    }
    auto proj_grad_norm = [&]() {
        double s = 0.0;
        for (int i = 0; i < n; ++i) {
            double gi = g[i];
            if (nbd[i] != 0) {
                if (gi < 0.0) { if (nbd[i] >= 2) gi = std::max(x[i] - u[i], gi); }
                else { if (nbd[i] <= 2) gi = std::min(x[i] - l[i], gi); }
            }
            s = std::max(s, fabs(gi));
        }
        return s;
    };
    auto form_B = [&]() {                             // theta I with the stored pairs applied as BFGS updates
        std::fill(B.begin(), B.end(), 0.0);
        for (int i = 0; i < n; ++i) B[i * n + i] = theta;
        std::vector<double> Bs(n);
        for (size_t p = 0; p < S.size(); ++p) {
            const std::vector<double>&s = S[p], &y = Y[p];
            double sBs = 0.0, ys = 0.0;
            for (int i = 0; i < n; ++i) {
                double t = 0.0;
                for (int j = 0; j < n; ++j) t += B[i * n + j] * s[j];
                Bs[i] = t;
            }
            for (int i = 0; i < n; ++i) { sBs += s[i] * Bs[i]; ys += y[i] * s[i]; }
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) B[i * n + j] += y[i] * y[j] / ys - Bs[i] * Bs[j] / sBs;
        }
    };

    if (fun(x, &f, g.data())) { res.status = -1; return res; }
    res.nfev = 1;
    double sbgnrm = proj_grad_norm();
    if (sbgnrm <= opt.pgtol) { res.f = f; res.pgnorm = sbgnrm; res.status = 0; return res; }

    int iter = 0;
    bool fresh = true;                                // no pairs stored
    for (;;) {
        // ---- generalised Cauchy point ------------------------------------------------------------------------------
        const int col = (int)S.size();
        if (!cnstnd && col > 0) {
            xcp = std::vector<double>(x, x + n);
        } else {
            form_B();
            // classify, search direction, breakpoints
            bool bnded = true;
            int nbreak = 0, nfree_nobreak = 0;
            double f1 = 0.0;
            for (int i = 0; i < n; ++i) {
                const double neggi = -g[i];
                double tl = 0.0, tu = 0.0;
                if (iwhere[i] != 3 && iwhere[i] != -1) {
                    if (nbd[i] <= 2) tl = x[i] - l[i];
                    if (nbd[i] >= 2) tu = u[i] - x[i];
                    const bool xlower = nbd[i] <= 2 && tl <= 0.0, xupper = nbd[i] >= 2 && tu <= 0.0;
                    iwhere[i] = 0;
                    if (xlower) { if (neggi <= 0.0) iwhere[i] = 1; }
                    else if (xupper) { if (neggi >= 0.0) iwhere[i] = 2; }
                    else if (fabs(neggi) <= 0.0) iwhere[i] = -3;
                }
                tbrk[i] = INFINITY;
                if (iwhere[i] != 0 && iwhere[i] != -1) {
                    d[i] = 0.0;
                } else {
                    d[i] = neggi;
                    f1 -= neggi * neggi;
                    if (nbd[i] <= 2 && nbd[i] != 0 && neggi < 0.0) { tbrk[i] = tl / (-neggi); order[nbreak++] = i; }
                    else if (nbd[i] >= 2 && neggi > 0.0) { tbrk[i] = tu / neggi; order[nbreak++] = i; }
                    else { ++nfree_nobreak; if (fabs(neggi) > 0.0) bnded = false; }
                }
            }
            xcp = std::vector<double>(x, x + n);
            if (!(nbreak == 0 && nfree_nobreak == 0)) {
                std::stable_sort(order.begin(), order.begin() + nbreak, [&](int a, int b) { return tbrk[a] < tbrk[b]; });
                const double f2_org = -theta * f1;
                auto quad = [&](double& f1o, double& f2o) {      // slope and curvature of the model along d from xcp
                    double s1 = 0.0, s2 = 0.0;
                    for (int i = 0; i < n; ++i) {
                        if (d[i] == 0.0) continue;
                        double bz = 0.0, bd = 0.0;
                        for (int j = 0; j < n; ++j) { bz += B[i * n + j] * (xcp[j] - x[j]); bd += B[i * n + j] * d[j]; }
                        s1 += d[i] * (g[i] + bz);
                        s2 += d[i] * bd;
                    }
                    f1o = s1; f2o = s2;
                };
                double f2;
                quad(f1, f2);
                f2 = std::max(kEps * f2_org, f2);
                double dtm = -f1 / f2, tj0 = 0.0;
                int nleft = nbreak;
                bool all_fixed = false;
                for (int ib = 0; ib < nbreak; ++ib) {
                    const int ibp = order[ib];
                    const double tj = tbrk[ibp], dt = tj - tj0;
                    if (dtm < dt) break;                          // the minimiser lies inside this segment
                    // advance to the breakpoint and fix the variable
                    for (int i = 0; i < n; ++i) if (d[i] != 0.0 && i != ibp) xcp[i] += dt * d[i];
                    --nleft;
                    const double dibp = d[ibp];
                    d[ibp] = 0.0;
                    if (dibp > 0.0) { xcp[ibp] = u[ibp]; iwhere[ibp] = 2; }
                    else { xcp[ibp] = l[ibp]; iwhere[ibp] = 1; }
                    tj0 = tj;
                    if (nleft == 0 && nbreak == n) { dtm = 0.0; all_fixed = true; break; }
                    quad(f1, f2);
                    f2 = std::max(kEps * f2_org, f2);
                    if (nleft > 0) dtm = -f1 / f2;
                    else if (bnded) dtm = 0.0;
                    else dtm = -f1 / f2;
                }
                if (!all_fixed) {
                    if (dtm <= 0.0) dtm = 0.0;
                    for (int i = 0; i < n; ++i) if (d[i] != 0.0) xcp[i] += dtm * d[i];
                }
            }
        }
        // ---- subspace minimisation over the free variables ---------------------------------------------------------
        z = xcp;
        std::vector<int> freev;
        for (int i = 0; i < n; ++i) if (iwhere[i] <= 0) freev.push_back(i);
        const int nf = (int)freev.size();
        if (nf > 0 && col > 0) {
            if (!cnstnd) form_B();
            std::vector<double> A(nf * nf), rhs(nf);
            for (int a = 0; a < nf; ++a) {
                const int i = freev[a];
                double bz = 0.0;
                for (int j = 0; j < n; ++j) bz += B[i * n + j] * (xcp[j] - x[j]);
                rhs[a] = -(g[i] + bz);
                for (int b = 0; b < nf; ++b) A[a * nf + b] = B[i * n + freev[b]];
            }
            if (!chol_solve(A, nf, rhs)) {               // (singular triangular system: refresh the memory and restart)
                S.clear(); Y.clear(); theta = 1.0; fresh = true;
                continue;
            }
            bool hit = false;
            for (int a = 0; a < nf; ++a) {               // projected Newton step (version 3.0)
                const int k = freev[a];
                const double xk = xcp[k], dk = rhs[a];
                if (nbd[k] == 0) z[k] = xk + dk;
                else if (nbd[k] == 1) { z[k] = std::max(l[k], xk + dk); if (z[k] == l[k]) hit = true; }
                else if (nbd[k] == 2) { z[k] = std::min(u[k], std::max(l[k], xk + dk)); if (z[k] == l[k] || z[k] == u[k]) hit = true; }
                else { z[k] = std::min(u[k], xk + dk); if (z[k] == u[k]) hit = true; }
            }
            if (hit) {
                double ddp = 0.0;
                for (int i = 0; i < n; ++i) ddp += (z[i] - x[i]) * g[i];
                if (ddp > 0.0) {                          // not a descent direction: truncated step from the Cauchy point instead
                    z = xcp;
                    double alpha = 1.0, temp1 = alpha;
                    int ibd = -1;
                    for (int a = 0; a < nf; ++a) {
                        const int k = freev[a];
                        const double dk = rhs[a];
                        if (nbd[k] != 0) {
                            if (dk < 0.0 && nbd[k] <= 2) {
                                const double temp2 = l[k] - xcp[k];
                                if (temp2 >= 0.0) temp1 = 0.0;
                                else if (dk * alpha < temp2) temp1 = temp2 / dk;
                            } else if (dk > 0.0 && nbd[k] >= 2) {
                                const double temp2 = u[k] - xcp[k];
                                if (temp2 <= 0.0) temp1 = 0.0;
                                else if (dk * alpha > temp2) temp1 = temp2 / dk;
                            }
                            if (temp1 < alpha) { alpha = temp1; ibd = a; }
                        }
                    }
                    if (alpha < 1.0 && ibd >= 0) {
                        const int k = freev[ibd];
                        if (rhs[ibd] > 0.0) { z[k] = u[k]; rhs[ibd] = 0.0; }
                        else if (rhs[ibd] < 0.0) { z[k] = l[k]; rhs[ibd] = 0.0; }
                    }
                    for (int a = 0; a < nf; ++a) z[freev[a]] += alpha * rhs[a];
                }
            }
        }
        // ---- line search along d = z - x ------------------------------------------------------------------------------
        double dtd = 0.0;
        for (int i = 0; i < n; ++i) { d[i] = z[i] - x[i]; dtd += d[i] * d[i]; }
        const double dnorm = sqrt(dtd);
        double stpmx = 1.0e10;
        if (cnstnd) {
            if (iter == 0) stpmx = 1.0;
            else {
                for (int i = 0; i < n; ++i) {
                    const double a1 = d[i];
                    if (nbd[i] != 0) {
                        if (a1 < 0.0 && nbd[i] <= 2) {
                            const double a2 = l[i] - x[i];
                            if (a2 >= 0.0) stpmx = 0.0;
                            else if (a1 * stpmx < a2) stpmx = a2 / a1;
                        } else if (a1 > 0.0 && nbd[i] >= 2) {
                            const double a2 = u[i] - x[i];
                            if (a2 <= 0.0) stpmx = 0.0;
                            else if (a1 * stpmx > a2) stpmx = a2 / a1;
                        }
                    }
                }
            }
        }
        double stp = (iter == 0 && !boxed) ? std::min(1.0 / dnorm, stpmx) : 1.0;
        xold.assign(x, x + n);
        gold = g;
        const double fold = f;
        int ifun = 0, iback = 0;
        double gd = 0.0, gdold = 0.0;
        LineSearch ls;
        bool ls_failed = false;
        for (;;) {
            gd = 0.0;
            for (int i = 0; i < n; ++i) gd += g[i] * d[i];
            if (ifun == 0) {
                gdold = gd;
                if (gd >= 0.0) { ls_failed = true; break; }       // not a descent direction
            }
            dcsrch(ls, f, gd, stp, 1e-3, 0.9, 0.1, 0.0, stpmx);
            if (ls.task == LineSearch::CONVERGENCE || ls.task == LineSearch::WARNING) break;
            if (ls.task == LineSearch::ERROR) { ls_failed = true; break; }
            ++ifun;
            iback = ifun - 1;
            if (iback >= opt.maxls) { ls_failed = true; break; }
            if (stp == 1.0) for (int i = 0; i < n; ++i) x[i] = z[i];
            else for (int i = 0; i < n; ++i) x[i] = stp * d[i] + xold[i];
            if (fun(x, &f, g.data())) { res.status = -1; res.f = f; return res; }
            ++res.nfev;
        }
        if (ls_failed) {
            for (int i = 0; i < n; ++i) x[i] = xold[i];
            g = gold;
            f = fold;
            if (fresh) {                                 // nothing to discard: abnormal termination
                res.status = 3;
                break;
            }
            S.clear(); Y.clear(); theta = 1.0; fresh = true;   // refresh the memory and try the steepest-descent direction
            continue;
        }
        ++iter;
        res.nit = iter;
        // ---- stopping tests --------------------------------------------------------------------------------------------
        sbgnrm = proj_grad_norm();
        if (sbgnrm <= opt.pgtol) { res.status = 0; break; }
        const double ddum = std::max(fabs(fold), std::max(fabs(f), 1.0));
        if (fold - f <= kEps * opt.factr * ddum) { res.status = 1; break; }
        if (iter >= opt.maxiter || res.nfev >= opt.maxfun) { res.status = 2; break; }
        // ---- new pair ----------------------------------------------------------------------------------------------------
        double rr = 0.0;
        for (int i = 0; i < n; ++i) { r[i] = g[i] - gold[i]; rr += r[i] * r[i]; }
        double dr, dd;
        if (stp == 1.0) { dr = gd - gdold; dd = -gdold; }
        else { dr = (gd - gdold) * stp; for (int i = 0; i < n; ++i) d[i] *= stp; dd = -gdold * stp; }
        if (dr > kEps * dd) {
            if ((int)S.size() == m) { S.erase(S.begin()); Y.erase(Y.begin()); }
            S.push_back(d);
            Y.push_back(r);
            theta = rr / dr;
            fresh = false;
        }
    }
    res.f = f;
    res.pgnorm = sbgnrm;
    return res;
}

}  // namespace ttm_opt

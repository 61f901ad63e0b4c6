// Compiled CPU baseline of the MiSTI composite-likelihood path (C++17 + OpenMP over candidates).
//
// TEST INFRASTRUCTURE: a reported baseline and a second checker.  Nothing under misti_amd/ includes, links or loads
// this; only tests/ and the cpu_baseline leg of bench.py (through oracle/cpu_baseline.py) do.
//
// It restates the REFERENCE's algorithm - not the GPU engine's - in compiled form (cites: /root/reference):
//   JAFSLikelihood   MigrationInference.py:566-614     eval_candidate()
//   CorrectLambdas   MigrationInference.py:305-378     correct_lambdas(): dense 3x3 expm per residual evaluation,
//                                                      2-point finite differences, trust-region-reflective solves
//   CorrectLambda    CorrectLambda.py:29-317           PairChain
//   Smooth           MigrationInference.py:380-405     smooth_const()
//   JAFSpectrum / SolveDifEq / CollapsePops  :467-540  jaf_spectrum(): dense generator (44x44, 37x37 without migration,
//                                                      8x8 after the split), expm(M T) by Pade-13 scaling and squaring
//                                                      (what scipy.linalg.expm does), the integral as M^-1 (P1 - P0),
//                                                      stationary states deleted and restored when mu = 0
//   TwoPopulations / OnePopulation                      tables.inc (generated from oracle/misti_oracle.py)
// SciPy's least_squares(method='trf') is restated (trf_no_bounds / trf_bounds, exact 2-D subproblem via SVD, 2-point
// Jacobian, gtol = xtol = 1e-10, ftol = 1e-8, max_nfev = 100 n).  Pinned by tests/test_cpu_baseline.py against the
// reference-generated golden vectors (determined cases to 1e-9, the others to the per-case contract).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <limits>
#include <vector>

#include "tables.inc"

namespace {

constexpr double EPS = 2.220446049250313e-16, SQRT_EPS = 1.4901161193847656e-08;
constexpr double FTOL = 1e-8, XTOL = 1e-10, GTOL = 1e-10;
constexpr int NS2 = 44, NS1 = 8;
const double INF = std::numeric_limits<double>::infinity();

// ------------------------------------------------------------------ dense linear algebra (row-major, n <= 44) ----
struct Mat {
    int n;
    std::vector<double> a;
    explicit Mat(int n_ = 0) : n(n_), a((size_t)n_ * n_, 0.0) {}
    double& operator()(int i, int j) { return a[(size_t)i * n + j]; }
    double operator()(int i, int j) const { return a[(size_t)i * n + j]; }
};

Mat matmul(const Mat& A, const Mat& B) {
    const int n = A.n;
    Mat C(n);
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < n; ++k) {
            const double aik = A(i, k);
            if (aik == 0.0) continue;
            const double* b = &B.a[(size_t)k * n];
            double* c = &C.a[(size_t)i * n];
            for (int j = 0; j < n; ++j) c[j] += aik * b[j];
        }
    return C;
}

// LU with partial pivoting; solves A X = B for nrhs right-hand sides stored column-major in B (n x nrhs, row-major array)
bool lu_solve(Mat A, std::vector<double>& B, int nrhs) {
    const int n = A.n;
    std::vector<int> piv(n);
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = std::fabs(A(k, k));
        for (int i = k + 1; i < n; ++i) if (std::fabs(A(i, k)) > best) { best = std::fabs(A(i, k)); p = i; }
        if (best == 0.0 || !(best == best)) return false;
        if (p != k) {
            for (int j = 0; j < n; ++j) std::swap(A(k, j), A(p, j));
            for (int j = 0; j < nrhs; ++j) std::swap(B[(size_t)k * nrhs + j], B[(size_t)p * nrhs + j]);
        }
        const double inv = 1.0 / A(k, k);
        for (int i = k + 1; i < n; ++i) {
            const double l = A(i, k) * inv;
            if (l == 0.0) continue;
            A(i, k) = l;
            for (int j = k + 1; j < n; ++j) A(i, j) -= l * A(k, j);
            for (int j = 0; j < nrhs; ++j) B[(size_t)i * nrhs + j] -= l * B[(size_t)k * nrhs + j];
        }
    }
    for (int k = n - 1; k >= 0; --k) {
        const double inv = 1.0 / A(k, k);
        for (int j = 0; j < nrhs; ++j) {
            double s = B[(size_t)k * nrhs + j];
            for (int i = k + 1; i < n; ++i) s -= A(k, i) * B[(size_t)i * nrhs + j];
            B[(size_t)k * nrhs + j] = s * inv;
        }
    }
    return true;
}

bool inverse(const Mat& A, Mat& out) {
    const int n = A.n;
    std::vector<double> B((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) B[(size_t)i * n + i] = 1.0;
    if (!lu_solve(A, B, n)) return false;
    out = Mat(n);
    out.a = B;
    return true;
}

double norm1(const Mat& A) {
    double best = 0.0;
    for (int j = 0; j < A.n; ++j) {
        double s = 0.0;
        for (int i = 0; i < A.n; ++i) s += std::fabs(A(i, j));
        best = std::max(best, s);
    }
    return best;
}

// expm by Pade-13 with scaling and squaring (Higham 2005, the method of scipy.linalg.expm; degree 13 throughout)
bool expm(const Mat& A0, Mat& E) {
    static const double b[14] = {64764752532480000., 32382376266240000., 7771770303897600., 1187353796428800., 129060195264000., 10559470521600.,
                                 670442572800., 33522128640., 1323241920., 40840800., 960960., 16380., 182., 1.};
    const int n = A0.n;
    const double nrm = norm1(A0);
    if (!(nrm < 1e300)) return false;
    int s = 0;
    if (nrm > 5.371920351148152) s = std::max(0, (int)std::ceil(std::log2(nrm / 5.371920351148152)));
    Mat A = A0;
    const double sc = std::ldexp(1.0, -s);
    for (auto& v : A.a) v *= sc;
    const Mat A2 = matmul(A, A), A4 = matmul(A2, A2), A6 = matmul(A4, A2);
    Mat W1(n), W2(n), Z1(n), Z2(n);
    for (size_t i = 0; i < A.a.size(); ++i) {
        W1.a[i] = b[13] * A6.a[i] + b[11] * A4.a[i] + b[9] * A2.a[i];
        W2.a[i] = b[7] * A6.a[i] + b[5] * A4.a[i] + b[3] * A2.a[i];
        Z1.a[i] = b[12] * A6.a[i] + b[10] * A4.a[i] + b[8] * A2.a[i];
        Z2.a[i] = b[6] * A6.a[i] + b[4] * A4.a[i] + b[2] * A2.a[i];
    }
    for (int i = 0; i < n; ++i) { W2(i, i) += b[1]; Z2(i, i) += b[0]; }
    Mat W = matmul(A6, W1);
    for (size_t i = 0; i < W.a.size(); ++i) W.a[i] += W2.a[i];
    const Mat U = matmul(A, W);
    Mat V = matmul(A6, Z1);
    for (size_t i = 0; i < V.a.size(); ++i) V.a[i] += Z2.a[i];
    Mat P(n);
    std::vector<double> Q((size_t)n * n);
    for (size_t i = 0; i < U.a.size(); ++i) { P.a[i] = V.a[i] - U.a[i]; Q[i] = V.a[i] + U.a[i]; }
    if (!lu_solve(P, Q, n)) return false;
    E = Mat(n);
    E.a = Q;
    for (int i = 0; i < s; ++i) E = matmul(E, E);
    return true;
}

std::vector<double> matvec(const Mat& A, const std::vector<double>& x) {
    std::vector<double> y(A.n, 0.0);
    for (int i = 0; i < A.n; ++i) { double s = 0.0; for (int j = 0; j < A.n; ++j) s += A(i, j) * x[j]; y[i] = s; }
    return y;
}

// ------------------------------------------------------------------------------ least squares (SciPy TRF) ----
typedef std::function<bool(const double*, double*)> Residual;   // f(x) -> residuals; false: not finite

struct LsqResult { int nfev = 0, status = 0; };

double vnorm(const double* v, int n) { double a = 0; for (int i = 0; i < n; ++i) a += v[i] * v[i]; return std::sqrt(a); }

// thin SVD of an m x n (n <= 2) matrix by a one-sided Jacobi rotation: singular values s (descending), V, U^T f
void svd_small(const double* A, int m, int n, const double* f, double* s, double* V, double* uf) {
    if (n == 1) {
        double nn = 0, fa = 0;
        for (int r = 0; r < m; ++r) { nn += A[r] * A[r]; fa += A[r] * f[r]; }
        s[0] = std::sqrt(nn); V[0] = 1.0; uf[0] = s[0] > 0 ? fa / s[0] : 0.0;
        return;
    }
    double al = 0, be = 0, ga = 0;
    for (int r = 0; r < m; ++r) { al += A[2 * r] * A[2 * r]; be += A[2 * r + 1] * A[2 * r + 1]; ga += A[2 * r] * A[2 * r + 1]; }
    double c = 1.0, sn = 0.0;
    if (ga != 0.0) {
        const double zeta = (be - al) / (2.0 * ga);
        const double t = std::copysign(1.0, zeta) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        c = 1.0 / std::sqrt(1.0 + t * t);
        sn = c * t;
    }
    double n1 = 0, n2 = 0, f1 = 0, f2 = 0;
    for (int r = 0; r < m; ++r) {
        const double a1 = c * A[2 * r] - sn * A[2 * r + 1], a2 = sn * A[2 * r] + c * A[2 * r + 1];
        n1 += a1 * a1; n2 += a2 * a2; f1 += a1 * f[r]; f2 += a2 * f[r];
    }
    const double s1 = std::sqrt(n1), s2 = std::sqrt(n2);
    const double u1 = s1 > 0 ? f1 / s1 : 0.0, u2 = s2 > 0 ? f2 / s2 : 0.0;
    // V columns: v1 = (c, -sn), v2 = (sn, c); V stored row-major V[r][i]
    if (s1 >= s2) { s[0] = s1; s[1] = s2; V[0] = c; V[2] = -sn; V[1] = sn; V[3] = c; uf[0] = u1; uf[1] = u2; }
    else { s[0] = s2; s[1] = s1; V[0] = sn; V[2] = c; V[1] = c; V[3] = -sn; uf[0] = u2; uf[1] = u1; }
}

// common.py solve_lsq_trust_region :57-166
void solve_tr(int n, int m, const double* s, const double* V, const double* uf, double Delta, double& alpha, double* p) {
    double suf[2];
    for (int i = 0; i < n; ++i) suf[i] = s[i] * uf[i];
    const bool full_rank = (m >= n) && (s[n - 1] > EPS * m * s[0]);
    if (full_rank) {
        for (int r = 0; r < n; ++r) { double a = 0; for (int i = 0; i < n; ++i) a += V[r * n + i] * (uf[i] / s[i]); p[r] = -a; }
        if (vnorm(p, n) <= Delta) { alpha = 0.0; return; }
    }
    double alpha_upper = vnorm(suf, n) / Delta;
    auto phi_fn = [&](double al, double& phi, double& dphi) {
        double q[2], acc = 0;
        for (int i = 0; i < n; ++i) { const double e = s[i] * s[i] + al; q[i] = suf[i] / e; acc += suf[i] * suf[i] / (e * e * e); }
        const double pn = vnorm(q, n);
        phi = pn - Delta; dphi = -acc / pn;
    };
    double alpha_lower = 0.0;
    if (full_rank) { double ph, dp; phi_fn(0.0, ph, dp); alpha_lower = -ph / dp; }
    double al = alpha;
    if (!full_rank && al == 0.0) al = std::max(0.001 * alpha_upper, std::sqrt(alpha_lower * alpha_upper));
    for (int it = 0; it < 10; ++it) {
        if (al < alpha_lower || al > alpha_upper) al = std::max(0.001 * alpha_upper, std::sqrt(alpha_lower * alpha_upper));
        double ph, dp; phi_fn(al, ph, dp);
        if (ph < 0) alpha_upper = al;
        const double ratio = ph / dp;
        alpha_lower = std::max(alpha_lower, al - ratio);
        al -= (ph + Delta) * ratio / Delta;
        if (std::fabs(ph) < 0.01 * Delta) break;
    }
    for (int r = 0; r < n; ++r) { double a = 0; for (int i = 0; i < n; ++i) a += V[r * n + i] * (suf[i] / (s[i] * s[i] + al)); p[r] = -a; }
    const double sc = Delta / vnorm(p, n);
    for (int r = 0; r < n; ++r) p[r] *= sc;
    alpha = al;
}

int check_term(double dF, double F, double dx, double xn, double ratio) {
    const bool f_ok = dF < FTOL * F && ratio > 0.25, x_ok = dx < XTOL * (XTOL + xn);
    return (f_ok && x_ok) ? 4 : f_ok ? 2 : x_ok ? 3 : 0;
}

double fd_step(double x) { return SQRT_EPS * (x >= 0 ? 1.0 : -1.0) * std::max(1.0, std::fabs(x)); }

// residuals + 2-point Jacobian (J row-major m x n, m == n here); lb = -inf: no bound
bool eval_fj(const Residual& fun, int n, const double* x, double lb, double* f, double* J) {
    bool ok = fun(x, f);
    for (int j = 0; j < n; ++j) {
        double h = fd_step(x[j]);
        if (x[j] + h < lb) h = -h;
        double x1[2] = {x[0], n > 1 ? x[1] : 0.0}, f1[2];
        x1[j] = x[j] + h;
        const double dx = x1[j] - x[j];
        fun(x1, f1);
        for (int r = 0; r < n; ++r) J[r * n + j] = (f1[r] - f[r]) / dx;
    }
    return ok;
}

// trf.py trf_no_bounds :401-560, n = 2
LsqResult trf_unbounded(const Residual& fun, double* x) {
    const int n = 2, max_nfev = 200;
    LsqResult res;
    double f[2], J[4], g[2];
    if (!eval_fj(fun, n, x, -INF, f, J)) { res.status = -1; return res; }
    int nfev = 1;
    double cost = 0.5 * (f[0] * f[0] + f[1] * f[1]);
    auto grad = [&]() { g[0] = J[0] * f[0] + J[2] * f[1]; g[1] = J[1] * f[0] + J[3] * f[1]; };
    grad();
    double Delta = vnorm(x, n);
    if (Delta == 0) Delta = 1.0;
    double alpha = 0.0;
    int term = 0;
    for (;;) {
        const double g_norm = std::max(std::fabs(g[0]), std::fabs(g[1]));
        if (g_norm < GTOL) term = 1;
        if (term != 0 || nfev >= max_nfev || !(g_norm < INF)) break;
        double s[2], V[4], uf[2];
        svd_small(J, 2, 2, f, s, V, uf);
        double actual = -1.0, xn[2], fn[2], Jn[4], cost_new = cost;
        while (actual <= 0 && nfev < max_nfev) {
            double p[2];
            solve_tr(n, 2, s, V, uf, Delta, alpha, p);
            const double Js0 = J[0] * p[0] + J[1] * p[1], Js1 = J[2] * p[0] + J[3] * p[1];
            const double predicted = -(0.5 * (Js0 * Js0 + Js1 * Js1) + (p[0] * g[0] + p[1] * g[1]));
            xn[0] = x[0] + p[0]; xn[1] = x[1] + p[1];
            const bool fin = eval_fj(fun, n, xn, -INF, fn, Jn);
            ++nfev;
            const double sn = vnorm(p, n);
            if (!fin || !std::isfinite(fn[0]) || !std::isfinite(fn[1])) { Delta = 0.25 * sn; continue; }
            cost_new = 0.5 * (fn[0] * fn[0] + fn[1] * fn[1]);
            actual = cost - cost_new;
            double ratio;
            if (predicted > 0) ratio = actual / predicted; else if (predicted == 0 && actual == 0) ratio = 1.0; else ratio = 0.0;
            double Delta_new = Delta;
            if (ratio < 0.25) Delta_new = 0.25 * sn; else if (ratio > 0.75 && sn > 0.95 * Delta) Delta_new = 2.0 * Delta;
            term = check_term(actual, cost, sn, vnorm(x, n), ratio);
            if (term != 0) break;
            alpha *= Delta / Delta_new;
            Delta = Delta_new;
        }
        if (actual > 0) {
            x[0] = xn[0]; x[1] = xn[1]; f[0] = fn[0]; f[1] = fn[1];
            std::memcpy(J, Jn, sizeof J);
            cost = cost_new;
            grad();
        }
    }
    res.nfev = nfev; res.status = term;
    return res;
}

// trf.py trf_bounds :205-400 with a lower bound lb on every variable and no upper bound; N = 1 or 2
LsqResult trf_bounded(const Residual& fun, int N, double* x, double lb) {
    LsqResult res;
    const int max_nfev = 100 * N;
    for (int i = 0; i < N; ++i) { const double th = 1e-10 * std::max(1.0, std::fabs(lb)); if (x[i] - lb <= th) x[i] = lb + th; }
    double f[2], J[4], g[2] = {0, 0};
    eval_fj(fun, N, x, lb, f, J);
    int nfev = 1;
    double cost = 0;
    for (int i = 0; i < N; ++i) cost += f[i] * f[i];
    cost *= 0.5;
    auto grad = [&]() { for (int c = 0; c < N; ++c) { g[c] = 0; for (int r = 0; r < N; ++r) g[c] += J[r * N + c] * f[r]; } };
    grad();
    double v[2], dv[2];
    auto CL = [&]() { for (int i = 0; i < N; ++i) { if (g[i] > 0) { v[i] = x[i] - lb; dv[i] = 1.0; } else { v[i] = 1.0; dv[i] = 0.0; } } };
    CL();
    double Delta;
    { double t[2]; for (int i = 0; i < N; ++i) t[i] = x[i] / std::sqrt(v[i]); Delta = vnorm(t, N); if (Delta == 0) Delta = 1.0; }
    double alpha = 0.0;
    int term = 0;
    auto step_to_bound = [&](const double* xx, const double* s, int* hits) {
        double steps[2], mn = INF;
        for (int i = 0; i < N; ++i) { steps[i] = INF; if (s[i] != 0) steps[i] = std::max((lb - xx[i]) / s[i], s[i] > 0 ? INF : -INF); mn = std::min(mn, steps[i]); }
        for (int i = 0; i < N; ++i) hits[i] = (steps[i] == mn) ? (s[i] > 0 ? 1 : (s[i] < 0 ? -1 : 0)) : 0;
        return mn;
    };
    for (;;) {
        CL();
        double g_norm = 0;
        for (int i = 0; i < N; ++i) g_norm = std::max(g_norm, std::fabs(g[i] * v[i]));
        if (g_norm < GTOL) term = 1;
        if (term != 0 || nfev >= max_nfev || !(g_norm < INF)) break;
        double d[2], diag[2], gh[2], Jh[4], A[8], fa[4];
        for (int i = 0; i < N; ++i) { d[i] = std::sqrt(v[i]); diag[i] = g[i] * dv[i]; gh[i] = d[i] * g[i]; }
        for (int r = 0; r < N; ++r) {
            fa[r] = f[r]; fa[N + r] = 0.0;
            for (int c = 0; c < N; ++c) { Jh[r * N + c] = J[r * N + c] * d[c]; A[r * N + c] = Jh[r * N + c]; A[(N + r) * N + c] = (r == c) ? std::sqrt(diag[r]) : 0.0; }
        }
        double s[2], V[4], uf[2];
        svd_small(A, 2 * N, N, fa, s, V, uf);
        const double theta = std::max(0.995, 1.0 - g_norm);
        auto eval_quad = [&](const double* st) {
            double q = 0, l = 0;
            for (int r = 0; r < N; ++r) { double js = 0; for (int c = 0; c < N; ++c) js += Jh[r * N + c] * st[c]; q += js * js; }
            for (int i = 0; i < N; ++i) { q += st[i] * diag[i] * st[i]; l += st[i] * gh[i]; }
            return 0.5 * q + l;
        };
        auto build_quad = [&](const double* st, const double* s0, double& a, double& b, double& c) {
            double vv[2];
            a = b = c = 0;
            for (int r = 0; r < N; ++r) { vv[r] = 0; for (int k = 0; k < N; ++k) vv[r] += Jh[r * N + k] * st[k]; a += vv[r] * vv[r]; }
            for (int i = 0; i < N; ++i) { a += st[i] * diag[i] * st[i]; b += gh[i] * st[i]; }
            a *= 0.5;
            if (s0) {
                double uu = 0;
                for (int r = 0; r < N; ++r) { double u = 0; for (int k = 0; k < N; ++k) u += Jh[r * N + k] * s0[k]; b += u * vv[r]; uu += u * u; }
                c = 0.5 * uu;
                for (int i = 0; i < N; ++i) { c += gh[i] * s0[i]; b += s0[i] * diag[i] * st[i]; }
                for (int i = 0; i < N; ++i) c += 0.5 * s0[i] * diag[i] * s0[i];
            }
        };
        auto min_quad = [&](double a, double b, double lo, double hi, double c, double& y) {
            double t = lo; y = lo * (a * lo + b) + c;
            const double yh = hi * (a * hi + b) + c;
            if (yh < y) { y = yh; t = hi; }
            if (a != 0) { const double e = -0.5 * b / a; if (lo < e && e < hi) { const double ye = e * (a * e + b) + c; if (ye < y) { y = ye; t = e; } } }
            return t;
        };
        double actual = -1.0, xn[2], fn[2], Jn[4], cost_new = cost;
        while (actual <= 0 && nfev < max_nfev) {
            double ph[2], p[2], step[2], steph[2];
            solve_tr(N, N, s, V, uf, Delta, alpha, ph);
            for (int i = 0; i < N; ++i) p[i] = d[i] * ph[i];
            // select_step, trf.py:128-203
            double predicted;
            bool inb = true;
            for (int i = 0; i < N; ++i) inb = inb && (x[i] + p[i] >= lb);
            if (inb) { for (int i = 0; i < N; ++i) { step[i] = p[i]; steph[i] = ph[i]; } predicted = -eval_quad(ph); }
            else {
                int hits[2], h2[2];
                const double pstride = step_to_bound(x, p, hits);
                double rh[2], r[2], xb[2];
                for (int i = 0; i < N; ++i) { rh[i] = hits[i] != 0 ? -ph[i] : ph[i]; r[i] = d[i] * rh[i]; }
                for (int i = 0; i < N; ++i) { p[i] *= pstride; ph[i] *= pstride; xb[i] = x[i] + p[i]; }
                double to_tr;
                {
                    double a = 0, b = 0, c = -Delta * Delta;
                    for (int i = 0; i < N; ++i) { a += rh[i] * rh[i]; b += ph[i] * rh[i]; c += ph[i] * ph[i]; }
                    const double dd = std::sqrt(b * b - a * c), q = -(b + std::copysign(dd, b));
                    to_tr = std::max(q / a, c / q);
                }
                const double to_bound = step_to_bound(xb, r, h2);
                const double rs = std::min(to_bound, to_tr);
                double rl, ru;
                if (rs > 0) { rl = (1 - theta) * pstride / rs; ru = (rs == to_bound) ? theta * to_bound : to_tr; } else { rl = 0; ru = -1; }
                double rval = INF;
                if (rl <= ru) {
                    double a, b, c;
                    build_quad(rh, ph, a, b, c);
                    const double t = min_quad(a, b, rl, ru, c, rval);
                    for (int i = 0; i < N; ++i) { rh[i] = rh[i] * t + ph[i]; r[i] = rh[i] * d[i]; }
                }
                for (int i = 0; i < N; ++i) { p[i] *= theta; ph[i] *= theta; }
                const double pval = eval_quad(ph);
                double agh[2], ag[2];
                for (int i = 0; i < N; ++i) { agh[i] = -gh[i]; ag[i] = d[i] * agh[i]; }
                const double to_tr2 = Delta / vnorm(agh, N), to_b2 = step_to_bound(x, ag, h2);
                double ags = to_b2 < to_tr2 ? theta * to_b2 : to_tr2, a, b, c, agval;
                build_quad(agh, nullptr, a, b, c);
                ags = min_quad(a, b, 0.0, ags, 0.0, agval);
                for (int i = 0; i < N; ++i) { agh[i] *= ags; ag[i] *= ags; }
                if (pval < rval && pval < agval) { for (int i = 0; i < N; ++i) { step[i] = p[i]; steph[i] = ph[i]; } predicted = -pval; }
                else if (rval < pval && rval < agval) { for (int i = 0; i < N; ++i) { step[i] = r[i]; steph[i] = rh[i]; } predicted = -rval; }
                else { for (int i = 0; i < N; ++i) { step[i] = ag[i]; steph[i] = agh[i]; } predicted = -agval; }
            }
            for (int i = 0; i < N; ++i) { xn[i] = x[i] + step[i]; if (xn[i] <= lb) xn[i] = std::nextafter(lb, INF); }
            eval_fj(fun, N, xn, lb, fn, Jn);
            ++nfev;
            const double shn = vnorm(steph, N);
            bool fin = true;
            for (int i = 0; i < N; ++i) fin = fin && std::isfinite(fn[i]);
            if (!fin) { Delta = 0.25 * shn; continue; }
            cost_new = 0;
            for (int i = 0; i < N; ++i) cost_new += fn[i] * fn[i];
            cost_new *= 0.5;
            actual = cost - cost_new;
            double ratio;
            if (predicted > 0) ratio = actual / predicted; else if (predicted == 0 && actual == 0) ratio = 1.0; else ratio = 0.0;
            double Delta_new = Delta;
            if (ratio < 0.25) Delta_new = 0.25 * shn; else if (ratio > 0.75 && shn > 0.95 * Delta) Delta_new = 2.0 * Delta;
            term = check_term(actual, cost, vnorm(step, N), vnorm(x, N), ratio);
            if (term != 0) break;
            alpha *= Delta / Delta_new;
            Delta = Delta_new;
        }
        if (actual > 0) {
            for (int i = 0; i < N; ++i) { x[i] = xn[i]; f[i] = fn[i]; }
            std::memcpy(J, Jn, sizeof J);
            cost = cost_new;
            grad();
        }
    }
    res.nfev = nfev; res.status = term;
    return res;
}

// ---- study hook: one ulp of noise in the pair chain's matrix exponential ------------------------------------------------
// tests/golden/internal_noise.py re-runs the REFERENCE with scipy.linalg.expm as CorrectLambda.py:62 sees it returning every entry
// moved by -1, 0 or +1 ulp at random: how well the reference's llh is determined by its own arithmetic.  The same study on this
// restatement (misti_cpu_set_expm_noise; tools/uniform_spread.py) - a per-candidate generator, so results do not depend on threads.
int g_noise_seed = -1;                           // < 0: off
thread_local uint64_t t_noise_state = 0;
inline uint64_t noise_next() {                   // splitmix64
    uint64_t z = (t_noise_state += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
inline void noise_apply(Mat& E) {
    if (g_noise_seed < 0) return;
    for (auto& v : E.a) {
        const uint64_t k = noise_next() % 3;     // 0: as it is, 1: one ulp up, 2: one ulp down
        if (k == 1) v = std::nextafter(v, INFINITY);
        else if (k == 2) v = std::nextafter(v, -INFINITY);
    }
}

// ------------------------------------------------------------------------- the pair chain (CorrectLambda.py) ----
struct PairChain {
    double mu[2] = {0, 0}, lh[2] = {0, 0}, T = 0, mixtureTH = 0;
    double P0[2][3];

    Mat matrix(const double* l) const {                                   // SetMatrix :55-56
        Mat M(3);
        M(0, 0) = -2 * mu[0] - l[0]; M(0, 2) = mu[1];
        M(1, 1) = -2 * mu[1] - l[1]; M(1, 2) = mu[0];
        M(2, 0) = 2 * mu[0]; M(2, 1) = 2 * mu[1]; M(2, 2) = -mu[0] - mu[1];
        return M;
    }
    bool met(const double* l, Mat& E) const {                             // MatrixExponent :58-62
        Mat M = matrix(l);
        for (auto& v : M.a) v *= T;
        if (!expm(M, E)) return false;
        noise_apply(E);
        return true;
    }
    double ect_one_pop(double lam) const { const double r = lam > 100 ? 0.0 : T / (std::exp(lam * T) - 1.0); return 1.0 / lam - r; }   // :67-72
    double ect_one_pop_tmp(double lam) const { const double pnc = std::exp(-lam * T); return 1.0 / lam - T / (1.0 / pnc - 1.0); }      // :74-77
    double ect_noncond(double lam) const { return (1 - std::exp(-lam * T) * (1 + lam * T)) / lam; }                                     // :79-80

    bool residual_cp(const double* l, double* out) const {               // LambdaSystem1 :169-173,135-144
        Mat E;
        if (!met(l, E)) { out[0] = out[1] = NAN; return false; }
        for (int k = 0; k < 2; ++k) {
            const double s = (P0[k][0] + P0[k][1]) + P0[k][2];
            const double nch = std::exp(-lh[k] * T) * s;
            double tot = 0;
            for (int i = 0; i < 3; ++i) tot += E(i, 0) * P0[k][0] + E(i, 1) * P0[k][1] + E(i, 2) * P0[k][2];
            out[k] = tot - nch;
        }
        return std::isfinite(out[0]) && std::isfinite(out[1]);
    }
    bool residual_ect(const double* l, double* out) const {              // LambdaSystem :151-157,94-110
        Mat M = matrix(l), E, Mi;
        if (!met(l, E) || !inverse(M, Mi)) { out[0] = out[1] = NAN; return false; }
        for (int k = 0; k < 2; ++k) {
            const double s = (P0[k][0] + P0[k][1]) + P0[k][2];
            std::vector<double> pn = {P0[k][0] / s, P0[k][1] / s, P0[k][2] / s};
            Mat EmI = E;
            for (int i = 0; i < 3; ++i) EmI(i, i) -= 1.0;
            std::vector<double> vec1 = matvec(Mi, matvec(Mi, matvec(EmI, pn)));
            std::vector<double> vec2 = matvec(E, pn);
            const double pnc = (vec2[0] + vec2[1]) + vec2[2];
            vec2 = matvec(Mi, vec2);
            for (auto& v : vec2) v *= T;
            const double ect2 = (l[0] * (vec2[0] - vec1[0]) + l[1] * (vec2[1] - vec1[1])) / (1 - pnc);
            out[k] = ect2 - ect_one_pop_tmp(lh[k]);
        }
        return std::isfinite(out[0]) && std::isfinite(out[1]);
    }

    // SolveLambdaSystem :266-317.  lc[2], P1[2][3] out; returns false on a numerical failure
    bool solve(bool cpfit, double* lc, double P1[2][3]) {
        const double s0 = (P0[0][0] + P0[0][1]) + P0[0][2], s1 = (P0[1][0] + P0[1][1]) + P0[1][2];
        double mix = 0;
        for (int i = 0; i < 3; ++i) { const double d = P0[0][i] / s0 - P0[1][i] / s1; mix += d * d; }
        if (std::sqrt(mix) < mixtureTH) { lc[0] = lc[1] = -1; std::memcpy(P1, P0, sizeof(double) * 6); return true; }
        if (mu[0] + mu[1] < 1e-10) {
            if (cpfit) {                                                  // SolveNoMigration1 :213-235
                const double A1 = P0[0][0] / s0, A2 = P0[0][1] / s0, A3 = P0[1][0] / s1, A4 = P0[1][1] / s1, C1 = P0[0][2] / s0, C2 = P0[1][2] / s1;
                const double D = A1 * A4 - A2 * A3, B1 = A4 / D, B2 = -A2 / D, B3 = -A3 / D, B4 = A1 / D;
                const double X1 = std::exp(-lh[0] * T) - C1, X2 = std::exp(-lh[1] * T) - C2;
                if (B1 * X1 + B2 * X2 > 0 && B3 * X1 + B4 * X2 > 0) { lc[0] = -std::log(B1 * X1 + B2 * X2) / T; lc[1] = -std::log(B3 * X1 + B4 * X2) / T; }
                else lc[0] = lc[1] = -1;
            } else {                                                      // SolveNoMigration :253-264
                double pr[2][3];
                for (int i = 0; i < 3; ++i) { pr[0][i] = P0[0][i] / s0; pr[1][i] = P0[1][i] / s1; }
                Residual fun = [&](const double* l, double* ff) {        // LambdaSystemNoMigration :237-251
                    for (int k = 0; k < 2; ++k) {
                        const double pnc = pr[k][0] * std::exp(-l[0] * T) + pr[k][1] * std::exp(-l[1] * T) + pr[k][2];
                        const double ct = (pr[k][0] * ect_noncond(l[0]) + pr[k][1] * ect_noncond(l[1])) / (1 - pnc);
                        ff[k] = ct - ect_one_pop(lh[k]);
                    }
                    return true;
                };
                double x[2] = {lh[0], lh[1]};
                trf_bounded(fun, 2, x, 0.01 * std::min(lh[0], lh[1]));
                lc[0] = x[0]; lc[1] = x[1];
            }
            for (int k = 0; k < 2; ++k) { P1[k][0] = P0[k][0] * std::exp(-lc[0] * T); P1[k][1] = P0[k][1] * std::exp(-lc[1] * T); P1[k][2] = P0[k][2]; }
            return true;
        }
        double n0 = 0, n1 = 0, nd = 0;
        for (int i = 0; i < 3; ++i) { n0 += P0[0][i] * P0[0][i]; n1 += P0[1][i] * P0[1][i]; const double d = P0[0][i] - P0[1][i]; nd += d * d; }
        if (std::sqrt(nd) < 0.02 * std::min(std::sqrt(n0), std::sqrt(n1))) { const double mean = (lh[0] + lh[1]) / 2.0; lh[0] = lh[1] = mean; }
        const double Tk = T;                                              // stretch to the unit interval :293-298
        T = T / Tk; mu[0] *= Tk; mu[1] *= Tk; lh[0] *= Tk; lh[1] *= Tk;
        Residual fun = cpfit ? Residual([&](const double* l, double* o) { return residual_cp(l, o); })
                             : Residual([&](const double* l, double* o) { return residual_ect(l, o); });
        double x[2] = {lh[0], lh[1]};
        LsqResult r = trf_unbounded(fun, x);
        T = Tk; mu[0] /= Tk; mu[1] /= Tk; lh[0] /= Tk; lh[1] /= Tk;
        if (r.status < 0) return false;
        lc[0] = x[0] / Tk; lc[1] = x[1] / Tk;
        Mat E;
        if (!met(lc, E)) return false;
        for (int k = 0; k < 2; ++k) for (int i = 0; i < 3; ++i) P1[k][i] = E(i, 0) * P0[k][0] + E(i, 1) * P0[k][1] + E(i, 2) * P0[k][2];
        return true;
    }
};

// ------------------------------------------------------------------------------------- the model, one candidate ----
struct Band { int pop, start, end, param; double value; };
struct Pulse { int pop, time, param; double value; };
struct Model {
    int numT, sample_date, n_band, n_pulse, n_param;
    bool cpfit, true_eps, smooth, unfolded;
    double mixture_th;
    const double* times; const double* lh;
    const Band* bands; const Pulse* pulses;
};

Mat two_pop_generator(const double* la, const double* mu) {               // UpdateMatrixCol order, TwoPopulations.py:336-359
    const double rate[4] = {la[0], la[1], mu[0], mu[1]};
    Mat M(NS2);
    for (int src = 0; src < NS2; ++src) {
        double total = 0.0;
        for (int e = TP_EV_OFF[src]; e < TP_EV_OFF[src + 1]; ++e) {
            const int dst = TP_EV[2 * e], kind = TP_EV[2 * e + 1];
            if (dst >= 0) M(dst, src) += rate[kind];
            total += rate[kind];
        }
        M(src, src) -= total;
    }
    return M;
}
Mat one_pop_generator(double la) {                                        // OnePopulation.py:160-178
    Mat M(NS1);
    for (int src = 0; src < NS1; ++src) {
        double total = 0.0;
        for (int e = OP_EV_OFF[src]; e < OP_EV_OFF[src + 1]; ++e) { const int dst = OP_EV[e]; if (dst >= 0) M(dst, src) += la; total += la; }
        M(src, src) -= total;
    }
    return M;
}

// status: 0 ok, 1 negative parameter, 2 correction failed, 3 infinite coalescence time, 4 bad structure, 5 numeric
int eval_candidate(const Model& m, double split_in, const double* par, const double* jsfs, int n_rep, double* llk, double* jafs_out, double* runaway) {
    for (int r = 0; r < n_rep; ++r) llk[r] = -INF;
    if (jafs_out) for (int i = 0; i < 7; ++i) jafs_out[i] = NAN;
    if (runaway) *runaway = 0.0;
    // grid: a fractional split time splits interval floor(st) in two (MigrationInference.py:89-99)
    if (!(split_in >= 0) || split_in < m.sample_date) return 4;
    std::vector<double> times(m.times, m.times + m.numT - 1);
    std::vector<double> lh(m.lh, m.lh + 2 * m.numT);
    int split = (int)std::floor(split_in);
    const double frac = split_in - split;
    if (split - 1 > (int)times.size()) return 4;
    if (frac != 0.0) {
        if (split > m.numT - 2) return 4;
        const double t1 = frac * times[split], t2 = times[split] - t1;
        times[split] = t1;
        times.insert(times.begin() + split + 1, t2);
        lh.insert(lh.begin() + 2 * (split + 1), {lh[2 * split], lh[2 * split + 1]});
        split += 1;
    }
    const int numT = (int)lh.size() / 2;
    if (split > numT) return 4;
    for (int i = 0; i < m.n_param; ++i) if (par[i] < 0) return 1;
    std::vector<double> mi(2 * numT, 0.0), pu(2 * numT, 0.0);
    for (int b = 0; b < m.n_band; ++b) {
        const Band& B = m.bands[b];
        const int end = B.end < 0 ? split : B.end;
        if (end <= B.start || B.start < m.sample_date || end > numT) return 4;
        for (int t = B.start; t < end; ++t) mi[2 * t + B.pop] = B.param >= 0 ? par[B.param] : B.value;
    }
    for (int p = 0; p < m.n_pulse; ++p) { const Pulse& P = m.pulses[p]; if (P.time >= numT) return 4; pu[2 * P.time + P.pop] = P.param >= 0 ? par[P.param] : P.value; }
    if (split >= numT) return 3;

    // ---- CorrectLambdas (:305-378) ----
    std::vector<double> lc(2 * numT, 1.0);
    PairChain cl;
    cl.mixtureTH = m.mixture_th;
    double p0[2][3] = {{1, 0, 0}, {0, 1, 0}}, nc[2] = {0, 0};
    for (int t = 0; t < split; ++t) {
        const double r = pu[2 * t] + pu[2 * t + 1];                         // :315-323
        if (r > 0) {
            const int a = pu[2 * t] > 0 ? 0 : 1, b = 1 - a;
            for (int k = 0; k < 2; ++k) {
                const double pa = p0[k][a], pb = p0[k][b], pc = p0[k][2];
                p0[k][a] = pa * (1 - r) * (1 - r);
                p0[k][b] = pa * r * r + pb + pc * r;
                p0[k][2] = pa * 2 * (1 - r) * r + pc * (1 - r);
            }
        }
        cl.mu[0] = mi[2 * t]; cl.mu[1] = mi[2 * t + 1];
        if (m.true_eps) { lc[2 * t] = lh[2 * t]; lc[2 * t + 1] = lh[2 * t + 1]; }
        else {
            cl.lh[0] = lh[2 * t]; cl.lh[1] = lh[2 * t + 1]; cl.T = times[t];
            std::memcpy(cl.P0, p0, sizeof p0);
            double sol[2], P1[2][3];
            if (!cl.solve(m.cpfit, sol, P1)) return 5;
            lc[2 * t] = sol[0]; lc[2 * t + 1] = sol[1];
            if (!(sol[0] > 0) || !(sol[1] > 0)) return (sol[0] != sol[0] || sol[1] != sol[1]) ? 5 : 2;
            if (runaway) *runaway = std::max(*runaway, std::max(sol[0], sol[1]) * times[t]);
            std::memcpy(p0, P1, sizeof p0);
        }
        nc[0] = (p0[0][0] + p0[0][1]) + p0[0][2];                           // a probability, used as a log below (:353-354)
        nc[1] = (p0[1][0] + p0[1][1]) + p0[1][2];
    }
    for (int t = split; t < numT - 1; ++t) {                                // :355-370
        const double T = times[t];
        if (T == 0) { lc[2 * t] = lc[2 * t + 1] = 1; continue; }
        double lam;
        if (!m.cpfit) {                                                     // FitSinglePop :82-92
            cl.lh[0] = lh[2 * t]; cl.lh[1] = lh[2 * t + 1]; cl.T = T;
            const double pa = std::exp(nc[0]), pb = std::exp(nc[1]);
            const double w0 = pa / (pa + pb), w1 = pb / (pa + pb);
            const double Te = w0 * cl.ect_one_pop(cl.lh[0]) + w1 * cl.ect_one_pop(cl.lh[1]);
            double x[1] = {w0 * cl.lh[0] + w1 * cl.lh[1]};
            Residual fun = [&](const double* l, double* f) { f[0] = cl.ect_one_pop(l[0]) - Te; return true; };
            trf_bounded(fun, 1, x, 0.01 * std::min(cl.lh[0], cl.lh[1]));
            lam = x[0];
        } else {
            const double pnc = (std::exp(-T * lh[2 * t]) + std::exp(nc[1] - nc[0] - T * lh[2 * t + 1])) / (1 + std::exp(nc[1] - nc[0]));
            lam = -std::log(pnc) / T;
        }
        lc[2 * t] = lc[2 * t + 1] = lam;
        nc[0] -= T * lam; nc[1] -= T * lam;
    }
    {
        const int t = numT - 1;
        const double pr0 = std::exp(nc[0]), pr1 = std::exp(nc[1]);
        const double lam = (pr0 + pr1) / (pr0 / lh[2 * t] + pr1 / lh[2 * t + 1]);
        lc[2 * t] = lc[2 * t + 1] = lam;
    }
    if (m.smooth)                                                           // SmoothConst :387-405
        for (int k = 0; k < 2; ++k) {
            int i = 0;
            double lam = lh[k];
            while (i < split) {
                int j = i;
                double acc = 0, tsum = 0;
                while (std::fabs(lh[2 * j + k] - lam) < 1e-10 && j < numT - 1) { acc += lc[2 * j + k] * times[j]; tsum += times[j]; ++j; if (j == split) break; }
                for (int q = i; q < j; ++q) lc[2 * q + k] = acc / tsum;
                lam = lh[2 * j + k];
                i = j;
            }
        }
    for (double v : lc) if (!(v == v)) return 5;

    // ---- JAFSpectrum (:467-506) ----
    std::vector<double> jafs(7, 0.0), P0v(NS2, 0.0);
    P0v[2] = 1.0;
    for (int it = 0; it < numT; ++it) {
        const bool two = it < split;
        const double mu[2] = {mi[2 * it], mi[2 * it + 1]}, la[2] = {lc[2 * it], lc[2 * it + 1]};
        if (two && it == numT - 1 && mu[0] + mu[1] == 0.0) return 3;
        if (it == m.sample_date) {                                          // AncientSampleP0 :246-262
            std::vector<double> q(NS2, 0.0);
            for (int e = 0; e < TP_N_ANC; ++e) q[TP_ANC[2 * e]] += P0v[TP_ANC[2 * e + 1]];
            P0v = q;
        }
        const double prate = pu[2 * it] + pu[2 * it + 1];
        if (two && prate > 0) {                                             // PulseMigration :361-377
            const int from = pu[2 * it] > 0 ? 0 : 1;
            const int* E = from == 0 ? TP_PULSE0 : TP_PULSE1;
            const int ne = from == 0 ? TP_N_PULSE0 : TP_N_PULSE1;
            std::vector<double> q(NS2, 0.0);
            for (int e = 0; e < ne; ++e) q[E[4 * e]] += std::pow(1.0 - prate, E[4 * e + 2]) * std::pow(prate, E[4 * e + 3]) * P0v[E[4 * e + 1]];
            P0v = q;
        }
        if (it == split) {                                                  // CollapsePops :518-528
            std::vector<double> q(NS1, 0.0);
            for (int g = 0; g < NS1; ++g) for (int i = TP_COLLAPSE[2 * g]; i < TP_COLLAPSE[2 * g + 1]; ++i) q[g] += P0v[i];
            P0v = q;
        }
        Mat M = two ? two_pop_generator(la, mu) : one_pop_generator(la[0]);
        const int* jaf = two ? TP_JAF : OP_JAF;
        const int N = two ? NS2 : NS1;
        std::vector<int> keep;
        const bool del = two && (mu[0] + mu[1] == 0);                       // TwoPopulations.py:231-244
        for (int i = 0; i < N; ++i) { bool d = false; if (del) for (int s = 0; s < TP_N_STATIONARY; ++s) d = d || TP_STATIONARY[s] == i; if (!d) keep.push_back(i); }
        const int n = (int)keep.size();
        Mat Mr(n);
        std::vector<double> P0r(n);
        for (int a = 0; a < n; ++a) { P0r[a] = P0v[keep[a]]; for (int b = 0; b < n; ++b) Mr(a, b) = M(keep[a], keep[b]); }
        const double T = it < numT - 1 ? times[it] : 0.0;
        std::vector<double> P1r(n, 0.0);
        if (it < numT - 1) {                                                // SolveDifEq :530-540
            Mat Ms = Mr, E;
            for (auto& v : Ms.a) v *= T;
            if (!expm(Ms, E)) return 5;
            P1r = matvec(E, P0r);
        }
        std::vector<double> rhs(n);
        for (int a = 0; a < n; ++a) rhs[a] = P1r[a] - P0r[a];
        if (!lu_solve(Mr, rhs, 1)) return 5;                                // integral = M^-1 (P1 - P0)
        std::vector<double> P1f(N, 0.0), If(N, 0.0);
        for (int a = 0; a < n; ++a) { P1f[keep[a]] = P1r[a]; If[keep[a]] = rhs[a]; }
        if (del) {                                                          // mass of the deleted stationary states :264-309
            std::vector<double> nP = P1f, nI = If;
            for (int s = 0; s < TP_N_STATIONARY; ++s) {
                const int ind = TP_STATIONARY[s];
                for (int i = 0; i < N; ++i)
                    if (TP_SIG[2 * i] == TP_SIG[2 * ind] && TP_SIG[2 * i + 1] == TP_SIG[2 * ind + 1]) { nP[ind] += P0v[i] - P1f[i]; nI[ind] += T * P0v[i] - If[i]; }
            }
            P1f = nP; If = nI;
        }
        P0v = P1f;
        for (int i = 0; i < N; ++i)
            for (int c = 0; c < 7; ++c) {
                if (it < m.sample_date && c >= 2) continue;                 // classes of the second genome before its sample date :503-505
                jafs[c] += jaf[7 * i + c] * If[i];
            }
    }
    double norm = 0;
    for (double v : jafs) norm += v;
    for (double& v : jafs) v /= norm;
    for (double v : jafs) if (!(v == v)) return 5;
    if (jafs_out) for (int i = 0; i < 7; ++i) jafs_out[i] = jafs[i];
    // ---- multinomial log-likelihood (:600-609) with llh_const (:217-227) ----
    for (int r = 0; r < n_rep; ++r) {
        const double* d = jsfs + 8 * r + 1;
        double snps = 0;
        for (int i = 0; i < 7; ++i) snps += d[i];
        double c = std::lgamma(snps + 1);
        double v = 0;
        if (m.unfolded) { for (int i = 0; i < 7; ++i) c -= std::lgamma(d[i] + 1); v = c; for (int i = 0; i < 7; ++i) v += d[i] * std::log(jafs[i]); }
        else {
            c -= std::lgamma(d[0] + d[6] + 1); c -= std::lgamma(d[1] + d[5] + 1); c -= std::lgamma(d[2] + d[4] + 1); c -= std::lgamma(d[3] + 1);
            v = c;
            v += (d[0] + d[6]) * std::log(jafs[0] + jafs[6]);
            v += (d[1] + d[5]) * std::log(jafs[1] + jafs[5]);
            v += (d[2] + d[4]) * std::log(jafs[2] + jafs[4]);
            v += d[3] * std::log(jafs[3]);
        }
        llk[r] = v;
    }
    return 0;
}

}  // namespace

extern "C" {

struct misti_cpu_band { int32_t pop, start, end, param; double value; };
struct misti_cpu_pulse { int32_t pop, time, param, pad; double value; };

// One candidate per OpenMP task.  flags: 1 cpfit, 2 trueEPS, 4 smooth, 8 unfolded (as include/misti_hip.h).
// llk [n_cand][n_rep], jafs [n_cand][7] or NULL, status [n_cand], runaway [n_cand] or NULL.  Returns the threads used.
int misti_cpu_eval(int numT, int sample_date, unsigned flags, double mixture_th, const double* times, const double* lh,
                   int n_band, const misti_cpu_band* bands, int n_pulse, const misti_cpu_pulse* pulses, int n_param,
                   int64_t n_cand, const double* split_time, const double* params, int64_t n_rep, const double* jsfs,
                   double* llk, double* jafs, int32_t* status, double* runaway, int threads) {
    std::vector<Band> B(n_band);
    std::vector<Pulse> P(n_pulse);
    for (int i = 0; i < n_band; ++i) B[i] = {bands[i].pop, bands[i].start, bands[i].end, bands[i].param, bands[i].value};
    for (int i = 0; i < n_pulse; ++i) P[i] = {pulses[i].pop, pulses[i].time, pulses[i].param, pulses[i].value};
    Model m{numT, sample_date, n_band, n_pulse, n_param, (flags & 1) != 0, (flags & 2) != 0, (flags & 4) != 0, (flags & 8) != 0,
            mixture_th, times, lh, B.data(), P.data()};
    int used = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
    for (int64_t c = 0; c < n_cand; ++c) {
        if (g_noise_seed >= 0) t_noise_state = (uint64_t)g_noise_seed * 0x100000001b3ull + (uint64_t)c * 0x9e3779b97f4a7c15ull + 1;
        status[c] = eval_candidate(m, split_time[c], n_param ? params + c * n_param : nullptr, jsfs, (int)n_rep, llk + c * n_rep,
                                   jafs ? jafs + 7 * c : nullptr, runaway ? runaway + c : nullptr);
    }
#ifdef _OPENMP
    used = threads > 0 ? threads : 1;
#endif
    return used;
}

// seed >= 0: every following misti_cpu_eval moves each entry of the pair chain's expm by -1, 0 or +1 ulp (generator = seed x candidate);
// seed < 0: off.  Process-wide; a study hook of the checker, never used on a timed or compared path.
void misti_cpu_set_expm_noise(int seed) { g_noise_seed = seed; }

}  // extern "C"

// hmx_oracle_body.hpp -- TEST INFRASTRUCTURE (part of the CPU oracle, see hmx_oracle.cpp).  The coefficient-typed part
// of the restatement (generator, compressors, block payloads, leaf products), written against `scalar` and included
// twice: namespace orc::f64 (scalar = double) and orc::f32 (scalar = float, htool's HMatrix<float,double>).

// ---------------------------------------------------------------------------------------------
// Generator: K(x,y) = 1/(delta + scale*|x-y|); examples/use_hmatrix.cpp:24-35,
// testing/generator_test.hpp:155-187; column-major output (hmatrix/interfaces/virtual_generator.hpp:24)
// evaluated through the permutations (virtual_generator.hpp:46-48).
// ---------------------------------------------------------------------------------------------
struct Generator {
    int dim;
    const double *xt, *xs;
    const int *pt, *ps;
    double delta, scale;
    // complex coefficients: (cre + i cim sgn) / (delta + scale |x-y|), sgn = 1 (complex symmetric form of
    // testing/generator_test.hpp:163-170,189-196) or sign(x_t[0] - x_s[0]) (Hermitian form, :198-205)
    double cre = 1, cim = 0;
    int hermitian = 0;
    // kernel family (include/hmx.h hmx_kernel): 0 the inverse distance above; 1 Helmholtz exp(i k r) / (delta + scale r) (real types:
    // its real part); 2 Laplace single layer (cre + i cim) / (4 pi (delta + r)).  Taken from orc_set_kernel_family at construction.
    int family        = orc_kernel_family();
    double wavenumber = orc_kernel_wavenumber();
    inline double distance(int i, int j) const {
        double s = 0;
        for (int p = 0; p < dim; p++) {
            double d = xt[dim * i + p] - xs[dim * j + p];
            s        = s + d * d;
        }
        return std::sqrt(s);
    }
    inline double denominator(int i, int j) const { // user numbering
        double s = 0;
        for (int p = 0; p < dim; p++) {
            double d = xt[dim * i + p] - xs[dim * j + p];
            s        = s + d * d;
        }
        return delta + scale * std::sqrt(s);
    }
    inline scalar value(int i, int j) const {
        if (family == 1) {
            const double r = distance(i, j), den = delta + scale * r;
            double sn, cs;
            orc_sincos(wavenumber * r, sn, cs);
#if ORC_COMPLEX
            return scalar((real)(cs / den), (real)(sn / den));
#else
            return (scalar)(cs / den);
#endif
        }
        if (family == 2) {
            const double den = 12.566370614359172 * (delta + distance(i, j));
#if ORC_COMPLEX
            return scalar((real)(cre / den), (real)(cim / den));
#else
            return (scalar)(cre / den);
#endif
        }
#if ORC_COMPLEX
        const double u   = xt[dim * i] - xs[dim * j];
        const double sgn = hermitian ? (u > 0 ? 1. : (u < 0 ? -1. : 0.)) : 1.;
        return scalar(std::complex<double>(cre, cim * sgn) / denominator(i, j)); // complex<double> / double, then to scalar
#else
        return (scalar)(1. / denominator(i, j));
#endif
    }
    void copy_submatrix(int M, int N, int row_off, int col_off, scalar *ptr) const { // cluster numbering
        for (int j = 0; j < M; j++)
            for (int k = 0; k < N; k++)
                ptr[j + (size_t)M * k] = value(pt[row_off + j], ps[col_off + k]);
    }
};

// conj_if_complex / std::real on either kind of scalar (misc/misc.hpp)
#if ORC_COMPLEX
static inline scalar cj(scalar v) { return std::conj(v); }
static inline real re_part(scalar v) { return v.real(); }
#else
static inline scalar cj(scalar v) { return v; }
static inline real re_part(scalar v) { return v; }
#endif

// ---------------------------------------------------------------------------------------------
// Compressors
// ---------------------------------------------------------------------------------------------
struct LowRank {
    int M = 0, N = 0, rank = 0;
    std::vector<scalar> U; // M x r column-major
    std::vector<scalar> V; // r x N column-major
    std::vector<int> pivots; // (I,J) per accepted iteration, for parity checks
};

// Blas<T>::dot is htool's own loop, conjugating the FIRST argument (wrappers/wrapper_blas.hpp:152-157)
static scalar plain_dot(int n, const scalar *a, const scalar *b) {
    scalar s = scalar();
    for (int i = 0; i < n; i++)
        s += cj(a[i]) * b[i];
    return s;
}

// hmatrix/lrmat/partialACA.hpp:42-184.  Returns true on success.  int32 arithmetic in the
// "not advantageous" test is kept on purpose.
static bool partial_aca(const Generator &A, int M, int N, int row_off, int col_off, real epsilon, int reqrank, LowRank &lr) {
    int I = 0, J = 0, q = 0;
    std::vector<std::vector<scalar>> uu, vv;
    std::vector<bool> vrow(M, false), vcol(N, false);
    real frob = 0, aux = 0, pivot, tmp;
    std::vector<scalar> r(N), c(M);
    lr.pivots.clear();
    while (((reqrank > 0) && (q < std::min(reqrank, std::min(M, N)))) || ((reqrank < 0) && (q == 0 || sqrt(aux / frob) > epsilon))) {
        q += 1;
        if (q * (M + N) > (M * N)) {
            q = -1;
            break;
        }
        std::fill(r.begin(), r.end(), scalar(0));
        A.copy_submatrix(1, N, I + row_off, col_off, r.data());
        for (size_t j = 0; j < uu.size(); j++) {
            scalar coef = -uu[j][I];
            for (int k = 0; k < N; k++)
                r[k] += coef * vv[j][k]; // axpy
        }
        pivot = 0.;
        for (int k = 0; k < N; k++) {
            if (vcol[k])
                continue;
            tmp = std::abs(r[k]);
            if (tmp < pivot)
                continue;
            pivot = tmp;
            J     = k;
        }
        vrow[I]      = true;
        scalar gamma = scalar(1.) / r[J];
        if (std::abs(r[J]) > 1e-15) {
            std::fill(c.begin(), c.end(), scalar(0));
            A.copy_submatrix(M, 1, row_off, J + col_off, c.data());
            for (size_t k = 0; k < uu.size(); k++) {
                scalar coef = -vv[k][J];
                for (int i = 0; i < M; i++)
                    c[i] += coef * uu[k][i];
            }
            for (auto &v : c)
                v = v * gamma;
            lr.pivots.push_back(I);
            lr.pivots.push_back(J);
            pivot = 0.;
            for (int k = 0; k < M; k++) {
                if (vrow[k])
                    continue;
                tmp = std::abs(c[k]);
                if (tmp < pivot)
                    continue;
                pivot = tmp;
                I     = k;
            }
            vcol[J] = true;
            if (reqrank < 0) {
                scalar frob_aux = scalar(0);
                aux             = std::abs(plain_dot(M, c.data(), c.data())) * std::abs(plain_dot(N, r.data(), r.data()));
                for (size_t j = 0; j < uu.size(); j++)
                    frob_aux += plain_dot(N, vv[j].data(), r.data()) * plain_dot(M, uu[j].data(), c.data());
                frob += aux + 2 * re_part(frob_aux);
            }
            uu.push_back(c);
            vv.push_back(r);
        } else {
            q -= 1;
            if (q == 0)
                q = -1;
            break;
        }
    }
    lr.M = M;
    lr.N = N;
    if (q > 0) {
        lr.rank = q;
        lr.U.resize((size_t)M * q);
        lr.V.resize((size_t)q * N);
        for (int k = 0; k < q; k++) {
            std::copy(uu[k].begin(), uu[k].end(), lr.U.begin() + (size_t)k * M);
            for (int j = 0; j < N; j++)
                lr.V[k + (size_t)q * j] = vv[k][j];
        }
        return true;
    }
    lr.rank = 0;
    return false;
}

// hmatrix/lrmat/sympartialACA.hpp:41-216
static bool sympartial_aca(const Generator &A, int M, int N, int row_off, int col_off, real epsilon, int reqrank, LowRank &lr) {
    int n1, n2, i1, i2;
    bool rows_first = row_off >= col_off;
    if (rows_first) {
        n1 = M;
        n2 = N;
        i1 = row_off;
        i2 = col_off;
    } else {
        n1 = N;
        n2 = M;
        i1 = col_off;
        i2 = row_off;
    }
    int I1 = 0, I2 = 0, q = 0;
    std::vector<std::vector<scalar>> uu, vv;
    std::vector<bool> v1(n1, false), v2(n2, false);
    real frob = 0, aux = 0, pivot, tmp;
    std::vector<scalar> u1(n2), u2(n1);
    lr.pivots.clear();
    while (((reqrank > 0) && (q < std::min(reqrank, std::min(n1, n2)))) || ((reqrank < 0) && (q == 0 || sqrt(aux / frob) > epsilon))) {
        q += 1;
        if (q * (n1 + n2) > (n1 * n2)) {
            q = -1;
            break;
        }
        std::fill(u1.begin(), u1.end(), scalar(0));
        if (rows_first)
            A.copy_submatrix(1, n2, i1 + I1, i2, u1.data());
        else
            A.copy_submatrix(n2, 1, i2, i1 + I1, u1.data());
        for (size_t j = 0; j < uu.size(); j++) {
            scalar coef = -uu[j][I1];
            for (int k = 0; k < n2; k++)
                u1[k] += coef * vv[j][k];
        }
        pivot = 0.;
        for (int k = 0; k < n2; k++) {
            if (v2[k])
                continue;
            tmp = std::abs(u1[k]);
            if (tmp < pivot)
                continue;
            pivot = tmp;
            I2    = k;
        }
        v1[I1]       = true;
        scalar gamma = scalar(1.) / u1[I2];
        if (std::abs(u1[I2]) > 1e-15) {
            std::fill(u2.begin(), u2.end(), scalar(0));
            if (rows_first)
                A.copy_submatrix(n1, 1, i1, i2 + I2, u2.data());
            else
                A.copy_submatrix(1, n1, i2 + I2, i1, u2.data());
            for (size_t k = 0; k < uu.size(); k++) {
                scalar coef = -vv[k][I2];
                for (int i = 0; i < n1; i++)
                    u2[i] += coef * uu[k][i];
            }
            for (auto &v : u2)
                v = v * gamma;
            lr.pivots.push_back(I1);
            lr.pivots.push_back(I2);
            pivot = 0.;
            for (int k = 0; k < n1; k++) {
                if (v1[k])
                    continue;
                tmp = std::abs(u2[k]);
                if (tmp < pivot)
                    continue;
                pivot = tmp;
                I1    = k;
            }
            v2[I2] = true;
            if (reqrank < 0) {
                scalar frob_aux = scalar(0);
                aux             = std::abs(plain_dot(n1, u2.data(), u2.data())) * std::abs(plain_dot(n2, u1.data(), u1.data()));
                for (size_t j = 0; j < uu.size(); j++)
                    frob_aux += plain_dot(n2, u1.data(), vv[j].data()) * plain_dot(n1, u2.data(), uu[j].data());
                frob += aux + 2 * re_part(frob_aux);
            }
            uu.push_back(u2);
            vv.push_back(u1);
        } else {
            q -= 1;
            if (q == 0)
                q = -1;
            break;
        }
    }
    lr.M = M;
    lr.N = N;
    if (q > 0) {
        lr.rank = q;
        lr.U.resize((size_t)M * q);
        lr.V.resize((size_t)q * N);
        for (int k = 0; k < q; k++) {
            const auto &ucol = rows_first ? uu[k] : vv[k];
            const auto &vrow = rows_first ? vv[k] : uu[k];
            std::copy(ucol.begin(), ucol.end(), lr.U.begin() + (size_t)k * M);
            for (int j = 0; j < N; j++)
                lr.V[k + (size_t)q * j] = vrow[j];
        }
        return true;
    }
    lr.rank = 0;
    return false;
}

// matrix/utils/math.hpp:7-16
static real norm_frob(const std::vector<scalar> &mat, int M, int N) {
    real norm = 0;
    for (int j = 0; j < M; j++)
        for (int k = 0; k < N; k++)
            norm = norm + std::pow(std::abs(mat[j + (size_t)M * k]), 2);
    return sqrt(norm);
}

// hmatrix/lrmat/fullACA.hpp:38-88
static bool full_aca(const Generator &A, int M, int N, int row_off, int col_off, real epsilon, int reqrank, LowRank &lr) {
    std::vector<scalar> mat((size_t)M * N);
    A.copy_submatrix(M, N, row_off, col_off, mat.data());
    int q = 0;
    std::vector<std::vector<scalar>> uu, vv;
    real Norm = norm_frob(mat, M, N);
    lr.pivots.clear();
    while (((reqrank > 0) && (q < std::min(reqrank, std::min(M, N)))) || ((reqrank < 0) && (norm_frob(mat, M, N) / Norm > epsilon || q == 0))) {
        q += 1;
        if (q * (M + N) > (M * N)) {
            q = -1;
            break;
        }
        // matrix/utils/math.hpp:18-23: std::max_element => first maximum in column-major order
        int p        = std::max_element(mat.begin(), mat.end(), [](scalar a, scalar b) { return std::abs(a) < std::abs(b); }) - mat.begin();
        int pi       = p % M, pj = p / M;
        scalar pivot = mat[pi + (size_t)M * pj];
        if (std::abs(pivot) < 1e-15) {
            q += -1;
            break;
        }
        lr.pivots.push_back(pi);
        lr.pivots.push_back(pj);
        std::vector<scalar> col(M), row(N);
        for (int i = 0; i < M; i++)
            col[i] = mat[i + (size_t)M * pj];
        for (int j = 0; j < N; j++)
            row[j] = mat[pi + (size_t)M * j] / pivot;
        uu.push_back(col);
        vv.push_back(row);
        for (int i = 0; i < M; i++)
            for (int j = 0; j < N; j++)
                mat[i + (size_t)M * j] -= uu[q - 1][i] * vv[q - 1][j];
    }
    lr.M = M;
    lr.N = N;
    if (q > 0) {
        lr.rank = q;
        lr.U.resize((size_t)M * q);
        lr.V.resize((size_t)q * N);
        for (int k = 0; k < q; k++) {
            std::copy(uu[k].begin(), uu[k].end(), lr.U.begin() + (size_t)k * M);
            for (int j = 0; j < N; j++)
                lr.V[k + (size_t)q * j] = vv[k][j];
        }
        return true;
    }
    lr.rank = 0;
    return false;
}

// One-sided Jacobi SVD of an M x N column-major matrix, standing in for LAPACK gesvd('A','A')
// (matrix/utils/SVD_truncation.hpp:30-33).  LAPACK is a third-party dependency absent from
// /root/reference (vendor/version unpinned); gesvd's published contract -- singular
// values descending, A = u diag(s) vt -- is what is restated.  Singular vectors are unique only up to
// sign, so parity on U,V is checked through the product U*V and the singular values.
// Returns s (min(M,N)), u (M x min) and vt (min x N) -- the thin factors, which is all SVD.hpp uses.
static void jacobi_svd(int M, int N, const std::vector<scalar> &Ain, std::vector<real> &s, std::vector<scalar> &u, std::vector<scalar> &vt) {
    bool transposed = M < N;
    int m = transposed ? N : M, n = transposed ? M : N; // work on tall m x n
    std::vector<scalar> W((size_t)m * n), Vm((size_t)n * n, scalar(0));
    for (int i = 0; i < M; i++)
        for (int j = 0; j < N; j++) {
            scalar v = Ain[i + (size_t)M * j];
            if (transposed)
                W[j + (size_t)m * i] = v;
            else
                W[i + (size_t)m * j] = v;
        }
    for (int i = 0; i < n; i++)
        Vm[i + (size_t)n * i] = scalar(1);
    for (int sweep = 0; sweep < 60; sweep++) {
        real off = 0;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                scalar *wp = &W[(size_t)m * p], *wq = &W[(size_t)m * q];
                real app = 0, aqq = 0;
                scalar apq = scalar(0);
                for (int i = 0; i < m; i++) {
                    app += re_part(cj(wp[i]) * wp[i]);
                    aqq += re_part(cj(wq[i]) * wq[i]);
                    apq += cj(wp[i]) * wq[i];
                }
                const real absq = std::abs(apq);
                if (absq <= 1e-300 || absq <= 1e-17 * std::sqrt(app * aqq))
                    continue;
                off = std::max(off, absq / std::sqrt(app * aqq));
#if ORC_COMPLEX
                // turn column q by e^{-i phi} (a^H c = |apq| e^{i phi}) so that the inner product is real and positive
                const scalar ph = cj(apq) / absq;
                const real zeta = (aqq - app) / (real(2) * absq);
#else
                const scalar ph = scalar(1);
                const real zeta = (aqq - app) / (2. * apq);
#endif
                real t  = (zeta >= 0 ? 1. : -1.) / (std::abs(zeta) + std::sqrt(1. + zeta * zeta));
                real cs = 1. / std::sqrt(1. + t * t), sn = cs * t;
                for (int i = 0; i < m; i++) {
                    scalar a = wp[i], b = wq[i] * ph;
                    wp[i] = cs * a - sn * b;
                    wq[i] = sn * a + cs * b;
                }
                scalar *vp = &Vm[(size_t)n * p], *vq = &Vm[(size_t)n * q];
                for (int i = 0; i < n; i++) {
                    scalar a = vp[i], b = vq[i] * ph;
                    vp[i] = cs * a - sn * b;
                    vq[i] = sn * a + cs * b;
                }
            }
        if (off < 1e-15)
            break;
    }
    std::vector<real> sv(n);
    std::vector<int> order(n);
    for (int j = 0; j < n; j++) {
        real nn = 0;
        for (int i = 0; i < m; i++)
            nn += re_part(cj(W[i + (size_t)m * j]) * W[i + (size_t)m * j]);
        sv[j] = std::sqrt(nn);
    }
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return sv[a] > sv[b]; });
    int k = n; // = min(M,N)
    s.resize(k);
    u.assign((size_t)M * k, scalar(0));
    vt.assign((size_t)k * N, scalar(0));
    for (int jj = 0; jj < k; jj++) {
        int j     = order[jj];
        s[jj]     = sv[j];
        real is = sv[j] > 0 ? 1. / sv[j] : 0.;
        if (!transposed) { // A = (W/s) s Vm^H
            for (int i = 0; i < M; i++)
                u[i + (size_t)M * jj] = W[i + (size_t)m * j] * is;
            for (int c = 0; c < N; c++)
                vt[jj + (size_t)k * c] = cj(Vm[c + (size_t)n * j]);
        } else { // A^T = (W/s) s Vm^H  =>  A = conj(Vm) s (W/s)^T
            for (int i = 0; i < M; i++)
                u[i + (size_t)M * jj] = cj(Vm[i + (size_t)n * j]);
            for (int c = 0; c < N; c++)
                vt[jj + (size_t)k * c] = W[c + (size_t)m * j] * is;
        }
    }
}

// hmatrix/lrmat/SVD.hpp:27-62 (auto) and :64-92 (fixed rank); truncation rule
// matrix/utils/SVD_truncation.hpp:37-52
static bool svd_compress(const Generator &A, int M, int N, int row_off, int col_off, real epsilon, int reqrank, LowRank &lr, std::vector<real> *sing_out = nullptr) {
    std::vector<scalar> mat((size_t)M * N);
    A.copy_submatrix(M, N, row_off, col_off, mat.data());
    std::vector<real> s;
    std::vector<scalar> u, vt;
    jacobi_svd(M, N, mat, s, u, vt);
    if (sing_out)
        *sing_out = s;
    int k = s.size();
    int truncated_rank;
    {
        int j           = k;
        real svd_norm = 0, error = 0;
        for (auto &e : s)
            svd_norm += e * e;
        svd_norm = std::sqrt(svd_norm);
        do {
            j = j - 1;
            error += std::pow(std::abs(s[j]), 2);
        } while (j > 0 && std::sqrt(error) / svd_norm < epsilon);
        truncated_rank = j + 1;
    }
    lr.M = M;
    lr.N = N;
    lr.pivots.clear();
    if (reqrank > 0) {
        truncated_rank = std::min(reqrank, std::min(M, N));
    } else {
        if (truncated_rank * (M + N) > (M * N)) {
            lr.rank = 0;
            return false;
        }
        if (truncated_rank <= 0) {
            lr.rank = 0;
            return false;
        }
    }
    int r   = truncated_rank;
    lr.rank = r;
    lr.U.resize((size_t)M * r);
    lr.V.resize((size_t)r * N);
    for (int i = 0; i < M; i++)
        for (int j = 0; j < r; j++)
            lr.U[i + (size_t)M * j] = u[i + (size_t)M * j] * s[j];
    for (int i = 0; i < r; i++)
        for (int j = 0; j < N; j++)
            lr.V[i + (size_t)r * j] = vt[i + (size_t)k * j];
    return true;
}

// SVD_recompression (hmatrix/lrmat/utils/SVD_recompression.hpp:19-181), branch rank <= min(M,N):
// U = Q1 R (geqrf), V = L Q2 (gelqf), SVD(R L) = u S vt, truncation (SVD_truncation.hpp:37-52), and only if the rank
// drops: U' = Q1 (u sqrt(S)), V' = (sqrt(S) vt) Q2.  LAPACK's Householder QR/LQ are restated by two-pass modified
// Gram-Schmidt (explicit thin Q); the r x r SVD by the Jacobi routine above.
static void thin_qr(int m, int r, const std::vector<scalar> &A, std::vector<scalar> &Q, std::vector<scalar> &R) { // A m x r col-major
    Q = A;
    R.assign((size_t)r * r, scalar(0));
    for (int j = 0; j < r; j++) {
        for (int pass = 0; pass < 2; pass++)
            for (int i = 0; i < j; i++) {
                scalar d = scalar(0);
                for (int k = 0; k < m; k++)
                    d += cj(Q[k + (size_t)m * i]) * Q[k + (size_t)m * j];
                R[i + (size_t)r * j] += d;
                for (int k = 0; k < m; k++)
                    Q[k + (size_t)m * j] -= d * Q[k + (size_t)m * i];
            }
        real nn = 0;
        for (int k = 0; k < m; k++)
            nn += re_part(cj(Q[k + (size_t)m * j]) * Q[k + (size_t)m * j]);
        nn                  = std::sqrt(nn);
        R[j + (size_t)r * j] = scalar(nn);
        for (int k = 0; k < m; k++)
            Q[k + (size_t)m * j] = nn > 0 ? Q[k + (size_t)m * j] / nn : scalar(0);
    }
}
static void svd_recompression(LowRank &lr, real epsilon) {
    const int M = lr.M, N = lr.N, r = lr.rank;
    if (r <= 0 || r > std::min(M, N))
        return;
    std::vector<scalar> Q1, R, Vt((size_t)N * r), Q2t, Rv;
    thin_qr(M, r, lr.U, Q1, R);
    for (int k = 0; k < r; k++)
        for (int j = 0; j < N; j++)
            Vt[j + (size_t)N * k] = lr.V[k + (size_t)r * j];
    thin_qr(N, r, Vt, Q2t, Rv); // V^T = Q2^T Rv  =>  V = Rv^T Q2 = L Q2
    std::vector<scalar> RL((size_t)r * r, scalar(0)); // R * L, L = Rv^T
    for (int i = 0; i < r; i++)
        for (int j = 0; j < r; j++) {
            scalar s = scalar(0);
            for (int l = 0; l < r; l++)
                s += R[i + (size_t)r * l] * Rv[j + (size_t)r * l];
            RL[i + (size_t)r * j] = s;
        }
    std::vector<real> s;
    std::vector<scalar> u, vt;
    jacobi_svd(r, r, RL, s, u, vt);
    int k;
    {
        int j         = r;
        real svd_norm = 0, error = 0;
        for (auto &e : s)
            svd_norm += e * e;
        svd_norm = std::sqrt(svd_norm);
        do {
            j = j - 1;
            error += std::pow(std::abs(s[j]), 2);
        } while (j > 0 && std::sqrt(error) / svd_norm < epsilon);
        k = j + 1;
    }
    if (k >= r)
        return;
    std::vector<scalar> nU((size_t)M * k, scalar(0)), nV((size_t)k * N, scalar(0));
    for (int c = 0; c < k; c++) {
        const real rs = std::sqrt(s[c]);
        for (int l = 0; l < r; l++) {
            const scalar cu = u[l + (size_t)r * c] * rs, cv = vt[c + (size_t)r * l] * rs;
            for (int i = 0; i < M; i++)
                nU[i + (size_t)M * c] += Q1[i + (size_t)M * l] * cu;
            for (int j = 0; j < N; j++)
                nV[c + (size_t)k * j] += cv * Q2t[j + (size_t)N * l];
        }
    }
    lr.U.swap(nU);
    lr.V.swap(nV);
    lr.rank = k;
    lr.pivots.clear(); // the ACA pivots no longer describe these factors
}


enum Compressor { PARTIAL_ACA = 0,
                  SYMPARTIAL_ACA = 1,
                  FULL_ACA = 2,
                  SVD = 3 };

static bool compress(int kind, const Generator &A, int M, int N, int ro, int co, real eps, int reqrank, LowRank &lr) {
    switch (kind) {
    case PARTIAL_ACA:
        return partial_aca(A, M, N, ro, co, eps, reqrank, lr);
    case SYMPARTIAL_ACA:
        return sympartial_aca(A, M, N, ro, co, eps, reqrank, lr);
    case FULL_ACA:
        return full_aca(A, M, N, ro, co, eps, reqrank, lr);
    default:
        return svd_compress(A, M, N, ro, co, eps, reqrank, lr);
    }
}

// ---------------------------------------------------------------------------------------------
// Block tree: hmatrix/tree_builder/tree_builder.hpp:417-566
// ---------------------------------------------------------------------------------------------
struct Block {
    const Cluster *t, *s;
    char symmetry = 'N', uplo = 'N';
    std::vector<std::unique_ptr<Block>> children;
    // leaf payload
    int kind = 0; // 0 hierarchical, 1 dense, 2 low rank
    std::vector<scalar> dense;
    LowRank lr;
    bool admissible_task = false;
    bool is_leaf() const { return children.empty(); }
};

struct HMat {
    std::unique_ptr<Block> root;
    const Cluster *root_t = nullptr, *root_s = nullptr; // after reset_root_of_block_tree
    char sym = 'N', uplo = 'N';
    char sym_for_leaves = 'N', uplo_for_leaves = 'N';
    int false_positive = 0;
    // flat views
    struct Leaf {
        Block *b;
        bool mirror;
    };
    std::vector<Leaf> preorder;  // natural order (children in creation order)
    std::vector<Leaf> dfs_order; // get_leaves_from order (hmatrix/hmatrix.hpp:247-274): explicit stack, last child first
    // flat-construction storage (from_blocks): owns clusters
    std::vector<std::unique_ptr<Cluster>> owned_clusters;
    std::vector<std::unique_ptr<Block>> owned_blocks;
};

struct BuildParams {
    double eta;
    char sym, uplo;
    int mint, mins;
    int target_partition;
    int partition_for_symmetry;
    bool consistent;
    const Cluster *troot, *sroot;
};

// hmatrix/interfaces/virtual_admissibility_condition.hpp:20-23
static bool admissible(const Cluster &t, const Cluster &s, double eta) {
    std::vector<double> diff(t.center.size());
    for (size_t i = 0; i < diff.size(); i++)
        diff[i] = t.center[i] - s.center[i];
    return 2 * std::min(t.radius, s.radius) < eta * std::max((norm2(diff) - t.radius - s.radius), 0.);
}
// tree_builder.hpp:92-94
static bool in_partition(const BuildParams &P, const Cluster &c) { return P.target_partition == -1 ? true : (P.target_partition == c.rank); }
// tree_builder.hpp:95-111
static bool removed_by_symmetry(const BuildParams &P, const Cluster &t, const Cluster &s) {
    if (P.sym == 'N')
        return false;
    int ps = P.partition_for_symmetry;
    if (P.uplo == 'U' && t.offset >= (s.offset + s.size)) {
        if (ps == -1)
            return true;
        const Cluster *sp = P.sroot->td->on_partition[ps], *tp = P.troot->td->on_partition[ps];
        return s.offset >= sp->offset && tp->offset <= t.offset && t.offset + t.size <= tp->offset + tp->size;
    }
    if (P.uplo == 'L' && s.offset >= (t.offset + t.size)) {
        if (ps == -1)
            return true;
        const Cluster *sp = P.sroot->td->on_partition[ps], *tp = P.troot->td->on_partition[ps];
        return s.offset < sp->offset + sp->size && tp->offset <= t.offset && t.offset + t.size <= tp->offset + tp->size;
    }
    return false;
}
// tree_builder.hpp:125-132
static void set_symmetry(const BuildParams &P, Block &b) {
    if (P.sym != 'N' && b.t->offset == b.s->offset && b.t->size == b.s->size) {
        b.symmetry = P.sym;
        b.uplo     = P.uplo;
    }
}
// cluster_node.hpp:89-96
static bool contains(const Cluster &a, const Cluster &b) { return a.offset <= b.offset && a.size + a.offset >= b.size + b.offset; }

static Block *add_child(Block &parent, const Cluster *t, const Cluster *s) {
    auto b = std::make_unique<Block>();
    b->t   = t;
    b->s   = s;
    parent.children.push_back(std::move(b));
    return parent.children.back().get();
}

// tree_builder.hpp:417-531
static void build_block_tree(const BuildParams &P, Block *cur, std::vector<Block *> &adm, std::vector<Block *> &dense) {
    const Cluster &t = *cur->t, &s = *cur->s;
    bool is_adm      = admissible(t, s, P.eta);
    auto recurse     = [&](const Cluster *tc, const Cluster *sc) {
        Block *ch = add_child(*cur, tc, sc);
        set_symmetry(P, *ch);
        build_block_tree(P, ch, adm, dense);
    };
    auto tchild_ok = [&](const Cluster &tc) { return in_partition(P, tc) || tc.rank < 0; };
    if (is_adm && in_partition(P, t) && !removed_by_symmetry(P, t, s) && t.depth >= P.mint && s.depth >= P.mins && t.rank >= 0 && (!P.consistent || s.rank >= 0)) {
        adm.push_back(cur);
        cur->admissible_task = true;
    } else if (s.is_leaf() && t.is_leaf()) {
        dense.push_back(cur);
    } else if (s.is_leaf() && !t.is_leaf()) {
        for (auto &tc : t.children)
            if (tchild_ok(*tc) && !removed_by_symmetry(P, *tc, s))
                recurse(tc.get(), &s);
    } else if (!s.is_leaf() && t.is_leaf()) {
        for (auto &sc : s.children)
            if (!removed_by_symmetry(P, t, *sc))
                recurse(&t, sc.get());
    } else if (P.consistent) {
        if (t.rank < 0 && s.rank >= 0) {
            for (auto *tc : t.td->on_partition)
                if (tchild_ok(*tc) && !removed_by_symmetry(P, *tc, s) && contains(t, *tc))
                    recurse(tc, &s);
        } else if (s.rank < 0 && t.rank >= 0) {
            for (auto *sc : s.td->on_partition)
                if (!removed_by_symmetry(P, t, *sc) && contains(s, *sc))
                    recurse(&t, sc);
        } else {
            for (auto &tc : t.children)
                for (auto &sc : s.children)
                    if (tchild_ok(*tc) && !removed_by_symmetry(P, *tc, *sc))
                        recurse(tc.get(), sc.get());
        }
    } else {
        if (t.rank < 0) {
            for (auto *tc : t.td->on_partition)
                if (tchild_ok(*tc) && !removed_by_symmetry(P, *tc, s) && contains(t, *tc))
                    recurse(tc, &s);
        } else if (s.size > t.size) {
            for (auto &sc : s.children)
                if ((in_partition(P, t) || t.rank < 0) && !removed_by_symmetry(P, t, *sc))
                    recurse(&t, sc.get());
        } else if (t.size > s.size) {
            for (auto &tc : t.children)
                if (tchild_ok(*tc) && !removed_by_symmetry(P, *tc, s))
                    recurse(tc.get(), &s);
        } else {
            for (auto &tc : t.children)
                for (auto &sc : s.children)
                    if (tchild_ok(*tc) && !removed_by_symmetry(P, *tc, *sc))
                        recurse(tc.get(), sc.get());
        }
    }
}

// tree_builder.hpp:533-566
static void reset_root(const BuildParams &P, HMat &H) {
    Block &root = *H.root;
    if (!in_partition(P, *root.t)) {
        std::stack<Block *> st;
        st.push(&root);
        std::vector<std::unique_ptr<Block>> new_children;
        while (!st.empty()) {
            Block *cur = st.top();
            st.pop();
            for (auto &child : cur->children) {
                if (child->t->rank == P.target_partition)
                    new_children.push_back(std::move(child));
                else
                    st.push(child.get());
            }
        }
        // keep the detached intermediate nodes alive until we are done, then drop them
        root.children.clear();
        root.children = std::move(new_children);
        root.t        = root.t->td->on_partition[P.target_partition];
    }
}

static void collect_leaves(HMat &H) {
    H.preorder.clear();
    H.dfs_order.clear();
    std::function<void(Block *, bool)> pre = [&](Block *b, bool sym_anc) {
        if (b->is_leaf()) {
            H.preorder.push_back({b, sym_anc && b->t->offset != b->s->offset});
            return;
        }
        for (auto &c : b->children)
            pre(c.get(), sym_anc || b->symmetry != 'N');
    };
    pre(H.root.get(), H.root->symmetry != 'N');
    // hmatrix.hpp:247-274
    std::stack<std::pair<Block *, bool>> st;
    st.push({H.root.get(), H.root->symmetry != 'N'});
    while (!st.empty()) {
        auto cur = st.top();
        st.pop();
        if (cur.first->is_leaf())
            H.dfs_order.push_back({cur.first, cur.second && cur.first->t->offset != cur.first->s->offset});
        for (auto &c : cur.first->children)
            st.push({c.get(), cur.first->symmetry != 'N' || cur.second});
    }
}

// tree_builder.hpp:134-150 (symmetry_for_leaves of the root)
static char symmetry_for_leaves(const Block &b, char sym) {
    if (sym == 'N')
        return 'N';
    if (b.is_leaf())
        return b.symmetry != 'N' ? sym : 'N';
    char res = 'N';
    for (auto &c : b.children) {
        // postorder: children first; parent flagged if any child has symmetry != 'N'
        symmetry_for_leaves(*c, sym);
        if (c->symmetry != 'N')
            res = sym;
    }
    return res;
}

static std::unique_ptr<HMat> build_hmatrix(const ClusterTree &tt, const ClusterTree &st, const Generator &A, real eps, double eta, char sym, char uplo, int reqrank, int compressor, int mint, int mins, int target_partition, int partition_for_symmetry, bool consistent, bool parallel, int root_partition = -1) {
    auto H     = std::make_unique<HMat>();
    H->root    = std::make_unique<Block>();
    // root_partition >= 0: DefaultLocalApproximationBuilder (distributed_operator/utility.hpp:64-88) builds from the
    // partition clusters themselves
    H->root->t = root_partition >= 0 ? tt.td.on_partition[root_partition] : tt.root.get();
    H->root->s = root_partition >= 0 ? st.td.on_partition[root_partition] : st.root.get();
    H->sym     = sym;
    H->uplo    = uplo;
    BuildParams P{eta, sym, uplo, mint, mins, target_partition, partition_for_symmetry, consistent, tt.root.get(), st.root.get()};
    std::vector<Block *> adm, dense;
    build_block_tree(P, H->root.get(), adm, dense);
    reset_root(P, *H);
    set_symmetry(P, *H->root);
    H->root_t = H->root->t;
    H->root_s = H->root->s;
    // sequential_compute_blocks / openmp_compute_blocks (tree_builder.hpp:568-666)
    int fp = 0;
#pragma omp parallel for schedule(guided) reduction(+ : fp) if (parallel)
    for (int p = 0; p < (int)adm.size(); p++) {
        Block *b = adm[p];
        int M = b->t->size, N = b->s->size;
        bool ok = compress(compressor, A, M, N, b->t->offset, b->s->offset, eps, reqrank, b->lr);
        if (ok) {
            b->kind = 2;
        } else {
            b->lr = LowRank();
            b->dense.resize((size_t)M * N);
            A.copy_submatrix(M, N, b->t->offset, b->s->offset, b->dense.data());
            b->kind = 1;
            fp += 1;
        }
    }
#pragma omp parallel for schedule(guided) if (parallel)
    for (int p = 0; p < (int)dense.size(); p++) {
        Block *b = dense[p];
        int M = b->t->size, N = b->s->size;
        b->dense.resize((size_t)M * N);
        A.copy_submatrix(M, N, b->t->offset, b->s->offset, b->dense.data());
        b->kind = 1;
    }
    H->false_positive  = fp;
    H->sym_for_leaves  = symmetry_for_leaves(*H->root, sym);
    if (H->root->is_leaf() && H->root->symmetry != 'N')
        H->sym_for_leaves = sym;
    H->uplo_for_leaves = H->sym_for_leaves != 'N' ? uplo : 'N';
    collect_leaves(*H);
    return H;
}

// ---------------------------------------------------------------------------------------------
// Leaf products: matrix/linalg/add_matrix_vector_product.hpp:10-35 (gemv / symv semantics restated
// as plain loops; BLAS is a third-party dependency, summation order unspecified),
// hmatrix/lrmat/linalg/add_lrmat_vector_product.hpp:9-24
// ---------------------------------------------------------------------------------------------
static void gemv(char trans, int m, int n, scalar alpha, const scalar *A, const scalar *x, scalar beta, scalar *y) {
    if (!(m && n))
        return;
    if (trans == 'N') {
        if (beta != scalar(1))
            for (int i = 0; i < m; i++)
                y[i] = beta == scalar(0) ? scalar(0) : beta * y[i];
        for (int j = 0; j < n; j++) {
            scalar t        = alpha * x[j];
            const scalar *a = A + (size_t)m * j;
            for (int i = 0; i < m; i++)
                y[i] += t * a[i];
        }
    } else { // 'T', or 'C' (conjugated coefficients)
        for (int j = 0; j < n; j++) {
            const scalar *a = A + (size_t)m * j;
            scalar t        = scalar(0);
            for (int i = 0; i < m; i++)
                t += (trans == 'C' ? cj(a[i]) : a[i]) * x[i];
            y[j] = alpha * t + (beta == scalar(0) ? scalar(0) : beta * y[j]);
        }
    }
}
// symv / hemv: only the UPLO triangle of the n x n column-major matrix is referenced; hemv conjugates the mirrored
// entries and uses the real part of the diagonal (matrix/linalg/add_matrix_vector_product.hpp:26-52)
static void symv(char uplo, int n, scalar alpha, const scalar *A, const scalar *x, scalar beta, scalar *y, bool herm = false) {
    if (!n)
        return;
    if (beta != scalar(1))
        for (int i = 0; i < n; i++)
            y[i] = beta == scalar(0) ? scalar(0) : beta * y[i];
    auto diag = [&](int j) { return herm ? scalar(re_part(A[j + (size_t)n * j])) : A[j + (size_t)n * j]; };
    auto mir  = [&](scalar v) { return herm ? cj(v) : v; };
    for (int j = 0; j < n; j++) {
        scalar t1 = alpha * x[j], t2 = scalar(0);
        if (uplo == 'L') {
            y[j] += t1 * diag(j);
            for (int i = j + 1; i < n; i++) {
                y[i] += t1 * A[i + (size_t)n * j];
                t2 += mir(A[i + (size_t)n * j]) * x[i];
            }
        } else {
            for (int i = 0; i < j; i++) {
                y[i] += t1 * A[i + (size_t)n * j];
                t2 += mir(A[i + (size_t)n * j]) * x[i];
            }
            y[j] += t1 * diag(j);
        }
        y[j] += alpha * t2;
    }
}
static void lrmat_vec(char trans, scalar alpha, const LowRank &lr, const scalar *in, scalar beta, scalar *out) {
    int r = lr.rank;
    if (r == 0)
        return; // beta NOT applied (add_lrmat_vector_product.hpp:11)
    std::vector<scalar> a(r);
    if (trans == 'N') {
        gemv('N', r, lr.N, scalar(1), lr.V.data(), in, scalar(0), a.data());
        gemv('N', lr.M, r, alpha, lr.U.data(), a.data(), beta, out);
    } else {
        gemv(trans, lr.M, r, scalar(1), lr.U.data(), in, scalar(0), a.data());
        gemv(trans, r, lr.N, alpha, lr.V.data(), a.data(), beta, out);
    }
}
// hmatrix/linalg/add_hmatrix_vector_product.hpp:17-33
static void leaf_vec(char trans, scalar alpha, const Block &b, const scalar *in, scalar beta, scalar *out) {
    if (b.kind == 1) {
        int M = b.t->size, N = b.s->size;
        if (b.symmetry == 'N')
            gemv(trans, M, N, alpha, b.dense.data(), in, beta, out);
        else
            symv(b.uplo, M, alpha, b.dense.data(), in, beta, out, b.symmetry == 'H');
    } else if (b.kind == 2) {
        lrmat_vec(trans, alpha, b.lr, in, beta, out);
    }
}

// add_hmatrix_vector_product.hpp:57-104 (sequential) -- cluster numbering, local offsets
static void matvec_seq(const HMat &H, char trans, scalar alpha, const scalar *in, scalar beta, scalar *out) {
    int out_size = H.root_t->size;
    int lin = H.root_s->offset, lout = H.root_t->offset;
    char trans_sym = H.sym_for_leaves == 'S' ? 'T' : 'C'; // add_hmatrix_vector_product.hpp:70
    bool tr        = trans != 'N';
    if (tr) {
        out_size  = H.root_s->size;
        lin       = H.root_t->offset;
        lout      = H.root_s->offset;
        trans_sym = 'N';
    }
    if (beta != scalar(1))
        for (int i = 0; i < out_size; i++)
            out[i] = beta * out[i]; // scal
    for (auto &l : H.dfs_order) {
        int io = tr ? l.b->t->offset : l.b->s->offset;
        int oo = tr ? l.b->s->offset : l.b->t->offset;
        leaf_vec(trans, alpha, *l.b, in + io - lin, scalar(1), out + (oo - lout));
    }
    if (H.sym_for_leaves != 'N') {
        for (auto &l : H.dfs_order) {
            if (!l.mirror)
                continue;
            int io = tr ? l.b->t->offset : l.b->s->offset;
            int oo = tr ? l.b->s->offset : l.b->t->offset;
            leaf_vec(trans_sym, alpha, *l.b, in + oo - lin, scalar(1), out + (io - lout));
        }
    }
}
// add_hmatrix_vector_product.hpp:107-170 (OpenMP): per-thread temp with alpha=1 per leaf, critical axpy
static void matvec_omp(const HMat &H, char trans, scalar alpha, const scalar *in, scalar beta, scalar *out) {
    int out_size = H.root_t->size;
    int lin = H.root_s->offset, lout = H.root_t->offset;
    char trans_sym = H.sym_for_leaves == 'S' ? 'T' : 'C'; // add_hmatrix_vector_product.hpp:70
    bool tr        = trans != 'N';
    if (tr) {
        out_size  = H.root_s->size;
        lin       = H.root_t->offset;
        lout      = H.root_s->offset;
        trans_sym = 'N';
    }
    if (beta != scalar(1))
        for (int i = 0; i < out_size; i++)
            out[i] = beta * out[i];
    std::vector<const HMat::Leaf *> mirrors;
    for (auto &l : H.dfs_order)
        if (l.mirror)
            mirrors.push_back(&l);
#pragma omp parallel
    {
        std::vector<scalar> temp(out_size, scalar(0));
#pragma omp for schedule(guided) nowait
        for (int b = 0; b < (int)H.dfs_order.size(); b++) {
            auto &l = H.dfs_order[b];
            int io  = tr ? l.b->t->offset : l.b->s->offset;
            int oo  = tr ? l.b->s->offset : l.b->t->offset;
            leaf_vec(trans, scalar(1), *l.b, in + io - lin, scalar(1), temp.data() + (oo - lout));
        }
        if (H.sym_for_leaves != 'N') {
#pragma omp for schedule(guided) nowait
            for (int b = 0; b < (int)mirrors.size(); b++) {
                auto &l = *mirrors[b];
                int io  = tr ? l.b->t->offset : l.b->s->offset;
                int oo  = tr ? l.b->s->offset : l.b->t->offset;
                leaf_vec(trans_sym, scalar(1), *l.b, in + oo - lin, scalar(1), temp.data() + (io - lout));
            }
        }
#pragma omp critical
        for (int i = 0; i < out_size; i++)
            out[i] += alpha * temp[i];
    }
}

// Row-major multi-RHS: hmatrix/linalg/add_hmatrix_matrix_product_row_major.hpp:58-109,
// matrix/linalg/add_matrix_matrix_product_row_major.hpp:23-46,87-106,
// hmatrix/lrmat/linalg/add_lrmat_matrix_product_row_major.hpp:11-27.  X[n][mu], Y[m][mu] (mu fastest).
static void leaf_mat_rm(char trans, const Block &b, const scalar *in, scalar *out, int mu) {
    int M = b.t->size, N = b.s->size;
    auto dense_rm = [&](char tr, int m, int n, const scalar *A, const scalar *X, scalar *Y) {
        // Y[(out idx)][mu] += op(A) X
        if (tr == 'N') {
            for (int j = 0; j < n; j++)
                for (int i = 0; i < m; i++) {
                    scalar a = A[i + (size_t)m * j];
                    for (int c = 0; c < mu; c++)
                        Y[(size_t)i * mu + c] += a * X[(size_t)j * mu + c];
                }
        } else {
            for (int j = 0; j < n; j++)
                for (int i = 0; i < m; i++) {
                    scalar a = tr == 'C' ? cj(A[i + (size_t)m * j]) : A[i + (size_t)m * j];
                    for (int c = 0; c < mu; c++)
                        Y[(size_t)j * mu + c] += a * X[(size_t)i * mu + c];
                }
        }
    };
    if (b.kind == 1) {
        if (b.symmetry == 'N') {
            dense_rm(trans, M, N, b.dense.data(), in, out);
        } else { // symm: only the UPLO triangle referenced
            for (int j = 0; j < N; j++)
                for (int i = 0; i < M; i++) {
                    bool stored = b.uplo == 'L' ? i >= j : i <= j;
                    scalar a    = stored ? b.dense[i + (size_t)M * j] : (b.symmetry == 'H' ? cj(b.dense[j + (size_t)M * i]) : b.dense[j + (size_t)M * i]);
                    if (b.symmetry == 'H' && i == j)
                        a = scalar(re_part(a));
                    for (int c = 0; c < mu; c++)
                        out[(size_t)i * mu + c] += a * in[(size_t)j * mu + c];
                }
        }
    } else if (b.kind == 2 && b.lr.rank > 0) {
        int r = b.lr.rank;
        std::vector<scalar> a((size_t)r * mu, scalar(0));
        if (trans == 'N') {
            dense_rm('N', r, N, b.lr.V.data(), in, a.data());
            dense_rm('N', M, r, b.lr.U.data(), a.data(), out);
        } else {
            dense_rm(trans, M, r, b.lr.U.data(), in, a.data());
            dense_rm(trans, r, N, b.lr.V.data(), a.data(), out);
        }
    }
}
static void matmat_rm_seq(const HMat &H, char trans, scalar alpha, const scalar *in, scalar beta, scalar *out, int mu) {
    int out_size = H.root_t->size;
    int lin = H.root_s->offset, lout = H.root_t->offset;
    char trans_sym = H.sym_for_leaves == 'S' ? 'T' : 'C'; // add_hmatrix_vector_product.hpp:70
    bool tr        = trans != 'N';
    if (tr) {
        out_size  = H.root_s->size;
        lin       = H.root_t->offset;
        lout      = H.root_s->offset;
        trans_sym = 'N';
    }
    size_t tot = (size_t)out_size * mu;
    if (beta != scalar(1))
        for (size_t i = 0; i < tot; i++)
            out[i] = beta * out[i];
    std::vector<scalar> temp(tot, scalar(0));
    for (auto &l : H.dfs_order) {
        int io = tr ? l.b->t->offset : l.b->s->offset;
        int oo = tr ? l.b->s->offset : l.b->t->offset;
        leaf_mat_rm(trans, *l.b, in + (size_t)(io - lin) * mu, temp.data() + (size_t)(oo - lout) * mu, mu);
    }
    if (H.sym_for_leaves != 'N')
        for (auto &l : H.dfs_order) {
            if (!l.mirror)
                continue;
            int io = tr ? l.b->t->offset : l.b->s->offset;
            int oo = tr ? l.b->s->offset : l.b->t->offset;
            leaf_mat_rm(trans_sym, *l.b, in + (size_t)(oo - lin) * mu, temp.data() + (size_t)(io - lout) * mu, mu);
        }
    for (size_t i = 0; i < tot; i++)
        out[i] += alpha * temp[i];
}


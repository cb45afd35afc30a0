/* host_generator.c -- a user's VirtualGenerator as compiled host code, for the host-generator route of libhmx
 * (hmx_hmatrix_set_callback, include/hmx.h).
 *
 * htool users describe their matrix by subclassing VirtualGenerator and overriding copy_submatrix
 * (hmatrix/interfaces/virtual_generator.hpp:17-31); examples/use_hmatrix.cpp:12-35 is the canonical one:
 * coefficient (j, k) = 1 / (1e-5 + |x_j - y_k|) on a point cloud, the block written column-major.  This file is that
 * generator behind the C callback signature libhmx takes, so that bench.py (--generator callback), the tests and the
 * examples can run the literal drop-in path -- a host generator libhmx knows nothing about -- at full size.  The
 * arithmetic follows the example's sequence (squared differences accumulated from 0 in coordinate order, one sqrt, one
 * multiply-free add of delta, one divide), which is also what the built-in device kernel HMX_KERNEL_INV_DIST does:
 * both routes must produce bit-identical blocks.  Compile without FMA contraction (-ffp-contract=off).
 *
 * Thread-safe (read-only state): libhmx calls it concurrently from all host cores, as htool's OpenMP build loop does
 * (hmatrix/tree_builder/tree_builder.hpp:603-648).
 */
#include <math.h>
#include <stdint.h>

typedef struct {
    int32_t dim;             /* spatial dimension: 2 or 3 */
    int32_t pad;
    const double *target;    /* target points, user numbering, AoS (dim doubles per point) */
    const double *source;    /* source points */
    double delta, scale;     /* K(x, y) = 1 / (delta + scale * |x - y|) */
    double cre, cim;         /* complex variants: numerator cre + i * cim * sgn */
    int32_t hermitian;       /* sgn = sign(x[0] - y[0]) instead of 1 (testing/generator_test.hpp:185-205) */
    int32_t pad2;
} hostgen_inv_dist;

static inline double denominator(const hostgen_inv_dist *g, const double *x, const double *y) {
    double s = 0.0;
    if (g->dim == 3) { /* the common case, unrolled: same operations in the same order */
        const double d0 = x[0] - y[0], d1 = x[1] - y[1], d2 = x[2] - y[2];
        s               = s + d0 * d0;
        s               = s + d1 * d1;
        s               = s + d2 * d2;
        return g->delta + g->scale * sqrt(s);
    }
    for (int p = 0; p < g->dim; p++) {
        const double d = x[p] - y[p];
        s              = s + d * d;
    }
    return g->delta + g->scale * sqrt(s);
}

/* HMatrix<double>: hmx_generator_fn */
void hostgen_inv_dist_f64(void *user, int M, int N, const int32_t *rows, const int32_t *cols, double *out) {
    const hostgen_inv_dist *g = (const hostgen_inv_dist *)user;
    for (int k = 0; k < N; k++) {
        const double *y = g->source + (int64_t)g->dim * cols[k];
        for (int j = 0; j < M; j++)
            out[j + (int64_t)M * k] = 1.0 / denominator(g, g->target + (int64_t)g->dim * rows[j], y);
    }
}

/* HMatrix<float, double>: hmx_generator_fn_s (the coefficient is rounded once, from the fp64 value) */
void hostgen_inv_dist_f32(void *user, int M, int N, const int32_t *rows, const int32_t *cols, float *out) {
    const hostgen_inv_dist *g = (const hostgen_inv_dist *)user;
    for (int k = 0; k < N; k++) {
        const double *y = g->source + (int64_t)g->dim * cols[k];
        for (int j = 0; j < M; j++)
            out[j + (int64_t)M * k] = (float)(1.0 / denominator(g, g->target + (int64_t)g->dim * rows[j], y));
    }
}

/* HMatrix<std::complex<double>>: interleaved (re, im); component-wise division as std::complex<double> / double does */
void hostgen_inv_dist_z64(void *user, int M, int N, const int32_t *rows, const int32_t *cols, double *out) {
    const hostgen_inv_dist *g = (const hostgen_inv_dist *)user;
    for (int k = 0; k < N; k++) {
        const double *y = g->source + (int64_t)g->dim * cols[k];
        for (int j = 0; j < M; j++) {
            const double *x   = g->target + (int64_t)g->dim * rows[j];
            const double den  = denominator(g, x, y);
            const double u    = x[0] - y[0];
            const double sgn  = g->hermitian ? (u > 0 ? 1.0 : (u < 0 ? -1.0 : 0.0)) : 1.0;
            double *o         = out + 2 * (j + (int64_t)M * k);
            o[0]              = g->cre / den;
            o[1]              = (g->cim * sgn) / den;
        }
    }
}

/* HMatrix<std::complex<float>> */
void hostgen_inv_dist_c32(void *user, int M, int N, const int32_t *rows, const int32_t *cols, float *out) {
    const hostgen_inv_dist *g = (const hostgen_inv_dist *)user;
    for (int k = 0; k < N; k++) {
        const double *y = g->source + (int64_t)g->dim * cols[k];
        for (int j = 0; j < M; j++) {
            const double *x   = g->target + (int64_t)g->dim * rows[j];
            const double den  = denominator(g, x, y);
            const double u    = x[0] - y[0];
            const double sgn  = g->hermitian ? (u > 0 ? 1.0 : (u < 0 ? -1.0 : 0.0)) : 1.0;
            float *o          = out + 2 * (j + (int64_t)M * k);
            o[0]              = (float)(g->cre / den);
            o[1]              = (float)((g->cim * sgn) / den);
        }
    }
}

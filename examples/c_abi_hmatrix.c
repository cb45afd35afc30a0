/* examples/c_abi_hmatrix.c -- the flow of the reference's examples/use_hmatrix.cpp (lines 12-109) written against the C ABI of include/hmx.h:
 * what a C (or cgo / JNI / ctypes) user of libhmx.so writes.  As there: a 2d ellipse in 3d, a binary cluster tree with leaves of at most 500
 * points (Partitioning_N / ComputeLargestExtent / RegularSplitting), epsilon = 0.01, eta = 200, symmetric storage of the lower triangle, a
 * generator written by the user (here `fill_block`: host code the library knows nothing about, called concurrently from its worker threads --
 * the role of the example's VirtualGenerator subclass; every cross it produces is compressed on the GPU), the product with the vector of ones in
 * user numbering and its relative error against the dense product.  Then the same operator from the built-in device kernel, whose product must
 * be the same bit for bit.  The factorisation at the end of the reference's example is outside this library's path.
 *
 *   gcc -O2 -ffp-contract=off -I include examples/c_abi_hmatrix.c -o examples/c_abi_hmatrix -L htool_amd -lhmx -Wl,-rpath,$PWD/htool_amd -lm
 *   ./examples/c_abi_hmatrix [number of points, default 10000] */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "hmx.h"

typedef struct {
    int dim;
    const double *target, *source; /* user numbering, dim doubles per point */
} point_kernel;

static double kernel_entry(const point_kernel *A, int k, int j) {
    double d2 = 0;
    for (int p = 0; p < A->dim; p++) {
        const double d = A->target[A->dim * k + p] - A->source[A->dim * j + p];
        d2 += d * d;
    }
    return 1. / (1e-5 + sqrt(d2));
}

/* what VirtualGenerator<double>::copy_submatrix does (hmatrix/interfaces/virtual_generator.hpp:24): M x N entries, column-major */
static void fill_block(void *user, int M, int N, const int32_t *rows, const int32_t *cols, double *out) {
    const point_kernel *K = (const point_kernel *)user;
    for (int k = 0; k < N; k++)
        for (int j = 0; j < M; j++)
            out[j + (size_t)M * k] = kernel_entry(K, rows[j], cols[k]);
}

static double now(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}

#define CHECK(call)                                                                     \
    do {                                                                                \
        const int rc_ = (call);                                                         \
        if (rc_ != HMX_OK) {                                                            \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, hmx_last_error());      \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 10000, dim = 3;
    double *pts = (double *)malloc(sizeof(double) * dim * n);
    CHECK(hmx_geometry("ellipse", n, 0., pts)); /* create_rotated_ellipse(3, 4., 1., 0., 0., n, ...): testing/geometry.hpp */

    /* cluster tree: leaves of at most 500 points, Partitioning_N<ComputeLargestExtent, RegularSplitting>, binary, one partition */
    hmx_cluster_tree *tree = NULL;
    CHECK(hmx_cluster_tree_create(n, dim, pts, NULL, NULL, 500, 2, 1, HMX_DIR_LARGEST_EXTENT, HMX_SPLIT_REGULAR, 1, &tree));

    /* block tree: eta = 200, symmetric storage of the lower triangle, htool's default minimal depths (0) and consistency (on) */
    const double epsilon = 0.01, eta = 200;
    hmx_block_tree *btree = NULL;
    CHECK(hmx_block_tree_create(tree, tree, eta, 'S', 'L', 0, 0, -1, -1, 1, &btree));

    /* the H-matrix: the user's generator, sympartialACA (HMatrixTreeBuilder's compressor for symmetric storage) */
    point_kernel A = {dim, pts, pts};
    hmx_hmatrix *op = NULL;
    CHECK(hmx_hmatrix_create(btree, 0, &op));
    CHECK(hmx_hmatrix_set_callback(op, fill_block, &A));
    double t = now();
    CHECK(hmx_hmatrix_compress(op, HMX_SYMPARTIAL_ACA, epsilon, -1));
    const double t_host = now() - t;

    hmx_stats st;
    CHECK(hmx_hmatrix_stats(op, &st));
    printf("Number of points: %d, host cores used by the generator: %d\n", n, hmx_host_cores());
    printf("Number of dense btree: %lld, low rank btree: %lld, rank min / mean / max: %d / %.2f / %d\n", (long long)st.n_dense, (long long)st.n_lowrank, st.rank_min,
           st.rank_mean, st.rank_max);
    printf("Compression ratio: %.2f, operator in HBM: %.1f MB, build %.3f s\n",
           (double)n * n / (double)(st.cgen_dense + st.cgen_lowrank), st.stream_bytes / 1e6, t_host);

    /* y = A x in the user's numbering (add_op_vector_product), host vectors */
    double *x = (double *)malloc(sizeof(double) * n), *y = (double *)calloc(n, sizeof(double)), *ref = (double *)calloc(n, sizeof(double));
    for (int i = 0; i < n; i++)
        x[i] = 1;
    CHECK(hmx_hmatrix_matvec_user(op, 'N', 1., x, 0., y, HMX_MEM_HOST, NULL));
    double err = 0, nrm = 0;
    for (int j = 0; j < n; j++) {
        for (int k = 0; k < n; k++)
            ref[j] += kernel_entry(&A, j, k) * x[k];
        err += (ref[j] - y[j]) * (ref[j] - y[j]);
        nrm += ref[j] * ref[j];
    }
    const double rel = sqrt(err / nrm);
    printf("relative error on matrix vector product : %.3e\n", rel);

    /* the same operator from the built-in device kernel 1 / (1e-5 + r): no callbacks at all */
    hmx_hmatrix *op_dev = NULL;
    const double params[2] = {1e-5, 1.0};
    CHECK(hmx_hmatrix_create(btree, 0, &op_dev));
    CHECK(hmx_hmatrix_set_kernel(op_dev, HMX_KERNEL_INV_DIST, params, 2, dim, pts, pts));
    t = now();
    CHECK(hmx_hmatrix_compress(op_dev, HMX_SYMPARTIAL_ACA, epsilon, -1));
    const double t_dev = now() - t;
    double *y2 = (double *)calloc(n, sizeof(double));
    CHECK(hmx_hmatrix_matvec_user(op_dev, 'N', 1., x, 0., y2, HMX_MEM_HOST, NULL));
    int same = 1;
    for (int i = 0; i < n; i++)
        same = same && y[i] == y2[i];
    printf("device-kernel build %.3f s, product %s the host-generator operator's\n", t_dev, same ? "bit-identical to" : "DIFFERS from");

    hmx_hmatrix_destroy(op_dev);
    hmx_hmatrix_destroy(op);
    hmx_block_tree_destroy(btree);
    hmx_cluster_tree_destroy(tree);
    free(pts), free(x), free(y), free(y2), free(ref);
    return rel < epsilon && same ? 0 : 2;
}

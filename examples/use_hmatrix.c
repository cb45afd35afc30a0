/* examples/use_hmatrix.c -- the reference's examples/use_hmatrix.cpp (lines 12-109: the user's own generator class, a cluster tree with
 * Partitioning_N / ComputeLargestExtent / RegularSplitting and leaves of 500 points, HMatrixTreeBuilder(epsilon = 0.01, eta = 200, 'S', 'L'),
 * the product with the vector of ones and its relative error against the dense product) written against the C ABI of include/hmx.h:
 * what a C (or cgo / JNI / ctypes) user of libhmx.so writes.  The generator below is the reference's UserOperator::copy_submatrix -- host
 * code the library knows nothing about, called concurrently from its worker threads; every cross it produces is compressed on the GPU, the
 * product runs on the GPU.  The Cholesky solve at the end of the reference's example is outside this library's path.
 *
 *   gcc -O2 -I include examples/use_hmatrix.c -o examples/use_hmatrix -L htool_amd -lhmx -Wl,-rpath,$PWD/htool_amd -lm
 *   ./examples/use_hmatrix [number of points, default 10000] */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "hmx.h"

typedef struct {
    int dim;
    const double *target, *source; /* user numbering, dim doubles per point */
} user_operator;

static double get_coef(const user_operator *A, int k, int j) {
    double d2 = 0;
    for (int p = 0; p < A->dim; p++) {
        const double d = A->target[A->dim * k + p] - A->source[A->dim * j + p];
        d2 += d * d;
    }
    return 1. / (1e-5 + sqrt(d2));
}

/* VirtualGenerator<double>::copy_submatrix (hmatrix/interfaces/virtual_generator.hpp:24): M x N entries, column-major */
static void copy_submatrix(void *user, int M, int N, const int32_t *rows, const int32_t *cols, double *ptr) {
    const user_operator *A = (const user_operator *)user;
    for (int k = 0; k < N; k++)
        for (int j = 0; j < M; j++)
            ptr[j + (size_t)M * k] = get_coef(A, rows[j], cols[k]);
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
    const int number_points = argc > 1 ? atoi(argv[1]) : 10000, spatial_dimension = 3;
    double *coordinates = (double *)malloc(sizeof(double) * spatial_dimension * number_points);
    CHECK(hmx_geometry("ellipse", number_points, 0., coordinates)); /* create_rotated_ellipse(3, 4., 1., 0., 0., n, ...): testing/geometry.hpp */

    /* cluster tree: leaves of at most 500 points, Partitioning_N<ComputeLargestExtent, RegularSplitting>, binary, one partition */
    hmx_cluster_tree *tree = NULL;
    CHECK(hmx_cluster_tree_create(number_points, spatial_dimension, coordinates, NULL, NULL, 500, 2, 1, HMX_DIR_LARGEST_EXTENT, HMX_SPLIT_REGULAR, 1, &tree));

    /* block tree: eta = 200, symmetric storage of the lower triangle, htool's default minimal depths (0) and consistency (on) */
    const double epsilon = 0.01, eta = 200;
    hmx_block_tree *blocks = NULL;
    CHECK(hmx_block_tree_create(tree, tree, eta, 'S', 'L', 0, 0, -1, -1, 1, &blocks));

    /* the H-matrix: the user's generator, sympartialACA (HMatrixTreeBuilder's compressor for symmetric storage) */
    user_operator A = {spatial_dimension, coordinates, coordinates};
    hmx_hmatrix *hmatrix = NULL;
    CHECK(hmx_hmatrix_create(blocks, 0, &hmatrix));
    CHECK(hmx_hmatrix_set_callback(hmatrix, copy_submatrix, &A));
    double t = now();
    CHECK(hmx_hmatrix_compress(hmatrix, HMX_SYMPARTIAL_ACA, epsilon, -1));
    const double build_s = now() - t;

    hmx_stats st;
    CHECK(hmx_hmatrix_stats(hmatrix, &st));
    printf("Number of points: %d, host cores used by the generator: %d\n", number_points, hmx_host_cores());
    printf("Number of dense blocks: %lld, low rank blocks: %lld, rank min / mean / max: %d / %.2f / %d\n", (long long)st.n_dense, (long long)st.n_lowrank, st.rank_min,
           st.rank_mean, st.rank_max);
    printf("Compression ratio: %.2f, operator in HBM: %.1f MB, build %.3f s\n",
           (double)number_points * number_points / (double)(st.cgen_dense + st.cgen_lowrank), st.stream_bytes / 1e6, build_s);

    /* y = A x in the user's numbering (add_hmatrix_vector_product), host vectors */
    double *x = (double *)malloc(sizeof(double) * number_points), *y = (double *)calloc(number_points, sizeof(double)), *ref = (double *)calloc(number_points, sizeof(double));
    for (int i = 0; i < number_points; i++)
        x[i] = 1;
    CHECK(hmx_hmatrix_matvec_user(hmatrix, 'N', 1., x, 0., y, HMX_MEM_HOST, NULL));
    double err = 0, nrm = 0;
    for (int j = 0; j < number_points; j++) {
        for (int k = 0; k < number_points; k++)
            ref[j] += get_coef(&A, j, k) * x[k];
        err += (ref[j] - y[j]) * (ref[j] - y[j]);
        nrm += ref[j] * ref[j];
    }
    const double rel = sqrt(err / nrm);
    printf("relative error on matrix vector product : %.3e\n", rel);

    /* the same operator from the built-in device kernel 1 / (1e-5 + r): no callbacks at all */
    hmx_hmatrix *on_device = NULL;
    const double params[2] = {1e-5, 1.0};
    CHECK(hmx_hmatrix_create(blocks, 0, &on_device));
    CHECK(hmx_hmatrix_set_kernel(on_device, HMX_KERNEL_INV_DIST, params, 2, spatial_dimension, coordinates, coordinates));
    t = now();
    CHECK(hmx_hmatrix_compress(on_device, HMX_SYMPARTIAL_ACA, epsilon, -1));
    const double build_dev_s = now() - t;
    double *y2 = (double *)calloc(number_points, sizeof(double));
    CHECK(hmx_hmatrix_matvec_user(on_device, 'N', 1., x, 0., y2, HMX_MEM_HOST, NULL));
    int same = 1;
    for (int i = 0; i < number_points; i++)
        same = same && y[i] == y2[i];
    printf("device-kernel build %.3f s, product %s the host-generator operator's\n", build_dev_s, same ? "bit-identical to" : "DIFFERS from");

    hmx_hmatrix_destroy(on_device);
    hmx_hmatrix_destroy(hmatrix);
    hmx_block_tree_destroy(blocks);
    hmx_cluster_tree_destroy(tree);
    free(coordinates), free(x), free(y), free(y2), free(ref);
    return rel < epsilon && same ? 0 : 2;
}

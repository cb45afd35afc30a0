// distributed_device_check.cpp -- TEST INFRASTRUCTURE.  examples/use_distributed_operator.cpp with the local operator on the
// GPU: htool's own MPI DistributedOperator (PartitionFromCluster, Allgatherv / Allreduce, numbering) is untouched, the rank's
// block rows are a hmx_htool::GlobalToLocalHmx registered through CustomApproximationBuilder
// (distributed_operator/utility.hpp:22-36).  Compared, on every rank, with htool's DefaultApproximationBuilder (CPU).
// Built in the dev container against the real htool headers + MPICH + libhmx.so (make -C oracle ref); run on the GPU box by
// tests/test_gpu_adaptor_end_to_end.py with `mpiexec -n {1,2}` (all ranks share device 0 there).
#include <htool/clustering/tree_builder/tree_builder.hpp>
#include <htool/distributed_operator/distributed_operator.hpp>
#include <htool/matrix/linalg/transpose.hpp> // used, but not included, by the matrix-product headers
#include <htool/distributed_operator/linalg/add_distributed_operator_matrix_product_global_to_global.hpp>
#include <htool/distributed_operator/linalg/add_distributed_operator_matrix_product_row_major_global_to_global.hpp>
#include <htool/distributed_operator/linalg/add_distributed_operator_matrix_product_row_major_local_to_local.hpp>
#include <htool/distributed_operator/linalg/add_distributed_operator_vector_product_global_to_global.hpp>
#include <htool/distributed_operator/linalg/add_distributed_operator_vector_product_local_to_local.hpp>
#include <htool/distributed_operator/utility.hpp>
#include <htool/hmatrix/lrmat/sympartialACA.hpp>
#include <htool/hmatrix/tree_builder/tree_builder.hpp>
#include <htool/testing/geometry.hpp>

#include "hmx/htool_adaptor.hpp"

#include <cmath>
#include <cstdio>
#include <vector>
using namespace htool;

struct UserOperator : public VirtualGenerator<double> { // the generator of examples/use_distributed_operator.cpp:17-43
    const std::vector<double> &x;
    explicit UserOperator(const std::vector<double> &x_) : x(x_) {}
    void copy_submatrix(int M, int N, const int *rows, const int *cols, double *ptr) const override {
        for (int j = 0; j < M; j++)
            for (int k = 0; k < N; k++) {
                double s = 0;
                for (int p = 0; p < 3; p++) {
                    const double d = x[3 * rows[j] + p] - x[3 * cols[k] + p];
                    s              = s + d * d;
                }
                ptr[j + (size_t)M * k] = 1. / (1 + std::sqrt(s));
            }
    }
};
static double rel(const std::vector<double> &a, const std::vector<double> &b) {
    double num = 0, den = 0;
    for (size_t i = 0; i < a.size(); i++) {
        num += (a[i] - b[i]) * (a[i] - b[i]);
        den += b[i] * b[i];
    }
    return std::sqrt(num / den);
}

int main(int argc, char **argv) {
    MPI_Init(&argc, &argv);
    int sizeWorld, rankWorld;
    MPI_Comm_rank(MPI_COMM_WORLD, &rankWorld);
    MPI_Comm_size(MPI_COMM_WORLD, &sizeWorld);
    const int n = 10000, children = 2;
    std::vector<double> x(3 * n);
    create_rotated_ellipse(3, 4., 1., 0., 0., n, x.data());
    ClusterTreeBuilder<double> ctb;
    ctb.set_maximal_leaf_size(100);
    Cluster<double> cluster = ctb.create_cluster_tree(n, 3, x.data(), children, sizeWorld);
    const double epsilon = 0.001, eta = 100;
    int failures = 0;
    for (char sym : {'N', 'S'}) {
        const char uplo = sym == 'N' ? 'N' : 'U';
        UserOperator A(x);
        // htool alone
        DefaultApproximationBuilder<double, double> reference(A, cluster, cluster, HMatrixTreeBuilder<double, double>(epsilon, eta, sym, uplo), MPI_COMM_WORLD);
        // the same DistributedOperator with this rank's block rows on the GPU
        hmx_htool::ClusterOptions opt;
        opt.maximal_leaf_size = 100, opt.number_of_children = children, opt.size_of_partition = sizeWorld;
        hmx_htool::Engine engine(cluster, n, x.data(), cluster, n, x.data(), 3, opt);
        const double p[2] = {1.0, 1.0};
        if (!engine.setup_block_tree(eta, sym, uplo, 0, 0, rankWorld, rankWorld, 0) || !engine.compress_on_device(HMX_KERNEL_INV_DIST, p, 2, 3, x.data(), x.data(), HMX_SYMPARTIAL_ACA, epsilon, -1)) {
            MPI_Abort(MPI_COMM_WORLD, 3);
        }
        hmx_htool::GlobalToLocalHmx local_op(engine, n);
        CustomApproximationBuilder<double> device(cluster, cluster, MPI_COMM_WORLD, local_op);

        std::vector<double> in(n), y0(n);
        for (int i = 0; i < n; i++) {
            in[i] = std::sin(0.37 * i) + 0.1;
            y0[i] = std::cos(0.11 * i);
        }
        double *work = nullptr;
        for (char trans : {'N', 'T'}) {
            std::vector<double> yref = y0, y = y0;
            add_distributed_operator_vector_product_global_to_global(trans, 1.5, reference.distributed_operator, in.data(), 0.5, yref.data(), work);
            add_distributed_operator_vector_product_global_to_global(trans, 1.5, device.distributed_operator, in.data(), 0.5, y.data(), work);
            const double e = rel(y, yref);
            if (rankWorld == 0)
                std::printf("np=%d sym=%c trans=%c global_to_global, GPU local operator vs htool: %.3e %s\n", sizeWorld, sym, trans, e, e < 1e-10 ? "ok" : "FAIL");
            failures += !(e < 1e-10);
            // local to local (the Krylov-side contract)
            const int off = cluster.get_cluster_on_partition(rankWorld).get_offset(), sz = cluster.get_cluster_on_partition(rankWorld).get_size();
            std::vector<double> inl(in.begin() + off, in.begin() + off + sz), ylr(sz, 0.), yl(sz, 0.);
            internal_add_distributed_operator_vector_product_local_to_local(trans, 1., reference.distributed_operator, inl.data(), 0., ylr.data(), work);
            internal_add_distributed_operator_vector_product_local_to_local(trans, 1., device.distributed_operator, inl.data(), 0., yl.data(), work);
            double el = rel(yl, ylr), emax = 0;
            MPI_Allreduce(&el, &emax, 1, MPI_DOUBLE, MPI_MAX, MPI_COMM_WORLD);
            if (rankWorld == 0)
                std::printf("np=%d sym=%c trans=%c local_to_local,   GPU local operator vs htool: %.3e %s\n", sizeWorld, sym, trans, emax, emax < 1e-10 ? "ok" : "FAIL");
            failures += !(emax < 1e-10);
            // mu = 5: the Krylov-side block product (HPDDMOperator::GMV calls exactly this for mu != 1, wrappers/wrapper_hpddm.hpp:126),
            // the row-major global-to-global product and the column-major user-numbering front end
            const int mu = 5;
            Matrix<double> Xl(mu, sz), Ylr(mu, sz), Yl(mu, sz), Xg(mu, n), Ygr(mu, n), Yg(mu, n), Xc(n, mu), Ycr(n, mu), Yc(n, mu);
            for (int i = 0; i < n; i++)
                for (int c = 0; c < mu; c++) {
                    Xg(c, i) = Xc(i, c) = std::sin(0.37 * i + 0.61 * c) + 0.1;
                    Ygr(c, i) = Yg(c, i) = Ycr(i, c) = Yc(i, c) = std::cos(0.11 * i - 0.3 * c);
                    if (i >= off && i < off + sz) {
                        Xl(c, i - off)  = Xg(c, i);
                        Ylr(c, i - off) = Yl(c, i - off) = Ygr(c, i);
                    }
                }
            internal_add_distributed_operator_matrix_product_row_major_local_to_local(trans, 1.5, reference.distributed_operator, Xl, 0.5, Ylr, work);
            internal_add_distributed_operator_matrix_product_row_major_local_to_local(trans, 1.5, device.distributed_operator, Xl, 0.5, Yl, work);
            internal_add_distributed_operator_matrix_product_row_major_global_to_global(trans, 1.5, reference.distributed_operator, Xg, 0.5, Ygr, work);
            internal_add_distributed_operator_matrix_product_row_major_global_to_global(trans, 1.5, device.distributed_operator, Xg, 0.5, Yg, work);
            add_distributed_operator_matrix_product_global_to_global(trans, 1.5, reference.distributed_operator, Xc, 0.5, Ycr, work);
            add_distributed_operator_matrix_product_global_to_global(trans, 1.5, device.distributed_operator, Xc, 0.5, Yc, work);
            auto relm = [](const Matrix<double> &a, const Matrix<double> &b) {
                return rel(std::vector<double>(a.data(), a.data() + (size_t)a.nb_rows() * a.nb_cols()), std::vector<double>(b.data(), b.data() + (size_t)b.nb_rows() * b.nb_cols()));
            };
            double em[3] = {relm(Yl, Ylr), relm(Yg, Ygr), relm(Yc, Ycr)}, emx[3];
            MPI_Allreduce(em, emx, 3, MPI_DOUBLE, MPI_MAX, MPI_COMM_WORLD);
            const char *names[3] = {"row-major local_to_local mu=5", "row-major global_to_global mu=5", "column-major global_to_global mu=5 (user numbering)"};
            for (int k = 0; k < 3; k++) {
                if (rankWorld == 0)
                    std::printf("np=%d sym=%c trans=%c %s, GPU local operator vs htool: %.3e %s\n", sizeWorld, sym, trans, names[k], emx[k], emx[k] < 1e-10 ? "ok" : "FAIL");
                failures += !(emx[k] < 1e-10);
            }
        }
    }
    if (rankWorld == 0)
        std::printf(failures ? "distributed device check: %d FAILED\n" : "distributed device check: all ok\n", failures);
    MPI_Finalize();
    return failures ? 1 : 0;
}

// adaptor_device_check.cpp -- TEST INFRASTRUCTURE.  The drop-in claim, end to end, against the REAL reference:
// compiled in the dev container against the htool headers where they lie (+ the image's MKL) and libhmx.so through
// htool_amd/include/hmx/htool_adaptor.hpp; the binary travels to the GPU box (like oracle/_ref/ref_driver) and is run
// there by tests/test_gpu_adaptor_end_to_end.py.  Nothing from /root/reference is read at run time.
//
//   (0) htool alone (CPU): cluster tree, HMatrixTreeBuilder + partialACA, add_hmatrix_vector_product  -> the reference result
//   (a) device compression with the built-in kernel, product on the GPU (hmx_hmatrix_matvec_user)
//   (b) htool's OWN builder fed by the plugin classes DeviceLowRankGenerator / DeviceDenseBlocksGenerator: an htool
//       HMatrix holding GPU-compressed blocks, multiplied by htool's own CPU leaf loop
//   (c) device compression driven by the user's VirtualGenerator (host callback), product on the GPU
//   (d) the htool-built H-matrix uploaded leaf by leaf (Engine::upload), product on the GPU
//   (e) GlobalToLocalHmx (htool's VirtualGlobalToLocalOperator) on the row slab of partition 1, vs htool's restricted build
//   (f) the same as (a),(d) for std::complex<double> with Hermitian storage
#include <htool/clustering/tree_builder/tree_builder.hpp>
#include <htool/hmatrix/hmatrix.hpp>
#include <htool/distributed_operator/implementations/local_to_local_operators/hmatrix.hpp>
#include <htool/hmatrix/linalg/add_hmatrix_vector_product.hpp>
#include <htool/hmatrix/lrmat/partialACA.hpp>
#include <htool/hmatrix/lrmat/sympartialACA.hpp>
#include <htool/hmatrix/tree_builder/tree_builder.hpp>
#include <htool/testing/geometry.hpp>

#include "hmx/htool_adaptor.hpp"
extern "C" { // the HIP runtime entry points this check needs (libamdhip64, which libhmx.so links)
int hipMalloc(void **, size_t);
int hipFree(void *);
int hipMemcpy(void *, const void *, size_t, int /* 1 = host to device, 2 = device to host */);
int hipDeviceSynchronize(void);
}

#include <cmath>
#include <complex>
#include <cstdio>
#include <vector>
using namespace htool;

template <typename T>
struct Kernel : public VirtualGenerator<T> {
    const std::vector<double> &x;
    bool herm;
    Kernel(const std::vector<double> &x_, bool herm_ = false) : x(x_), herm(herm_) {}
    void copy_submatrix(int M, int N, const int *rows, const int *cols, T *ptr) const override {
        for (int j = 0; j < M; j++)
            for (int k = 0; k < N; k++) {
                double s = 0;
                for (int p = 0; p < 3; p++) {
                    const double d = x[3 * rows[j] + p] - x[3 * cols[k] + p];
                    s              = s + d * d;
                }
                const double den = 1e-5 + std::sqrt(s);
                if constexpr (std::is_same<T, double>::value) {
                    ptr[j + (size_t)M * k] = 1. / den;
                } else {
                    const double u   = x[3 * rows[j]] - x[3 * cols[k]];
                    const double sgn = herm ? (u > 0 ? 1. : (u < 0 ? -1. : 0.)) : 1.;
                    ptr[j + (size_t)M * k] = T(std::complex<double>(1., 0.5 * sgn) / den);
                }
            }
    }
};
template <typename T>
static double rel(const std::vector<T> &a, const std::vector<T> &b) {
    double num = 0, den = 0;
    for (size_t i = 0; i < a.size(); i++) {
        num += std::norm(a[i] - b[i]);
        den += std::norm(b[i]);
    }
    return std::sqrt(num / den);
}
static int failures = 0;
static void report(const char *what, double err, double tol) {
    std::printf("%-78s %.3e  (< %.0e) %s\n", what, err, tol, err < tol ? "ok" : "FAIL");
    if (!(err < tol))
        failures++;
}

// `adaptor_device_check --time N`: how long the drop-in routes take at size N next to htool's own OpenMP build on the same box
// (same generator class, same cluster tree, eps = 1e-4, eta = 10, leaf 100, minimal block depth as bench.py's).  Routes: (a) device kernel,
// (b) htool's builder fed by the device generators (bulk download of every block), (c) the user's VirtualGenerator through the host
// callback on all cores.  Ranks of (a) and (c) must agree leaf by leaf; (b) must reproduce htool's product.
#include <chrono>
#include <omp.h>
static double seconds_since(const std::chrono::steady_clock::time_point &t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
static int timing(int n) {
    // htool's OpenMP loops on as many threads as the process really has cores (a container quota of 16 cores on a 256-thread box: 256
    // OpenMP threads are throttled to a crawl; hmx_host_cores reads the quota) -- what a user would set OMP_NUM_THREADS to
    if (!std::getenv("OMP_NUM_THREADS"))
        omp_set_num_threads(hmx_host_cores());
    std::printf("host cores %d (OpenMP threads %d), N = %d\n", hmx_host_cores(), omp_get_max_threads(), n);
    std::vector<double> x(3 * (size_t)n);
    create_rotated_ellipse(3, 4., 1., 0., 0., n, x.data());
    auto t0 = std::chrono::steady_clock::now();
    ClusterTreeBuilder<double> ctb;
    ctb.set_maximal_leaf_size(100);
    Cluster<double> T = ctb.create_cluster_tree(n, 3, x.data(), 2, 2);
    std::printf("htool cluster tree (host, htool's own builder)                         %8.3f s\n", seconds_since(t0));
    const double eps = 1e-4, eta = 10.;
    int depth = 0; // bench.py's minimal_depth: blocks of at most 31 250 rows
    for (long long m = n; m > 31250; m /= 2)
        depth++;
    Kernel<double> A(x);
    std::vector<double> in(n), yref(n, 0.), y(n, 0.);
    for (int i = 0; i < n; i++)
        in[i] = std::sin(0.37 * i) + 0.1;
    t0 = std::chrono::steady_clock::now();
    HMatrixTreeBuilder<double> tb(eps, eta, 'N', 'N');
    tb.set_minimal_target_depth(depth);
    tb.set_minimal_source_depth(depth);
    tb.set_low_rank_generator(std::make_shared<partialACA<double>>(A, T.get_permutation().data(), T.get_permutation().data()));
    HMatrix<double> Href = tb.openmp_build(A, T, T);
    const double t_ref   = seconds_since(t0);
    std::printf("(0) htool's own build (OpenMP, partialACA on the host)                  %8.3f s\n", t_ref);
    add_hmatrix_vector_product('N', 1., Href, in.data(), 0., yref.data());

    t0 = std::chrono::steady_clock::now();
    hmx_htool::Engine E(T, T, 3);
    if (!E.setup_block_tree(eta, 'N', 'N', depth, depth, -1, -1, 0))
        return 3;
    const double p[2] = {1e-5, 1.0};
    if (!E.compress_on_device(HMX_KERNEL_INV_DIST, p, 2, 3, x.data(), x.data(), HMX_PARTIAL_ACA, eps, -1))
        return 3;
    hipDeviceSynchronize();
    std::printf("(a) tree import + block tree + device compression, built-in kernel      %8.3f s\n", seconds_since(t0));

    t0 = std::chrono::steady_clock::now();
    HMatrixTreeBuilder<double> tb2(eps, eta, 'N', 'N');
    tb2.set_minimal_target_depth(depth);
    tb2.set_minimal_source_depth(depth);
    auto lrgen = std::make_shared<hmx_htool::DeviceLowRankGenerator>(E);
    tb2.set_low_rank_generator(lrgen);
    tb2.set_dense_blocks_generator(std::make_shared<hmx_htool::DeviceDenseBlocksGenerator>(E));
    HMatrix<double> Hdev = tb2.openmp_build(A, T, T);
    std::printf("(b) htool's builder fed by the device generators                        %8.3f s  (of which the one bulk download of the low-rank blocks: %.3f s)\n", seconds_since(t0), lrgen->prefetch_seconds());
    add_hmatrix_vector_product('N', 1., Hdev, in.data(), 0., y.data());
    report("(b) htool CPU product on the GPU-compressed blocks vs htool's own operator", rel(y, yref), 1e-10);

    t0 = std::chrono::steady_clock::now();
    hmx_htool::Engine Ec(T, T, 3);
    if (!Ec.setup_block_tree(eta, 'N', 'N', depth, depth, -1, -1, 0) || !Ec.compress_with_generator(A, HMX_PARTIAL_ACA, eps, -1))
        return 3;
    hipDeviceSynchronize();
    const double t_c = seconds_since(t0);
    std::printf("(c) tree import + block tree + device ACA on the user's VirtualGenerator %8.3f s  (%d host cores; htool's own build / this = %.2f)\n", t_c, hmx_host_cores(), t_ref / t_c);
    std::vector<int32_t> ra(E.number_of_leaves()), rc(Ec.number_of_leaves());
    hmx_hmatrix_leaf_ranks(E.hmatrix(), ra.data());
    hmx_hmatrix_leaf_ranks(Ec.hmatrix(), rc.data());
    report("(c) ranks of the host-generator build = ranks of the device-kernel build, leaf by leaf", ra == rc ? 0. : 1., 1e-300);
    std::fill(y.begin(), y.end(), 0.);
    hmx_hmatrix_matvec_user(Ec.hmatrix(), 'N', 1., in.data(), 0., y.data(), HMX_MEM_HOST, nullptr);
    report("(c) device product of the host-generator build vs htool on the CPU", rel(y, yref), 1e-10);
    std::printf(failures ? "adaptor timing: %d FAILED\n" : "adaptor timing: all ok\n", failures);
    return failures ? 1 : 0;
}

int main(int argc, char **argv) {
    if (argc >= 3 && std::string(argv[1]) == "--time")
        return timing(std::atoi(argv[2]));
    const int n = 4000;
    std::vector<double> x(3 * n);
    create_rotated_ellipse(3, 4., 1., 0., 0., n, x.data());
    ClusterTreeBuilder<double> ctb;
    ctb.set_maximal_leaf_size(100);
    Cluster<double> T = ctb.create_cluster_tree(n, 3, x.data(), 2, 2);
    hmx_htool::ClusterOptions opt;
    opt.maximal_leaf_size = 100, opt.number_of_children = 2, opt.size_of_partition = 2;
    const double eps = 1e-4, eta = 10.;
    std::vector<double> in(n), y0(n);
    for (int i = 0; i < n; i++) {
        in[i] = std::sin(0.37 * i) + 0.1;
        y0[i] = std::cos(0.11 * i);
    }
    {
        Kernel<double> A(x);
        // (0) the reference
        HMatrixTreeBuilder<double> tb(eps, eta, 'N', 'N');
        tb.set_low_rank_generator(std::make_shared<partialACA<double>>(A, T.get_permutation().data(), T.get_permutation().data()));
        HMatrix<double> Href = tb.sequential_build(A, T, T);
        std::vector<double> yref = y0;
        add_hmatrix_vector_product('N', 1.5, Href, in.data(), 0.5, yref.data());
        std::vector<double> yrefT = y0;
        add_hmatrix_vector_product('T', 1.5, Href, in.data(), 0.5, yrefT.data());

        // (a) device kernel + device product
        hmx_htool::Engine E(T, n, x.data(), T, n, x.data(), 3, opt);
        if (!E.setup_block_tree(eta, 'N', 'N', 0, 0, -1, -1, 0))
            return 3;
        const double p[2] = {1e-5, 1.0};
        if (!E.compress_on_device(HMX_KERNEL_INV_DIST, p, 2, 3, x.data(), x.data(), HMX_PARTIAL_ACA, eps, -1))
            return 3;
        std::vector<double> y = y0;
        hmx_hmatrix_matvec_user(E.hmatrix(), 'N', 1.5, in.data(), 0.5, y.data(), HMX_MEM_HOST, nullptr);
        report("(a) device compression + device product vs htool on the CPU, trans N", rel(y, yref), 1e-10);
        y = y0;
        hmx_hmatrix_matvec_user(E.hmatrix(), 'T', 1.5, in.data(), 0.5, y.data(), HMX_MEM_HOST, nullptr);
        report("(a) ... trans T", rel(y, yrefT), 1e-10);

        // (b) htool's builder with the plugin classes, htool's CPU product
        HMatrixTreeBuilder<double> tb2(eps, eta, 'N', 'N');
        tb2.set_low_rank_generator(std::make_shared<hmx_htool::DeviceLowRankGenerator>(E));
        tb2.set_dense_blocks_generator(std::make_shared<hmx_htool::DeviceDenseBlocksGenerator>(E));
        HMatrix<double> Hdev = tb2.sequential_build(A, T, T);
        y                    = y0;
        add_hmatrix_vector_product('N', 1.5, Hdev, in.data(), 0.5, y.data());
        report("(b) htool builder fed by DeviceLowRankGenerator/DeviceDenseBlocksGenerator, htool CPU product", rel(y, yref), 1e-10);

        // (b') the user-numbering flavour of the plug-in (VirtualLowRankGenerator: rows / cols point into the permutations)
        HMatrixTreeBuilder<double> tb2u(eps, eta, 'N', 'N');
        tb2u.set_low_rank_generator(std::make_shared<hmx_htool::DeviceUserLowRankGenerator>(E, T.get_permutation().data(), T.get_permutation().data()));
        HMatrix<double> Hdevu = tb2u.sequential_build(A, T, T);
        y                     = y0;
        add_hmatrix_vector_product('N', 1.5, Hdevu, in.data(), 0.5, y.data());
        report("(b') htool builder fed by DeviceUserLowRankGenerator (dense leaves by htool), htool CPU product", rel(y, yref), 1e-10);

        // (c) user generator through the host callback
        hmx_htool::Engine Ec(T, n, x.data(), T, n, x.data(), 3, opt);
        if (!Ec.setup_block_tree(eta, 'N', 'N', 0, 0, -1, -1, 0) || !Ec.compress_with_generator(A, HMX_PARTIAL_ACA, eps, -1))
            return 3;
        y = y0;
        hmx_hmatrix_matvec_user(Ec.hmatrix(), 'N', 1.5, in.data(), 0.5, y.data(), HMX_MEM_HOST, nullptr);
        report("(c) device ACA driven by the user's VirtualGenerator (host callback), device product", rel(y, yref), 1e-10);

        // (d) upload the htool-built H-matrix
        hmx_htool::Engine Eu(T, n, x.data(), T, n, x.data(), 3, opt);
        if (!Eu.setup_block_tree(eta, 'N', 'N', 0, 0, -1, -1, 0) || !Eu.upload(Href))
            return 3;
        y = y0;
        hmx_hmatrix_matvec_user(Eu.hmatrix(), 'N', 1.5, in.data(), 0.5, y.data(), HMX_MEM_HOST, nullptr);
        report("(d) htool-compressed leaves uploaded, device product (same blocks: rounding only)", rel(y, yref), 1e-13);

        // (e) the distributed local operator: row slab of partition 1
        HMatrixTreeBuilder<double> tb3(eps, eta, 'N', 'N');
        tb3.set_low_rank_generator(std::make_shared<partialACA<double>>(A, T.get_permutation().data(), T.get_permutation().data()));
        HMatrix<double> Hloc = tb3.sequential_build(A, T, T, 1, 1);
        const int nloc       = Hloc.get_target_cluster().get_size();
        std::vector<double> xin_cluster(n), yl(nloc, 0.), ylref(nloc, 0.);
        for (int i = 0; i < n; i++)
            xin_cluster[i] = in[T.get_permutation()[i]];
        sequential_internal_add_hmatrix_vector_product('N', 1., Hloc, xin_cluster.data(), 0., ylref.data());
        hmx_htool::Engine El(T, n, x.data(), T, n, x.data(), 3, opt);
        if (!El.setup_block_tree(eta, 'N', 'N', 0, 0, 1, 1, 0) || !El.compress_on_device(HMX_KERNEL_INV_DIST, p, 2, 3, x.data(), x.data(), HMX_PARTIAL_ACA, eps, -1))
            return 3;
        hmx_htool::GlobalToLocalHmx op(El, n);
        op.add_vector_product('N', 1., xin_cluster.data(), 0., yl.data());
        report("(e) GlobalToLocalHmx (row slab of partition 1) vs htool's restricted H-matrix", rel(yl, ylref), 1e-10);
        { // the same product on DEVICE pointers (add_vector_product_device / add_matrix_product_row_major_device): no host staging
            void *d_in = nullptr, *d_out = nullptr;
            if (hipMalloc(&d_in, n * sizeof(double)) != 0 || hipMalloc(&d_out, nloc * sizeof(double)) != 0)
                return 4;
            std::vector<double> yd(nloc, 0.);
            hipMemcpy(d_in, xin_cluster.data(), n * sizeof(double), 1);
            hipMemcpy(d_out, yd.data(), nloc * sizeof(double), 1);
            const bool okd = op.add_vector_product_device('N', 1., static_cast<const double *>(d_in), 0., static_cast<double *>(d_out));
            hipDeviceSynchronize();
            hipMemcpy(yd.data(), d_out, nloc * sizeof(double), 2);
            report("(e) ... on device pointers: identical to the host-pointer product", okd && yd == yl ? 0. : 1., 1e-300);
            const int mu = 3;
            std::vector<double> Xm((size_t)n * mu), Ym((size_t)nloc * mu, 0.), Yd((size_t)nloc * mu, 0.);
            for (size_t i = 0; i < Xm.size(); i++)
                Xm[i] = std::sin(0.013 * (double)i) + 0.2;
            op.add_matrix_product_row_major('N', 1., Xm.data(), 0., Ym.data(), mu);
            void *d_X = nullptr, *d_Y = nullptr;
            if (hipMalloc(&d_X, Xm.size() * sizeof(double)) != 0 || hipMalloc(&d_Y, Yd.size() * sizeof(double)) != 0)
                return 4;
            hipMemcpy(d_X, Xm.data(), Xm.size() * sizeof(double), 1);
            const bool okm = op.add_matrix_product_row_major_device('N', 1., static_cast<const double *>(d_X), 0., static_cast<double *>(d_Y), mu);
            hipDeviceSynchronize();
            hipMemcpy(Yd.data(), d_Y, Yd.size() * sizeof(double), 2);
            report("(e) ... row-major multi-RHS on device pointers: identical to the host-pointer product", okm && Yd == Ym ? 0. : 1., 1e-300);
            hipFree(d_in), hipFree(d_out), hipFree(d_X), hipFree(d_Y);
        }

        // (e') the block-diagonal operator of partition 1 (DefaultLocalApproximationBuilder) as a VirtualLocalToLocalOperator
        const Cluster<double> &part = T.get_cluster_on_partition(1);
        HMatrixTreeBuilder<double> tb4(eps, eta, 'N', 'N');
        tb4.set_low_rank_generator(std::make_shared<partialACA<double>>(A, T.get_permutation().data(), T.get_permutation().data()));
        HMatrix<double> Hdiag = tb4.sequential_build(A, part, part);
        const int nd          = part.get_size();
        std::vector<double> xd(xin_cluster.begin() + part.get_offset(), xin_cluster.begin() + part.get_offset() + nd), yd(nd, 0.), ydref(nd, 0.);
        sequential_internal_add_hmatrix_vector_product('N', 1., Hdiag, xd.data(), 0., ydref.data());
        hmx_htool::Engine Ed(T, n, x.data(), T, n, x.data(), 3, opt);
        if (!Ed.setup_local_block_tree(eta, 'N', 'N', 0, 0, 1, 1, 0) || !Ed.compress_on_device(HMX_KERNEL_INV_DIST, p, 2, 3, x.data(), x.data(), HMX_PARTIAL_ACA, eps, -1))
            return 3;
        hmx_htool::LocalToLocalHmx opd(Ed, nd);
        opd.add_vector_product('N', 1., xd.data(), 0., yd.data());
        report("(e') LocalToLocalHmx (block-diagonal operator of partition 1) vs htool's build on the partition clusters", rel(yd, ydref), 1e-10);
        // (e'') add_sub_matrix_product_to_local on partition 1 (source offset > 0): a window of the GLOBAL source numbering that
        // overlaps the local cluster only partly, the whole cluster, and a window outside it -- against htool's own LocalToLocalHMatrix
        {
            LocalToLocalHMatrix<double> href(Hdiag);
            const int so = part.get_offset();
            const int windows[3][2] = {{so - 37, nd / 2 + 37}, {so, nd}, {0, std::max(1, so - 1)}};
            for (const auto &w : windows) {
                const int off = std::max(0, w[0]), size = w[1];
                std::vector<double> sub(xin_cluster.begin() + off, xin_cluster.begin() + off + size), a(nd, 0.25), b(nd, 0.25);
                href.add_sub_matrix_product_to_local(sub.data(), a.data(), 1, off, size);
                opd.add_sub_matrix_product_to_local(sub.data(), b.data(), 1, off, size);
                report("(e'') LocalToLocalHmx::add_sub_matrix_product_to_local, window of the global source numbering", rel(b, a), 1e-10);
            }
        }
    }
    {
        using Z = std::complex<double>;
        Kernel<Z> A(x, true);
        HMatrixTreeBuilder<Z> tb(eps, eta, 'H', 'L');
        tb.set_low_rank_generator(std::make_shared<sympartialACA<Z>>(A, T.get_permutation().data(), T.get_permutation().data()));
        HMatrix<Z> Href = tb.sequential_build(A, T, T);
        std::vector<Z> zin(n), zy0(n);
        for (int i = 0; i < n; i++) {
            zin[i] = Z(std::sin(0.37 * i), std::cos(0.21 * i));
            zy0[i] = Z(std::cos(0.11 * i), 0.3);
        }
        std::vector<Z> yref = zy0;
        add_hmatrix_vector_product('N', Z(1.5, -0.5), Href, zin.data(), Z(0.5, 0.25), yref.data());
        hmx_htool::EngineT<Z> E(T, n, x.data(), T, n, x.data(), 3, opt);
        if (!E.setup_block_tree(eta, 'H', 'L', 0, 0, -1, -1, 0))
            return 3;
        const double p[5] = {1e-5, 1.0, 1.0, 0.5, 1.0};
        if (!E.compress_on_device(HMX_KERNEL_INV_DIST, p, 5, 3, x.data(), x.data(), HMX_SYMPARTIAL_ACA, eps, -1))
            return 3;
        std::vector<Z> y = zy0;
        const Z al(1.5, -0.5), be(0.5, 0.25);
        hmx_hmatrix_matvec_user_z(E.hmatrix(), 'N', reinterpret_cast<const double *>(&al), reinterpret_cast<const double *>(zin.data()), reinterpret_cast<const double *>(&be), reinterpret_cast<double *>(y.data()), HMX_MEM_HOST, nullptr);
        report("(f) complex Hermitian: device compression + device product vs htool on the CPU", rel(y, yref), 1e-10);
        hmx_htool::EngineT<Z> Eu(T, n, x.data(), T, n, x.data(), 3, opt);
        if (!Eu.setup_block_tree(eta, 'H', 'L', 0, 0, -1, -1, 0) || !Eu.upload(Href))
            return 3;
        y = zy0;
        hmx_hmatrix_matvec_user_z(Eu.hmatrix(), 'N', reinterpret_cast<const double *>(&al), reinterpret_cast<const double *>(zin.data()), reinterpret_cast<const double *>(&be), reinterpret_cast<double *>(y.data()), HMX_MEM_HOST, nullptr);
        report("(f) complex Hermitian: htool-compressed leaves uploaded, device product", rel(y, yref), 1e-13);
    }
    std::printf(failures ? "adaptor device check: %d FAILED\n" : "adaptor device check: all ok\n", failures);
    return failures ? 1 : 0;
}

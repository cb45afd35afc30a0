// TEST INFRASTRUCTURE (dev container only: needs the htool headers under /root/reference/include and the image's MPICH).
// Runs the REAL reference's DistributedOperator under MPI -- DefaultApproximationBuilder (row-restricted H-matrix per rank,
// distributed_operator/utility.hpp:38-61) or DefaultLocalApproximationBuilder (block-diagonal, :64-88) -- through EVERY product
// family of distributed_operator/linalg/ on closed-form inputs (oracle/oracle.py hashed_vector) and dumps what each rank gets:
//   vector   g2g / l2l, user ("add_") and partition ("internal_") numbering                      ..._vector_product_*.hpp
//   row-major multi-RHS g2g / l2l (what HPDDMOperator::GMV calls for mu != 1, wrappers/wrapper_hpddm.hpp:126)
//   column-major multi-RHS g2g / l2l, user and partition numbering                                ..._matrix_product_*.hpp
//   sub product global-to-local                                                                  ..._vector_sub_product_global_to_local.hpp
// tests/golden/make_golden.py stores the dumps of all ranks as ONE fixture; nothing here is shipped or run on the GPU box.
//   mpiexec -n P oracle/_ref/dist_products n=3000 geom=ellipse leaf=60 eps=1e-6 eta=10 sym=N uplo=N mu=5 given=local out=/tmp/dp
//   -> /tmp/dp.rank<k> (record format of ref_driver.cpp)
#include <mpi.h>
#define main ref_driver_main
#include "ref_driver.cpp" // geometry, generator, option parsing and the dump format of the fixture driver
#undef main
#include <htool/matrix/linalg/transpose.hpp> // the matrix-product headers below use transpose() without including it
#include <htool/distributed_operator/distributed_operator.hpp>
#include <htool/distributed_operator/linalg/add_distributed_operator_matrix_product_global_to_global.hpp>
#include <htool/distributed_operator/linalg/add_distributed_operator_matrix_product_local_to_local.hpp>
#include <htool/distributed_operator/linalg/add_distributed_operator_matrix_product_row_major_global_to_global.hpp>
#include <htool/distributed_operator/linalg/add_distributed_operator_matrix_product_row_major_local_to_local.hpp>
#include <htool/distributed_operator/linalg/add_distributed_operator_vector_product_global_to_global.hpp>
#include <htool/distributed_operator/linalg/add_distributed_operator_vector_product_local_to_local.hpp>
#include <htool/distributed_operator/linalg/add_distributed_operator_vector_sub_product_global_to_local.hpp>
#include <htool/distributed_operator/utility.hpp>

template <typename T>
static std::vector<T> hashed_values(size_t n, unsigned salt) { // = ref_driver's `hashed` lambda (complex: imaginary part with salt + 16)
    std::vector<T> v(n);
    for (size_t i = 0; i < n; i++) {
        const double re = (double)(uint32_t)((uint32_t)(i + 1) * 2654435761u + salt * 40503u) / 4294967296.0;
        const double im = (double)(uint32_t)((uint32_t)(i + 1) * 2654435761u + (salt + 16) * 40503u) / 4294967296.0;
        v[i]            = make_value<T>(re, im);
    }
    return v;
}

template <typename T>
static int run(std::map<std::string, std::string> &kv) {
    int rank, world;
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &world);
    const int n            = geti(kv, "n", 3000);
    const int mu           = geti(kv, "mu", 5);
    const std::string geom = gets(kv, "geom", "ellipse"), sym = gets(kv, "sym", "N"), uplo = gets(kv, "uplo", "N"), comp = gets(kv, "compressor", "default");
    const std::string given = gets(kv, "given", "none");
    const int block_diagonal = geti(kv, "local", 0);
    const int dim          = geometry_dim(geom);
    const T alpha = make_value<T>(getd(kv, "alpha", 3.), is_cplx<T>::value ? 0.5 : 0.), beta = make_value<T>(getd(kv, "beta", 2.), is_cplx<T>::value ? -0.25 : 0.);
    std::vector<double> x;
    make_geometry(geom, n, 0., x);
    ClusterTreeBuilder<double> ctb;
    ctb.set_maximal_leaf_size(geti(kv, "leaf", 60));
    std::vector<int> part(2 * world);
    for (int p = 0; p < world; p++) {
        const int lo = (int)((long long)n * p / world), hi = (int)((long long)n * (p + 1) / world);
        part[2 * p] = lo, part[2 * p + 1] = hi - lo;
    }
    Cluster<double> Tc = given == "local" ? ctb.create_cluster_tree_from_local_partition(n, dim, x.data(), geti(kv, "children", 2), world, part.data())
                                          : ctb.create_cluster_tree(n, dim, x.data(), geti(kv, "children", 2), world);
    const bool local_numbering = Tc.is_permutation_local();
    InvDistGenerator<T> A(dim, x, x, getd(kv, "delta", 1e-5), getd(kv, "scale", 1.), getd(kv, "cre", 1.), getd(kv, "cim", is_cplx<T>::value ? 0.5 : 0.), sym == "H");
    HMatrixTreeBuilder<T, double> tb(getd(kv, "eps", 1e-6), getd(kv, "eta", 10), sym[0], uplo[0], geti(kv, "reqrank", -1));
    if (comp == "partialACA")
        tb.set_low_rank_generator(std::make_shared<partialACA<T>>(A, Tc.get_permutation().data(), Tc.get_permutation().data()));
    else if (comp == "sympartialACA")
        tb.set_low_rank_generator(std::make_shared<sympartialACA<T>>(A, Tc.get_permutation().data(), Tc.get_permutation().data()));
    std::unique_ptr<DefaultApproximationBuilder<T, double>> full;
    std::unique_ptr<DefaultLocalApproximationBuilder<T, double>> diag;
    if (block_diagonal)
        diag = std::make_unique<DefaultLocalApproximationBuilder<T, double>>(A, Tc, Tc, tb, MPI_COMM_WORLD);
    else
        full = std::make_unique<DefaultApproximationBuilder<T, double>>(A, Tc, Tc, tb, MPI_COMM_WORLD);
    const DistributedOperator<T> &Op = block_diagonal ? diag->distributed_operator : full->distributed_operator;
    const HMatrix<T, double> &H      = block_diagonal ? diag->hmatrix : full->hmatrix;

    Dump D(gets(kv, "out", "/tmp/dist_products") + ".rank" + std::to_string(rank));
    D.i32("world_rank_n_mu_localnumbering", {world, rank, n, mu, (int)local_numbering});
    D.f64("alpha_beta", {std::real(alpha), std::imag(alpha), std::real(beta), std::imag(beta)});
    D.i32("perm", Tc.get_permutation());
    std::vector<int> pp;
    for (auto *c : Tc.get_clusters_on_partition()) {
        pp.push_back(c->get_offset());
        pp.push_back(c->get_size());
    }
    D.i32("partition", pp, {pp.size() / 2, 2});
    { // this rank's leaves with ranks (structure pin), htool's save_leaves_with_rank order
        std::vector<int> leaves;
        preorder_leaves(H, false, [&](const HMatrix<T, double> &l, bool) {
            leaves.push_back(l.get_target_cluster().get_offset());
            leaves.push_back(l.get_target_cluster().get_size());
            leaves.push_back(l.get_source_cluster().get_offset());
            leaves.push_back(l.get_source_cluster().get_size());
            leaves.push_back(l.get_rank());
        });
        D.i32("leaves", leaves, {leaves.size() / 5, 5});
    }
    const int off = pp[2 * rank], sz = pp[2 * rank + 1];
    // inputs: whole vectors / matrices (column-major n x mu = salt 21 / 22), local parts are their rows [off, off + sz)
    const std::vector<T> xin = hashed_values<T>(n, 17), y0 = hashed_values<T>(n, 18);
    const std::vector<T> Xcm = hashed_values<T>((size_t)n * mu, 21), Y0cm = hashed_values<T>((size_t)n * mu, 22);
    auto rows_cm = [&](const std::vector<T> &M, int lo, int cnt) { // rows [lo, lo + cnt) of a column-major n x mu matrix, column-major
        std::vector<T> r((size_t)cnt * mu);
        for (int j = 0; j < mu; j++)
            for (int i = 0; i < cnt; i++)
                r[i + (size_t)cnt * j] = M[lo + i + (size_t)n * j];
        return r;
    };
    auto to_row_major = [&](const std::vector<T> &M, int rows) { // column-major rows x mu -> row-major (mu fastest)
        std::vector<T> r(M.size());
        for (int j = 0; j < mu; j++)
            for (int i = 0; i < rows; i++)
                r[(size_t)i * mu + j] = M[i + (size_t)rows * j];
        return r;
    };
    T *work = nullptr;
    for (char trans : std::string(is_cplx<T>::value ? (sym == "H" ? "NC" : (sym == "S" ? "NT" : "NTC")) : "NT")) {
        const std::string t(1, trans);
        { // vectors
            std::vector<T> y = y0;
            add_distributed_operator_vector_product_global_to_global(trans, alpha, Op, xin.data(), beta, y.data(), work);
            if (rank == 0)
                D.vec("g2g_user_" + t, y);
            y = y0;
            internal_add_distributed_operator_vector_product_global_to_global(trans, alpha, Op, xin.data(), beta, y.data(), work);
            if (rank == 0)
                D.vec("g2g_internal_" + t, y);
            std::vector<T> xl(xin.begin() + off, xin.begin() + off + sz), yl(y0.begin() + off, y0.begin() + off + sz);
            internal_add_distributed_operator_vector_product_local_to_local(trans, alpha, Op, xl.data(), beta, yl.data(), work);
            D.vec("l2l_internal_" + t, yl);
            if (local_numbering) {
                yl.assign(y0.begin() + off, y0.begin() + off + sz);
                add_distributed_operator_vector_product_local_to_local(trans, alpha, Op, xl.data(), beta, yl.data(), work);
                D.vec("l2l_user_" + t, yl);
            }
        }
        { // row-major multi-RHS (mu fastest)
            Matrix<T> X(mu, n), Y(mu, n);
            const std::vector<T> xr = to_row_major(Xcm, n), yr = to_row_major(Y0cm, n);
            std::copy(xr.begin(), xr.end(), X.data());
            std::copy(yr.begin(), yr.end(), Y.data());
            internal_add_distributed_operator_matrix_product_row_major_global_to_global(trans, alpha, Op, X, beta, Y, work);
            if (rank == 0)
                D.f64p("g2g_rm_" + t, Y.data(), {(uint64_t)n, (uint64_t)mu});
            for (int with_beta = 1; with_beta >= 0; with_beta--) {
                Matrix<T> Xl(mu, sz), Yl(mu, sz);
                std::copy(xr.begin() + (size_t)off * mu, xr.begin() + (size_t)(off + sz) * mu, Xl.data());
                std::copy(yr.begin() + (size_t)off * mu, yr.begin() + (size_t)(off + sz) * mu, Yl.data());
                internal_add_distributed_operator_matrix_product_row_major_local_to_local(trans, alpha, Op, Xl, with_beta ? beta : T(0), Yl, work);
                D.f64p(std::string(with_beta ? "l2l_rm_" : "l2l_rm_beta0_") + t, Yl.data(), {(uint64_t)sz, (uint64_t)mu});
            }
        }
        { // column-major multi-RHS
            Matrix<T> X(n, mu), Y(n, mu);
            std::copy(Xcm.begin(), Xcm.end(), X.data());
            std::copy(Y0cm.begin(), Y0cm.end(), Y.data());
            add_distributed_operator_matrix_product_global_to_global(trans, alpha, Op, X, beta, Y, work);
            if (rank == 0)
                D.f64p("g2g_cm_user_" + t, Y.data(), {(uint64_t)mu, (uint64_t)n});
            std::copy(Y0cm.begin(), Y0cm.end(), Y.data());
            internal_add_distributed_operator_matrix_product_global_to_global(trans, alpha, Op, X, beta, Y, work);
            if (rank == 0)
                D.f64p("g2g_cm_internal_" + t, Y.data(), {(uint64_t)mu, (uint64_t)n});
            std::copy(Y0cm.begin(), Y0cm.end(), Y.data());
            add_distributed_operator_matrix_product_global_to_global(trans, alpha, Op, X, T(0), Y, work);
            if (rank == 0)
                D.f64p("g2g_cm_user_beta0_" + t, Y.data(), {(uint64_t)mu, (uint64_t)n});
            const std::vector<T> xl = rows_cm(Xcm, off, sz);
            std::vector<T> yl       = rows_cm(Y0cm, off, sz);
            internal_add_distributed_operator_matrix_product_local_to_local(trans, alpha, Op, xl.data(), beta, yl.data(), mu, work);
            D.f64p("l2l_cm_internal_" + t, yl.data(), {(uint64_t)mu, (uint64_t)sz});
            if (local_numbering) {
                yl = rows_cm(Y0cm, off, sz);
                add_distributed_operator_matrix_product_local_to_local(trans, alpha, Op, xl.data(), beta, yl.data(), mu, work);
                D.f64p("l2l_cm_user_" + t, yl.data(), {(uint64_t)mu, (uint64_t)sz});
            }
        }
    }
    { // sub product (virtual_global_to_local_operator.hpp:33): `in` = rows [offset, offset + size) of the source numbering, zero-extended.
        // As its caller does (solvers/geneo/coarse_operator_builder.hpp:99) with the rows of one partition after the other, accumulating;
        // and once with a range that overlaps the partitions partially, for mu = 1 only: the partial-overlap branch of the reference
        // advances `in` by rows instead of rows * mu (local_to_local_operators/hmatrix.hpp:45, restricted_operator.hpp:184).
        std::vector<T> yl = to_row_major(rows_cm(Y0cm, off, sz), sz);
        for (int i = 0; i < world; i++) {
            const std::vector<T> xs = to_row_major(rows_cm(Xcm, pp[2 * i], pp[2 * i + 1]), pp[2 * i + 1]);
            internal_add_distributed_operator_vector_sub_product_global_to_local(Op, xs.data(), yl.data(), mu, pp[2 * i], pp[2 * i + 1]);
        }
        D.f64p("sub_g2l", yl.data(), {(uint64_t)sz, (uint64_t)mu});
        const int s_off = n / 3, s_size = n / 4;
        std::vector<T> y1(y0.begin() + off, y0.begin() + off + sz);
        internal_add_distributed_operator_vector_sub_product_global_to_local(Op, xin.data() + s_off, y1.data(), 1, s_off, s_size);
        D.i32("sub_offset_size", {s_off, s_size});
        D.vec("sub_g2l_partial_mu1", y1);
    }
    return 0;
}

int main(int argc, char **argv) {
    MPI_Init(&argc, &argv);
    auto kv                = parse(argc, argv);
    const std::string prec = gets(kv, "prec", "f64");
    int rc                 = prec == "z64" ? run<std::complex<double>>(kv) : run<double>(kv);
    MPI_Finalize();
    return rc;
}

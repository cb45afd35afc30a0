// TEST / BENCH INFRASTRUCTURE (built in the dev container only: needs the htool headers under /root/reference/include and the image's
// MPICH; the BINARY travels to the GPU box, where bench.py --gpus N times it as `cpu_baseline` of kind "reference-mpi").
// htool's own MPI + OpenMP CPU path on the bench's configuration: every rank builds the block rows of its partition
// (HMatrixTreeBuilder::openmp_build with target_partition_number = rank: what DefaultApproximationBuilder holds,
// distributed_operator/utility.hpp:38-61), wraps them in RestrictedGlobalToLocalHMatrix + DistributedOperator and the product
// internal_add_distributed_operator_vector_product_global_to_global (distributed_operator/linalg/
// add_distributed_operator_vector_product_global_to_global.hpp:18-85: openmp leaf loop per rank + MPI_Allgatherv) is timed between
// barriers, maximum over the ranks, best of `reps`.
//   mpiexec -n P oracle/_ref/dist_bench n=1000000 geom=ellipse leaf=100 eps=1e-4 eta=10 mindepth=5 reps=5      (OMP_NUM_THREADS = cores / P)
// Rank 0 prints: ranks=P threads=T cgen=<sum over ranks> build=<max s> matvec=<best of max-over-ranks s>
#include <mpi.h>
#define main ref_driver_main
#include "ref_driver.cpp" // geometry, generator and option parsing of the fixture driver
#undef main
#include <htool/distributed_operator/distributed_operator.hpp>
#include <htool/distributed_operator/linalg/add_distributed_operator_vector_product_global_to_global.hpp>
#include <htool/distributed_operator/utility.hpp>
#include <omp.h>

int main(int argc, char **argv) {
    int provided = 0;
    MPI_Init_thread(&argc, &argv, MPI_THREAD_FUNNELED, &provided);
    int rank, world;
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &world);
    auto kv                = parse(argc, argv);
    const int n            = geti(kv, "n", 100000);
    const int reps         = geti(kv, "reps", 5);
    const int mindepth     = geti(kv, "mindepth", 0);
    const std::string geom = gets(kv, "geom", "ellipse"), sym = gets(kv, "sym", "N"), uplo = gets(kv, "uplo", "N"), comp = gets(kv, "compressor", "partialACA");
    const int dim          = geometry_dim(geom);
    std::vector<double> x;
    make_geometry(geom, n, 0., x);
    ClusterTreeBuilder<double> ctb;
    ctb.set_maximal_leaf_size(geti(kv, "leaf", 100));
    Cluster<double> T = ctb.create_cluster_tree(n, dim, x.data(), 2, world);
    InvDistGenerator<double> A(dim, x, x, getd(kv, "delta", 1e-5), getd(kv, "scale", 1.), 1., 0., false);
    HMatrixTreeBuilder<double, double> tb(getd(kv, "eps", 1e-4), getd(kv, "eta", 10), sym[0], uplo[0], -1);
    tb.set_minimal_target_depth(mindepth);
    tb.set_minimal_source_depth(mindepth);
    if (comp == "partialACA")
        tb.set_low_rank_generator(std::make_shared<partialACA<double>>(A, T.get_permutation().data(), T.get_permutation().data()));
    else
        tb.set_low_rank_generator(std::make_shared<sympartialACA<double>>(A, T.get_permutation().data(), T.get_permutation().data()));
    MPI_Barrier(MPI_COMM_WORLD);
    const double tb0  = MPI_Wtime();
    HMatrix<double> H = tb.openmp_build(A, T, T, rank, rank);
    double t_build    = MPI_Wtime() - tb0, t_build_max = 0;
    MPI_Reduce(&t_build, &t_build_max, 1, MPI_DOUBLE, MPI_MAX, 0, MPI_COMM_WORLD);
    long long cgen = 0, cgen_all = 0;
    preorder_leaves(H, false, [&](const HMatrix<double> &l, bool) {
        const long long m = l.get_target_cluster().get_size(), k = l.get_source_cluster().get_size();
        cgen += l.is_low_rank() ? (long long)l.get_rank() * (m + k) : m * k;
    });
    MPI_Reduce(&cgen, &cgen_all, 1, MPI_LONG_LONG, MPI_SUM, 0, MPI_COMM_WORLD);
    const RestrictedGlobalToLocalHMatrix<double, double> local(H, H.get_target_cluster(), H.get_source_cluster(), false, false);
    CustomApproximationBuilder<double> holder(T, T, MPI_COMM_WORLD, local);
    std::vector<double> in(n), out(n, 0.), work(3 * (size_t)n);
    for (int i = 0; i < n; i++)
        in[i] = (double)(uint32_t)((uint32_t)(i + 1) * 2654435761u + 40503u) / 4294967296.0; // oracle.hashed_vector(n, 1)
    double best = 1e30;
    for (int r = 0; r < reps + 1; r++) { // first call untimed
        MPI_Barrier(MPI_COMM_WORLD);
        const double t0 = MPI_Wtime();
        internal_add_distributed_operator_vector_product_global_to_global('N', 1., holder.distributed_operator, in.data(), 0., out.data(), work.data());
        double t = MPI_Wtime() - t0, tmax = 0;
        MPI_Allreduce(&t, &tmax, 1, MPI_DOUBLE, MPI_MAX, MPI_COMM_WORLD);
        if (r > 0)
            best = std::min(best, tmax);
    }
    if (rank == 0) {
        double checksum = 0;
        for (int i = 0; i < n; i++)
            checksum += out[i];
        printf("ranks=%d threads=%d n=%d cgen=%lld build=%.3fs matvec=%.5fs checksum=%.10e\n", world, omp_get_max_threads(), n, cgen_all, t_build_max, best, checksum);
    }
    MPI_Finalize();
    return 0;
}

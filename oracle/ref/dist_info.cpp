// TEST INFRASTRUCTURE (dev container only: needs the htool headers under /root/reference/include and the image's MPICH).
// Runs the REAL reference under MPI -- every rank builds its block rows on the CPU exactly as use_distributed_operator.cpp
// does -- and writes what print_distributed_hmatrix_information (hmatrix/hmatrix_distributed_output.hpp:218-243) prints on
// rank 0.  tests/golden/make_golden.py stores that text as a fixture; nothing here is shipped or run on the GPU box.
//   mpiexec -n P oracle/_ref/dist_info n=4000 geom=ellipse leaf=100 eps=1e-4 eta=10 sym=N uplo=N compressor=partialACA out=/tmp/info.txt
#include <mpi.h>
#define main ref_driver_main
#include "ref_driver.cpp" // geometry, generator and option parsing of the fixture driver
#undef main
#include <htool/hmatrix/hmatrix_distributed_output.hpp>

int main(int argc, char **argv) {
    MPI_Init(&argc, &argv);
    int rank, world;
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &world);
    auto kv                = parse(argc, argv);
    const int n            = geti(kv, "n", 4000);
    const std::string geom = gets(kv, "geom", "ellipse"), sym = gets(kv, "sym", "N"), uplo = gets(kv, "uplo", "N"), comp = gets(kv, "compressor", "partialACA");
    const int dim          = geometry_dim(geom);
    std::vector<double> x;
    make_geometry(geom, n, 0., x);
    ClusterTreeBuilder<double> ctb;
    ctb.set_maximal_leaf_size(geti(kv, "leaf", 100));
    Cluster<double> T = ctb.create_cluster_tree(n, dim, x.data(), geti(kv, "children", 2), world);
    InvDistGenerator<double> A(dim, x, x, getd(kv, "delta", 1e-5), getd(kv, "scale", 1.), 1., 0., false);
    HMatrixTreeBuilder<double, double> tb(getd(kv, "eps", 1e-4), getd(kv, "eta", 10), sym[0], uplo[0], -1);
    if (comp == "partialACA")
        tb.set_low_rank_generator(std::make_shared<partialACA<double>>(A, T.get_permutation().data(), T.get_permutation().data()));
    else
        tb.set_low_rank_generator(std::make_shared<sympartialACA<double>>(A, T.get_permutation().data(), T.get_permutation().data()));
    HMatrix<double> H = tb.sequential_build(A, T, T, rank, rank);
    std::ostringstream text;
    print_distributed_hmatrix_information(H, text, MPI_COMM_WORLD);
    if (rank == 0) {
        std::ofstream f(gets(kv, "out", "/tmp/dist_info.txt"));
        f << text.str();
    }
    MPI_Finalize();
    return 0;
}

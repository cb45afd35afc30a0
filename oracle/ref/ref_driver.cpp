// ref_driver.cpp -- TEST INFRASTRUCTURE (oracle side), never linked into the product.
//
// Drives the REAL reference (htool headers where they lie under /root/reference/include, compiled
// in place; nothing is copied) to (1) dump golden fixtures for tests/golden/ and (2) cross-check the
// CPU restatement in oracle/hmx_oracle.cpp.  Built only in the dev container (oracle/Makefile target
// `ref`), output binary goes to oracle/_ref/ (git-ignored; the BINARY travels to the GPU box with the snapshot, where
// bench.py times it as the `cpu_baseline` of kind "reference" -- the sources and the htool headers never do).
//
// Usage: ref_driver <mode> key=value ... out=<file>
//   mode=hmat  : cluster tree + block tree + compression + H-matvec fixture
//   mode=lrmat : the 500x100 two-disk block of tests/functional_tests/hmatrix/lrmat (all 4 compressors)
//
// Dump format (little endian): repeated records
//   u32 name_len | name bytes | u8 dtype ('i' int32,'l' int64,'d' float64) | u32 ndim | u64 dims[ndim] | payload
#include <htool/clustering/cluster_node.hpp>
#include <htool/clustering/cluster_output.hpp>
#include <htool/hmatrix/hmatrix_output.hpp>
#include <htool/matrix/utils/output.hpp>
#include <htool/clustering/implementations/partitioning.hpp>
#include <htool/clustering/tree_builder/tree_builder.hpp>
#include <htool/hmatrix/hmatrix.hpp>
#include <htool/hmatrix/linalg/add_hmatrix_matrix_product_row_major.hpp>
#include <htool/hmatrix/linalg/add_hmatrix_vector_product.hpp>
#include <htool/hmatrix/lrmat/SVD.hpp>
#include <htool/hmatrix/lrmat/fullACA.hpp>
#include <htool/hmatrix/lrmat/partialACA.hpp>
#include <htool/hmatrix/lrmat/sympartialACA.hpp>
#include <htool/hmatrix/tree_builder/tree_builder.hpp>
#include <htool/hmatrix/utils/recompression.hpp>
#include <htool/testing/geometry.hpp>

#include <chrono>
#include <complex>
#include <type_traits>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <random>
#include <string>

using namespace htool;

struct Dump {
    FILE *f;
    explicit Dump(const std::string &path) : f(fopen(path.c_str(), "wb")) {
        if (!f) {
            perror("open dump");
            exit(2);
        }
    }
    ~Dump() { fclose(f); }
    void rec(const std::string &name, char dtype, const std::vector<uint64_t> &dims, const void *data, size_t elem) {
        uint32_t nl = name.size();
        fwrite(&nl, 4, 1, f);
        fwrite(name.data(), 1, nl, f);
        fwrite(&dtype, 1, 1, f);
        uint32_t nd = dims.size();
        fwrite(&nd, 4, 1, f);
        size_t n = 1;
        for (auto d : dims) {
            fwrite(&d, 8, 1, f);
            n *= d;
        }
        if (n)
            fwrite(data, elem, n, f);
    }
    void i32(const std::string &name, const std::vector<int> &v, std::vector<uint64_t> dims = {}) {
        if (dims.empty())
            dims = {v.size()};
        rec(name, 'i', dims, v.data(), 4);
    }
    void f64(const std::string &name, const std::vector<double> &v, std::vector<uint64_t> dims = {}) {
        if (dims.empty())
            dims = {v.size()};
        rec(name, 'd', dims, v.data(), 8);
    }
    void f64p(const std::string &name, const double *p, std::vector<uint64_t> dims) { rec(name, 'd', dims, p, 8); }
    void f64p(const std::string &name, const float *p, std::vector<uint64_t> dims) { // fp32 payloads are stored as float64 (exact)
        size_t n = 1;
        for (auto d : dims)
            n *= d;
        std::vector<double> tmp(p, p + n);
        rec(name, 'd', dims, tmp.data(), 8);
    }
    template <typename T>
    void vec(const std::string &name, const std::vector<T> &v, std::vector<uint64_t> dims = {}) {
        if (dims.empty())
            dims = {v.size()};
        std::vector<double> tmp(v.begin(), v.end());
        rec(name, 'd', dims, tmp.data(), 8);
    }
    // complex payloads: float64 pairs, trailing dimension 2 (numpy: .view(complex128))
    template <typename U>
    void f64p(const std::string &name, const std::complex<U> *p, std::vector<uint64_t> dims) {
        size_t n = 1;
        for (auto d : dims)
            n *= d;
        std::vector<double> tmp(2 * n);
        for (size_t i = 0; i < n; i++) {
            tmp[2 * i]     = p[i].real();
            tmp[2 * i + 1] = p[i].imag();
        }
        dims.push_back(2);
        rec(name, 'd', dims, tmp.data(), 8);
    }
    template <typename U>
    void vec(const std::string &name, const std::vector<std::complex<U>> &v, std::vector<uint64_t> dims = {}) {
        if (dims.empty())
            dims = {v.size()};
        f64p(name, v.data(), dims);
    }
};

template <typename T>
struct is_cplx : std::false_type {};
template <typename U>
struct is_cplx<std::complex<U>> : std::true_type {};
template <typename T>
static T make_value(double re, double im) {
    if constexpr (is_cplx<T>::value)
        return T((typename T::value_type)re, (typename T::value_type)im);
    else
        return (T)re;
}

// Kernel family K(x,y) = 1 / (delta + scale * |x-y|), evaluated with the same operation order as the
// reference's own generators (examples/use_hmatrix.cpp:33, testing/generator_test.hpp:159,185):
// squared differences accumulated left to right from 0, one sqrt, one multiply, one add, one divide.
// Two more families (kernel=helmholtz, kernel=laplace: include/hmx.h HMX_KERNEL_HELMHOLTZ / HMX_KERNEL_LAPLACE_SL), written as a user would
// write them on top of the same distance: exp(i k r) / (delta + scale r) and (cre + i cim) / (4 pi (delta + r)).  sin / cos of the phase are
// the documented IEEE sequence of the device kernel (Cody-Waite reduction with three 33-bit pieces of pi/2, minimax polynomials with
// fdlibm's coefficients), restated here so that both sides produce the same bits (compiled with -ffp-contract=off).
static inline void ref_sincos(double x, double &sn, double &cs) {
    const double fn = std::rint(x * 6.36619772367581382433e-01);
    double r        = x - fn * 1.57079632673412561417e+00;
    r               = r - fn * 6.07710050630396597660e-11;
    r               = r - fn * 2.02226624871116645580e-21;
    const double z  = r * r;
    const double ps = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    const double s  = r + (z * r) * (-1.66666666666666324348e-01 + z * ps);
    const double pc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    const double hz = 0.5 * z, w = 1.0 - hz;
    const double c  = w + (((1.0 - w) - hz) + z * pc);
    const int q     = (int)((long long)fn & 3);
    sn              = (q == 0) ? s : (q == 1) ? c : (q == 2) ? -s : -c;
    cs              = (q == 0) ? c : (q == 1) ? -s : (q == 2) ? -c : s;
}
static int g_kernel_family  = 0; // 0: inverse distance, 1: Helmholtz, 2: Laplace single layer (set from the command line: kernel=...)
static double g_wavenumber = 0;

template <typename T>
class InvDistGenerator : public VirtualGenerator<T> {
    int m_dim;
    const std::vector<double> &m_xt, &m_xs;
    double m_delta, m_scale;
    // complex coefficients: (cre + i * cim * sgn) / (delta + scale |x-y|), sgn = 1 (complex symmetric, the form of
    // testing/generator_test.hpp:163-170,189-196) or sign(x_target[0] - x_source[0]) (Hermitian, :198-205)
    double m_cre = 1, m_cim = 0;
    bool m_hermitian = false;

  public:
    InvDistGenerator(int dim, const std::vector<double> &xt, const std::vector<double> &xs, double delta, double scale, double cre = 1, double cim = 0, bool hermitian = false) : m_dim(dim), m_xt(xt), m_xs(xs), m_delta(delta), m_scale(scale), m_cre(cre), m_cim(cim), m_hermitian(hermitian) {}
    double denominator(int i, int j) const {
        double s = 0;
        for (int p = 0; p < m_dim; p++) {
            double d = m_xt[m_dim * i + p] - m_xs[m_dim * j + p];
            s        = s + d * d;
        }
        return m_delta + m_scale * std::sqrt(s);
    }
    double get_coef(int i, int j) const { return 1. / denominator(i, j); }
    double distance(int i, int j) const {
        double s = 0;
        for (int p = 0; p < m_dim; p++) {
            double d = m_xt[m_dim * i + p] - m_xs[m_dim * j + p];
            s        = s + d * d;
        }
        return std::sqrt(s);
    }
    void copy_submatrix(int M, int N, const int *rows, const int *cols, T *ptr) const override {
        for (int j = 0; j < M; j++)
            for (int k = 0; k < N; k++) {
                if (g_kernel_family == 1) { // exp(i k r) / (delta + scale r)
                    const double r = distance(rows[j], cols[k]), den = m_delta + m_scale * r;
                    double sn, cs;
                    ref_sincos(g_wavenumber * r, sn, cs);
                    ptr[j + (size_t)M * k] = make_value<T>(cs / den, sn / den);
                    continue;
                }
                if (g_kernel_family == 2) { // (cre + i cim) / (4 pi (delta + r))
                    const double den       = 12.566370614359172 * (m_delta + distance(rows[j], cols[k]));
                    ptr[j + (size_t)M * k] = make_value<T>(m_cre / den, m_cim / den);
                    continue;
                }
                if constexpr (is_cplx<T>::value) {
                    const double u   = m_xt[m_dim * rows[j]] - m_xs[m_dim * cols[k]];
                    const double sgn = m_hermitian ? (u > 0 ? 1. : (u < 0 ? -1. : 0.)) : 1.;
                    ptr[j + (size_t)M * k] = T(std::complex<double>(m_cre, m_cim * sgn) / denominator(rows[j], cols[k])); // complex<double> / double, then to T
                } else
                    ptr[j + (size_t)M * k] = get_coef(rows[j], cols[k]); // double expression assigned to T, as a user generator would
            }
    }
};

static std::map<std::string, std::string> parse(int argc, char **argv) {
    std::map<std::string, std::string> kv;
    for (int i = 1; i < argc; i++) {
        std::string a(argv[i]);
        auto p = a.find('=');
        if (p == std::string::npos)
            kv["mode"] = a;
        else
            kv[a.substr(0, p)] = a.substr(p + 1);
    }
    return kv;
}
static std::string gets(std::map<std::string, std::string> &kv, const std::string &k, const std::string &d) { return kv.count(k) ? kv[k] : d; }
static double getd(std::map<std::string, std::string> &kv, const std::string &k, double d) { return kv.count(k) ? atof(kv[k].c_str()) : d; }
static int geti(std::map<std::string, std::string> &kv, const std::string &k, int d) { return kv.count(k) ? atoi(kv[k].c_str()) : d; }

static int geometry_dim(const std::string &geom) { return geom == "disk2d" ? 2 : 3; }
static void make_geometry(const std::string &geom, int n, double z, std::vector<double> &x) {
    x.assign(geometry_dim(geom) * (size_t)n, 0.);
    if (geom == "disk2d")
        create_disk(2, z, n, x.data());
    else if (geom == "ellipse")
        create_rotated_ellipse(3, 4., 1., 0., z, n, x.data());
    else if (geom == "disk")
        create_disk(3, z, n, x.data());
    else if (geom == "ball")
        create_sphere(n, x.data());
    else {
        fprintf(stderr, "unknown geometry %s\n", geom.c_str());
        exit(2);
    }
}

static void dump_cluster_tree(Dump &D, const std::string &prefix, const Cluster<double> &root) {
    // preorder, children in order: depth, offset, size, rank, counter, nchildren | radius, center[3]
    std::vector<int> ints;
    std::vector<double> reals;
    preorder_tree_traversal(root, [&](const Cluster<double> &c) {
        ints.push_back(c.get_depth());
        ints.push_back(c.get_offset());
        ints.push_back(c.get_size());
        ints.push_back(c.get_rank());
        ints.push_back(c.get_counter());
        ints.push_back((int)c.get_children().size());
        reals.push_back(c.get_radius());
        for (int p = 0; p < 3; p++)
            reals.push_back(p < (int)c.get_center().size() ? c.get_center()[p] : 0.);
    });
    D.i32(prefix + "nodes_int", ints, {ints.size() / 6, 6});
    D.f64(prefix + "nodes_real", reals, {reals.size() / 4, 4});
    D.i32(prefix + "perm", root.get_permutation());
    std::vector<int> part;
    for (auto *c : root.get_clusters_on_partition()) {
        part.push_back(c->get_offset());
        part.push_back(c->get_size());
    }
    D.i32(prefix + "partition", part, {part.size() / 2, 2});
}

template <typename HM, typename F>
static void preorder_leaves(const HM &h, bool sym_anc, F &&f) {
    if (h.is_leaf()) {
        f(h, sym_anc);
        return;
    }
    for (auto &c : h.get_children())
        preorder_leaves(*c, sym_anc || h.get_symmetry() != 'N', f);
}

static std::shared_ptr<VirtualPartitioning<double>> make_partitioning(const std::string &s) {
    if (s == "pca_regular")
        return std::make_shared<Partitioning<double, ComputeLargestExtent<double>, RegularSplitting<double>>>();
    if (s == "pca_geometric")
        return std::make_shared<Partitioning<double, ComputeLargestExtent<double>, GeometricSplitting<double>>>();
    if (s == "bbox_regular")
        return std::make_shared<Partitioning<double, ComputeBoundingBox<double>, RegularSplitting<double>>>();
    if (s == "bbox_geometric")
        return std::make_shared<Partitioning<double, ComputeBoundingBox<double>, GeometricSplitting<double>>>();
    if (s == "n_pca_regular")
        return std::make_shared<Partitioning_N<double, ComputeLargestExtent<double>, RegularSplitting<double>>>();
    if (s == "n_bbox_regular")
        return std::make_shared<Partitioning_N<double, ComputeBoundingBox<double>, RegularSplitting<double>>>();
    fprintf(stderr, "unknown partitioning %s\n", s.c_str());
    exit(2);
}

template <typename T>
static int run_hmat(std::map<std::string, std::string> &kv) {
    using HM = HMatrix<T, double>;
    int n                 = geti(kv, "n", 2000);
    int nsrc              = geti(kv, "nsrc", 0); // 0 => square, source == target geometry
    std::string geom      = gets(kv, "geom", "ellipse");
    std::string sgeom     = gets(kv, "sgeom", geom);
    double sz             = getd(kv, "sz", 0.);
    int leaf              = geti(kv, "leaf", 100);
    int children          = geti(kv, "children", 2);
    int partitions        = geti(kv, "partitions", 2);
    std::string partstr   = gets(kv, "partitioning", "pca_regular");
    double eps            = getd(kv, "eps", 1e-4);
    double eta            = getd(kv, "eta", 10);
    std::string sym       = gets(kv, "sym", "N");
    std::string uplo      = gets(kv, "uplo", "N");
    std::string comp      = gets(kv, "compressor", "partialACA");
    double delta          = getd(kv, "delta", 1e-5);
    double scale          = getd(kv, "scale", 1.);
    const std::string kernel_name = gets(kv, "kernel", "invdist");
    g_kernel_family               = kernel_name == "helmholtz" ? 1 : (kernel_name == "laplace" ? 2 : 0);
    g_wavenumber                  = getd(kv, "wavenumber", 0.);
    int mindepth          = geti(kv, "mindepth", 0);
    int rank              = geti(kv, "rank", -1); // target_partition_number (and symmetry partition)
    int reqrank           = geti(kv, "reqrank", -1);
    int dump_blocks       = geti(kv, "dump_blocks", 2);
    int dump_all          = geti(kv, "dump_all", 0); // dump every block's payload (U,V / dense), concatenated
    int time_reps         = geti(kv, "time_reps", 0);
    int par               = geti(kv, "par", 0); // 1: openmp_build + openmp_internal_add_hmatrix_vector_product (the reference's MPI+OpenMP CPU path, one rank)
    double alpha          = getd(kv, "alpha", 3.);
    double beta           = getd(kv, "beta", 2.);
    std::string out       = gets(kv, "out", "/tmp/ref_hmat.bin");
    bool square           = (nsrc == 0);
    int consistent        = geti(kv, "consistent", 1);
    int local             = geti(kv, "local", -1); // >= 0: block-diagonal operator rooted at partition clusters (DefaultLocalApproximationBuilder)
    const int dim         = geometry_dim(geom);

    std::vector<double> xt, xs_store;
    make_geometry(geom, n, 0., xt);
    if (!square)
        make_geometry(sgeom, nsrc, sz, xs_store);
    const std::vector<double> &xs = square ? xt : xs_store;
    int ns                        = square ? n : nsrc;

    Dump D(out);
    D.f64("xt", xt, {(uint64_t)n, (uint64_t)dim});
    if (!square)
        D.f64("xs", xs, {(uint64_t)ns, (uint64_t)dim});

    ClusterTreeBuilder<double> ctb;
    ctb.set_maximal_leaf_size(leaf);
    ctb.set_partitioning_strategy(make_partitioning(partstr));
    if (geti(kv, "complete", 0))
        ctb.set_is_complete(true);
    // user-given partitions (tree_builder.hpp:46-48): "global" = a scattered assignment point -> part (hash of the index),
    // "local" = contiguous chunks given as (offset, size)
    const std::string given = gets(kv, "given", "none");
    std::vector<int> given_partition;
    if (given == "global") {
        given_partition.resize(n);
        for (int i = 0; i < n; i++)
            given_partition[i] = (int)(((uint32_t)(i + 1) * 2654435761u >> 7) % (uint32_t)partitions);
    } else if (given == "local") {
        given_partition.resize(2 * partitions);
        for (int p = 0; p < partitions; p++) {
            const int lo = (int)((long long)n * p / partitions), hi = (int)((long long)n * (p + 1) / partitions);
            given_partition[2 * p]     = lo;
            given_partition[2 * p + 1] = hi - lo;
        }
    }
    auto t0             = std::chrono::steady_clock::now();
    Cluster<double> tct = given == "global" ? ctb.create_cluster_tree_from_global_partition(n, dim, xt.data(), children, partitions, given_partition.data())
                          : given == "local" ? ctb.create_cluster_tree_from_local_partition(n, dim, xt.data(), children, partitions, given_partition.data())
                                             : ctb.create_cluster_tree(n, dim, xt.data(), children, partitions);
    auto t1             = std::chrono::steady_clock::now();
    std::unique_ptr<Cluster<double>> sct_store;
    if (!square)
        sct_store = std::make_unique<Cluster<double>>(ctb.create_cluster_tree(ns, dim, xs.data(), children, partitions));
    const Cluster<double> &sct = square ? tct : *sct_store;
    dump_cluster_tree(D, "t_", tct);
    if (!square)
        dump_cluster_tree(D, "s_", sct);

    const bool cplx       = is_cplx<T>::value;
    const double cre = getd(kv, "cre", 1.), cim = getd(kv, "cim", cplx ? 1. : 0.);
    InvDistGenerator<T> A(dim, xt, xs, delta, scale, cre, cim, sym == "H");
    HMatrixTreeBuilder<T, double> tb(eps, eta, sym[0], uplo[0], reqrank);
    if (!consistent)
        tb.set_block_tree_consistency(false);
    if (comp == "partialACA")
        tb.set_low_rank_generator(std::make_shared<partialACA<T>>(A, tct.get_permutation().data(), sct.get_permutation().data()));
    else if (comp == "sympartialACA")
        tb.set_low_rank_generator(std::make_shared<sympartialACA<T>>(A, tct.get_permutation().data(), sct.get_permutation().data()));
    else if (comp == "fullACA")
        tb.set_low_rank_generator(std::make_shared<fullACA<T>>(A, tct.get_permutation().data(), sct.get_permutation().data()));
    else if (comp == "SVD")
        tb.set_low_rank_generator(std::make_shared<SVD<T>>(A, tct.get_permutation().data(), sct.get_permutation().data()));
    else if (comp != "default") {
        fprintf(stderr, "unknown compressor\n");
        return 2;
    }
    tb.set_minimal_target_depth(mindepth);
    tb.set_minimal_source_depth(mindepth);

    auto t2           = std::chrono::steady_clock::now();
    HM H = local >= 0 ? tb.sequential_build(A, tct.get_cluster_on_partition(local), sct.get_cluster_on_partition(local))
                      : (par ? tb.openmp_build(A, tct, sct, rank, rank) : tb.sequential_build(A, tct, sct, rank, rank));
    if (geti(kv, "recompress", 0))
        recompression(H); // hmatrix/utils/recompression.hpp:8-13 (SVD_recompression of every low-rank leaf) // hmatrix/utils/recompression.hpp:8-13 (SVD_recompression of every low-rank leaf)
    auto t3           = std::chrono::steady_clock::now();

    // Leaves in natural preorder (children in creation order)
    std::vector<int> leaves; // t_off t_size s_off s_size rank(-1 dense) mirror
    std::vector<const HM *> leaf_ptr;
    preorder_leaves(H, H.get_symmetry() != 'N', [&](const HM &l, bool sym_anc) {
        leaves.push_back(l.get_target_cluster().get_offset());
        leaves.push_back(l.get_target_cluster().get_size());
        leaves.push_back(l.get_source_cluster().get_offset());
        leaves.push_back(l.get_source_cluster().get_size());
        leaves.push_back(l.get_rank());
        leaves.push_back((sym_anc && l.get_target_cluster().get_offset() != l.get_source_cluster().get_offset()) ? 1 : 0);
        leaf_ptr.push_back(&l);
    });
    D.i32("leaves", leaves, {leaves.size() / 6, 6});
    std::vector<int> rootinfo = {H.get_target_cluster().get_offset(), H.get_target_cluster().get_size(), H.get_source_cluster().get_offset(), H.get_source_cluster().get_size(), tb.get_false_positive(), (int)H.get_symmetry_for_leaves(), (int)H.get_UPLO_for_leaves()};
    D.i32("rootinfo", rootinfo);

    // the reference's own on-disk formats (clustering/cluster_output.hpp:33-84,87-179, hmatrix/hmatrix_output.hpp:39-55):
    // <prefix>_cluster_tree{,_properties}.csv, the same tree after read_cluster_tree -> save (prefix_reread_*), the leaf list
    std::string save_prefix = gets(kv, "save_prefix", "");
    if (!save_prefix.empty()) {
        save_cluster_tree(tct, save_prefix);
        Cluster<double> reread = read_cluster_tree<double>(save_prefix + "_cluster_tree_properties.csv", save_prefix + "_cluster_tree.csv");
        save_cluster_tree(reread, save_prefix + "_reread");
        save_leaves_with_rank(H, save_prefix + "_leaves");
        { // print_tree_parameters + print_hmatrix_information (hmatrix/hmatrix_output.hpp:101-118,218-236), as use_hmatrix.cpp prints them
            std::ofstream info(save_prefix + "_information.txt");
            print_tree_parameters(H, info);
            print_hmatrix_information(H, info);
        }
        for (auto *l : leaf_ptr)
            if (l->is_dense()) {
                matrix_to_bytes(*l->get_dense_data(), save_prefix + "_dense0.bin");
                break;
            }
    }

    // Payload of the first few low-rank and dense leaves (or all of them, concatenated)
    {
        int nlr = 0, nd = 0;
        std::vector<T> allU, allV, allD;
        for (size_t b = 0; b < leaf_ptr.size(); b++) {
            auto *l = leaf_ptr[b];
            if (l->is_low_rank()) {
                auto &U = l->get_low_rank_data()->get_U();
                auto &V = l->get_low_rank_data()->get_V();
                if (dump_all) {
                    allU.insert(allU.end(), U.data(), U.data() + (size_t)U.nb_rows() * U.nb_cols());
                    allV.insert(allV.end(), V.data(), V.data() + (size_t)V.nb_rows() * V.nb_cols());
                } else if (nlr < dump_blocks) {
                    D.f64p("U_" + std::to_string(b), U.data(), {(uint64_t)U.nb_cols(), (uint64_t)U.nb_rows()}); // col-major M x r
                    D.f64p("V_" + std::to_string(b), V.data(), {(uint64_t)V.nb_cols(), (uint64_t)V.nb_rows()}); // col-major r x N
                }
                nlr++;
            } else if (l->is_dense()) {
                auto &M = *l->get_dense_data();
                if (dump_all)
                    allD.insert(allD.end(), M.data(), M.data() + (size_t)M.nb_rows() * M.nb_cols());
                else if (nd < std::min(dump_blocks, 1))
                    D.f64p("D_" + std::to_string(b), M.data(), {(uint64_t)M.nb_cols(), (uint64_t)M.nb_rows()});
                nd++;
            }
        }
        if (dump_all) {
            D.vec("allU", allU);
            D.vec("allV", allV);
            D.vec("allD", allD);
        }
    }

    // H-matvec in cluster ("internal") numbering: seeded inputs, alpha/beta as in
    // tests/functional_tests/hmatrix/test_task_based_hmatrix_vector_product.hpp:90-91
    int nrows = H.get_target_cluster().get_size();
    int ncols = H.get_source_cluster().get_size();
    // Inputs are a closed-form hash of the index (reproducible from numpy: oracle/oracle.py hashed_vector),
    // so fixtures only need to store outputs.
    auto hval = [](size_t i, unsigned salt) { return double((uint32_t)((uint32_t)(i + 1) * 2654435761u + salt * 40503u)) / 4294967296.0; };
    auto hashed = [&](size_t n, unsigned salt) { // complex: imaginary part = the same hash with salt + 16
        std::vector<T> v(n);
        for (size_t i = 0; i < n; i++)
            v[i] = make_value<T>(hval(i, salt), hval(i, salt + 16));
        return v;
    };
    std::vector<T> x = hashed(ncols, 1), xT = hashed(nrows, 2), y0 = hashed(nrows, 3), y0T = hashed(ncols, 4);
    const double alpha_im = getd(kv, "alpha_im", cplx ? 0.5 : 0.), beta_im = getd(kv, "beta_im", cplx ? -0.25 : 0.);
    const T al = make_value<T>(alpha, alpha_im), be = make_value<T>(beta, beta_im);
    D.f64("alphabeta", {alpha, beta, alpha_im, beta_im});
    {
        std::vector<T> y = y0;
        sequential_internal_add_hmatrix_vector_product('N', al, H, x.data(), be, y.data());
        D.vec("yN", y);
        if (geti(kv, "extra_ab", 0)) { // a second product with alpha = 1, beta = 0 (the timed form)
            std::vector<T> y1(nrows, T(0));
            if (par)
                openmp_internal_add_hmatrix_vector_product('N', T(1.), H, x.data(), T(0.), y1.data());
            else
                sequential_internal_add_hmatrix_vector_product('N', T(1.), H, x.data(), T(0.), y1.data());
            D.vec("yN_a1b0", y1);
        }
        if (sym != "H") { // trans='T' with 'H' leaves is refused by the reference (add_hmatrix_vector_product.hpp:59-62)
            std::vector<T> yt = y0T;
            sequential_internal_add_hmatrix_vector_product('T', al, H, xT.data(), be, yt.data());
            D.vec("yT", yt);
        }
        if (cplx && sym != "S") {
            std::vector<T> yc = y0T;
            sequential_internal_add_hmatrix_vector_product('C', al, H, xT.data(), be, yc.data());
            D.vec("yC", yc);
        }
    }
    if (rank < 0 && square && local < 0) {
        // user-numbering front end (a16)
        std::vector<T> y = y0;
        add_hmatrix_vector_product('N', al, H, x.data(), be, y.data());
        D.vec("yN_user", y);
    }
    // multi-RHS row-major (a18), mu = 2
    {
        int mu                = geti(kv, "mu", 2);
        std::vector<T> X = hashed(ncols * (size_t)mu, 5), Y = hashed(nrows * (size_t)mu, 6);
        if (par)
            openmp_internal_add_hmatrix_matrix_product_row_major('N', 'N', al, H, X.data(), be, Y.data(), mu);
        else
            sequential_internal_add_hmatrix_matrix_product_row_major('N', 'N', al, H, X.data(), be, Y.data(), mu);
        D.vec("YNrm", Y, {(uint64_t)nrows, (uint64_t)mu});
    }

    // stats (hmatrix_output.hpp:153-175 semantics)
    long long cgen_dense = 0, cgen_lr = 0;
    int n_dense = 0, n_lr = 0, rmin = 1 << 30, rmax = 0;
    double rsum = 0;
    for (auto *l : leaf_ptr) {
        long long m = l->get_target_cluster().get_size(), nn = l->get_source_cluster().get_size();
        if (l->is_dense()) {
            cgen_dense += m * nn;
            n_dense++;
        } else {
            int r = l->get_rank();
            cgen_lr += (long long)r * (m + nn);
            n_lr++;
            rmin = std::min(rmin, r);
            rmax = std::max(rmax, r);
            rsum += r;
        }
    }
    double t_tree  = std::chrono::duration<double>(t1 - t0).count();
    double t_build = std::chrono::duration<double>(t3 - t2).count();
    double t_mv    = 0;
    if (time_reps > 0) {
        std::vector<T> y(nrows, 0.);
        double best = 1e30;
        for (int r = 0; r < time_reps; r++) {
            auto a = std::chrono::steady_clock::now();
            if (par)
                openmp_internal_add_hmatrix_vector_product('N', T(1.), H, x.data(), T(0.), y.data());
            else
                sequential_internal_add_hmatrix_vector_product('N', T(1.), H, x.data(), T(0.), y.data());
            auto b = std::chrono::steady_clock::now();
            best   = std::min(best, std::chrono::duration<double>(b - a).count());
        }
        t_mv = best;
    }
    D.f64("stats", {(double)n_dense, (double)n_lr, (double)cgen_dense, (double)cgen_lr, (double)rmin, n_lr ? rsum / n_lr : 0., (double)rmax, t_tree, t_build, t_mv});
    printf("n=%d dense=%d lowrank=%d cgen=%lld+%lld rank=%d/%.2f/%d false_pos=%d tree=%.3fs build=%.3fs matvec=%.4fs\n", n, n_dense, n_lr, cgen_dense, cgen_lr, rmin, n_lr ? rsum / n_lr : 0., rmax, tb.get_false_positive(), t_tree, t_build, t_mv);
    return 0;
}

// The 500 x 100 block between two unit disks (tests/functional_tests/hmatrix/lrmat/lrmat_build/*.cpp).
// The source cluster is built from the TARGET coordinates there (test_lrmat_build_partialACA.cpp:44);
// mirrored here so the permutations match.
static int run_lrmat(std::map<std::string, std::string> &kv) {
    double distance = getd(kv, "distance", 15);
    double eps      = getd(kv, "eps", 1e-4);
    int nr          = geti(kv, "nr", 500);
    int nc          = geti(kv, "nc", 100);
    std::string out = gets(kv, "out", "/tmp/ref_lrmat.bin");
    std::vector<double> xt(3 * nr), xs(3 * nc);
    create_disk(3, 0., nr, xt.data());
    create_disk(3, distance, nc, xs.data());
    ClusterTreeBuilder<double> ctb;
    Cluster<double> t = ctb.create_cluster_tree(nr, 3, xt.data(), 2, 2);
    Cluster<double> s = ctb.create_cluster_tree(nc, 3, xt.data(), 2, 2);
    InvDistGenerator<double> A(3, xt, xs, 0., 4 * M_PI);
    Dump D(out);
    D.f64("xt", xt, {(uint64_t)nr, 3});
    D.f64("xs", xs, {(uint64_t)nc, 3});
    D.i32("t_perm", t.get_permutation());
    D.i32("s_perm", s.get_permutation());
    auto run = [&](const std::string &name, const VirtualInternalLowRankGenerator<double> &c) {
        LowRankMatrix<double> fixed(nr, nc, 10, eps);
        c.copy_low_rank_approximation(nr, nc, 0, 0, 10, fixed);
        LowRankMatrix<double> autol(nr, nc, eps);
        bool ok = c.copy_low_rank_approximation(nr, nc, 0, 0, autol);
        D.f64p(name + "_fixed_U", fixed.get_U().data(), {(uint64_t)fixed.get_U().nb_cols(), (uint64_t)nr});
        D.f64p(name + "_fixed_V", fixed.get_V().data(), {(uint64_t)nc, (uint64_t)fixed.get_V().nb_rows()});
        D.f64p(name + "_auto_U", autol.get_U().data(), {(uint64_t)autol.get_U().nb_cols(), (uint64_t)nr});
        D.f64p(name + "_auto_V", autol.get_V().data(), {(uint64_t)nc, (uint64_t)autol.get_V().nb_rows()});
        D.i32(name + "_info", {fixed.rank_of(), autol.rank_of(), ok ? 1 : 0});
        printf("%s: fixed rank %d, auto rank %d ok=%d\n", name.c_str(), fixed.rank_of(), autol.rank_of(), (int)ok);
    };
    run("partialACA", partialACA<double>(A, t.get_permutation().data(), s.get_permutation().data()));
    run("sympartialACA", sympartialACA<double>(A, t.get_permutation().data(), s.get_permutation().data()));
    run("fullACA", fullACA<double>(A, t.get_permutation().data(), s.get_permutation().data()));
    run("SVD", SVD<double>(A, t.get_permutation().data(), s.get_permutation().data()));
    return 0;
}

int main(int argc, char **argv) {
    auto kv          = parse(argc, argv);
    std::string mode = gets(kv, "mode", "hmat");
    if (mode == "hmat")
    {
        const std::string prec = gets(kv, "prec", "f64");
        if (prec == "f32")
            return run_hmat<float>(kv);
        if (prec == "z64")
            return run_hmat<std::complex<double>>(kv);
        if (prec == "c32")
            return run_hmat<std::complex<float>>(kv);
        return run_hmat<double>(kv);
    }
    if (mode == "lrmat")
        return run_lrmat(kv);
    fprintf(stderr, "unknown mode\n");
    return 2;
}

// Test program (dev container only: needs the htool headers): builds an htool cluster tree + H-matrix on the
// CPU and checks that the adaptor's hmx structure mirrors it leaf for leaf.  No GPU: the device step is expected
// to report an error through htool's Logger and return false.
#include <htool/clustering/tree_builder/tree_builder.hpp>
#include <htool/hmatrix/tree_builder/tree_builder.hpp>
#include <htool/hmatrix/lrmat/partialACA.hpp>
#include <htool/testing/geometry.hpp>
#include "hmx/htool_adaptor.hpp"
#include <complex>
#include <cstdio>
using namespace htool;
class Gen : public VirtualGenerator<double> {
    const std::vector<double> &x;
  public:
    explicit Gen(const std::vector<double> &x_) : x(x_) {}
    void copy_submatrix(int M, int N, const int *rows, const int *cols, double *ptr) const override {
        for (int j = 0; j < M; j++)
            for (int k = 0; k < N; k++) {
                double s = 0;
                for (int p = 0; p < 3; p++) { double d = x[3 * rows[j] + p] - x[3 * cols[k] + p]; s = s + d * d; }
                ptr[j + (size_t)M * k] = 1. / (1e-5 + std::sqrt(s));
            }
    }
};
// complex instantiation: Hermitian htool H-matrix mirrored leaf for leaf; the plugin classes must be constructible
// (i.e. implement every pure virtual of htool's interfaces for std::complex<double>)
class GenZ : public VirtualGenerator<std::complex<double>> {
    const std::vector<double> &x;
  public:
    explicit GenZ(const std::vector<double> &x_) : x(x_) {}
    void copy_submatrix(int M, int N, const int *rows, const int *cols, std::complex<double> *ptr) const override {
        for (int j = 0; j < M; j++)
            for (int k = 0; k < N; k++) {
                double s = 0;
                for (int p = 0; p < 3; p++) { double d = x[3 * rows[j] + p] - x[3 * cols[k] + p]; s = s + d * d; }
                const double u = x[3 * rows[j]] - x[3 * cols[k]];
                ptr[j + (size_t)M * k] = std::complex<double>(1., u > 0 ? 1. : (u < 0 ? -1. : 0.)) / (1e-5 + std::sqrt(s));
            }
    }
};
static int check_complex() {
    using Z = std::complex<double>;
    const int n = 1500;
    std::vector<double> x(3 * n);
    create_sphere(n, x.data());
    ClusterTreeBuilder<double> ctb;
    ctb.set_maximal_leaf_size(50);
    Cluster<double> T = ctb.create_cluster_tree(n, 3, x.data(), 2, 2);
    GenZ A(x);
    HMatrixTreeBuilder<Z> tb(1e-3, 10., 'H', 'U');
    HMatrix<Z> H = tb.sequential_build(A, T, T);
    hmx_htool::ClusterOptions opt;
    opt.maximal_leaf_size = 50; opt.number_of_children = 2; opt.size_of_partition = 2;
    hmx_htool::EngineT<Z> E(T, n, x.data(), T, n, x.data(), 3, opt);
    bool device = E.setup_block_tree(10., 'H', 'U', 0, 0, -1, -1, 0);
    hmx_htool::DeviceLowRankGeneratorT<Z> lrgen(E);
    hmx_htool::DeviceDenseBlocksGeneratorT<Z> dgen(E);
    hmx_htool::GlobalToLocalHmxT<Z> op(E, n);
    (void)lrgen; (void)dgen; (void)op;
    size_t nleaves = 0, missing = 0;
    std::vector<const HMatrix<Z> *> st{&H};
    while (!st.empty()) {
        auto *c = st.back(); st.pop_back();
        if (c->is_leaf()) {
            nleaves++;
            if (E.find_leaf(c->get_target_cluster().get_offset(), c->get_target_cluster().get_size(), c->get_source_cluster().get_offset(), c->get_source_cluster().get_size()) < 0) missing++;
        }
        for (auto &ch : c->get_children()) st.push_back(ch.get());
    }
    std::printf("complex: htool_leaves=%zu hmx_leaves=%zu missing=%zu device=%d\n", nleaves, E.number_of_leaves(), missing, (int)device);
    return (missing == 0 && nleaves == E.number_of_leaves()) ? 0 : 1;
}
// a user admissibility condition (twice as strict as Rjasanow-Steinbach) given to both htool's builder and the adaptor
class StrictCondition final : public VirtualAdmissibilityCondition<double> {
  public:
    mutable long calls = 0;
    bool ComputeAdmissibility(const Cluster<double> &t, const Cluster<double> &s, double eta) const override {
        calls++;
        return 2 * std::min(t.get_radius(), s.get_radius()) < 0.5 * eta * std::max(norm2(t.get_center() - s.get_center()) - t.get_radius() - s.get_radius(), 0.);
    }
};
static int check_user_admissibility() {
    const int n = 2500;
    std::vector<double> x(3 * n);
    create_sphere(n, x.data());
    ClusterTreeBuilder<double> ctb;
    ctb.set_maximal_leaf_size(40);
    Cluster<double> T = ctb.create_cluster_tree(n, 3, x.data(), 2, 2);
    Gen A(x);
    HMatrixTreeBuilder<double> tb(1e-3, 10., 'N', 'N');
    HMatrix<double> Hdefault = tb.sequential_build(A, T, T);
    auto cond = std::make_shared<StrictCondition>();
    tb.set_admissibility_condition(cond);
    HMatrix<double> H = tb.sequential_build(A, T, T);
    hmx_htool::ClusterOptions opt;
    opt.maximal_leaf_size = 40; opt.number_of_children = 2; opt.size_of_partition = 2;
    hmx_htool::Engine E(T, n, x.data(), T, n, x.data(), 3, opt);
    StrictCondition mine;
    E.setup_block_tree(10., 'N', 'N', 0, 0, -1, -1, 0, &mine);
    size_t nleaves = 0, ndefault = 0, missing = 0;
    std::vector<const HMatrix<double> *> st{&H};
    while (!st.empty()) {
        auto *c = st.back(); st.pop_back();
        if (c->is_leaf()) {
            nleaves++;
            int64_t leaf = E.find_leaf(c->get_target_cluster().get_offset(), c->get_target_cluster().get_size(), c->get_source_cluster().get_offset(), c->get_source_cluster().get_size());
            if (leaf < 0 || (c->is_low_rank() && !E.leaf_is_admissible(leaf))) missing++;
        }
        for (auto &ch : c->get_children()) st.push_back(ch.get());
    }
    st = {&Hdefault};
    while (!st.empty()) {
        auto *c = st.back(); st.pop_back();
        ndefault += c->is_leaf();
        for (auto &ch : c->get_children()) st.push_back(ch.get());
    }
    std::printf("user admissibility: htool_leaves=%zu (default condition %zu) hmx_leaves=%zu missing=%zu calls=%ld\n", nleaves, ndefault, E.number_of_leaves(), missing, mine.calls);
    return (missing == 0 && nleaves == E.number_of_leaves() && nleaves != ndefault && mine.calls > 0) ? 0 : 1;
}
// a user's partitioning strategy hmx has no builder for: cut the current cluster along the coordinate axis (depth mod 3) at the 40 %
// quantile (uneven children).  The adaptor must take htool's tree as it is.
class SkewedAxisPartitioning final : public VirtualPartitioning<double> {
  public:
    std::vector<std::pair<int, int>> compute_partitioning(Cluster<double> &c, int dim, const double *x, const double *, const double *, int nparts) override {
        auto &perm     = c.get_permutation();
        const int off = c.get_offset(), size = c.get_size(), axis = c.get_depth() % dim;
        std::stable_sort(perm.begin() + off, perm.begin() + off + size, [&](int a, int b) { return x[dim * a + axis] < x[dim * b + axis]; });
        std::vector<std::pair<int, int>> parts;
        int pos = off;
        for (int p = 0; p < nparts; p++) {
            int len = p == nparts - 1 ? off + size - pos : std::max(1, (int)(0.4 * (off + size - pos)));
            parts.emplace_back(pos, len);
            pos += len;
        }
        return parts;
    }
};
static int check_imported_tree() {
    const int n = 2200;
    std::vector<double> x(3 * n);
    create_sphere(n, x.data());
    ClusterTreeBuilder<double> ctb;
    ctb.set_maximal_leaf_size(45);
    ctb.set_partitioning_strategy(std::make_shared<SkewedAxisPartitioning>());
    Cluster<double> T = ctb.create_cluster_tree(n, 3, x.data(), 2, 2);
    Gen A(x);
    HMatrixTreeBuilder<double> tb(1e-3, 10., 'N', 'N');
    HMatrix<double> H = tb.sequential_build(A, T, T);
    hmx_htool::Engine E(T, T, 3); // no options, no coordinates: the tree is imported node for node
    E.setup_block_tree(10., 'N', 'N', 0, 0, -1, -1, 0);
    size_t nleaves = 0, missing = 0;
    std::vector<const HMatrix<double> *> st{&H};
    while (!st.empty()) {
        auto *c = st.back(); st.pop_back();
        if (c->is_leaf()) {
            nleaves++;
            int64_t leaf = E.find_leaf(c->get_target_cluster().get_offset(), c->get_target_cluster().get_size(), c->get_source_cluster().get_offset(), c->get_source_cluster().get_size());
            if (leaf < 0 || (c->is_low_rank() != E.leaf_is_admissible(leaf) && c->is_low_rank())) missing++;
        }
        for (auto &ch : c->get_children()) st.push_back(ch.get());
    }
    // the row-restricted block tree of partition 1 on the imported tree as well
    HMatrix<double> H1 = tb.sequential_build(A, T, T, 1, 1);
    hmx_htool::Engine E1(T, T, 3);
    E1.setup_block_tree(10., 'N', 'N', 0, 0, 1, 1, 0);
    size_t n1 = 0, missing1 = 0;
    st = {&H1};
    while (!st.empty()) {
        auto *c = st.back(); st.pop_back();
        if (c->is_leaf()) {
            n1++;
            if (E1.find_leaf(c->get_target_cluster().get_offset(), c->get_target_cluster().get_size(), c->get_source_cluster().get_offset(), c->get_source_cluster().get_size()) < 0) missing1++;
        }
        for (auto &ch : c->get_children()) st.push_back(ch.get());
    }
    std::printf("imported tree (user partitioning): htool_leaves=%zu hmx_leaves=%zu missing=%zu | partition 1: htool_leaves=%zu hmx_leaves=%zu missing=%zu\n", nleaves, E.number_of_leaves(), missing, n1, E1.number_of_leaves(), missing1);
    return (missing == 0 && nleaves == E.number_of_leaves() && missing1 == 0 && n1 == E1.number_of_leaves()) ? 0 : 1;
}
int main() {
    if (check_imported_tree() != 0)
        return 4;
    if (check_complex() != 0)
        return 2;
    if (check_user_admissibility() != 0)
        return 3;
    const int n = 3000;
    std::vector<double> x(3 * n);
    create_rotated_ellipse(3, 4., 1., 0., 0., n, x.data());
    ClusterTreeBuilder<double> ctb;
    ctb.set_maximal_leaf_size(100);
    Cluster<double> T = ctb.create_cluster_tree(n, 3, x.data(), 2, 2);
    Gen A(x);
    HMatrixTreeBuilder<double> tb(1e-4, 10., 'S', 'L');
    HMatrix<double> H = tb.sequential_build(A, T, T);
    hmx_htool::ClusterOptions opt;
    opt.maximal_leaf_size = 100; opt.number_of_children = 2; opt.size_of_partition = 2;
    hmx_htool::Engine E(T, n, x.data(), T, n, x.data(), 3, opt);
    bool device = E.setup_block_tree(10., 'S', 'L', 0, 0, -1, -1, 0);
    size_t nleaves = 0, missing = 0;
    std::vector<const HMatrix<double> *> st{&H};
    while (!st.empty()) {
        auto *c = st.back(); st.pop_back();
        if (c->is_leaf()) {
            nleaves++;
            if (E.find_leaf(c->get_target_cluster().get_offset(), c->get_target_cluster().get_size(), c->get_source_cluster().get_offset(), c->get_source_cluster().get_size()) < 0) missing++;
        }
        for (auto &ch : c->get_children()) st.push_back(ch.get());
    }
    std::printf("htool_leaves=%zu hmx_leaves=%zu missing=%zu device=%d\n", nleaves, E.number_of_leaves(), missing, (int)device);
    return (missing == 0 && nleaves == E.number_of_leaves()) ? 0 : 1;
}

// htool_adaptor.hpp -- header-only C++14 adaptor that plugs libhmx (C ABI: include/hmx.h) into htool's
// own plugin surface.  Compiled only where the htool headers are available (-I<htool>/include); nothing
// from htool is copied here.  See INTEGRATION.md for the call sites a maintainer would touch.
//
//   hmx_htool::Engine                  owns the hmx cluster trees / block tree / device H-matrix that mirror an
//                                      htool cluster pair (verified bit-exact against htool's permutation)
//   hmx_htool::DeviceLowRankGenerator  htool::VirtualInternalLowRankGenerator<double>: htool's build loop
//                                      (HMatrix::compute_low_rank_data, hmatrix/hmatrix.hpp:228-237) receives blocks
//                                      that were compressed on the GPU in one batched launch
//   hmx_htool::GlobalToLocalHmx        htool::VirtualGlobalToLocalOperator<double>: registered with
//                                      DistributedOperator::add_global_to_local_operator
//                                      (distributed_operator/distributed_operator.hpp:47-53); products run on the GPU
//   Engine::upload(const HMatrix&)     htool-compressed leaves (any VirtualGenerator / compressor) -> device streams
// Every class is a template on htool's CoefficientPrecision T in {double, float, std::complex<double>, std::complex<float>}
// (EngineT<T>, DeviceLowRankGeneratorT<T>, ...); the un-suffixed names are the double instantiations.
#ifndef HMX_HTOOL_ADAPTOR_HPP
#define HMX_HTOOL_ADAPTOR_HPP

#include <htool/clustering/cluster_node.hpp>
#include <htool/distributed_operator/interfaces/virtual_global_to_local_operator.hpp>
#include <htool/distributed_operator/interfaces/virtual_local_to_local_operator.hpp>
#include <htool/hmatrix/hmatrix.hpp>
#include <htool/hmatrix/interfaces/virtual_dense_blocks_generator.hpp>
#include <htool/hmatrix/interfaces/virtual_generator.hpp>
#include <htool/hmatrix/interfaces/virtual_lrmat_generator.hpp>
#include <htool/misc/logger.hpp>

#include <algorithm>
#include <chrono>
#include <complex>
#include <cstdlib>
#include <memory>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../../include/hmx.h"

namespace hmx_htool {

inline bool ok(int rc, const char *what) {
    if (rc != HMX_OK) { // htool's convention: log and continue (misc/logger.hpp:74-76)
        htool::Logger::get_instance().log(htool::LogLevel::ERROR, std::string("[hmx] ") + what + ": " + hmx_last_error());
        return false;
    }
    return true;
}

// The C entry points of one coefficient type (include/hmx.h: plain, _s, _z, _c).  Complex values cross the ABI as interleaved
// (re, im) pairs, which is the layout of std::complex<T>.
template <typename T>
struct Abi;
template <>
struct Abi<double> {
    using R = double;
    static int create(const hmx_block_tree *bt, int dev, hmx_hmatrix **h) { return hmx_hmatrix_create(bt, dev, h); }
    static int set_callback(hmx_hmatrix *h, void (*fn)(void *, int, int, const int32_t *, const int32_t *, R *), void *u) { return hmx_hmatrix_set_callback(h, fn, u); }
    static int set_lowrank(hmx_hmatrix *h, int64_t leaf, int r, const double *U, const double *V) { return hmx_hmatrix_set_block_lowrank(h, leaf, r, U, V); }
    static int set_dense(hmx_hmatrix *h, int64_t leaf, const double *D) { return hmx_hmatrix_set_block_dense(h, leaf, D); }
    static int get_block(const hmx_hmatrix *h, int64_t leaf, double *U, double *V) { return hmx_hmatrix_get_block(h, leaf, U, V); }
    static int get_blocks(const hmx_hmatrix *h, int64_t n, const int64_t *leaves, double *const *U, double *const *V) { return hmx_hmatrix_get_blocks(h, n, leaves, U, V); }
    static int matvec(hmx_hmatrix *h, char tr, double a, const double *in, double b, double *out, int mem = HMX_MEM_HOST, void *stream = nullptr) { return hmx_hmatrix_matvec(h, tr, a, in, b, out, mem, stream); }
    static int matmat(hmx_hmatrix *h, char tr, double a, const double *in, double b, double *out, int mu, int mem = HMX_MEM_HOST, void *stream = nullptr) { return hmx_hmatrix_matmat_row_major(h, tr, a, in, b, out, mu, mem, stream); }
};
template <>
struct Abi<float> {
    using R = float;
    static int create(const hmx_block_tree *bt, int dev, hmx_hmatrix **h) { return hmx_hmatrix_create_s(bt, dev, h); }
    static int set_callback(hmx_hmatrix *h, void (*fn)(void *, int, int, const int32_t *, const int32_t *, R *), void *u) { return hmx_hmatrix_set_callback_s(h, fn, u); }
    static int set_lowrank(hmx_hmatrix *h, int64_t leaf, int r, const float *U, const float *V) { return hmx_hmatrix_set_block_lowrank_s(h, leaf, r, U, V); }
    static int set_dense(hmx_hmatrix *h, int64_t leaf, const float *D) { return hmx_hmatrix_set_block_dense_s(h, leaf, D); }
    static int get_block(const hmx_hmatrix *h, int64_t leaf, float *U, float *V) { return hmx_hmatrix_get_block_s(h, leaf, U, V); }
    static int get_blocks(const hmx_hmatrix *h, int64_t n, const int64_t *leaves, float *const *U, float *const *V) { return hmx_hmatrix_get_blocks_s(h, n, leaves, U, V); }
    static int matvec(hmx_hmatrix *h, char tr, float a, const float *in, float b, float *out, int mem = HMX_MEM_HOST, void *stream = nullptr) { return hmx_hmatrix_matvec_s(h, tr, a, in, b, out, mem, stream); }
    static int matmat(hmx_hmatrix *h, char tr, float a, const float *in, float b, float *out, int mu, int mem = HMX_MEM_HOST, void *stream = nullptr) { return hmx_hmatrix_matmat_row_major_s(h, tr, a, in, b, out, mu, mem, stream); }
};
template <>
struct Abi<std::complex<double>> {
    using R = double;
    using Z = std::complex<double>;
    static const double *p(const Z *v) { return reinterpret_cast<const double *>(v); }
    static double *p(Z *v) { return reinterpret_cast<double *>(v); }
    static int create(const hmx_block_tree *bt, int dev, hmx_hmatrix **h) { return hmx_hmatrix_create_z(bt, dev, h); }
    static int set_callback(hmx_hmatrix *h, void (*fn)(void *, int, int, const int32_t *, const int32_t *, R *), void *u) { return hmx_hmatrix_set_callback_z(h, fn, u); }
    static int set_lowrank(hmx_hmatrix *h, int64_t leaf, int r, const Z *U, const Z *V) { return hmx_hmatrix_set_block_lowrank_z(h, leaf, r, p(U), p(V)); }
    static int set_dense(hmx_hmatrix *h, int64_t leaf, const Z *D) { return hmx_hmatrix_set_block_dense_z(h, leaf, p(D)); }
    static int get_block(const hmx_hmatrix *h, int64_t leaf, Z *U, Z *V) { return hmx_hmatrix_get_block_z(h, leaf, p(U), p(V)); }
    static int get_blocks(const hmx_hmatrix *h, int64_t n, const int64_t *leaves, Z *const *U, Z *const *V) { return hmx_hmatrix_get_blocks_z(h, n, leaves, reinterpret_cast<double *const *>(U), reinterpret_cast<double *const *>(V)); }
    static int matvec(hmx_hmatrix *h, char tr, Z a, const Z *in, Z b, Z *out, int mem = HMX_MEM_HOST, void *stream = nullptr) { return hmx_hmatrix_matvec_z(h, tr, p(&a), p(in), p(&b), p(out), mem, stream); }
    static int matmat(hmx_hmatrix *h, char tr, Z a, const Z *in, Z b, Z *out, int mu, int mem = HMX_MEM_HOST, void *stream = nullptr) { return hmx_hmatrix_matmat_row_major_z(h, tr, p(&a), p(in), p(&b), p(out), mu, mem, stream); }
};
template <>
struct Abi<std::complex<float>> {
    using R = float;
    using Z = std::complex<float>;
    static const float *p(const Z *v) { return reinterpret_cast<const float *>(v); }
    static float *p(Z *v) { return reinterpret_cast<float *>(v); }
    static int create(const hmx_block_tree *bt, int dev, hmx_hmatrix **h) { return hmx_hmatrix_create_c(bt, dev, h); }
    static int set_callback(hmx_hmatrix *h, void (*fn)(void *, int, int, const int32_t *, const int32_t *, R *), void *u) { return hmx_hmatrix_set_callback_c(h, fn, u); }
    static int set_lowrank(hmx_hmatrix *h, int64_t leaf, int r, const Z *U, const Z *V) { return hmx_hmatrix_set_block_lowrank_c(h, leaf, r, p(U), p(V)); }
    static int set_dense(hmx_hmatrix *h, int64_t leaf, const Z *D) { return hmx_hmatrix_set_block_dense_c(h, leaf, p(D)); }
    static int get_block(const hmx_hmatrix *h, int64_t leaf, Z *U, Z *V) { return hmx_hmatrix_get_block_c(h, leaf, p(U), p(V)); }
    static int get_blocks(const hmx_hmatrix *h, int64_t n, const int64_t *leaves, Z *const *U, Z *const *V) { return hmx_hmatrix_get_blocks_c(h, n, leaves, reinterpret_cast<float *const *>(U), reinterpret_cast<float *const *>(V)); }
    static int matvec(hmx_hmatrix *h, char tr, Z a, const Z *in, Z b, Z *out, int mem = HMX_MEM_HOST, void *stream = nullptr) { return hmx_hmatrix_matvec_c(h, tr, p(&a), p(in), p(&b), p(out), mem, stream); }
    static int matmat(hmx_hmatrix *h, char tr, Z a, const Z *in, Z b, Z *out, int mu, int mem = HMX_MEM_HOST, void *stream = nullptr) { return hmx_hmatrix_matmat_row_major_c(h, tr, p(&a), p(in), p(&b), p(out), mu, mem, stream); }
};

struct ClusterOptions { // the arguments the caller gave to htool's ClusterTreeBuilder
    int maximal_leaf_size  = 10;
    int number_of_children = 2;
    int size_of_partition  = 1;
    int direction          = HMX_DIR_LARGEST_EXTENT;
    int splitting          = HMX_SPLIT_REGULAR;
    bool partitioning_n    = false;
};

template <typename T>
class EngineT {
    hmx_cluster_tree *m_target = nullptr, *m_source = nullptr;
    hmx_block_tree *m_block_tree = nullptr;
    hmx_hmatrix *m_hmatrix       = nullptr;
    bool m_square                = false;
    std::map<std::tuple<int, int, int, int>, int64_t> m_leaf_of; // (t_off, t_size, s_off, s_size) -> leaf
    std::vector<hmx_leaf> m_leaves;
    const htool::Cluster<double> *m_htool_target = nullptr, *m_htool_source = nullptr;

    // a user's htool::VirtualAdmissibilityCondition, called on htool's own Cluster nodes from the hmx block-tree recursion
    struct AdmissibilityBridge {
        const htool::VirtualAdmissibilityCondition<double> *condition = nullptr;
        std::map<std::tuple<int, int, int>, const htool::Cluster<double> *> target, source; // (depth, offset, size) -> node
        static void index(const htool::Cluster<double> &root, std::map<std::tuple<int, int, int>, const htool::Cluster<double> *> &out) {
            std::vector<const htool::Cluster<double> *> stack{&root};
            while (!stack.empty()) {
                const htool::Cluster<double> *c = stack.back();
                stack.pop_back();
                out[std::make_tuple(c->get_depth(), c->get_offset(), c->get_size())] = c;
                for (const auto &child : c->get_children())
                    stack.push_back(child.get());
            }
        }
        static int call(void *user, const hmx_cluster_node *t, const hmx_cluster_node *s, double eta) {
            const auto *self = static_cast<const AdmissibilityBridge *>(user);
            const auto a = self->target.find(std::make_tuple(t->depth, t->offset, t->size)), b = self->source.find(std::make_tuple(s->depth, s->offset, s->size));
            if (a == self->target.end() || b == self->source.end()) {
                htool::Logger::get_instance().log(htool::LogLevel::ERROR, "[hmx] admissibility bridge: cluster not found in htool's tree");
                return 0;
            }
            return self->condition->ComputeAdmissibility(*a->second, *b->second, eta) ? 1 : 0;
        }
    };

    static hmx_cluster_tree *make_tree(int n, int dim, const double *x, const ClusterOptions &o, const htool::Cluster<double> &check) {
        hmx_cluster_tree *t = nullptr;
        if (!ok(hmx_cluster_tree_create(n, dim, x, nullptr, nullptr, o.maximal_leaf_size, o.number_of_children, o.size_of_partition, o.direction, o.splitting, o.partitioning_n, &t), "cluster tree"))
            return nullptr;
        const int32_t *p = hmx_cluster_tree_permutation(t);
        if (!std::equal(p, p + n, check.get_permutation().begin()))
            htool::Logger::get_instance().log(htool::LogLevel::ERROR, "[hmx] cluster permutation differs from htool's: options do not match the ClusterTreeBuilder that built the htool cluster");
        return t;
    }

    // htool's own Cluster -> hmx_cluster_tree, node for node (preorder walk): works for ANY tree, whatever built it -- a user's
    // VirtualPartitioning (clustering/interfaces/virtual_partitioning.hpp:9-14), trees read from disk, given partitions
    static hmx_cluster_tree *import_tree(const htool::Cluster<double> &root, int dim) {
        std::vector<hmx_cluster_node> nodes;
        std::map<const htool::Cluster<double> *, int32_t> index;
        std::vector<const htool::Cluster<double> *> stack{&root};
        while (!stack.empty()) { // preorder: a node, then its children's subtrees in order
            const htool::Cluster<double> *c = stack.back();
            stack.pop_back();
            index[c] = (int32_t)nodes.size();
            hmx_cluster_node nd;
            nd.depth = c->get_depth(), nd.offset = c->get_offset(), nd.size = c->get_size(), nd.rank = c->get_rank(), nd.counter = c->get_counter();
            nd.n_children = (int32_t)c->get_children().size();
            nd.radius     = c->get_radius();
            for (int p = 0; p < 3; p++)
                nd.center[p] = p < (int)c->get_center().size() ? c->get_center()[p] : 0.;
            nodes.push_back(nd);
            for (size_t k = c->get_children().size(); k-- > 0;)
                stack.push_back(c->get_children()[k].get());
        }
        std::vector<int32_t> parts;
        for (const auto *c : root.get_clusters_on_partition())
            parts.push_back(index.at(c));
        std::vector<int32_t> perm(root.get_permutation().begin(), root.get_permutation().end());
        hmx_cluster_tree *t = nullptr;
        if (!ok(hmx_cluster_tree_from_nodes(root.get_size(), dim, perm.data(), (int)nodes.size(), nodes.data(), (int)parts.size(), parts.data(), root.get_maximal_leaf_size(), root.is_permutation_local() ? 1 : 0, &t), "cluster tree import"))
            return nullptr;
        return t;
    }

  public:
    // htool's cluster trees are taken as they are (no rebuild, no options to match)
    EngineT(const htool::Cluster<double> &target, const htool::Cluster<double> &source, int dim) {
        m_square       = (&target == &source);
        m_htool_target = &target;
        m_htool_source = &source;
        m_target       = import_tree(target, dim);
        m_source       = m_square ? m_target : import_tree(source, dim);
    }
    // the tree is REBUILT by hmx's own cluster-tree builder from the coordinates and the options the caller gave htool's
    // ClusterTreeBuilder (an error is logged when the permutations differ)
    EngineT(const htool::Cluster<double> &target, int nt, const double *xt, const htool::Cluster<double> &source, int ns, const double *xs, int dim, const ClusterOptions &opt) {
        m_square       = (&target == &source);
        m_htool_target = &target;
        m_htool_source = &source;
        m_target       = make_tree(nt, dim, xt, opt, target);
        m_source = m_square ? m_target : make_tree(ns, dim, xs, opt, source);
    }
    ~EngineT() {
        hmx_hmatrix_destroy(m_hmatrix);
        hmx_block_tree_destroy(m_block_tree);
        if (!m_square)
            hmx_cluster_tree_destroy(m_source);
        hmx_cluster_tree_destroy(m_target);
    }
    EngineT(const EngineT &)            = delete;
    EngineT &operator=(const EngineT &) = delete;

    // same arguments as HMatrixTreeBuilder's constructor + build() (hmatrix/tree_builder/tree_builder.hpp:180-210)
    // `condition`: what the caller gave to HMatrixTreeBuilder::set_admissibility_condition (tree_builder.hpp:243-246), NULL for the
    // default Rjasanow-Steinbach condition
    bool setup_block_tree(double eta, char symmetry, char UPLO, int min_target_depth, int min_source_depth, int target_partition_number, int partition_number_for_symmetry, int device, const htool::VirtualAdmissibilityCondition<double> *condition = nullptr) {
        AdmissibilityBridge bridge;
        if (condition) {
            bridge.condition = condition;
            AdmissibilityBridge::index(*m_htool_target, bridge.target);
            AdmissibilityBridge::index(*m_htool_source, bridge.source);
        }
        if (!ok(hmx_block_tree_create_adm(m_target, m_source, eta, symmetry, UPLO, min_target_depth, min_source_depth, target_partition_number, partition_number_for_symmetry, 1, condition ? &AdmissibilityBridge::call : nullptr, &bridge, &m_block_tree), "block tree"))
            return false;
        m_leaves.resize(hmx_block_tree_num_leaves(m_block_tree));
        hmx_block_tree_leaves(m_block_tree, m_leaves.data());
        for (size_t b = 0; b < m_leaves.size(); b++)
            m_leaf_of[std::make_tuple(m_leaves[b].t_offset, m_leaves[b].t_size, m_leaves[b].s_offset, m_leaves[b].s_size)] = (int64_t)b;
        return ok(Abi<T>::create(m_block_tree, device, &m_hmatrix), "device H-matrix"); // needs a GPU: no CPU path
    }

    // block tree rooted at (target partition, source partition): the block-diagonal / local-to-local operator
    // (DefaultLocalApproximationBuilder, distributed_operator/utility.hpp:64-88)
    bool setup_local_block_tree(double eta, char symmetry, char UPLO, int min_target_depth, int min_source_depth, int target_partition, int source_partition, int device, const htool::VirtualAdmissibilityCondition<double> *condition = nullptr) {
        AdmissibilityBridge bridge;
        if (condition) {
            bridge.condition = condition;
            AdmissibilityBridge::index(*m_htool_target, bridge.target);
            AdmissibilityBridge::index(*m_htool_source, bridge.source);
        }
        if (!ok(hmx_block_tree_create_local_adm(m_target, m_source, eta, symmetry, UPLO, min_target_depth, min_source_depth, target_partition, source_partition, 1, condition ? &AdmissibilityBridge::call : nullptr, &bridge, &m_block_tree), "local block tree"))
            return false;
        m_leaves.resize(hmx_block_tree_num_leaves(m_block_tree));
        hmx_block_tree_leaves(m_block_tree, m_leaves.data());
        for (size_t b = 0; b < m_leaves.size(); b++)
            m_leaf_of[std::make_tuple(m_leaves[b].t_offset, m_leaves[b].t_size, m_leaves[b].s_offset, m_leaves[b].s_size)] = (int64_t)b;
        return ok(Abi<T>::create(m_block_tree, device, &m_hmatrix), "device H-matrix");
    }

    // device compression with a built-in kernel (the generator must be the same function as the user's VirtualGenerator)
    bool compress_on_device(int kernel, const double *params, int nparams, int dim, const double *xt, const double *xs, int compressor, double epsilon, int reqrank) {
        return ok(hmx_hmatrix_set_kernel(m_hmatrix, kernel, params, nparams, dim, xt, xs), "set kernel") && ok(hmx_hmatrix_compress(m_hmatrix, compressor, epsilon, reqrank), "compress");
    }

    // compression on the device with the USER's generator: htool::VirtualGenerator<T>::copy_submatrix is called on
    // the host for one cross row / column per block and ACA iteration (and for the dense leaves); the ACA arithmetic
    // and every later product run on the GPU.  `A` must outlive this call only.
    // `generator_threads`: host threads that may call A.copy_submatrix CONCURRENTLY.  The default follows htool's own build loop
    // (hmatrix/tree_builder/tree_builder.hpp:603-648): it calls the generator from an OpenMP parallel for only when htool is compiled
    // with OpenMP and without HTOOL_WITH_PYTHON_INTERFACE (:606) -- a generator written for such a build is thread-safe by htool's own
    // contract, so all cores are used (0); in every other build the generator has only ever been called from one thread (it may hold
    // the GIL, or simply not be re-entrant), and it stays on the calling thread (1).
#if defined(_OPENMP) && !defined(HTOOL_WITH_PYTHON_INTERFACE)
    static constexpr int default_generator_threads = 0;
#else
    static constexpr int default_generator_threads = 1;
#endif
    bool compress_with_generator(const htool::VirtualGenerator<T> &A, int compressor, double epsilon, int reqrank, int generator_threads = default_generator_threads) {
        auto thunk = [](void *user, int M, int N, const int32_t *rows, const int32_t *cols, typename Abi<T>::R *out) {
            static_cast<const htool::VirtualGenerator<T> *>(user)->copy_submatrix(M, N, rows, cols, reinterpret_cast<T *>(out));
        };
        return ok(Abi<T>::set_callback(m_hmatrix, thunk, const_cast<htool::VirtualGenerator<T> *>(&A)), "set callback") &&
               ok(hmx_hmatrix_set_callback_threads(m_hmatrix, generator_threads), "set callback threads") && ok(hmx_hmatrix_compress(m_hmatrix, compressor, epsilon, reqrank), "compress");
    }

    // htool-built H-matrix (any generator, any compressor) -> device
    bool upload(const htool::HMatrix<T> &H) {
        std::vector<const htool::HMatrix<T> *> stack{&H};
        while (!stack.empty()) {
            const htool::HMatrix<T> *cur = stack.back();
            stack.pop_back();
            if (cur->is_leaf()) {
                auto it = m_leaf_of.find(std::make_tuple(cur->get_target_cluster().get_offset(), cur->get_target_cluster().get_size(), cur->get_source_cluster().get_offset(), cur->get_source_cluster().get_size()));
                if (it == m_leaf_of.end()) {
                    htool::Logger::get_instance().log(htool::LogLevel::ERROR, "[hmx] htool leaf not present in the hmx block tree (builder parameters differ)");
                    return false;
                }
                if (cur->is_low_rank()) {
                    const auto &lr = *cur->get_low_rank_data();
                    if (!ok(Abi<T>::set_lowrank(m_hmatrix, it->second, lr.rank_of(), lr.get_U().data(), lr.get_V().data()), "upload low rank"))
                        return false;
                } else if (cur->is_dense()) {
                    if (!ok(Abi<T>::set_dense(m_hmatrix, it->second, cur->get_dense_data()->data()), "upload dense"))
                        return false;
                }
            }
            for (auto &c : cur->get_children())
                stack.push_back(c.get());
        }
        return ok(hmx_hmatrix_finalize(m_hmatrix), "finalize");
    }

    hmx_hmatrix *hmatrix() const { return m_hmatrix; }
    // root block of the block tree: {target offset, target size, source offset, source size} in cluster numbering
    bool root(int32_t out[4]) const {
        char sym = 'N', uplo = 'N';
        return m_block_tree && ok(hmx_block_tree_root(m_block_tree, out, &sym, &uplo), "block tree root");
    }
    size_t number_of_leaves() const { return m_leaves.size(); }
    const hmx_leaf &leaf(size_t b) const { return m_leaves[b]; }
    bool leaf_is_admissible(int64_t leaf) const { return m_leaves[leaf].admissible != 0; }
    int64_t find_leaf(int row_offset, int M, int col_offset, int N) const {
        auto it = m_leaf_of.find(std::make_tuple(row_offset, M, col_offset, N));
        return it == m_leaf_of.end() ? -1 : it->second;
    }
};
using Engine = EngineT<double>;

// Plugged in with HMatrixTreeBuilder::set_low_rank_generator(std::shared_ptr<VirtualInternalLowRankGenerator>)
// (hmatrix/tree_builder/tree_builder.hpp:251-254).  Called concurrently from OpenMP threads (tree_builder.hpp:606-617).
template <typename T>
class DeviceLowRankGeneratorT final : public htool::VirtualInternalLowRankGenerator<T> {
    const EngineT<T> &m_engine;
    mutable std::mutex m_mutex;
    mutable std::vector<int32_t> m_ranks;
    // every low-rank block of the device operator, downloaded in ONE bulk call (hmx_hmatrix_get_blocks: gathered on the device, a few
    // large copies) when htool's build loop asks for the first one; handed out by memcpy afterwards -- 324 418 admissible blocks at
    // N = 1e6 would otherwise be as many blocking device-to-host copies under a mutex.  Host memory: the factors once more (released
    // by release_cache(), or with the generator).
    struct FreeDeleter {
        void operator()(T *p) const { std::free(p); }
    };
    mutable std::unique_ptr<T[], FreeDeleter> m_cache; // malloc'ed: not zero-filled, the pages are first touched by the download's threads
    mutable std::vector<int64_t> m_cache_off;          // per leaf: first entry of U in m_cache (V follows), -1 = not low rank
    mutable bool m_cached = false, m_cache_ok = false;
    mutable double m_prefetch_seconds = 0;

    void prefetch() const {
        const auto t0 = std::chrono::steady_clock::now();
        const size_t nl = m_engine.number_of_leaves();
        m_ranks.resize(nl + 1);
        hmx_hmatrix_leaf_ranks(m_engine.hmatrix(), m_ranks.data());
        m_cache_off.assign(nl, -1);
        std::vector<int64_t> leaves;
        int64_t total = 0;
        for (size_t b = 0; b < nl; b++)
            if (m_ranks[b] > 0) {
                m_cache_off[b] = total;
                total += (int64_t)m_ranks[b] * ((int64_t)m_engine.leaf(b).t_size + m_engine.leaf(b).s_size);
                leaves.push_back((int64_t)b);
            }
        m_cache.reset(static_cast<T *>(std::malloc(std::max<size_t>((size_t)total, 1) * sizeof(T))));
        if (!m_cache) {
            htool::Logger::get_instance().log(htool::LogLevel::ERROR, "[hmx] out of host memory for the downloaded low-rank blocks");
            m_cache_ok = false;
            m_cached   = true;
            return;
        }
        std::vector<T *> pu(leaves.size()), pv(leaves.size());
        for (size_t k = 0; k < leaves.size(); k++) {
            const int64_t b = leaves[k];
            pu[k]           = m_cache.get() + m_cache_off[b];
            pv[k]           = pu[k] + (int64_t)m_ranks[b] * m_engine.leaf(b).t_size;
        }
        m_cache_ok = leaves.empty() || ok(Abi<T>::get_blocks(m_engine.hmatrix(), (int64_t)leaves.size(), leaves.data(), pu.data(), pv.data()), "get blocks");
        m_cached   = true;
        m_prefetch_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }

  public:
    explicit DeviceLowRankGeneratorT(const EngineT<T> &engine) : m_engine(engine) {}
    void release_cache() const {
        std::lock_guard<std::mutex> lock(m_mutex);
        m_cache.reset();
        m_cached = false;
    }
    double prefetch_seconds() const { return m_prefetch_seconds; } // wall time of the one bulk download

    bool copy_low_rank_approximation(int M, int N, int row_offset, int col_offset, htool::LowRankMatrix<T> &lrmat) const override {
        const int64_t leaf = m_engine.find_leaf(row_offset, M, col_offset, N);
        if (leaf < 0)
            return false; // unknown block -> htool falls back to a dense block (tree_builder.hpp:572-577)
        {
            std::lock_guard<std::mutex> lock(m_mutex); // the first caller downloads everything; the others wait for it here
            if (!m_cached)
                prefetch();
        }
        const int r = m_ranks[leaf];
        if (r <= 0 || !m_cache_ok)
            return false; // compressor failed on the device as well
        lrmat.get_U().resize(M, r);
        lrmat.get_V().resize(r, N);
        const T *src = m_cache.get() + m_cache_off[leaf];
        std::copy_n(src, (size_t)M * r, lrmat.get_U().data());
        std::copy_n(src + (size_t)M * r, (size_t)r * N, lrmat.get_V().data());
        return true;
    }
    bool copy_low_rank_approximation(int M, int N, int row_offset, int col_offset, int, htool::LowRankMatrix<T> &lrmat) const override {
        return copy_low_rank_approximation(M, N, row_offset, col_offset, lrmat); // the device build already used reqrank
    }
};
using DeviceLowRankGenerator = DeviceLowRankGeneratorT<double>;

// The user-numbering flavour, htool::VirtualLowRankGenerator<T> (hmatrix/interfaces/virtual_lrmat_generator.hpp:24-35), plugged in
// with HMatrixTreeBuilder::set_low_rank_generator(std::shared_ptr<VirtualLowRankGenerator>) (tree_builder.hpp:247-250): htool passes
// `rows` / `cols` as pointers INTO the cluster permutation arrays (:49-55), so the block offsets are rows - perm_base.
template <typename T>
class DeviceUserLowRankGeneratorT final : public htool::VirtualLowRankGenerator<T> {
    DeviceLowRankGeneratorT<T> m_internal;
    const int *m_target_permutation, *m_source_permutation;

  public:
    DeviceUserLowRankGeneratorT(const EngineT<T> &engine, const int *target_permutation, const int *source_permutation) : m_internal(engine), m_target_permutation(target_permutation), m_source_permutation(source_permutation) {}
    bool copy_low_rank_approximation(int M, int N, const int *rows, const int *cols, htool::LowRankMatrix<T> &lrmat) const override {
        return m_internal.copy_low_rank_approximation(M, N, (int)(rows - m_target_permutation), (int)(cols - m_source_permutation), lrmat);
    }
    bool copy_low_rank_approximation(int M, int N, const int *rows, const int *cols, int, htool::LowRankMatrix<T> &lrmat) const override {
        return copy_low_rank_approximation(M, N, rows, cols, lrmat);
    }
};
using DeviceUserLowRankGenerator = DeviceUserLowRankGeneratorT<double>;

// Plugged in with HMatrixTreeBuilder::set_dense_blocks_generator (hmatrix/tree_builder/tree_builder.hpp:258):
// htool hands over ALL dense leaves in one batched call (tree_builder.hpp:585-600) with zero-filled destinations;
// they are filled from the blocks the device assembled.
template <typename T>
class DeviceDenseBlocksGeneratorT final : public htool::VirtualDenseBlocksGenerator<T> {
    const EngineT<T> &m_engine;

  public:
    explicit DeviceDenseBlocksGeneratorT(const EngineT<T> &engine) : m_engine(engine) {}
    void copy_dense_blocks(const std::vector<int> &M, const std::vector<int> &N, const std::vector<int> &rows, const std::vector<int> &cols, std::vector<T *> &ptr) const override {
        // one bulk download straight into htool's destinations (hmx_hmatrix_get_blocks)
        std::vector<int64_t> leaves;
        std::vector<T *> dst;
        for (size_t b = 0; b < ptr.size(); b++) {
            const int64_t leaf = m_engine.find_leaf(rows[b], M[b], cols[b], N[b]);
            if (leaf < 0) {
                htool::Logger::get_instance().log(htool::LogLevel::ERROR, "[hmx] dense block not present in the hmx block tree");
                continue;
            }
            leaves.push_back(leaf);
            dst.push_back(ptr[b]);
        }
        if (!leaves.empty())
            ok(Abi<T>::get_blocks(m_engine.hmatrix(), (int64_t)leaves.size(), leaves.data(), dst.data(), nullptr), "get dense blocks");
    }
};
using DeviceDenseBlocksGenerator = DeviceDenseBlocksGeneratorT<double>;

// Same contract as RestrictedGlobalToLocalHMatrix (distributed_operator/implementations/global_to_local_operators/hmatrix.hpp:15-35)
// for a local H-matrix whose source cluster is the whole source tree.
template <typename T>
class GlobalToLocalHmxT final : public htool::VirtualGlobalToLocalOperator<T> {
    const EngineT<T> &m_engine;
    int m_source_size;

  public:
    GlobalToLocalHmxT(const EngineT<T> &engine, int source_size) : m_engine(engine), m_source_size(source_size) {}
    void add_vector_product(char trans, T alpha, const T *const in, T beta, T *const out) const override {
        ok(Abi<T>::matvec(m_engine.hmatrix(), trans, alpha, in, beta, out), "matvec");
    }
    void add_matrix_product_row_major(char trans, T alpha, const T *const in, T beta, T *const out, int mu) const override {
        ok(Abi<T>::matmat(m_engine.hmatrix(), trans, alpha, in, beta, out, mu), "matmat");
    }
    // For callers that already hold their vectors in device memory (a GPU Krylov solver, another HIP library): the same products on
    // DEVICE pointers, enqueued on `stream` (a hipStream_t) -- no host staging, no PCIe copies.  Not part of htool's interface
    // (its contract is host pointers); `in` / `out` have the sizes add_vector_product / add_matrix_product_row_major document.
    bool add_vector_product_device(char trans, T alpha, const T *d_in, T beta, T *d_out, void *stream = nullptr) const {
        return ok(Abi<T>::matvec(m_engine.hmatrix(), trans, alpha, d_in, beta, d_out, HMX_MEM_DEVICE, stream), "matvec (device pointers)");
    }
    bool add_matrix_product_row_major_device(char trans, T alpha, const T *d_in, T beta, T *d_out, int mu, void *stream = nullptr) const {
        return ok(Abi<T>::matmat(m_engine.hmatrix(), trans, alpha, d_in, beta, d_out, mu, HMX_MEM_DEVICE, stream), "matmat (device pointers)");
    }
    void add_sub_matrix_product_to_local(const T *const in, T *const out, int mu, int offset, int size) const override {
        // restricted_operator.hpp:170-193: zero-extend the sub-vector to the whole source range
        std::vector<T> temp((size_t)m_source_size * mu, T(0));
        const int lo = std::max(offset, 0), hi = std::min(offset + size, m_source_size);
        if (hi > lo)
            std::copy_n(in + (size_t)(lo - offset) * mu, (size_t)(hi - lo) * mu, temp.data() + (size_t)lo * mu);
        ok(Abi<T>::matmat(m_engine.hmatrix(), 'N', T(1), temp.data(), T(1), out, mu), "sub matmat");
    }
};
using GlobalToLocalHmx = GlobalToLocalHmxT<double>;

// htool::VirtualLocalToLocalOperator<T> (distributed_operator/interfaces/virtual_local_to_local_operator.hpp:9-35) for an engine whose
// block tree is rooted at (target partition k, source partition k) -- Engine::setup_local_block_tree -- i.e. the block-diagonal
// operator of DefaultLocalApproximationBuilder (distributed_operator/utility.hpp:64-88); registered with
// DistributedOperator::add_local_to_local_operator or CustomApproximationBuilder (utility.hpp:32-34).  Local slices in and out.
template <typename T>
class LocalToLocalHmxT final : public htool::VirtualLocalToLocalOperator<T> {
    const EngineT<T> &m_engine;
    int m_source_offset = 0, m_source_size = 0; // the local source cluster, GLOBAL cluster numbering (the block tree's root)

  public:
    // `local_source_size` is kept for source compatibility; offset and size are taken from the block tree's root
    explicit LocalToLocalHmxT(const EngineT<T> &engine, int local_source_size = -1) : m_engine(engine), m_source_size(local_source_size) {
        int32_t r[4];
        if (engine.root(r)) {
            m_source_offset = r[2];
            m_source_size   = r[3];
        }
    }
    void add_vector_product(char trans, T alpha, const T *const in, T beta, T *const out) const override {
        ok(Abi<T>::matvec(m_engine.hmatrix(), trans, alpha, in, beta, out), "matvec");
    }
    void add_matrix_product_row_major(char trans, T alpha, const T *const in, T beta, T *const out, int mu) const override {
        ok(Abi<T>::matmat(m_engine.hmatrix(), trans, alpha, in, beta, out, mu), "matmat");
    }
    // For callers that already hold their vectors in device memory (a GPU Krylov solver, another HIP library): the same products on
    // DEVICE pointers, enqueued on `stream` (a hipStream_t) -- no host staging, no PCIe copies.  Not part of htool's interface
    // (its contract is host pointers); `in` / `out` have the sizes add_vector_product / add_matrix_product_row_major document.
    bool add_vector_product_device(char trans, T alpha, const T *d_in, T beta, T *d_out, void *stream = nullptr) const {
        return ok(Abi<T>::matvec(m_engine.hmatrix(), trans, alpha, d_in, beta, d_out, HMX_MEM_DEVICE, stream), "matvec (device pointers)");
    }
    bool add_matrix_product_row_major_device(char trans, T alpha, const T *d_in, T beta, T *d_out, int mu, void *stream = nullptr) const {
        return ok(Abi<T>::matmat(m_engine.hmatrix(), trans, alpha, d_in, beta, d_out, mu, HMX_MEM_DEVICE, stream), "matmat (device pointers)");
    }
    // local_to_local_operators/hmatrix.hpp:33-51: `in` holds rows [offset, offset + size) of the GLOBAL source numbering; the part
    // inside the local source cluster is zero-extended to the cluster and multiplied.  (For mu > 1 the reference advances `in` by
    // rows, not rows * mu, :46 -- the row-major meaning of the argument is followed here; both agree for mu = 1.)
    void add_sub_matrix_product_to_local(const T *const in, T *const out, int mu, int offset, int size) const override {
        const int lo = std::max(offset, m_source_offset), hi = std::min(offset + size, m_source_offset + m_source_size);
        if (offset == m_source_offset && hi == m_source_offset + m_source_size) {
            add_matrix_product_row_major('N', T(1), in, T(1), out, mu);
            return;
        }
        if (hi <= lo)
            return;
        std::vector<T> temp((size_t)m_source_size * mu, T(0));
        std::copy_n(in + (size_t)(lo - offset) * mu, (size_t)(hi - lo) * mu, temp.data() + (size_t)(lo - m_source_offset) * mu);
        add_matrix_product_row_major('N', T(1), temp.data(), T(1), out, mu);
    }
};
using LocalToLocalHmx = LocalToLocalHmxT<double>;

} // namespace hmx_htool
#endif

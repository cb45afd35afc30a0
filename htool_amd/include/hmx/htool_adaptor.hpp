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
#ifndef HMX_HTOOL_ADAPTOR_HPP
#define HMX_HTOOL_ADAPTOR_HPP

#include <htool/clustering/cluster_node.hpp>
#include <htool/distributed_operator/interfaces/virtual_global_to_local_operator.hpp>
#include <htool/hmatrix/hmatrix.hpp>
#include <htool/hmatrix/interfaces/virtual_dense_blocks_generator.hpp>
#include <htool/hmatrix/interfaces/virtual_generator.hpp>
#include <htool/hmatrix/interfaces/virtual_lrmat_generator.hpp>
#include <htool/misc/logger.hpp>

#include <algorithm>
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

struct ClusterOptions { // the arguments the caller gave to htool's ClusterTreeBuilder
    int maximal_leaf_size  = 10;
    int number_of_children = 2;
    int size_of_partition  = 1;
    int direction          = HMX_DIR_LARGEST_EXTENT;
    int splitting          = HMX_SPLIT_REGULAR;
    bool partitioning_n    = false;
};

class Engine {
    hmx_cluster_tree *m_target = nullptr, *m_source = nullptr;
    hmx_block_tree *m_block_tree = nullptr;
    hmx_hmatrix *m_hmatrix       = nullptr;
    bool m_square                = false;
    std::map<std::tuple<int, int, int, int>, int64_t> m_leaf_of; // (t_off, t_size, s_off, s_size) -> leaf
    std::vector<hmx_leaf> m_leaves;

    static hmx_cluster_tree *make_tree(int n, int dim, const double *x, const ClusterOptions &o, const htool::Cluster<double> &check) {
        hmx_cluster_tree *t = nullptr;
        if (!ok(hmx_cluster_tree_create(n, dim, x, nullptr, nullptr, o.maximal_leaf_size, o.number_of_children, o.size_of_partition, o.direction, o.splitting, o.partitioning_n, &t), "cluster tree"))
            return nullptr;
        const int32_t *p = hmx_cluster_tree_permutation(t);
        if (!std::equal(p, p + n, check.get_permutation().begin()))
            htool::Logger::get_instance().log(htool::LogLevel::ERROR, "[hmx] cluster permutation differs from htool's: options do not match the ClusterTreeBuilder that built the htool cluster");
        return t;
    }

  public:
    Engine(const htool::Cluster<double> &target, int nt, const double *xt, const htool::Cluster<double> &source, int ns, const double *xs, int dim, const ClusterOptions &opt) {
        m_square = (&target == &source);
        m_target = make_tree(nt, dim, xt, opt, target);
        m_source = m_square ? m_target : make_tree(ns, dim, xs, opt, source);
    }
    ~Engine() {
        hmx_hmatrix_destroy(m_hmatrix);
        hmx_block_tree_destroy(m_block_tree);
        if (!m_square)
            hmx_cluster_tree_destroy(m_source);
        hmx_cluster_tree_destroy(m_target);
    }
    Engine(const Engine &)            = delete;
    Engine &operator=(const Engine &) = delete;

    // same arguments as HMatrixTreeBuilder's constructor + build() (hmatrix/tree_builder/tree_builder.hpp:180-210)
    bool setup_block_tree(double eta, char symmetry, char UPLO, int min_target_depth, int min_source_depth, int target_partition_number, int partition_number_for_symmetry, int device) {
        if (!ok(hmx_block_tree_create(m_target, m_source, eta, symmetry, UPLO, min_target_depth, min_source_depth, target_partition_number, partition_number_for_symmetry, 1, &m_block_tree), "block tree"))
            return false;
        m_leaves.resize(hmx_block_tree_num_leaves(m_block_tree));
        hmx_block_tree_leaves(m_block_tree, m_leaves.data());
        for (size_t b = 0; b < m_leaves.size(); b++)
            m_leaf_of[std::make_tuple(m_leaves[b].t_offset, m_leaves[b].t_size, m_leaves[b].s_offset, m_leaves[b].s_size)] = (int64_t)b;
        return ok(hmx_hmatrix_create(m_block_tree, device, &m_hmatrix), "device H-matrix"); // needs a GPU: no CPU path
    }

    // device compression with a built-in kernel (the generator must be the same function as the user's VirtualGenerator)
    bool compress_on_device(int kernel, const double *params, int nparams, int dim, const double *xt, const double *xs, int compressor, double epsilon, int reqrank) {
        return ok(hmx_hmatrix_set_kernel(m_hmatrix, kernel, params, nparams, dim, xt, xs), "set kernel") && ok(hmx_hmatrix_compress(m_hmatrix, compressor, epsilon, reqrank), "compress");
    }

    // compression on the device with the USER's generator: htool::VirtualGenerator<double>::copy_submatrix is called on
    // the host for one cross row / column per block and ACA iteration (and for the dense leaves); the ACA arithmetic
    // and every later product run on the GPU.  `A` must outlive this call only.
    bool compress_with_generator(const htool::VirtualGenerator<double> &A, int compressor, double epsilon, int reqrank) {
        auto thunk = [](void *user, int M, int N, const int32_t *rows, const int32_t *cols, double *out) {
            static_cast<const htool::VirtualGenerator<double> *>(user)->copy_submatrix(M, N, rows, cols, out);
        };
        return ok(hmx_hmatrix_set_callback(m_hmatrix, thunk, const_cast<htool::VirtualGenerator<double> *>(&A)), "set callback") && ok(hmx_hmatrix_compress(m_hmatrix, compressor, epsilon, reqrank), "compress");
    }

    // htool-built H-matrix (any generator, any compressor) -> device
    bool upload(const htool::HMatrix<double> &H) {
        std::vector<const htool::HMatrix<double> *> stack{&H};
        while (!stack.empty()) {
            const htool::HMatrix<double> *cur = stack.back();
            stack.pop_back();
            if (cur->is_leaf()) {
                auto it = m_leaf_of.find(std::make_tuple(cur->get_target_cluster().get_offset(), cur->get_target_cluster().get_size(), cur->get_source_cluster().get_offset(), cur->get_source_cluster().get_size()));
                if (it == m_leaf_of.end()) {
                    htool::Logger::get_instance().log(htool::LogLevel::ERROR, "[hmx] htool leaf not present in the hmx block tree (builder parameters differ)");
                    return false;
                }
                if (cur->is_low_rank()) {
                    const auto &lr = *cur->get_low_rank_data();
                    if (!ok(hmx_hmatrix_set_block_lowrank(m_hmatrix, it->second, lr.rank_of(), lr.get_U().data(), lr.get_V().data()), "upload low rank"))
                        return false;
                } else if (cur->is_dense()) {
                    if (!ok(hmx_hmatrix_set_block_dense(m_hmatrix, it->second, cur->get_dense_data()->data()), "upload dense"))
                        return false;
                }
            }
            for (auto &c : cur->get_children())
                stack.push_back(c.get());
        }
        return ok(hmx_hmatrix_finalize(m_hmatrix), "finalize");
    }

    hmx_hmatrix *hmatrix() const { return m_hmatrix; }
    size_t number_of_leaves() const { return m_leaves.size(); }
    int64_t find_leaf(int row_offset, int M, int col_offset, int N) const {
        auto it = m_leaf_of.find(std::make_tuple(row_offset, M, col_offset, N));
        return it == m_leaf_of.end() ? -1 : it->second;
    }
};

// Plugged in with HMatrixTreeBuilder::set_low_rank_generator(std::shared_ptr<VirtualInternalLowRankGenerator>)
// (hmatrix/tree_builder/tree_builder.hpp:251-254).  Called concurrently from OpenMP threads (tree_builder.hpp:606-617).
class DeviceLowRankGenerator final : public htool::VirtualInternalLowRankGenerator<double> {
    const Engine &m_engine;
    mutable std::mutex m_mutex;
    mutable std::vector<int32_t> m_ranks;

  public:
    explicit DeviceLowRankGenerator(const Engine &engine) : m_engine(engine) {}

    bool copy_low_rank_approximation(int M, int N, int row_offset, int col_offset, htool::LowRankMatrix<double> &lrmat) const override {
        const int64_t leaf = m_engine.find_leaf(row_offset, M, col_offset, N);
        if (leaf < 0)
            return false; // unknown block -> htool falls back to a dense block (tree_builder.hpp:572-577)
        std::lock_guard<std::mutex> lock(m_mutex);
        if (m_ranks.empty()) {
            hmx_stats st;
            hmx_hmatrix_stats(m_engine.hmatrix(), &st);
            m_ranks.resize(st.n_dense + st.n_lowrank + 1);
            hmx_hmatrix_leaf_ranks(m_engine.hmatrix(), m_ranks.data());
        }
        const int r = m_ranks[leaf];
        if (r <= 0)
            return false; // compressor failed on the device as well
        lrmat.get_U().resize(M, r);
        lrmat.get_V().resize(r, N);
        return ok(hmx_hmatrix_get_block(m_engine.hmatrix(), leaf, lrmat.get_U().data(), lrmat.get_V().data()), "get block");
    }
    bool copy_low_rank_approximation(int M, int N, int row_offset, int col_offset, int, htool::LowRankMatrix<double> &lrmat) const override {
        return copy_low_rank_approximation(M, N, row_offset, col_offset, lrmat); // the device build already used reqrank
    }
};

// Plugged in with HMatrixTreeBuilder::set_dense_blocks_generator (hmatrix/tree_builder/tree_builder.hpp:258):
// htool hands over ALL dense leaves in one batched call (tree_builder.hpp:585-600) with zero-filled destinations;
// they are filled from the blocks the device assembled.
class DeviceDenseBlocksGenerator final : public htool::VirtualDenseBlocksGenerator<double> {
    const Engine &m_engine;

  public:
    explicit DeviceDenseBlocksGenerator(const Engine &engine) : m_engine(engine) {}
    void copy_dense_blocks(const std::vector<int> &M, const std::vector<int> &N, const std::vector<int> &rows, const std::vector<int> &cols, std::vector<double *> &ptr) const override {
        for (size_t b = 0; b < ptr.size(); b++) {
            const int64_t leaf = m_engine.find_leaf(rows[b], M[b], cols[b], N[b]);
            if (leaf < 0) {
                htool::Logger::get_instance().log(htool::LogLevel::ERROR, "[hmx] dense block not present in the hmx block tree");
                continue;
            }
            ok(hmx_hmatrix_get_block(m_engine.hmatrix(), leaf, ptr[b], nullptr), "get dense block");
        }
    }
};

// Same contract as RestrictedGlobalToLocalHMatrix (distributed_operator/implementations/global_to_local_operators/hmatrix.hpp:15-35)
// for a local H-matrix whose source cluster is the whole source tree.
class GlobalToLocalHmx final : public htool::VirtualGlobalToLocalOperator<double> {
    const Engine &m_engine;
    int m_source_size;

  public:
    GlobalToLocalHmx(const Engine &engine, int source_size) : m_engine(engine), m_source_size(source_size) {}
    void add_vector_product(char trans, double alpha, const double *const in, double beta, double *const out) const override {
        ok(hmx_hmatrix_matvec(m_engine.hmatrix(), trans, alpha, in, beta, out, HMX_MEM_HOST, nullptr), "matvec");
    }
    void add_matrix_product_row_major(char trans, double alpha, const double *const in, double beta, double *const out, int mu) const override {
        ok(hmx_hmatrix_matmat_row_major(m_engine.hmatrix(), trans, alpha, in, beta, out, mu, HMX_MEM_HOST, nullptr), "matmat");
    }
    void add_sub_matrix_product_to_local(const double *const in, double *const out, int mu, int offset, int size) const override {
        // restricted_operator.hpp:170-193: zero-extend the sub-vector to the whole source range
        std::vector<double> temp((size_t)m_source_size * mu, 0.0);
        const int lo = std::max(offset, 0), hi = std::min(offset + size, m_source_size);
        if (hi > lo)
            std::copy_n(in + (size_t)(lo - offset) * mu, (size_t)(hi - lo) * mu, temp.data() + (size_t)lo * mu);
        ok(hmx_hmatrix_matmat_row_major(m_engine.hmatrix(), 'N', 1.0, temp.data(), 1.0, out, mu, HMX_MEM_HOST, nullptr), "sub matmat");
    }
};

} // namespace hmx_htool
#endif

// hmx_host.hpp -- host-side structures of libhmx (cluster tree, block tree).  C++14.
//
// The host structure layer reproduces htool's indexing bit-for-bit (permutation, cluster table, leaf
// order) but is laid out for the device engine: flat arrays, no pointer trees, level-parallel build.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/hmx.h"

namespace hmx {

struct ClusterNode {
    int parent      = -1;
    int first_child = -1; // children are contiguous in ClusterTree::nodes
    int n_children  = 0;
    int depth       = 0;
    int offset      = 0;
    int size        = 0;
    int rank        = -1;
    int counter     = 0;
    double radius   = 0;
    double center[3] = {0, 0, 0};
};

struct ClusterTreeOptions {
    int maximal_leaf_size  = 10; // htool default, clustering/tree_builder/tree_builder.hpp:25
    int number_of_children = 2;
    int size_of_partition  = 1;
    int direction          = HMX_DIR_LARGEST_EXTENT;
    int splitting          = HMX_SPLIT_REGULAR;
    bool partitioning_n    = false;
    bool is_complete       = false; // ClusterTreeBuilder::set_is_complete (tree_builder.hpp:26,39,176-192)
};

} // namespace hmx

// The C ABI's opaque types are these structs.
struct hmx_cluster_tree {
    int n = 0, dim = 3;
    hmx::ClusterTreeOptions opt;
    std::vector<int32_t> perm;             // cluster position -> user index
    std::vector<hmx::ClusterNode> nodes;   // nodes[0] = root; children contiguous
    std::vector<int> on_partition;         // node id of partition k
    bool permutation_is_local = false;
    std::vector<int> preorder() const;     // node ids in preorder (children in creation order)
    bool is_leaf(int v) const { return nodes[v].n_children == 0; }
};

struct hmx_block_tree {
    const hmx_cluster_tree *target = nullptr, *source = nullptr;
    double eta = 10;
    char symmetry = 'N', uplo = 'N';
    int min_target_depth = 0, min_source_depth = 0;
    int target_partition = -1, partition_for_symmetry = -1;
    int target_root_partition = -1, source_root_partition = -1; // >= 0: the block tree is rooted at these partition clusters
    bool consistent = true;
    int (*admissibility)(void *, const hmx_cluster_node *, const hmx_cluster_node *, double) = nullptr; // user condition (NULL: Rjasanow-Steinbach)
    void *admissibility_user = nullptr;
    // root after reset_root_of_block_tree
    int root_t_offset = 0, root_t_size = 0, root_s_offset = 0, root_s_size = 0;
    char symmetry_for_leaves = 'N', uplo_for_leaves = 'N';
    std::vector<hmx_leaf> leaves; // htool build order (preorder of the block tree)
};

namespace hmx {
void set_error(const std::string &msg);
// Cores this process may really use: the hardware threads, capped by the cgroup CPU quota (containers: cpu.max = "1600000 100000" means
// 16 cores' worth of time however many threads run; more threads than that only get throttled -- measured on a 256-thread box with that
// quota: 16 threads 6.7 G entries/s, 128 threads 5.1, 256 threads 4.3).  capi_host.cpp.
int host_cores();
// partition_kind: 0 none ("simple" partition from size_of_partition), 1 global (partition[i] = part of point i),
// 2 local (partition[2p], partition[2p+1] = offset, size of part p) -- tree_builder.hpp:87-123
int build_cluster_tree(int n, int dim, const double *coords, const double *radii, const double *weights,
                       const ClusterTreeOptions &opt, hmx_cluster_tree &out, const int32_t *partition = nullptr, int partition_kind = 0);
int build_block_tree(hmx_block_tree &bt);
void make_geometry(const std::string &name, int n, double z, double *coords);
// io.cpp: htool's CSV formats
int save_cluster_tree(const hmx_cluster_tree &T, const std::string &prefix);
int load_cluster_tree(const std::string &props_file, const std::string &tree_file, hmx_cluster_tree &T);
void cluster_tree_depths(const hmx_cluster_tree &T, int &max_depth, int &min_depth);
int save_leaves_with_rank(const std::vector<hmx_leaf> &leaves, const int32_t *rank, int t0, int nt, int s0, int ns, const std::string &name);
} // namespace hmx

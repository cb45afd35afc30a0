// capi_host.cpp -- C ABI entry points that need no GPU: geometry, cluster tree, block tree.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>

#include <algorithm>
#include "hmx_host.hpp"

namespace hmx {
static thread_local std::string g_error;
void set_error(const std::string &msg) { g_error = msg; }
int host_cores() {
    static const int cores = [] {
        int n = (int)std::max(1u, std::thread::hardware_concurrency());
        long long quota = -1, period = -1;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) { // cgroup v2: "<quota|max> <period>"
            char q[64] = {0};
            if (fscanf(f, "%63s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0)
                quota = atoll(q);
            fclose(f);
        } else { // cgroup v1
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
                if (fscanf(g, "%lld", &quota) != 1)
                    quota = -1;
                fclose(g);
            }
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (fscanf(g, "%lld", &period) != 1)
                    period = -1;
                fclose(g);
            }
        }
        if (quota > 0 && period > 0)
            n = std::min<long long>(n, std::max<long long>(1, (quota + period - 1) / period));
        if (const char *e = getenv("HMX_HOST_CORES"))
            if (atoi(e) > 0)
                n = atoi(e);
        return n;
    }();
    return cores;
}
} // namespace hmx

extern "C" {

int hmx_host_cores(void) { return hmx::host_cores(); }


const char *hmx_last_error(void) { return hmx::g_error.c_str(); }

int hmx_geometry(const char *name, int n, double z, double *coords) {
    if (!name || !coords || n < 0) {
        hmx::set_error("hmx_geometry: invalid arguments");
        return HMX_ERR_INVALID;
    }
    const std::string g(name);
    if (g != "ellipse" && g != "disk" && g != "ball" && g != "disk2d") {
        hmx::set_error("hmx_geometry: unknown geometry " + g);
        return HMX_ERR_INVALID;
    }
    hmx::make_geometry(g, n, z, coords);
    return HMX_OK;
}

int hmx_cluster_tree_create(int n, int dim, const double *coords, const double *radii, const double *weights,
                            int maximal_leaf_size, int number_of_children, int size_of_partition, int direction,
                            int splitting, int partitioning_n, hmx_cluster_tree **out) {
    if (!out) {
        hmx::set_error("hmx_cluster_tree_create: out is NULL");
        return HMX_ERR_INVALID;
    }
    hmx::ClusterTreeOptions opt;
    opt.maximal_leaf_size  = maximal_leaf_size;
    opt.number_of_children = number_of_children;
    opt.size_of_partition  = size_of_partition;
    opt.direction          = direction;
    opt.splitting          = splitting;
    opt.partitioning_n     = partitioning_n != 0;
    auto *T                = new (std::nothrow) hmx_cluster_tree();
    if (!T)
        return HMX_ERR_INVALID;
    const int rc = hmx::build_cluster_tree(n, dim, coords, radii, weights, opt, *T);
    if (rc != HMX_OK) {
        delete T;
        return rc;
    }
    *out = T;
    return HMX_OK;
}
int hmx_cluster_tree_create_ex(int n, int dim, const double *coords, const double *radii, const double *weights, int maximal_leaf_size,
                               int number_of_children, int size_of_partition, int direction, int splitting, int partitioning_n,
                               int is_complete, const int32_t *partition, int partition_kind, hmx_cluster_tree **out) {
    if (!out) {
        hmx::set_error("hmx_cluster_tree_create_ex: out is NULL");
        return HMX_ERR_INVALID;
    }
    hmx::ClusterTreeOptions opt;
    opt.maximal_leaf_size  = maximal_leaf_size;
    opt.number_of_children = number_of_children;
    opt.size_of_partition  = size_of_partition;
    opt.direction          = direction;
    opt.splitting          = splitting;
    opt.partitioning_n     = partitioning_n != 0;
    opt.is_complete        = is_complete != 0;
    auto *T                = new (std::nothrow) hmx_cluster_tree();
    if (!T)
        return HMX_ERR_INVALID;
    const int rc = hmx::build_cluster_tree(n, dim, coords, radii, weights, opt, *T, partition, partition_kind);
    if (rc != HMX_OK) {
        delete T;
        return rc;
    }
    *out = T;
    return HMX_OK;
}
// An EXISTING cluster tree (any builder: a user's VirtualPartitioning, a tree read from disk, another library), given as its
// preorder node table.  Everything downstream (block tree, device engine) only needs this structure.
int hmx_cluster_tree_from_nodes(int n, int dim, const int32_t *permutation, int num_nodes, const hmx_cluster_node *nodes, int num_partitions,
                                const int32_t *partition_nodes, int maximal_leaf_size, int permutation_is_local, hmx_cluster_tree **out) {
    if (!out || !permutation || !nodes || n < 1 || num_nodes < 1 || (dim != 2 && dim != 3) || num_partitions < 0 || (num_partitions > 0 && !partition_nodes)) {
        hmx::set_error("hmx_cluster_tree_from_nodes: invalid arguments");
        return HMX_ERR_INVALID;
    }
    auto fail = [](const std::string &why) {
        hmx::set_error("hmx_cluster_tree_from_nodes: " + why);
        return HMX_ERR_INVALID;
    };
    try {
        // the permutation must be one, the root must cover [0, n)
        std::vector<char> seen((size_t)n, 0);
        for (int i = 0; i < n; i++) {
            if (permutation[i] < 0 || permutation[i] >= n || seen[permutation[i]])
                return fail("the permutation is not a permutation of 0.." + std::to_string(n - 1));
            seen[permutation[i]] = 1;
        }
        if (nodes[0].offset != 0 || nodes[0].size != n)
            return fail("the first node (root) must cover all points");
        // preorder -> (parent, children): a node with c children is followed by c subtrees
        std::vector<int> parent(num_nodes, -1), subtree_end(num_nodes, 0);
        {
            std::vector<std::pair<int, int>> stack; // (node, children still to come)
            for (int v = 0; v < num_nodes; v++) {
                if (nodes[v].n_children < 0 || nodes[v].size < 0 || nodes[v].offset < 0 || (long long)nodes[v].offset + nodes[v].size > n)
                    return fail("node " + std::to_string(v) + " is out of range");
                while (!stack.empty() && stack.back().second == 0)
                    stack.pop_back();
                if (v > 0) {
                    if (stack.empty())
                        return fail("more nodes than the children counts account for");
                    parent[v] = stack.back().first;
                    stack.back().second--;
                }
                stack.emplace_back(v, nodes[v].n_children);
            }
            while (!stack.empty() && stack.back().second == 0)
                stack.pop_back();
            if (!stack.empty())
                return fail("fewer nodes than the children counts account for");
        }
        std::vector<std::vector<int>> children(num_nodes);
        for (int v = 1; v < num_nodes; v++)
            children[parent[v]].push_back(v);
        for (int v = 0; v < num_nodes; v++) { // the children of a node tile it, in order
            int pos = nodes[v].offset;
            for (int c : children[v]) {
                if (nodes[c].offset != pos || nodes[c].depth != nodes[v].depth + 1)
                    return fail("the children of node " + std::to_string(v) + " do not tile it in order (or their depth is not parent + 1)");
                pos += nodes[c].size;
            }
            if (!children[v].empty() && pos != nodes[v].offset + nodes[v].size)
                return fail("the children of node " + std::to_string(v) + " do not cover it");
        }
        auto *T = new hmx_cluster_tree();
        T->n    = n;
        T->dim  = dim;
        T->perm.assign(permutation, permutation + n);
        T->permutation_is_local = permutation_is_local != 0;
        // internal layout: root first, the children of a node contiguous
        std::vector<int> id(num_nodes, -1), order;
        order.reserve(num_nodes);
        order.push_back(0);
        id[0] = 0;
        T->nodes.resize(num_nodes);
        for (size_t q = 0; q < order.size(); q++) {
            const int v = order[q];
            hmx::ClusterNode &c = T->nodes[id[v]];
            c.parent     = parent[v] < 0 ? -1 : id[parent[v]];
            c.n_children = (int)children[v].size();
            c.first_child = c.n_children ? (int)order.size() : -1;
            c.depth = nodes[v].depth, c.offset = nodes[v].offset, c.size = nodes[v].size, c.rank = nodes[v].rank, c.counter = nodes[v].counter;
            c.radius = nodes[v].radius;
            for (int p = 0; p < 3; p++)
                c.center[p] = nodes[v].center[p];
            for (int ch : children[v]) {
                id[ch] = (int)order.size();
                order.push_back(ch);
            }
        }
        for (int k = 0; k < num_partitions; k++) {
            if (partition_nodes[k] < 0 || partition_nodes[k] >= num_nodes) {
                delete T;
                return fail("partition " + std::to_string(k) + " names a node that does not exist");
            }
            T->on_partition.push_back(id[partition_nodes[k]]);
        }
        int pos = 0; // the partition clusters tile the points in order (what the row distribution relies on)
        for (int k = 0; k < num_partitions; k++) {
            const hmx::ClusterNode &c = T->nodes[T->on_partition[k]];
            if (c.offset != pos) {
                delete T;
                return fail("the partition clusters do not tile the points in order");
            }
            pos += c.size;
        }
        if (num_partitions > 0 && pos != n) {
            delete T;
            return fail("the partition clusters do not cover the points");
        }
        T->opt.maximal_leaf_size  = maximal_leaf_size;
        T->opt.size_of_partition  = std::max(1, num_partitions);
        T->opt.number_of_children = T->nodes[0].n_children;
        *out = T;
        return HMX_OK;
    } catch (...) {
        return fail("out of host memory");
    }
}
void hmx_cluster_tree_destroy(hmx_cluster_tree *T) { delete T; }
int hmx_cluster_tree_size(const hmx_cluster_tree *T) { return T ? T->n : 0; }
int hmx_cluster_tree_num_nodes(const hmx_cluster_tree *T) { return T ? (int)T->nodes.size() : 0; }
int hmx_cluster_tree_num_partitions(const hmx_cluster_tree *T) { return T ? (int)T->on_partition.size() : 0; }
const int32_t *hmx_cluster_tree_permutation(const hmx_cluster_tree *T) { return T ? T->perm.data() : nullptr; }
int hmx_cluster_tree_nodes(const hmx_cluster_tree *T, hmx_cluster_node *out) {
    if (!T || !out)
        return HMX_ERR_INVALID;
    size_t i = 0;
    for (int v : T->preorder()) {
        const hmx::ClusterNode &c = T->nodes[v];
        hmx_cluster_node &o       = out[i++];
        o.depth                   = c.depth;
        o.offset                  = c.offset;
        o.size                    = c.size;
        o.rank                    = c.rank;
        o.counter                 = c.counter;
        o.n_children              = c.n_children;
        o.radius                  = c.radius;
        for (int p = 0; p < 3; p++)
            o.center[p] = p < T->dim ? c.center[p] : 0.0;
    }
    return HMX_OK;
}
int hmx_cluster_tree_save(const hmx_cluster_tree *T, const char *prefix) {
    if (!T || !prefix) {
        hmx::set_error("hmx_cluster_tree_save: invalid arguments");
        return HMX_ERR_INVALID;
    }
    return hmx::save_cluster_tree(*T, prefix);
}
int hmx_cluster_tree_load(const char *properties_file, const char *tree_file, hmx_cluster_tree **out) {
    if (!properties_file || !tree_file || !out) {
        hmx::set_error("hmx_cluster_tree_load: invalid arguments");
        return HMX_ERR_INVALID;
    }
    auto *T = new (std::nothrow) hmx_cluster_tree();
    if (!T)
        return HMX_ERR_INVALID;
    const int rc = hmx::load_cluster_tree(properties_file, tree_file, *T);
    if (rc != HMX_OK) {
        delete T;
        return rc;
    }
    *out = T;
    return HMX_OK;
}
int hmx_cluster_tree_depths(const hmx_cluster_tree *T, int32_t *max_min_leafsize_local) {
    if (!T || !max_min_leafsize_local)
        return HMX_ERR_INVALID;
    int dmax, dmin;
    hmx::cluster_tree_depths(*T, dmax, dmin);
    max_min_leafsize_local[0] = dmax;
    max_min_leafsize_local[1] = dmin;
    max_min_leafsize_local[2] = T->opt.maximal_leaf_size;
    max_min_leafsize_local[3] = T->permutation_is_local ? 1 : 0;
    return HMX_OK;
}
int hmx_cluster_tree_partition(const hmx_cluster_tree *T, int32_t *offset_size) {
    if (!T || !offset_size)
        return HMX_ERR_INVALID;
    for (size_t k = 0; k < T->on_partition.size(); k++) {
        offset_size[2 * k]     = T->nodes[T->on_partition[k]].offset;
        offset_size[2 * k + 1] = T->nodes[T->on_partition[k]].size;
    }
    return HMX_OK;
}

int hmx_block_tree_create(const hmx_cluster_tree *target, const hmx_cluster_tree *source, double eta, char symmetry,
                          char uplo, int min_target_depth, int min_source_depth, int target_partition_number,
                          int partition_number_for_symmetry, int block_tree_consistency, hmx_block_tree **out) {
    if (!target || !source || !out) {
        hmx::set_error("hmx_block_tree_create: NULL argument");
        return HMX_ERR_INVALID;
    }
    auto *bt                   = new hmx_block_tree();
    bt->target                 = target;
    bt->source                 = source;
    bt->eta                    = eta;
    bt->symmetry               = symmetry;
    bt->uplo                   = uplo;
    bt->min_target_depth       = min_target_depth;
    bt->min_source_depth       = min_source_depth;
    bt->target_partition       = target_partition_number;
    bt->partition_for_symmetry = partition_number_for_symmetry;
    bt->consistent             = block_tree_consistency != 0;
    const int rc               = hmx::build_block_tree(*bt);
    if (rc != HMX_OK) {
        delete bt;
        return rc;
    }
    *out = bt;
    return HMX_OK;
}
int hmx_block_tree_create_adm(const hmx_cluster_tree *target, const hmx_cluster_tree *source, double eta, char symmetry, char uplo,
                              int min_target_depth, int min_source_depth, int target_partition_number, int partition_number_for_symmetry,
                              int block_tree_consistency, hmx_admissibility_fn fn, void *user, hmx_block_tree **out) {
    if (!target || !source || !out) {
        hmx::set_error("hmx_block_tree_create_adm: NULL argument");
        return HMX_ERR_INVALID;
    }
    auto *bt                   = new hmx_block_tree();
    bt->target                 = target;
    bt->source                 = source;
    bt->eta                    = eta;
    bt->symmetry               = symmetry;
    bt->uplo                   = uplo;
    bt->min_target_depth       = min_target_depth;
    bt->min_source_depth       = min_source_depth;
    bt->target_partition       = target_partition_number;
    bt->partition_for_symmetry = partition_number_for_symmetry;
    bt->consistent             = block_tree_consistency != 0;
    bt->admissibility          = fn;
    bt->admissibility_user     = user;
    const int rc               = hmx::build_block_tree(*bt);
    if (rc != HMX_OK) {
        delete bt;
        return rc;
    }
    *out = bt;
    return HMX_OK;
}
int hmx_block_tree_create_local(const hmx_cluster_tree *target, const hmx_cluster_tree *source, double eta, char symmetry, char uplo,
                                int min_target_depth, int min_source_depth, int target_partition, int source_partition,
                                int block_tree_consistency, hmx_block_tree **out) {
    return hmx_block_tree_create_local_adm(target, source, eta, symmetry, uplo, min_target_depth, min_source_depth, target_partition,
                                           source_partition, block_tree_consistency, nullptr, nullptr, out);
}
int hmx_block_tree_create_local_adm(const hmx_cluster_tree *target, const hmx_cluster_tree *source, double eta, char symmetry, char uplo,
                                    int min_target_depth, int min_source_depth, int target_partition, int source_partition,
                                    int block_tree_consistency, hmx_admissibility_fn fn, void *user, hmx_block_tree **out) {
    if (!target || !source || !out || target_partition < 0 || source_partition < 0) {
        hmx::set_error("hmx_block_tree_create_local: invalid argument");
        return HMX_ERR_INVALID;
    }
    auto *bt                  = new hmx_block_tree();
    bt->target                = target;
    bt->source                = source;
    bt->eta                   = eta;
    bt->symmetry              = symmetry;
    bt->uplo                  = uplo;
    bt->min_target_depth      = min_target_depth;
    bt->min_source_depth      = min_source_depth;
    bt->target_root_partition = target_partition;
    bt->source_root_partition = source_partition;
    bt->consistent            = block_tree_consistency != 0;
    bt->admissibility         = fn;
    bt->admissibility_user    = user;
    const int rc              = hmx::build_block_tree(*bt);
    if (rc != HMX_OK) {
        delete bt;
        return rc;
    }
    *out = bt;
    return HMX_OK;
}
void hmx_block_tree_destroy(hmx_block_tree *bt) { delete bt; }
int64_t hmx_block_tree_num_leaves(const hmx_block_tree *bt) { return bt ? (int64_t)bt->leaves.size() : 0; }
int hmx_block_tree_leaves(const hmx_block_tree *bt, hmx_leaf *out) {
    if (!bt || !out)
        return HMX_ERR_INVALID;
    std::memcpy(out, bt->leaves.data(), bt->leaves.size() * sizeof(hmx_leaf));
    return HMX_OK;
}
int hmx_block_tree_root(const hmx_block_tree *bt, int32_t *r, char *symmetry_for_leaves, char *uplo_for_leaves) {
    if (!bt || !r)
        return HMX_ERR_INVALID;
    r[0] = bt->root_t_offset;
    r[1] = bt->root_t_size;
    r[2] = bt->root_s_offset;
    r[3] = bt->root_s_size;
    if (symmetry_for_leaves)
        *symmetry_for_leaves = bt->symmetry_for_leaves;
    if (uplo_for_leaves)
        *uplo_for_leaves = bt->uplo_for_leaves;
    return HMX_OK;
}
int hmx_block_tree_save_leaves_with_rank(const hmx_block_tree *bt, const int32_t *rank, const char *name) {
    if (!bt || !name) {
        hmx::set_error("hmx_block_tree_save_leaves_with_rank: invalid arguments");
        return HMX_ERR_INVALID;
    }
    return hmx::save_leaves_with_rank(bt->leaves, rank, bt->root_t_offset, bt->root_t_size, bt->root_s_offset, bt->root_s_size, name);
}

} // extern "C"

// capi_host.cpp -- C ABI entry points that need no GPU: geometry, cluster tree, block tree.
#include <cstring>
#include <mutex>
#include <new>

#include "hmx_host.hpp"

namespace hmx {
static thread_local std::string g_error;
void set_error(const std::string &msg) { g_error = msg; }
} // namespace hmx

extern "C" {

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

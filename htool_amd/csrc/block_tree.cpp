// block_tree.cpp -- block cluster tree on the host: admissibility, symmetry pruning, restriction to
// one row partition, re-rooting; emits the flat leaf list the device engine consumes.
//
// Behavioural contract: HMatrixTreeBuilder::build_block_tree / reset_root_of_block_tree /
// set_hmatrix_symmetry / set_symmetry_for_leaves (hmatrix/tree_builder/tree_builder.hpp:92-150,417-566)
// and get_leaves_from's mirror rule (hmatrix/hmatrix.hpp:247-274).  The leaf ORDER equals htool's
// preorder (children in creation order), which is also the order of m_admissible_tasks / m_dense_tasks.
// Integer results are bit-exact with the reference (tests/test_host_structure.py).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <thread>

#include "hmx_host.hpp"

namespace hmx {
namespace {

struct BNode {
    int t, s; // cluster node ids
    bool symmetric = false;
    int kind       = 0; // 0 hierarchical, 1 admissible leaf, 2 dense leaf
    std::vector<int> children;
};

struct Walker {
    hmx_block_tree &bt;
    const hmx_cluster_tree &T, &S;
    std::vector<BNode> arena;
    // Sub-trees are independent of each other: the first walk stops at recursion depth `defer_depth` (below the partition level, where
    // all descendants keep their target's rank) and leaves those nodes to be expanded by one walker each, on several threads.
    int defer_depth = -1;
    std::vector<int> deferred;        // arena ids, in creation order
    std::vector<int> depth_of{};      // recursion depth per arena id (only kept while deferring)

    explicit Walker(hmx_block_tree &b) : bt(b), T(*b.target), S(*b.source) {}
    const ClusterNode &tn(int v) const { return T.nodes[v]; }
    const ClusterNode &sn(int v) const { return S.nodes[v]; }

    // RjasanowSteinbach (hmatrix/interfaces/virtual_admissibility_condition.hpp:20-23)
    static hmx_cluster_node public_node(const ClusterNode &c, int dim) {
        hmx_cluster_node o;
        o.depth = c.depth, o.offset = c.offset, o.size = c.size, o.rank = c.rank, o.counter = c.counter, o.n_children = c.n_children;
        o.radius = c.radius;
        for (int p = 0; p < 3; p++)
            o.center[p] = p < dim ? c.center[p] : 0.0;
        return o;
    }
    bool admissible(const ClusterNode &t, const ClusterNode &s) const {
        if (bt.admissibility) { // VirtualAdmissibilityCondition supplied by the user
            const hmx_cluster_node a = public_node(t, T.dim), b = public_node(s, S.dim);
            return bt.admissibility(bt.admissibility_user, &a, &b, bt.eta) != 0;
        }
        double d2 = 0.0;
        for (int p = 0; p < T.dim; p++) {
            const double u = t.center[p] - s.center[p];
            d2             = d2 + u * u;
        }
        const double gap = std::sqrt(std::fabs(d2)) - t.radius - s.radius;
        return 2 * std::min(t.radius, s.radius) < bt.eta * std::max(gap, 0.0);
    }
    bool in_partition(const ClusterNode &t) const { return bt.target_partition == -1 || bt.target_partition == t.rank; }
    bool may_descend(const ClusterNode &t) const { return in_partition(t) || t.rank < 0; }
    // is_removed_by_symmetry (tree_builder.hpp:95-111)
    bool pruned(const ClusterNode &t, const ClusterNode &s) const {
        if (bt.symmetry == 'N')
            return false;
        const int ps = bt.partition_for_symmetry;
        if (bt.uplo == 'U' && t.offset >= s.offset + s.size) {
            if (ps == -1)
                return true;
            const ClusterNode &sp = sn(S.on_partition[ps]), &tp = tn(T.on_partition[ps]);
            return s.offset >= sp.offset && tp.offset <= t.offset && t.offset + t.size <= tp.offset + tp.size;
        }
        if (bt.uplo == 'L' && s.offset >= t.offset + t.size) {
            if (ps == -1)
                return true;
            const ClusterNode &sp = sn(S.on_partition[ps]), &tp = tn(T.on_partition[ps]);
            return s.offset < sp.offset + sp.size && tp.offset <= t.offset && t.offset + t.size <= tp.offset + tp.size;
        }
        return false;
    }
    static bool covers(const ClusterNode &a, const ClusterNode &b) { return a.offset <= b.offset && a.offset + a.size >= b.offset + b.size; }
    bool diagonal(const ClusterNode &t, const ClusterNode &s) const { return bt.symmetry != 'N' && t.offset == s.offset && t.size == s.size; }

    int make(int t, int s) {
        BNode b;
        b.t         = t;
        b.s         = s;
        b.symmetric = diagonal(tn(t), sn(s));
        arena.push_back(b);
        return (int)arena.size() - 1;
    }
    void child(int parent, int t, int s) {
        const int c = make(t, s);
        arena[parent].children.push_back(c);
        if (defer_depth >= 0) {
            depth_of.resize(arena.size(), 0);
            depth_of[c] = depth_of[parent] + 1;
            if (depth_of[c] >= defer_depth && tn(t).rank >= 0) {
                deferred.push_back(c);
                return;
            }
        }
        descend(c);
    }

    void descend(int id) {
        const int ti = arena[id].t, si = arena[id].s;
        const ClusterNode &t = tn(ti), &s = sn(si);
        const bool t_leaf = T.is_leaf(ti), s_leaf = S.is_leaf(si);
        if (admissible(t, s) && in_partition(t) && !pruned(t, s) && t.depth >= bt.min_target_depth && s.depth >= bt.min_source_depth && t.rank >= 0 && (!bt.consistent || s.rank >= 0)) {
            arena[id].kind = 1;
            return;
        }
        if (s_leaf && t_leaf) {
            arena[id].kind = 2;
            return;
        }
        auto each_t_child = [&](auto &&fn) {
            for (int c = 0; c < t.n_children; c++)
                fn(t.first_child + c);
        };
        auto each_s_child = [&](auto &&fn) {
            for (int c = 0; c < s.n_children; c++)
                fn(s.first_child + c);
        };
        auto split_t = [&]() { each_t_child([&](int tc) { if (may_descend(tn(tc)) && !pruned(tn(tc), s)) child(id, tc, si); }); };
        auto split_s = [&]() { each_s_child([&](int sc) { if (!pruned(t, sn(sc))) child(id, ti, sc); }); };
        auto split_both = [&]() {
            each_t_child([&](int tc) { each_s_child([&](int sc) { if (may_descend(tn(tc)) && !pruned(tn(tc), sn(sc))) child(id, tc, sc); }); });
        };
        auto jump_t_to_partition = [&]() {
            for (int tc : T.on_partition)
                if (may_descend(tn(tc)) && !pruned(tn(tc), s) && covers(t, tn(tc)))
                    child(id, tc, si);
        };
        if (s_leaf) {
            split_t();
        } else if (t_leaf) {
            split_s();
        } else if (bt.consistent) {
            if (t.rank < 0 && s.rank >= 0) {
                jump_t_to_partition();
            } else if (s.rank < 0 && t.rank >= 0) {
                for (int sc : S.on_partition)
                    if (!pruned(t, sn(sc)) && covers(s, sn(sc)))
                        child(id, ti, sc);
            } else {
                split_both();
            }
        } else {
            if (t.rank < 0) {
                jump_t_to_partition();
            } else if (s.size > t.size) {
                each_s_child([&](int sc) { if (may_descend(t) && !pruned(t, sn(sc))) child(id, ti, sc); });
            } else if (t.size > s.size) {
                split_t();
            } else {
                split_both();
            }
        }
    }
};

} // namespace

int build_block_tree(hmx_block_tree &bt) {
    const hmx_cluster_tree &T = *bt.target, &S = *bt.source;
    const bool sym_ok = (bt.symmetry == 'N' && bt.uplo == 'N') || ((bt.symmetry == 'S' || bt.symmetry == 'H') && (bt.uplo == 'L' || bt.uplo == 'U'));
    if (!sym_ok) { // check_inputs (tree_builder.hpp:79-91); 'H' (Hermitian) is meant for the complex instantiations
        set_error("hmx_block_tree_create: symmetry/UPLO must be ('N','N'), ('S'|'H','L') or ('S'|'H','U')");
        return HMX_ERR_INVALID;
    }
    if (bt.symmetry != 'N' && !bt.consistent) {
        set_error("hmx_block_tree_create: block tree consistency cannot be false when symmetry is not N");
        return HMX_ERR_INVALID;
    }
    const int np = (int)T.on_partition.size();
    if ((bt.target_partition != -1 && bt.target_partition >= np) || (bt.partition_for_symmetry != -1 && bt.partition_for_symmetry >= np)) {
        set_error("hmx_block_tree_create: partition number exceeds number of partitions");
        return HMX_ERR_INVALID;
    }
    if (bt.target_root_partition >= np || bt.source_root_partition >= (int)S.on_partition.size()) {
        set_error("hmx_block_tree_create_local: partition number exceeds number of partitions");
        return HMX_ERR_INVALID;
    }
    // DefaultLocalApproximationBuilder (distributed_operator/utility.hpp:64-88) builds from the partition clusters
    // themselves: same recursion, different starting pair
    const int t_start = bt.target_root_partition >= 0 ? T.on_partition[bt.target_root_partition] : 0;
    const int s_start = bt.source_root_partition >= 0 ? S.on_partition[bt.source_root_partition] : 0;
    Walker W(bt);
    // a user-supplied admissibility condition is a callback into the caller's code (Python, through ctypes): one thread
    const int hw      = hmx::host_cores();
    const int threads = bt.admissibility ? 1 : std::min(hw, 32);
    if (threads > 1 && (int64_t)T.nodes.size() + (int64_t)S.nodes.size() > 4096)
        W.defer_depth = 6;
    if (const char *e = std::getenv("HMX_BT_DEFER_DEPTH")) // tests: sub-trees from this recursion depth on small trees too (-1: one walk)
        W.defer_depth = bt.admissibility ? -1 : std::atoi(e);
    if (W.defer_depth >= 0)
        W.depth_of.assign(1, 0);
    const int root = W.make(t_start, s_start);
    W.arena[root].symmetric = false; // the root is flagged after re-rooting (tree_builder.hpp:408-410)
    W.descend(root);
    // the deferred sub-trees, each in its own arena
    std::vector<std::unique_ptr<Walker>> sub(W.deferred.size());
    std::vector<int> sub_of(W.deferred.empty() ? 0 : W.arena.size(), -1);
    for (size_t k = 0; k < W.deferred.size(); k++)
        sub_of[W.deferred[k]] = (int)k;
    if (!W.deferred.empty()) {
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (size_t k = next.fetch_add(1); k < sub.size(); k = next.fetch_add(1)) {
                sub[k].reset(new Walker(bt));
                const BNode &b = W.arena[W.deferred[k]];
                const int r    = sub[k]->make(b.t, b.s);
                sub[k]->descend(r);
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < std::min<int>(std::max(threads, 2), (int)sub.size()); t++)
            th.emplace_back(work);
        work();
        for (auto &x : th)
            x.join();
    }

    // reset_root_of_block_tree (tree_builder.hpp:533-566): when the root's target cluster is not the
    // requested partition, the new root adopts every node whose target cluster has that rank, in the
    // order an explicit LIFO stack discovers them.
    int root_t = t_start;
    std::vector<int> top = W.arena[root].children;
    if (!W.in_partition(T.nodes[t_start])) {
        std::vector<int> adopted, stack{root};
        while (!stack.empty()) {
            const int cur = stack.back();
            stack.pop_back();
            for (int c : W.arena[cur].children) {
                if (T.nodes[W.arena[c].t].rank == bt.target_partition)
                    adopted.push_back(c);
                else
                    stack.push_back(c);
            }
        }
        top    = adopted;
        root_t = T.on_partition[bt.target_partition];
        W.arena[root].kind = 0;
    }
    W.arena[root].children  = top;
    W.arena[root].t         = root_t;
    W.arena[root].symmetric = W.diagonal(T.nodes[root_t], S.nodes[s_start]);

    bt.root_t_offset = T.nodes[root_t].offset;
    bt.root_t_size   = T.nodes[root_t].size;
    bt.root_s_offset = S.nodes[s_start].offset;
    bt.root_s_size   = S.nodes[s_start].size;

    // symmetry_for_leaves of the root (tree_builder.hpp:134-150)
    bool flagged = false;
    if (bt.symmetry != 'N') {
        if (W.arena[root].children.empty())
            flagged = W.arena[root].symmetric;
        for (int c : W.arena[root].children)
            flagged = flagged || W.arena[c].symmetric;
    }
    bt.symmetry_for_leaves = flagged ? bt.symmetry : 'N';
    bt.uplo_for_leaves     = flagged ? bt.uplo : 'N';

    // leaves in preorder; mirror flag = has a symmetric ancestor (or is itself the symmetric root) and is
    // off-diagonal (hmatrix.hpp:252-272)
    bt.leaves.clear();
    struct Item {
        int id;
        bool sym_anc;
    };
    // walks one arena from `start`; a deferred node of the first walk is only recorded (position in the list so far + inherited flag)
    struct Hole {
        size_t at;
        int sub;
        bool sym_anc;
    };
    std::vector<Hole> holes;
    auto emit = [&](const Walker &A, int start, bool start_flag, std::vector<hmx_leaf> &out, bool first_walk) {
        std::vector<Item> stack{{start, start_flag}};
        while (!stack.empty()) {
            const Item it = stack.back();
            stack.pop_back();
            const BNode &b = A.arena[it.id];
            if (first_walk && !sub_of.empty() && sub_of[it.id] >= 0) {
                holes.push_back({out.size(), sub_of[it.id], it.sym_anc});
                continue;
            }
            if (b.children.empty()) {
                if (b.kind == 0)
                    continue; // pruned interior node without leaves
                const ClusterNode &t = T.nodes[b.t], &s = S.nodes[b.s];
                hmx_leaf l;
                l.t_offset   = t.offset;
                l.t_size     = t.size;
                l.s_offset   = s.offset;
                l.s_size     = s.size;
                l.admissible = b.kind == 1;
                l.mirror     = (it.sym_anc && t.offset != s.offset) ? 1 : 0;
                l.symmetric  = b.symmetric ? 1 : 0;
                l.rank       = b.kind == 1 ? 0 : -1;
                out.push_back(l);
                continue;
            }
            for (int c = (int)b.children.size() - 1; c >= 0; c--)
                stack.push_back({b.children[c], it.sym_anc || b.symmetric});
        }
    };
    if (sub.empty()) {
        emit(W, root, W.arena[root].symmetric, bt.leaves, false);
        return HMX_OK;
    }
    std::vector<hmx_leaf> upper;
    emit(W, root, W.arena[root].symmetric, upper, true);
    std::vector<std::vector<hmx_leaf>> part(holes.size());
    {
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (size_t k = next.fetch_add(1); k < holes.size(); k = next.fetch_add(1))
                emit(*sub[holes[k].sub], 0, holes[k].sym_anc, part[k], false);
        };
        std::vector<std::thread> th;
        for (int t = 1; t < std::min<int>(threads, (int)holes.size()); t++)
            th.emplace_back(work);
        work();
        for (auto &x : th)
            x.join();
    }
    size_t total = upper.size();
    for (auto &v : part)
        total += v.size();
    bt.leaves.reserve(total);
    size_t done = 0;
    for (size_t k = 0; k < holes.size(); k++) {
        bt.leaves.insert(bt.leaves.end(), upper.begin() + done, upper.begin() + holes[k].at);
        done = holes[k].at;
        bt.leaves.insert(bt.leaves.end(), part[k].begin(), part[k].end());
    }
    bt.leaves.insert(bt.leaves.end(), upper.begin() + done, upper.end());
    return HMX_OK;
}

} // namespace hmx

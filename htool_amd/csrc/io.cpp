// io.cpp -- the reference's on-disk formats for the structures either side of the hot path.
//
//   save_cluster_tree / read_cluster_tree   clustering/cluster_output.hpp:33-84,87-179
//       <prefix>_cluster_tree_properties.csv : leaf size, max/min depth, permutation, "local permutation" flag
//       <prefix>_cluster_tree.csv            : one line per depth; nodes in preorder, fields joined by '|':
//                                              #children|rank|offset|size|radius|counter|on_partition|center...
//   save_leaves_with_rank                   hmatrix/hmatrix_output.hpp:39-55
//       <name>.csv : "nt,ns" then "t_off,t_size,s_off,s_size,rank" per leaf, offsets relative to the root block
//
// Numbers are written the way `std::ostringstream << x` writes them (misc/user.hpp:14-18), i.e. 6 significant
// digits for radius/center: files are byte-identical to htool's, and a tree loaded from a file carries the rounded
// geometry exactly as htool's read_cluster_tree would (tests/test_io_formats.py compares against files written by
// the reference itself).
#include <algorithm>
#include <fstream>
#include <sstream>

#include "hmx_host.hpp"

namespace hmx {
namespace {

template <typename T>
std::string to_str(T v) {
    std::ostringstream ss;
    ss << v;
    return ss.str();
}
template <typename T>
T from_str(const std::string &s) { // StrToNbr (misc/user.hpp:20-25): 0 on failure
    std::istringstream ss(s);
    T v;
    return (ss >> v) ? v : T(0);
}
std::vector<std::string> split(const std::string &s, const std::string &delim) { // misc/user.hpp:29-41 (keeps empty tokens)
    std::vector<std::string> out;
    size_t from = 0, pos;
    while ((pos = s.find(delim, from)) != std::string::npos) {
        out.push_back(s.substr(from, pos - from));
        from = pos + delim.size();
    }
    out.push_back(s.substr(from));
    return out;
}

} // namespace

void cluster_tree_depths(const hmx_cluster_tree &T, int &max_depth, int &min_depth) {
    // the builder records the depth of every cluster it does not split further (tree_builder.hpp:181-195)
    max_depth = 0;
    min_depth = -1;
    for (size_t v = 0; v < T.nodes.size(); v++)
        if (T.is_leaf((int)v)) {
            max_depth = std::max(max_depth, T.nodes[v].depth);
            min_depth = min_depth < 0 ? T.nodes[v].depth : std::min(min_depth, T.nodes[v].depth);
        }
    if (min_depth < 0)
        min_depth = 0;
}

int save_cluster_tree(const hmx_cluster_tree &T, const std::string &prefix) {
    std::ofstream props(prefix + "_cluster_tree_properties.csv");
    std::ofstream tree(prefix + "_cluster_tree.csv");
    if (!props || !tree) {
        set_error("hmx_cluster_tree_save: cannot create files with prefix " + prefix);
        return HMX_ERR_INVALID;
    }
    int dmax, dmin;
    cluster_tree_depths(T, dmax, dmin);
    props << "maximal leaf size: " << T.opt.maximal_leaf_size << "\n";
    props << "maximal depth: " << dmax << "\n";
    props << "minimal depth: " << dmin << "\n";
    props << "permutation: ";
    for (size_t i = 0; i < T.perm.size(); i++)
        props << T.perm[i] << (i + 1 == T.perm.size() ? "\n" : ",");
    props << "local permutation: " << T.permutation_is_local << "\n";

    const int partition_depth = T.on_partition.empty() ? -1 : T.nodes[T.on_partition[0]].depth; // is_cluster_on_partition (cluster_node.hpp:85-87)
    std::vector<std::string> level(dmax + 1);
    for (int v : T.preorder()) {
        const ClusterNode &c = T.nodes[v];
        std::string rec      = to_str(c.n_children) + "|" + to_str(c.rank) + "|" + to_str(c.offset) + "|" + to_str(c.size) + "|" + to_str(c.radius) + "|" + to_str(c.counter) + "|" + to_str(c.depth == partition_depth ? 1 : 0);
        for (int p = 0; p < T.dim; p++)
            rec += "|" + to_str(c.center[p]);
        std::string &line = level[c.depth];
        if (!line.empty())
            line += ",";
        line += rec;
    }
    for (const std::string &line : level)
        tree << line << "\n";
    return (props && tree) ? HMX_OK : HMX_ERR_INVALID;
}

int load_cluster_tree(const std::string &props_file, const std::string &tree_file, hmx_cluster_tree &T) {
    std::ifstream tree(tree_file), props(props_file);
    if (!tree) {
        set_error("hmx_cluster_tree_load: cannot open file containing tree: " + tree_file);
        return HMX_ERR_INVALID;
    }
    if (!props) {
        set_error("hmx_cluster_tree_load: cannot open file containing permutation: " + props_file);
        return HMX_ERR_INVALID;
    }
    std::vector<std::vector<std::string>> level;
    std::string line;
    while (std::getline(tree, line))
        level.push_back(split(line, ","));
    if (level.empty() || level[0].empty() || split(level[0][0], "|").size() < 8) {
        set_error("hmx_cluster_tree_load: malformed tree file " + tree_file);
        return HMX_ERR_INVALID;
    }
    struct Rec {
        int n_children, rank, offset, size, counter, on_partition;
        double radius, center[3];
    };
    int dim = 0;
    auto parse = [&](const std::string &s, Rec &r) {
        const std::vector<std::string> f = split(s, "|");
        if (f.size() < 8 || f.size() > 10 || (dim && (int)f.size() - 7 != dim))
            return false;
        dim            = (int)f.size() - 7;
        r.n_children   = std::stoi(f[0]);
        r.rank         = std::stoi(f[1]);
        r.offset       = std::stoi(f[2]);
        r.size         = std::stoi(f[3]);
        r.radius       = from_str<double>(f[4]);
        r.counter      = std::stoi(f[5]);
        r.on_partition = from_str<bool>(f[6]) ? 1 : 0;
        for (int p = 0; p < 3; p++)
            r.center[p] = p < dim ? from_str<double>(f[7 + p]) : 0.0;
        return true;
    };
    auto fill = [&](ClusterNode &c, const Rec &r, int parent, int depth) {
        c.parent  = parent;
        c.depth   = depth;
        c.rank    = r.rank;
        c.offset  = r.offset;
        c.size    = r.size;
        c.radius  = r.radius;
        c.counter = r.counter;
        std::copy(r.center, r.center + 3, c.center);
    };
    T = hmx_cluster_tree();
    Rec root;
    try {
        if (!parse(level[0][0], root))
            throw 0;
        T.nodes.emplace_back();
        fill(T.nodes[0], root, -1, 0);
        // nodes are consumed level by level in preorder, exactly like read_cluster_tree's counter_offset walk
        std::vector<size_t> used(level.size() + 1, 0);
        std::vector<std::pair<int, int>> stack{{0, root.n_children}};
        while (!stack.empty()) {
            const int v = stack.back().first, nc = stack.back().second;
            stack.pop_back();
            if (nc == 0)
                continue;
            const int d = T.nodes[v].depth + 1;
            if (d >= (int)level.size() || used[d] + nc > level[d].size())
                throw 0;
            const int first          = (int)T.nodes.size();
            T.nodes[v].first_child   = first;
            T.nodes[v].n_children    = nc;
            std::vector<int> next(nc);
            for (int p = 0; p < nc; p++) {
                Rec r;
                if (!parse(level[d][used[d] + p], r))
                    throw 0;
                T.nodes.emplace_back();
                fill(T.nodes.back(), r, v, d);
                next[p] = r.n_children;
                if (r.on_partition && r.rank >= 0) { // child constructor, cluster_node.hpp:35-42
                    if (r.rank + 1 > (int)T.on_partition.size())
                        T.on_partition.resize(r.rank + 1, -1);
                    T.on_partition[r.rank] = first + p;
                }
            }
            used[d] += nc;
            for (int p = nc - 1; p >= 0; p--)
                stack.push_back({first + p, next[p]});
        }

        T.n   = root.size;
        T.dim = dim;
        std::getline(props, line);
        T.opt.maximal_leaf_size = (int)std::stoul(split(line, " ").back());
        std::getline(props, line); // maximal depth: implied by the tree
        std::getline(props, line); // minimal depth
        std::getline(props, line);
        const std::vector<std::string> p = split(split(line, " ").back(), ",");
        if ((int)p.size() != T.n)
            throw 0;
        T.perm.resize(T.n);
        for (int i = 0; i < T.n; i++)
            T.perm[i] = std::stoi(p[i]);
        std::getline(props, line);
        T.permutation_is_local = std::stoi(split(line, " ").back()) != 0;
    } catch (...) {
        set_error("hmx_cluster_tree_load: malformed cluster tree files " + props_file + ", " + tree_file);
        return HMX_ERR_INVALID;
    }
    for (int id : T.on_partition)
        if (id < 0) {
            set_error("hmx_cluster_tree_load: partition table has holes");
            return HMX_ERR_INVALID;
        }
    // the permutation and the node ranges become gather / scatter indices on host and device: reject anything out of range
    {
        std::vector<char> seen((size_t)std::max(T.n, 0), 0);
        for (int i = 0; i < T.n; i++) {
            const int v = T.perm[i];
            if (v < 0 || v >= T.n || seen[v]) {
                set_error("hmx_cluster_tree_load: the permutation in " + props_file + " is not a permutation of 0.." + std::to_string(T.n - 1));
                return HMX_ERR_INVALID;
            }
            seen[v] = 1;
        }
        for (const auto &nd : T.nodes)
            if (nd.offset < 0 || nd.size < 0 || (long long)nd.offset + nd.size > (long long)T.n) {
                set_error("hmx_cluster_tree_load: a cluster of " + tree_file + " lies outside [0, " + std::to_string(T.n) + ")");
                return HMX_ERR_INVALID;
            }
    }
    T.opt.size_of_partition  = (int)T.on_partition.size();
    T.opt.number_of_children = T.nodes[0].n_children;
    return HMX_OK;
}

int save_leaves_with_rank(const std::vector<hmx_leaf> &leaves, const int32_t *rank, int t0, int nt, int s0, int ns, const std::string &name) {
    std::ofstream out(name + ".csv");
    if (!out) {
        set_error("hmx_save_leaves_with_rank: cannot create " + name + ".csv");
        return HMX_ERR_INVALID;
    }
    out << nt << "," << ns << "\n";
    for (size_t b = 0; b < leaves.size(); b++) {
        const hmx_leaf &l = leaves[b];
        out << l.t_offset - t0 << "," << l.t_size << "," << l.s_offset - s0 << "," << l.s_size << "," << (rank ? rank[b] : l.rank) << "\n";
    }
    return out ? HMX_OK : HMX_ERR_INVALID;
}

} // namespace hmx

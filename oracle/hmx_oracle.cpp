// hmx_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the reference's (htool-ddm/htool) hot path: geometric cluster tree, block cluster
// tree + admissibility, partialACA / sympartialACA / fullACA / SVD compression, and the blockwise
// H-matrix-vector product.  It exists only to check the HIP engine: only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load it.  The product (htool_amd/, libhmx.so) never links,
// imports or calls anything in this directory.
//
// Language: C++17 with a C ABI (ctypes).  C++ rather than plain C because bit-exact cluster
// permutations depend on libstdc++'s std::sort / std::mt19937 behaviour.
//
// PARITY PINNING: this restatement is pinned against outputs of the reference itself, produced in the
// dev container by oracle/ref/ref_driver.cpp (real htool headers + MKL) and committed as
// tests/golden/*.npz (generator: tests/golden/make_golden.py).  tests/test_oracle_vs_golden.py checks:
// permutation / cluster table / leaf list bit-exact, ranks + pivots equal, U/V and y to ~1e-12.
//
// Every function cites the reference file:line (relative to include/htool/) it follows.
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <numeric>
#include <random>
#include <stack>
#include <string>
#include <tuple>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace orc {

// ---------------------------------------------------------------------------------------------
// Geometry: testing/geometry.hpp:11-38 (rotated ellipse / disk), :46-61 (ball)
// ---------------------------------------------------------------------------------------------
static void rotated_ellipse(int space_dim, double a, double b, double alpha, double z, int nr, double *xt) {
    std::mt19937 eng(0);
    std::uniform_real_distribution<double> dist(0, 1);
    double ca = std::cos(alpha), sa = std::sin(alpha);
    for (int j = 0; j < nr; j++) {
        double rho   = dist(eng);
        double theta = dist(eng);
        double r     = std::sqrt(rho);
        double phi   = 2 * static_cast<double>(M_PI) * theta;
        double xp    = a * r * std::cos(phi);
        double yp    = b * r * std::sin(phi);
        xt[space_dim * j + 0] = ca * xp - sa * yp;
        xt[space_dim * j + 1] = sa * xp + ca * yp;
        if (space_dim == 3)
            xt[space_dim * j + 2] = z;
    }
}
static void ball(int nr, double *xt) {
    std::mt19937 eng(0);
    std::uniform_real_distribution<double> dist(0, 1);
    for (int j = 0; j < nr; j++) {
        double rho    = dist(eng);
        double theta  = 2 * M_PI * dist(eng);
        double phi    = std::acos(2 * dist(eng) - 1);
        xt[3 * j + 0] = 0. + std::cbrt(rho) * std::sin(phi) * std::cos(theta);
        xt[3 * j + 1] = 0. + std::cbrt(rho) * std::sin(phi) * std::sin(theta);
        xt[3 * j + 2] = 0. + std::cbrt(rho) * std::cos(phi);
    }
}

// ---------------------------------------------------------------------------------------------
// Cluster tree: clustering/cluster_node.hpp:17-82, clustering/tree_builder/tree_builder.hpp:52-253
// ---------------------------------------------------------------------------------------------
struct Cluster;
struct TreeData {
    std::vector<int> perm;
    std::vector<Cluster *> on_partition;
    bool perm_local = false;
    int leaf_size   = 10;
};
struct Cluster {
    double radius = 0;
    std::vector<double> center;
    int rank, offset, size, counter = 0, depth = 0;
    std::vector<std::unique_ptr<Cluster>> children;
    TreeData *td = nullptr;
    bool is_leaf() const { return children.empty(); }
    Cluster *add_child(double r, const std::vector<double> &c, int rk, int off, int sz, int cnt, bool on_part) {
        auto ch     = std::make_unique<Cluster>();
        ch->radius  = r;
        ch->center  = c;
        ch->rank    = rk;
        ch->offset  = off;
        ch->size    = sz;
        ch->counter = cnt;
        ch->depth   = depth + 1;
        ch->td      = td;
        if (on_part) { // cluster_node.hpp:35-42
            if (rk + 1 > (int)td->on_partition.size())
                td->on_partition.resize(rk + 1, nullptr);
            td->on_partition[rk] = ch.get();
        }
        children.push_back(std::move(ch));
        return children.back().get();
    }
};
struct ClusterTree {
    TreeData td;
    std::unique_ptr<Cluster> root;
};

// basic_types/vector.hpp:86-95
static double dprod(const std::vector<double> &a, const std::vector<double> &b) { return std::inner_product(a.begin(), a.end(), b.begin(), double(0)); }
static double norm2(const std::vector<double> &u) { return std::sqrt(std::abs(dprod(u, u))); }

// tree_builder.hpp:210-233
static std::vector<double> compute_center(int dim, const double *x, const double *w, int offset, int size, const int *perm_or_null) {
    std::vector<double> center(dim, 0);
    std::vector<int> iota_perm;
    const int *perm = perm_or_null;
    if (!perm) {
        iota_perm.resize(size);
        std::iota(iota_perm.begin(), iota_perm.end(), 0);
        perm = iota_perm.data();
    }
    double total = 0;
    for (int j = 0; j < size; j++)
        total += w[perm[j + offset]];
    for (int j = 0; j < size; j++)
        for (int p = 0; p < dim; p++)
            center[p] += w[perm[j + offset]] * x[dim * perm[j + offset] + p];
    double inv = 1. / total;
    for (auto &c : center)
        c = c * inv;
    return center;
}
// tree_builder.hpp:236-253
static double compute_radius(int dim, const double *x, const double *radii, const std::vector<double> &center, int offset, int size, const int *perm_or_null) {
    double radius = 0;
    std::vector<int> iota_perm;
    const int *perm = perm_or_null;
    if (!perm) {
        iota_perm.resize(size);
        std::iota(iota_perm.begin(), iota_perm.end(), 0);
        perm = iota_perm.data();
    }
    for (int j = 0; j < size; j++) {
        std::vector<double> u(dim, 0);
        for (int p = 0; p < dim; p++)
            u[p] = x[dim * perm[j + offset] + p] - center[p];
        radius = std::max(radius, norm2(u) + radii[perm[j + offset]]);
    }
    return radius;
}

struct Dirs {
    std::vector<double> m; // dim x dim column-major: m[p + dim*col]
    std::vector<double> w;
    int dim;
    std::vector<double> col(int c) const { return std::vector<double>(m.begin() + c * dim, m.begin() + (c + 1) * dim); }
};

// misc/evp.hpp:13-49
static Dirs solve_evp2(const std::vector<double> &cov) {
    Dirs d;
    d.dim = 2;
    d.m.assign(4, 0.);
    d.w.assign(2, 0.);
    auto C      = [&](int i, int j) { return cov[i + 2 * j]; };
    double trace = C(0, 0) + C(1, 1);
    double det   = C(0, 0) * C(1, 1) - C(0, 1) * C(1, 0);
    d.w[0]       = trace / 2. + std::sqrt((trace * trace / 4. - det));
    d.w[1]       = trace / 2. - std::sqrt((trace * trace / 4. - det));
    const double eps = std::numeric_limits<double>::epsilon();
    if (std::abs(d.w[0]) > eps) {
        for (int index : {0, 1}) {
            double lam = d.w[(index + 1) % 2];
            double prod[4];
            for (int i = 0; i < 2; i++)
                for (int j = 0; j < 2; j++)
                    prod[i + 2 * j] = C(i, j) - (i == j ? 1. : 0.) * lam;
            int ind        = 0;
            double dirnorm = 0;
            do {
                d.m[0 + 2 * index] = prod[0 + 2 * ind];
                d.m[1 + 2 * index] = prod[1 + 2 * ind];
                dirnorm            = std::sqrt(d.m[0 + 2 * index] * d.m[0 + 2 * index] + d.m[1 + 2 * index] * d.m[1 + 2 * index]);
                ind++;
            } while ((dirnorm < eps) && (ind < 2));
            if (dirnorm < eps) {
                d.m[0 + 2 * index] = 1;
                d.m[1 + 2 * index] = 0;
            } else {
                d.m[0 + 2 * index] /= dirnorm;
                d.m[1 + 2 * index] /= dirnorm;
            }
        }
    } else {
        d.m[0] = 1;
        d.m[3] = 1;
    }
    return d;
}

// misc/evp.hpp:52-159
static Dirs solve_evp3(const std::vector<double> &cov) {
    Dirs d;
    d.dim = 3;
    d.m.assign(9, 0.);
    d.w.assign(3, 0.);
    auto C    = [&](int i, int j) { return cov[i + 3 * j]; };
    double p1 = std::pow(C(0, 1), 2) + std::pow(C(0, 2), 2) + std::pow(C(1, 2), 2);
    const double eps = std::numeric_limits<double>::epsilon();
    if (p1 < eps) {
        std::vector<double> eigs = {C(0, 0), C(1, 1), C(2, 2)};
        std::array<int, 3> idx   = {0, 1, 2};
        std::sort(idx.begin(), idx.end(), [&eigs](int a, int b) { return eigs[a] < eigs[b]; });
        d.m[idx[2] + 3 * 0] = 1;
        d.m[idx[1] + 3 * 1] = 1;
        d.m[idx[0] + 3 * 2] = 1;
        d.w                 = {eigs[idx[2]], eigs[idx[1]], eigs[idx[0]]};
    } else {
        double q  = (C(0, 0) + C(1, 1) + C(2, 2)) / 3.;
        double p2 = std::pow(C(0, 0) - q, 2) + std::pow(C(1, 1) - q, 2) + std::pow(C(2, 2) - q, 2) + 2. * p1;
        double p  = std::sqrt(p2 / 6.);
        double B[9];
        double invp = 1. / p;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                B[i + 3 * j] = (C(i, j) - (i == j ? 1. : 0.) * q) * invp; // (cov - q*I) * (1/p), matrix.hpp:179-190,210-218
        auto Bm     = [&](int i, int j) { return B[i + 3 * j]; };
        double detB = Bm(0, 0) * (Bm(1, 1) * Bm(2, 2) - Bm(1, 2) * Bm(2, 1)) - Bm(0, 1) * (Bm(1, 0) * Bm(2, 2) - Bm(1, 2) * Bm(2, 0)) + Bm(0, 2) * (Bm(1, 0) * Bm(2, 1) - Bm(1, 1) * Bm(2, 0));
        double r    = detB / 2.;
        double phi;
        if (r <= -1)
            phi = 1.047197551196598;
        else if (r >= 1)
            phi = 0;
        else
            phi = std::acos(r) / 3.;
        d.w[0] = q + 2. * p * std::cos(phi);
        d.w[2] = q + 2. * p * std::cos(phi + 2.094395102393195);
        d.w[1] = 3. * q - d.w[0] - d.w[2];
        if (std::abs(d.w[0]) > eps) {
            for (int index : {0, 1, 2}) {
                double prod[9];
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++)
                        prod[i + 3 * j] = C(i, j) - (i == j ? 1. : 0.) * d.w[index];
                const double *c0 = prod, *c1 = prod + 3, *c2 = prod + 6;
                std::vector<double> c0xc1{c0[1] * c1[2] - c0[2] * c1[1], c0[2] * c1[0] - c0[0] * c1[2], c0[0] * c1[1] - c0[1] * c1[0]};
                std::vector<double> c0xc2{c0[1] * c2[2] - c0[2] * c2[1], c0[2] * c2[0] - c0[0] * c2[2], c0[0] * c2[1] - c0[1] * c2[0]};
                std::vector<double> c1xc2{c1[1] * c2[2] - c1[2] * c2[1], c1[2] * c2[0] - c1[0] * c2[2], c1[0] * c2[1] - c1[1] * c2[0]};
                double d0 = dprod(c0xc1, c0xc1), d1 = dprod(c0xc2, c0xc2), d2 = dprod(c1xc2, c1xc2);
                double dmax = d0;
                int imax    = 0;
                if (d1 > dmax) {
                    dmax = d1;
                    imax = 1;
                }
                if (d2 > dmax)
                    imax = 2;
                std::vector<double> *v = imax == 0 ? &c0xc1 : (imax == 1 ? &c0xc2 : &c1xc2);
                double s               = std::sqrt(imax == 0 ? d0 : (imax == 1 ? d1 : d2));
                for (int i = 0; i < 3; i++)
                    d.m[i + 3 * index] = (*v)[i] / s;
            }
        } else {
            d.m[0] = 1;
            d.m[4] = 1;
            d.m[8] = 1;
        }
    }
    return d;
}

// partitioning.hpp:160-193 (ComputeLargestExtent)
static Dirs direction_pca(const Cluster &c, int dim, const double *x, const double *w) {
    const auto &perm = c.td->perm;
    std::vector<double> cov(dim * dim, 0.);
    for (int j = 0; j < c.size; j++) {
        std::vector<double> u(dim, 0);
        for (int p = 0; p < dim; p++)
            u[p] = x[dim * perm[j + c.offset] + p] - c.center[p];
        for (int p = 0; p < dim; p++)
            for (int q = 0; q < dim; q++)
                cov[p + dim * q] += w[perm[j + c.offset]] * u[p] * u[q];
    }
    Dirs d = dim == 2 ? solve_evp2(cov) : solve_evp3(cov);
    for (auto &e : d.w)
        e = e > 0 ? std::sqrt(e) : 0;
    return d;
}
// partitioning.hpp:196-231 (ComputeBoundingBox; note max initialised with numeric_limits::min())
static Dirs direction_bbox(const Cluster &c, int dim, const double *x) {
    const auto &perm = c.td->perm;
    std::vector<double> mn(dim, std::numeric_limits<double>::max()), mx(dim, std::numeric_limits<double>::min());
    for (int j = 0; j < c.size; j++)
        for (int p = 0; p < dim; p++) {
            double v = x[dim * perm[j + c.offset] + p];
            if (mn[p] > v)
                mn[p] = v;
            if (mx[p] < v)
                mx[p] = v;
        }
    std::vector<int> idx(dim);
    std::iota(idx.begin(), idx.end(), 0);
    std::sort(idx.begin(), idx.end(), [&](int a, int b) { return (mx[a] - mn[a]) < (mx[b] - mn[b]); });
    Dirs d;
    d.dim = dim;
    d.m.assign(dim * dim, 0.);
    d.w.assign(dim, 0.);
    for (int k = 0; k < dim; k++) {
        d.m[idx[dim - 1 - k] + dim * k] = 1;
        d.w[k]                          = mx[idx[dim - 1 - k]] - mn[idx[dim - 1 - k]];
    }
    return d;
}

using Split = std::vector<std::pair<int, int>>;
// partitioning.hpp:234-249
static Split split_regular(int offset, int size, int k) {
    Split s(k);
    int cs = int(size / k);
    for (int p = 0; p < k - 1; p++)
        s[p] = {offset + cs * p, cs};
    s.back() = {offset + cs * (k - 1), size - cs * (k - 1)};
    return s;
}
// partitioning.hpp:253-296 (including the `result != permutation.end()` quirk)
static Split split_geometric(int offset, int size, int dim, const double *x, const std::vector<int> &perm, const std::vector<double> &dir, int k) {
    Split out;
    if (size > k) {
        out.resize(k);
        auto pt   = [&](int a) { return std::vector<double>(x + dim * a, x + dim * (a + 1)); };
        auto sub  = [&](const std::vector<double> &a, const std::vector<double> &b) {
            std::vector<double> r(a.size());
            for (size_t i = 0; i < a.size(); i++)
                r[i] = a[i] - b[i];
            return r;
        };
        std::vector<double> first = pt(perm[offset]);
        std::vector<double> last  = pt(perm[offset + size - 1]);
        double gdist              = dprod(dir, sub(last, first));
        double csize              = gdist / k;
        int count                 = offset;
        std::vector<int> offs(k, 0), sizes(k, 0);
        for (int p = 0; p < k - 1; p++) {
            int res = count;
            while (res < offset + size && !(dprod(dir, sub(pt(perm[res]), first)) > csize))
                res++;
            if (res != (int)perm.size()) {
                offs[p]  = count;
                sizes[p] = res - count;
                count    = res;
                first    = pt(perm[res]);
            } else {
                offs[p]  = 0;
                sizes[p] = 0;
                break;
            }
        }
        offs.back()  = count;
        sizes.back() = size - std::accumulate(sizes.begin(), sizes.end() - 1, 0);
        for (int p = 0; p < k; p++)
            out[p] = {offs[p], sizes[p]};
    }
    return out;
}

struct Strategy {
    int direction   = 0; // 0 pca (ComputeLargestExtent), 1 bounding box
    int splitting   = 0; // 0 regular, 1 geometric
    int partition_n = 0; // 0 Partitioning, 1 Partitioning_N
};

static void sort_slice(std::vector<int> &perm, int off, int size, int dim, const double *x, const double *dir) {
    // partitioning.hpp:27-31: unstable std::sort, comparator recomputes both projections
    std::sort(perm.begin() + off, perm.begin() + off + size, [&](int a, int b) {
        double c = std::inner_product(x + dim * a, x + dim * (1 + a), dir, double(0));
        double d = std::inner_product(x + dim * b, x + dim * (1 + b), dir, double(0));
        return c < d;
    });
}
static Split do_split(const Strategy &st, int off, int size, int dim, const double *x, const std::vector<int> &perm, const std::vector<double> &dir, int k) {
    return st.splitting == 0 ? split_regular(off, size, k) : split_geometric(off, size, dim, x, perm, dir, k);
}

// partitioning.hpp:43-57
static void backtrack(int rn, int rd, int start, std::vector<int> &cur, std::vector<std::vector<int>> &res) {
    if (rd == 1) {
        if (rn <= start && rn >= 1) {
            cur.push_back(rn);
            res.push_back(cur);
            cur.pop_back();
        }
        return;
    }
    for (int f = start; f >= 1; f--)
        if (rn % f == 0) {
            cur.push_back(f);
            backtrack(rn / f, rd - 1, f, cur, res);
            cur.pop_back();
        }
}
// partitioning.hpp:62-86
static std::vector<int> distributed_splittings(int ndim, int nparts, const std::vector<double> &weights) {
    std::vector<std::vector<int>> decs;
    std::vector<int> cur;
    backtrack(nparts, ndim, nparts, cur, decs);
    double cost = std::numeric_limits<double>::max();
    int index = 0, result = 0;
    std::vector<double> ar(ndim);
    for (auto &dec : decs) {
        for (int i = 0; i < (int)dec.size(); i++)
            ar[i] = weights[i] / double(dec[i]);
        double cc = *std::max_element(ar.begin(), ar.end()) / *std::min_element(ar.begin(), ar.end());
        if (cc < cost) {
            cost   = cc;
            result = index;
        }
        index++;
    }
    return decs[result];
}

// partitioning.hpp:15-35 (Partitioning) and :89-156 (Partitioning_N)
static Split compute_partitioning(const Strategy &st, Cluster &c, int dim, const double *x, const double *w, int k) {
    auto &perm = c.td->perm;
    Dirs dirs  = st.direction == 0 ? direction_pca(c, dim, x, w) : direction_bbox(c, dim, x);
    if (st.partition_n) {
        int nrel = 0;
        for (auto dw : dirs.w)
            if (dw > std::numeric_limits<double>::epsilon() * 10)
                nrel++;
        nrel           = std::max(1, nrel);
        auto nsplit    = distributed_splittings(nrel, k, dirs.w);
        nrel           = nsplit.size();
        std::stack<std::tuple<int, int, int>> stack;
        stack.push(std::make_tuple(c.offset, c.size, 0));
        Split result;
        while (!stack.empty()) {
            auto t = stack.top();
            stack.pop();
            int toff = std::get<0>(t), tsize = std::get<1>(t), tdim = std::get<2>(t);
            std::vector<double> direction = dirs.col(tdim);
            sort_slice(perm, toff, tsize, dim, x, direction.data());
            Split tmp = do_split(st, toff, tsize, dim, x, perm, direction, nsplit[tdim]);
            if ((tdim < nrel - 1) && (int)tmp.size() == nsplit[tdim]) {
                for (int p = nsplit[tdim] - 1; p >= 0; p--)
                    stack.push(std::make_tuple(tmp[p].first, tmp[p].second, tdim + 1));
            } else if ((tdim == nrel - 1) && (int)tmp.size() == nsplit[tdim]) {
                result.insert(result.end(), tmp.begin(), tmp.end());
            } else {
                break;
            }
        }
        if ((int)result.size() == k) {
            std::sort(result.begin(), result.end(), [](auto a, auto b) { return a.first < b.first; });
            return result;
        }
    }
    sort_slice(perm, c.offset, c.size, dim, x, dirs.m.data()); // first column
    return do_split(st, c.offset, c.size, dim, x, perm, dirs.col(0), k);
}

// tree_builder.hpp:52-207, "Simple" partition type only (no user-given partition)
static std::unique_ptr<ClusterTree> create_cluster_tree(int n, int dim, const double *x, int leaf_size, int nchildren, int size_partition, const Strategy &st) {
    auto tree = std::make_unique<ClusterTree>();
    std::vector<double> radii(n, 0.), weights(n, 1.);
    auto center      = compute_center(dim, x, weights.data(), 0, n, nullptr);
    double radius    = compute_radius(dim, x, radii.data(), center, 0, n, nullptr);
    tree->root       = std::make_unique<Cluster>();
    Cluster &root    = *tree->root;
    root.radius      = radius;
    root.center      = center;
    root.rank        = -1;
    root.offset      = 0;
    root.size        = n;
    root.td          = &tree->td;
    tree->td.perm.resize(n);
    std::iota(tree->td.perm.begin(), tree->td.perm.end(), 0);
    tree->td.leaf_size = leaf_size;
    auto &perm         = tree->td.perm;

    std::stack<Cluster *> stack;
    stack.push(&root);
    int depth_of_partition;
    int nchildren_on_partition_level = size_partition;
    int additional_children          = 0;
    if (size_partition >= nchildren) {
        depth_of_partition           = static_cast<int>(floor(log(size_partition) / log(nchildren)));
        nchildren_on_partition_level = nchildren;
        if (size_partition != std::pow(nchildren, depth_of_partition))
            additional_children = size_partition - std::pow(nchildren, depth_of_partition);
    } else {
        depth_of_partition = 1;
    }
    if (size_partition == 1)
        tree->td.perm_local = true;

    while (!stack.empty()) {
        Cluster *cur = stack.top();
        stack.pop();
        bool on_level = (cur->depth == depth_of_partition - 1);
        int k         = on_level ? nchildren_on_partition_level : nchildren;
        if (on_level && cur->counter == std::pow(nchildren, cur->depth) - 1)
            k += additional_children;
        Split split = compute_partitioning(st, *cur, dim, x, weights.data(), k);
        if ((int)split.size() == k && std::all_of(split.begin(), split.end(), [](auto a) { return a.second > 0; })) {
            std::vector<Cluster *> children;
            for (int p = 0; p < (int)split.size(); p++) {
                center            = compute_center(dim, x, weights.data(), split[p].first, split[p].second, perm.data());
                radius            = compute_radius(dim, x, radii.data(), center, split[p].first, split[p].second, perm.data());
                int rank_of_child = cur->rank;
                int cnt           = cur->counter * k + p;
                bool on_part      = false;
                if (on_level) {
                    rank_of_child = cur->counter * nchildren_on_partition_level + p;
                    cnt           = rank_of_child;
                    on_part       = true;
                }
                children.push_back(cur->add_child(radius, center, rank_of_child, split[p].first, split[p].second, cnt, on_part));
            }
            for (auto *ch : children)
                if (ch->size > leaf_size)
                    stack.push(ch);
        }
    }
    return tree;
}

// ---------------------------------------------------------------------------------------------
// Generator: K(x,y) = 1/(delta + scale*|x-y|); examples/use_hmatrix.cpp:24-35,
// testing/generator_test.hpp:155-187; column-major output (hmatrix/interfaces/virtual_generator.hpp:24)
// evaluated through the permutations (virtual_generator.hpp:46-48).
// ---------------------------------------------------------------------------------------------
struct Generator {
    int dim;
    const double *xt, *xs;
    const int *pt, *ps;
    double delta, scale;
    inline double coef(int i, int j) const { // user numbering
        double s = 0;
        for (int p = 0; p < dim; p++) {
            double d = xt[dim * i + p] - xs[dim * j + p];
            s        = s + d * d;
        }
        return 1. / (delta + scale * std::sqrt(s));
    }
    void copy_submatrix(int M, int N, int row_off, int col_off, double *ptr) const { // cluster numbering
        for (int j = 0; j < M; j++)
            for (int k = 0; k < N; k++)
                ptr[j + (size_t)M * k] = coef(pt[row_off + j], ps[col_off + k]);
    }
};

// ---------------------------------------------------------------------------------------------
// Compressors
// ---------------------------------------------------------------------------------------------
struct LowRank {
    int M = 0, N = 0, rank = 0;
    std::vector<double> U; // M x r column-major
    std::vector<double> V; // r x N column-major
    std::vector<int> pivots; // (I,J) per accepted iteration, for parity checks
};

static double plain_dot(int n, const double *a, const double *b) {
    double s = 0;
    for (int i = 0; i < n; i++)
        s += a[i] * b[i];
    return s;
}

// hmatrix/lrmat/partialACA.hpp:42-184.  Returns true on success.  int32 arithmetic in the
// "not advantageous" test is kept on purpose.
static bool partial_aca(const Generator &A, int M, int N, int row_off, int col_off, double epsilon, int reqrank, LowRank &lr) {
    int I = 0, J = 0, q = 0;
    std::vector<std::vector<double>> uu, vv;
    std::vector<bool> vrow(M, false), vcol(N, false);
    double frob = 0, aux = 0, pivot, tmp;
    std::vector<double> r(N), c(M);
    lr.pivots.clear();
    while (((reqrank > 0) && (q < std::min(reqrank, std::min(M, N)))) || ((reqrank < 0) && (q == 0 || sqrt(aux / frob) > epsilon))) {
        q += 1;
        if (q * (M + N) > (M * N)) {
            q = -1;
            break;
        }
        std::fill(r.begin(), r.end(), 0.);
        A.copy_submatrix(1, N, I + row_off, col_off, r.data());
        for (size_t j = 0; j < uu.size(); j++) {
            double coef = -uu[j][I];
            for (int k = 0; k < N; k++)
                r[k] += coef * vv[j][k]; // axpy
        }
        pivot = 0.;
        for (int k = 0; k < N; k++) {
            if (vcol[k])
                continue;
            tmp = std::abs(r[k]);
            if (tmp < pivot)
                continue;
            pivot = tmp;
            J     = k;
        }
        vrow[I]      = true;
        double gamma = 1. / r[J];
        if (std::abs(r[J]) > 1e-15) {
            std::fill(c.begin(), c.end(), 0.);
            A.copy_submatrix(M, 1, row_off, J + col_off, c.data());
            for (size_t k = 0; k < uu.size(); k++) {
                double coef = -vv[k][J];
                for (int i = 0; i < M; i++)
                    c[i] += coef * uu[k][i];
            }
            for (auto &v : c)
                v = v * gamma;
            lr.pivots.push_back(I);
            lr.pivots.push_back(J);
            pivot = 0.;
            for (int k = 0; k < M; k++) {
                if (vrow[k])
                    continue;
                tmp = std::abs(c[k]);
                if (tmp < pivot)
                    continue;
                pivot = tmp;
                I     = k;
            }
            vcol[J] = true;
            if (reqrank < 0) {
                double frob_aux = 0.;
                aux             = std::abs(plain_dot(M, c.data(), c.data())) * std::abs(plain_dot(N, r.data(), r.data()));
                for (size_t j = 0; j < uu.size(); j++)
                    frob_aux += plain_dot(N, vv[j].data(), r.data()) * plain_dot(M, uu[j].data(), c.data());
                frob += aux + 2 * frob_aux;
            }
            uu.push_back(c);
            vv.push_back(r);
        } else {
            q -= 1;
            if (q == 0)
                q = -1;
            break;
        }
    }
    lr.M = M;
    lr.N = N;
    if (q > 0) {
        lr.rank = q;
        lr.U.resize((size_t)M * q);
        lr.V.resize((size_t)q * N);
        for (int k = 0; k < q; k++) {
            std::copy(uu[k].begin(), uu[k].end(), lr.U.begin() + (size_t)k * M);
            for (int j = 0; j < N; j++)
                lr.V[k + (size_t)q * j] = vv[k][j];
        }
        return true;
    }
    lr.rank = 0;
    return false;
}

// hmatrix/lrmat/sympartialACA.hpp:41-216
static bool sympartial_aca(const Generator &A, int M, int N, int row_off, int col_off, double epsilon, int reqrank, LowRank &lr) {
    int n1, n2, i1, i2;
    bool rows_first = row_off >= col_off;
    if (rows_first) {
        n1 = M;
        n2 = N;
        i1 = row_off;
        i2 = col_off;
    } else {
        n1 = N;
        n2 = M;
        i1 = col_off;
        i2 = row_off;
    }
    int I1 = 0, I2 = 0, q = 0;
    std::vector<std::vector<double>> uu, vv;
    std::vector<bool> v1(n1, false), v2(n2, false);
    double frob = 0, aux = 0, pivot, tmp;
    std::vector<double> u1(n2), u2(n1);
    lr.pivots.clear();
    while (((reqrank > 0) && (q < std::min(reqrank, std::min(n1, n2)))) || ((reqrank < 0) && (q == 0 || sqrt(aux / frob) > epsilon))) {
        q += 1;
        if (q * (n1 + n2) > (n1 * n2)) {
            q = -1;
            break;
        }
        std::fill(u1.begin(), u1.end(), 0.);
        if (rows_first)
            A.copy_submatrix(1, n2, i1 + I1, i2, u1.data());
        else
            A.copy_submatrix(n2, 1, i2, i1 + I1, u1.data());
        for (size_t j = 0; j < uu.size(); j++) {
            double coef = -uu[j][I1];
            for (int k = 0; k < n2; k++)
                u1[k] += coef * vv[j][k];
        }
        pivot = 0.;
        for (int k = 0; k < n2; k++) {
            if (v2[k])
                continue;
            tmp = std::abs(u1[k]);
            if (tmp < pivot)
                continue;
            pivot = tmp;
            I2    = k;
        }
        v1[I1]       = true;
        double gamma = 1. / u1[I2];
        if (std::abs(u1[I2]) > 1e-15) {
            std::fill(u2.begin(), u2.end(), 0.);
            if (rows_first)
                A.copy_submatrix(n1, 1, i1, i2 + I2, u2.data());
            else
                A.copy_submatrix(1, n1, i2 + I2, i1, u2.data());
            for (size_t k = 0; k < uu.size(); k++) {
                double coef = -vv[k][I2];
                for (int i = 0; i < n1; i++)
                    u2[i] += coef * uu[k][i];
            }
            for (auto &v : u2)
                v = v * gamma;
            lr.pivots.push_back(I1);
            lr.pivots.push_back(I2);
            pivot = 0.;
            for (int k = 0; k < n1; k++) {
                if (v1[k])
                    continue;
                tmp = std::abs(u2[k]);
                if (tmp < pivot)
                    continue;
                pivot = tmp;
                I1    = k;
            }
            v2[I2] = true;
            if (reqrank < 0) {
                double frob_aux = 0.;
                aux             = std::abs(plain_dot(n1, u2.data(), u2.data())) * std::abs(plain_dot(n2, u1.data(), u1.data()));
                for (size_t j = 0; j < uu.size(); j++)
                    frob_aux += plain_dot(n2, u1.data(), vv[j].data()) * plain_dot(n1, u2.data(), uu[j].data());
                frob += aux + 2 * frob_aux;
            }
            uu.push_back(u2);
            vv.push_back(u1);
        } else {
            q -= 1;
            if (q == 0)
                q = -1;
            break;
        }
    }
    lr.M = M;
    lr.N = N;
    if (q > 0) {
        lr.rank = q;
        lr.U.resize((size_t)M * q);
        lr.V.resize((size_t)q * N);
        for (int k = 0; k < q; k++) {
            const auto &ucol = rows_first ? uu[k] : vv[k];
            const auto &vrow = rows_first ? vv[k] : uu[k];
            std::copy(ucol.begin(), ucol.end(), lr.U.begin() + (size_t)k * M);
            for (int j = 0; j < N; j++)
                lr.V[k + (size_t)q * j] = vrow[j];
        }
        return true;
    }
    lr.rank = 0;
    return false;
}

// matrix/utils/math.hpp:7-16
static double norm_frob(const std::vector<double> &mat, int M, int N) {
    double norm = 0;
    for (int j = 0; j < M; j++)
        for (int k = 0; k < N; k++)
            norm = norm + std::pow(std::abs(mat[j + (size_t)M * k]), 2);
    return sqrt(norm);
}

// hmatrix/lrmat/fullACA.hpp:38-88
static bool full_aca(const Generator &A, int M, int N, int row_off, int col_off, double epsilon, int reqrank, LowRank &lr) {
    std::vector<double> mat((size_t)M * N);
    A.copy_submatrix(M, N, row_off, col_off, mat.data());
    int q = 0;
    std::vector<std::vector<double>> uu, vv;
    double Norm = norm_frob(mat, M, N);
    lr.pivots.clear();
    while (((reqrank > 0) && (q < std::min(reqrank, std::min(M, N)))) || ((reqrank < 0) && (norm_frob(mat, M, N) / Norm > epsilon || q == 0))) {
        q += 1;
        if (q * (M + N) > (M * N)) {
            q = -1;
            break;
        }
        // matrix/utils/math.hpp:18-23: std::max_element => first maximum in column-major order
        int p        = std::max_element(mat.begin(), mat.end(), [](double a, double b) { return std::abs(a) < std::abs(b); }) - mat.begin();
        int pi       = p % M, pj = p / M;
        double pivot = mat[pi + (size_t)M * pj];
        if (std::abs(pivot) < 1e-15) {
            q += -1;
            break;
        }
        lr.pivots.push_back(pi);
        lr.pivots.push_back(pj);
        std::vector<double> col(M), row(N);
        for (int i = 0; i < M; i++)
            col[i] = mat[i + (size_t)M * pj];
        for (int j = 0; j < N; j++)
            row[j] = mat[pi + (size_t)M * j] / pivot;
        uu.push_back(col);
        vv.push_back(row);
        for (int i = 0; i < M; i++)
            for (int j = 0; j < N; j++)
                mat[i + (size_t)M * j] -= uu[q - 1][i] * vv[q - 1][j];
    }
    lr.M = M;
    lr.N = N;
    if (q > 0) {
        lr.rank = q;
        lr.U.resize((size_t)M * q);
        lr.V.resize((size_t)q * N);
        for (int k = 0; k < q; k++) {
            std::copy(uu[k].begin(), uu[k].end(), lr.U.begin() + (size_t)k * M);
            for (int j = 0; j < N; j++)
                lr.V[k + (size_t)q * j] = vv[k][j];
        }
        return true;
    }
    lr.rank = 0;
    return false;
}

// One-sided Jacobi SVD of an M x N column-major matrix, standing in for LAPACK gesvd('A','A')
// (matrix/utils/SVD_truncation.hpp:30-33).  LAPACK is a third-party dependency absent from
// /root/reference (vendor/version unpinned); gesvd's published contract -- singular
// values descending, A = u diag(s) vt -- is what is restated.  Singular vectors are unique only up to
// sign, so parity on U,V is checked through the product U*V and the singular values.
// Returns s (min(M,N)), u (M x min) and vt (min x N) -- the thin factors, which is all SVD.hpp uses.
static void jacobi_svd(int M, int N, const std::vector<double> &Ain, std::vector<double> &s, std::vector<double> &u, std::vector<double> &vt) {
    bool transposed = M < N;
    int m = transposed ? N : M, n = transposed ? M : N; // work on tall m x n
    std::vector<double> W((size_t)m * n), Vm((size_t)n * n, 0.);
    for (int i = 0; i < M; i++)
        for (int j = 0; j < N; j++) {
            double v = Ain[i + (size_t)M * j];
            if (transposed)
                W[j + (size_t)m * i] = v;
            else
                W[i + (size_t)m * j] = v;
        }
    for (int i = 0; i < n; i++)
        Vm[i + (size_t)n * i] = 1.;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double *wp = &W[(size_t)m * p], *wq = &W[(size_t)m * q];
                double app = 0, aqq = 0, apq = 0;
                for (int i = 0; i < m; i++) {
                    app += wp[i] * wp[i];
                    aqq += wq[i] * wq[i];
                    apq += wp[i] * wq[i];
                }
                if (std::abs(apq) <= 1e-300 || std::abs(apq) <= 1e-17 * std::sqrt(app * aqq))
                    continue;
                off          = std::max(off, std::abs(apq) / std::sqrt(app * aqq));
                double zeta  = (aqq - app) / (2. * apq);
                double t     = (zeta >= 0 ? 1. : -1.) / (std::abs(zeta) + std::sqrt(1. + zeta * zeta));
                double cs    = 1. / std::sqrt(1. + t * t), sn = cs * t;
                for (int i = 0; i < m; i++) {
                    double a = wp[i], b = wq[i];
                    wp[i] = cs * a - sn * b;
                    wq[i] = sn * a + cs * b;
                }
                double *vp = &Vm[(size_t)n * p], *vq = &Vm[(size_t)n * q];
                for (int i = 0; i < n; i++) {
                    double a = vp[i], b = vq[i];
                    vp[i] = cs * a - sn * b;
                    vq[i] = sn * a + cs * b;
                }
            }
        if (off < 1e-15)
            break;
    }
    std::vector<double> sv(n);
    std::vector<int> order(n);
    for (int j = 0; j < n; j++) {
        double nn = 0;
        for (int i = 0; i < m; i++)
            nn += W[i + (size_t)m * j] * W[i + (size_t)m * j];
        sv[j] = std::sqrt(nn);
    }
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return sv[a] > sv[b]; });
    int k = n; // = min(M,N)
    s.resize(k);
    u.assign((size_t)M * k, 0.);
    vt.assign((size_t)k * N, 0.);
    for (int jj = 0; jj < k; jj++) {
        int j     = order[jj];
        s[jj]     = sv[j];
        double is = sv[j] > 0 ? 1. / sv[j] : 0.;
        if (!transposed) { // A = (W/s) s Vm^T
            for (int i = 0; i < M; i++)
                u[i + (size_t)M * jj] = W[i + (size_t)m * j] * is;
            for (int c = 0; c < N; c++)
                vt[jj + (size_t)k * c] = Vm[c + (size_t)n * j];
        } else { // A^T = (W/s) s Vm^T  =>  A = Vm s (W/s)^T
            for (int i = 0; i < M; i++)
                u[i + (size_t)M * jj] = Vm[i + (size_t)n * j];
            for (int c = 0; c < N; c++)
                vt[jj + (size_t)k * c] = W[c + (size_t)m * j] * is;
        }
    }
}

// hmatrix/lrmat/SVD.hpp:27-62 (auto) and :64-92 (fixed rank); truncation rule
// matrix/utils/SVD_truncation.hpp:37-52
static bool svd_compress(const Generator &A, int M, int N, int row_off, int col_off, double epsilon, int reqrank, LowRank &lr, std::vector<double> *sing_out = nullptr) {
    std::vector<double> mat((size_t)M * N);
    A.copy_submatrix(M, N, row_off, col_off, mat.data());
    std::vector<double> s, u, vt;
    jacobi_svd(M, N, mat, s, u, vt);
    if (sing_out)
        *sing_out = s;
    int k = s.size();
    int truncated_rank;
    {
        int j           = k;
        double svd_norm = 0, error = 0;
        for (auto &e : s)
            svd_norm += e * e;
        svd_norm = std::sqrt(svd_norm);
        do {
            j = j - 1;
            error += std::pow(std::abs(s[j]), 2);
        } while (j > 0 && std::sqrt(error) / svd_norm < epsilon);
        truncated_rank = j + 1;
    }
    lr.M = M;
    lr.N = N;
    lr.pivots.clear();
    if (reqrank > 0) {
        truncated_rank = std::min(reqrank, std::min(M, N));
    } else {
        if (truncated_rank * (M + N) > (M * N)) {
            lr.rank = 0;
            return false;
        }
        if (truncated_rank <= 0) {
            lr.rank = 0;
            return false;
        }
    }
    int r   = truncated_rank;
    lr.rank = r;
    lr.U.resize((size_t)M * r);
    lr.V.resize((size_t)r * N);
    for (int i = 0; i < M; i++)
        for (int j = 0; j < r; j++)
            lr.U[i + (size_t)M * j] = u[i + (size_t)M * j] * s[j];
    for (int i = 0; i < r; i++)
        for (int j = 0; j < N; j++)
            lr.V[i + (size_t)r * j] = vt[i + (size_t)k * j];
    return true;
}

enum Compressor { PARTIAL_ACA = 0,
                  SYMPARTIAL_ACA = 1,
                  FULL_ACA = 2,
                  SVD = 3 };
static bool compress(int kind, const Generator &A, int M, int N, int ro, int co, double eps, int reqrank, LowRank &lr) {
    switch (kind) {
    case PARTIAL_ACA:
        return partial_aca(A, M, N, ro, co, eps, reqrank, lr);
    case SYMPARTIAL_ACA:
        return sympartial_aca(A, M, N, ro, co, eps, reqrank, lr);
    case FULL_ACA:
        return full_aca(A, M, N, ro, co, eps, reqrank, lr);
    default:
        return svd_compress(A, M, N, ro, co, eps, reqrank, lr);
    }
}

// ---------------------------------------------------------------------------------------------
// Block tree: hmatrix/tree_builder/tree_builder.hpp:417-566
// ---------------------------------------------------------------------------------------------
struct Block {
    const Cluster *t, *s;
    char symmetry = 'N', uplo = 'N';
    std::vector<std::unique_ptr<Block>> children;
    // leaf payload
    int kind = 0; // 0 hierarchical, 1 dense, 2 low rank
    std::vector<double> dense;
    LowRank lr;
    bool admissible_task = false;
    bool is_leaf() const { return children.empty(); }
};

struct HMat {
    std::unique_ptr<Block> root;
    const Cluster *root_t = nullptr, *root_s = nullptr; // after reset_root_of_block_tree
    char sym = 'N', uplo = 'N';
    char sym_for_leaves = 'N', uplo_for_leaves = 'N';
    int false_positive = 0;
    // flat views
    struct Leaf {
        Block *b;
        bool mirror;
    };
    std::vector<Leaf> preorder;  // natural order (children in creation order)
    std::vector<Leaf> dfs_order; // get_leaves_from order (hmatrix/hmatrix.hpp:247-274): explicit stack, last child first
    // flat-construction storage (from_blocks): owns clusters
    std::vector<std::unique_ptr<Cluster>> owned_clusters;
    std::vector<std::unique_ptr<Block>> owned_blocks;
};

struct BuildParams {
    double eta;
    char sym, uplo;
    int mint, mins;
    int target_partition;
    int partition_for_symmetry;
    bool consistent;
    const Cluster *troot, *sroot;
};

// hmatrix/interfaces/virtual_admissibility_condition.hpp:20-23
static bool admissible(const Cluster &t, const Cluster &s, double eta) {
    std::vector<double> diff(t.center.size());
    for (size_t i = 0; i < diff.size(); i++)
        diff[i] = t.center[i] - s.center[i];
    return 2 * std::min(t.radius, s.radius) < eta * std::max((norm2(diff) - t.radius - s.radius), 0.);
}
// tree_builder.hpp:92-94
static bool in_partition(const BuildParams &P, const Cluster &c) { return P.target_partition == -1 ? true : (P.target_partition == c.rank); }
// tree_builder.hpp:95-111
static bool removed_by_symmetry(const BuildParams &P, const Cluster &t, const Cluster &s) {
    if (P.sym == 'N')
        return false;
    int ps = P.partition_for_symmetry;
    if (P.uplo == 'U' && t.offset >= (s.offset + s.size)) {
        if (ps == -1)
            return true;
        const Cluster *sp = P.sroot->td->on_partition[ps], *tp = P.troot->td->on_partition[ps];
        return s.offset >= sp->offset && tp->offset <= t.offset && t.offset + t.size <= tp->offset + tp->size;
    }
    if (P.uplo == 'L' && s.offset >= (t.offset + t.size)) {
        if (ps == -1)
            return true;
        const Cluster *sp = P.sroot->td->on_partition[ps], *tp = P.troot->td->on_partition[ps];
        return s.offset < sp->offset + sp->size && tp->offset <= t.offset && t.offset + t.size <= tp->offset + tp->size;
    }
    return false;
}
// tree_builder.hpp:125-132
static void set_symmetry(const BuildParams &P, Block &b) {
    if (P.sym != 'N' && b.t->offset == b.s->offset && b.t->size == b.s->size) {
        b.symmetry = P.sym;
        b.uplo     = P.uplo;
    }
}
// cluster_node.hpp:89-96
static bool contains(const Cluster &a, const Cluster &b) { return a.offset <= b.offset && a.size + a.offset >= b.size + b.offset; }

static Block *add_child(Block &parent, const Cluster *t, const Cluster *s) {
    auto b = std::make_unique<Block>();
    b->t   = t;
    b->s   = s;
    parent.children.push_back(std::move(b));
    return parent.children.back().get();
}

// tree_builder.hpp:417-531
static void build_block_tree(const BuildParams &P, Block *cur, std::vector<Block *> &adm, std::vector<Block *> &dense) {
    const Cluster &t = *cur->t, &s = *cur->s;
    bool is_adm      = admissible(t, s, P.eta);
    auto recurse     = [&](const Cluster *tc, const Cluster *sc) {
        Block *ch = add_child(*cur, tc, sc);
        set_symmetry(P, *ch);
        build_block_tree(P, ch, adm, dense);
    };
    auto tchild_ok = [&](const Cluster &tc) { return in_partition(P, tc) || tc.rank < 0; };
    if (is_adm && in_partition(P, t) && !removed_by_symmetry(P, t, s) && t.depth >= P.mint && s.depth >= P.mins && t.rank >= 0 && (!P.consistent || s.rank >= 0)) {
        adm.push_back(cur);
        cur->admissible_task = true;
    } else if (s.is_leaf() && t.is_leaf()) {
        dense.push_back(cur);
    } else if (s.is_leaf() && !t.is_leaf()) {
        for (auto &tc : t.children)
            if (tchild_ok(*tc) && !removed_by_symmetry(P, *tc, s))
                recurse(tc.get(), &s);
    } else if (!s.is_leaf() && t.is_leaf()) {
        for (auto &sc : s.children)
            if (!removed_by_symmetry(P, t, *sc))
                recurse(&t, sc.get());
    } else if (P.consistent) {
        if (t.rank < 0 && s.rank >= 0) {
            for (auto *tc : t.td->on_partition)
                if (tchild_ok(*tc) && !removed_by_symmetry(P, *tc, s) && contains(t, *tc))
                    recurse(tc, &s);
        } else if (s.rank < 0 && t.rank >= 0) {
            for (auto *sc : s.td->on_partition)
                if (!removed_by_symmetry(P, t, *sc) && contains(s, *sc))
                    recurse(&t, sc);
        } else {
            for (auto &tc : t.children)
                for (auto &sc : s.children)
                    if (tchild_ok(*tc) && !removed_by_symmetry(P, *tc, *sc))
                        recurse(tc.get(), sc.get());
        }
    } else {
        if (t.rank < 0) {
            for (auto *tc : t.td->on_partition)
                if (tchild_ok(*tc) && !removed_by_symmetry(P, *tc, s) && contains(t, *tc))
                    recurse(tc, &s);
        } else if (s.size > t.size) {
            for (auto &sc : s.children)
                if ((in_partition(P, t) || t.rank < 0) && !removed_by_symmetry(P, t, *sc))
                    recurse(&t, sc.get());
        } else if (t.size > s.size) {
            for (auto &tc : t.children)
                if (tchild_ok(*tc) && !removed_by_symmetry(P, *tc, s))
                    recurse(tc.get(), &s);
        } else {
            for (auto &tc : t.children)
                for (auto &sc : s.children)
                    if (tchild_ok(*tc) && !removed_by_symmetry(P, *tc, *sc))
                        recurse(tc.get(), sc.get());
        }
    }
}

// tree_builder.hpp:533-566
static void reset_root(const BuildParams &P, HMat &H) {
    Block &root = *H.root;
    if (!in_partition(P, *root.t)) {
        std::stack<Block *> st;
        st.push(&root);
        std::vector<std::unique_ptr<Block>> new_children;
        while (!st.empty()) {
            Block *cur = st.top();
            st.pop();
            for (auto &child : cur->children) {
                if (child->t->rank == P.target_partition)
                    new_children.push_back(std::move(child));
                else
                    st.push(child.get());
            }
        }
        // keep the detached intermediate nodes alive until we are done, then drop them
        root.children.clear();
        root.children = std::move(new_children);
        root.t        = root.t->td->on_partition[P.target_partition];
    }
}

static void collect_leaves(HMat &H) {
    H.preorder.clear();
    H.dfs_order.clear();
    std::function<void(Block *, bool)> pre = [&](Block *b, bool sym_anc) {
        if (b->is_leaf()) {
            H.preorder.push_back({b, sym_anc && b->t->offset != b->s->offset});
            return;
        }
        for (auto &c : b->children)
            pre(c.get(), sym_anc || b->symmetry != 'N');
    };
    pre(H.root.get(), H.root->symmetry != 'N');
    // hmatrix.hpp:247-274
    std::stack<std::pair<Block *, bool>> st;
    st.push({H.root.get(), H.root->symmetry != 'N'});
    while (!st.empty()) {
        auto cur = st.top();
        st.pop();
        if (cur.first->is_leaf())
            H.dfs_order.push_back({cur.first, cur.second && cur.first->t->offset != cur.first->s->offset});
        for (auto &c : cur.first->children)
            st.push({c.get(), cur.first->symmetry != 'N' || cur.second});
    }
}

// tree_builder.hpp:134-150 (symmetry_for_leaves of the root)
static char symmetry_for_leaves(const Block &b, char sym) {
    if (sym == 'N')
        return 'N';
    if (b.is_leaf())
        return b.symmetry != 'N' ? sym : 'N';
    char res = 'N';
    for (auto &c : b.children) {
        // postorder: children first; parent flagged if any child has symmetry != 'N'
        symmetry_for_leaves(*c, sym);
        if (c->symmetry != 'N')
            res = sym;
    }
    return res;
}

static std::unique_ptr<HMat> build_hmatrix(const ClusterTree &tt, const ClusterTree &st, const Generator &A, double eps, double eta, char sym, char uplo, int reqrank, int compressor, int mint, int mins, int target_partition, int partition_for_symmetry, bool consistent, bool parallel) {
    auto H     = std::make_unique<HMat>();
    H->root    = std::make_unique<Block>();
    H->root->t = tt.root.get();
    H->root->s = st.root.get();
    H->sym     = sym;
    H->uplo    = uplo;
    BuildParams P{eta, sym, uplo, mint, mins, target_partition, partition_for_symmetry, consistent, tt.root.get(), st.root.get()};
    std::vector<Block *> adm, dense;
    build_block_tree(P, H->root.get(), adm, dense);
    reset_root(P, *H);
    set_symmetry(P, *H->root);
    H->root_t = H->root->t;
    H->root_s = H->root->s;
    // sequential_compute_blocks / openmp_compute_blocks (tree_builder.hpp:568-666)
    int fp = 0;
#pragma omp parallel for schedule(guided) reduction(+ : fp) if (parallel)
    for (int p = 0; p < (int)adm.size(); p++) {
        Block *b = adm[p];
        int M = b->t->size, N = b->s->size;
        bool ok = compress(compressor, A, M, N, b->t->offset, b->s->offset, eps, reqrank, b->lr);
        if (ok) {
            b->kind = 2;
        } else {
            b->lr = LowRank();
            b->dense.resize((size_t)M * N);
            A.copy_submatrix(M, N, b->t->offset, b->s->offset, b->dense.data());
            b->kind = 1;
            fp += 1;
        }
    }
#pragma omp parallel for schedule(guided) if (parallel)
    for (int p = 0; p < (int)dense.size(); p++) {
        Block *b = dense[p];
        int M = b->t->size, N = b->s->size;
        b->dense.resize((size_t)M * N);
        A.copy_submatrix(M, N, b->t->offset, b->s->offset, b->dense.data());
        b->kind = 1;
    }
    H->false_positive  = fp;
    H->sym_for_leaves  = symmetry_for_leaves(*H->root, sym);
    if (H->root->is_leaf() && H->root->symmetry != 'N')
        H->sym_for_leaves = sym;
    H->uplo_for_leaves = H->sym_for_leaves != 'N' ? uplo : 'N';
    collect_leaves(*H);
    return H;
}

// ---------------------------------------------------------------------------------------------
// Leaf products: matrix/linalg/add_matrix_vector_product.hpp:10-35 (gemv / symv semantics restated
// as plain loops; BLAS is a third-party dependency, summation order unspecified),
// hmatrix/lrmat/linalg/add_lrmat_vector_product.hpp:9-24
// ---------------------------------------------------------------------------------------------
static void gemv(char trans, int m, int n, double alpha, const double *A, const double *x, double beta, double *y) {
    if (!(m && n))
        return;
    if (trans == 'N') {
        if (beta != 1.)
            for (int i = 0; i < m; i++)
                y[i] = beta == 0. ? 0. : beta * y[i];
        for (int j = 0; j < n; j++) {
            double t        = alpha * x[j];
            const double *a = A + (size_t)m * j;
            for (int i = 0; i < m; i++)
                y[i] += t * a[i];
        }
    } else {
        for (int j = 0; j < n; j++) {
            const double *a = A + (size_t)m * j;
            double t        = 0;
            for (int i = 0; i < m; i++)
                t += a[i] * x[i];
            y[j] = alpha * t + (beta == 0. ? 0. : beta * y[j]);
        }
    }
}
// symv: only the UPLO triangle of the n x n column-major matrix is referenced
static void symv(char uplo, int n, double alpha, const double *A, const double *x, double beta, double *y) {
    if (!n)
        return;
    if (beta != 1.)
        for (int i = 0; i < n; i++)
            y[i] = beta == 0. ? 0. : beta * y[i];
    for (int j = 0; j < n; j++) {
        double t1 = alpha * x[j], t2 = 0;
        if (uplo == 'L') {
            y[j] += t1 * A[j + (size_t)n * j];
            for (int i = j + 1; i < n; i++) {
                y[i] += t1 * A[i + (size_t)n * j];
                t2 += A[i + (size_t)n * j] * x[i];
            }
        } else {
            for (int i = 0; i < j; i++) {
                y[i] += t1 * A[i + (size_t)n * j];
                t2 += A[i + (size_t)n * j] * x[i];
            }
            y[j] += t1 * A[j + (size_t)n * j];
        }
        y[j] += alpha * t2;
    }
}
static void lrmat_vec(char trans, double alpha, const LowRank &lr, const double *in, double beta, double *out) {
    int r = lr.rank;
    if (r == 0)
        return; // beta NOT applied (add_lrmat_vector_product.hpp:11)
    std::vector<double> a(r);
    if (trans == 'N') {
        gemv('N', r, lr.N, 1., lr.V.data(), in, 0., a.data());
        gemv('N', lr.M, r, alpha, lr.U.data(), a.data(), beta, out);
    } else {
        gemv('T', lr.M, r, 1., lr.U.data(), in, 0., a.data());
        gemv('T', r, lr.N, alpha, lr.V.data(), a.data(), beta, out);
    }
}
// hmatrix/linalg/add_hmatrix_vector_product.hpp:17-33
static void leaf_vec(char trans, double alpha, const Block &b, const double *in, double beta, double *out) {
    if (b.kind == 1) {
        int M = b.t->size, N = b.s->size;
        if (b.symmetry == 'N')
            gemv(trans, M, N, alpha, b.dense.data(), in, beta, out);
        else
            symv(b.uplo, M, alpha, b.dense.data(), in, beta, out);
    } else if (b.kind == 2) {
        lrmat_vec(trans, alpha, b.lr, in, beta, out);
    }
}

// add_hmatrix_vector_product.hpp:57-104 (sequential) -- cluster numbering, local offsets
static void matvec_seq(const HMat &H, char trans, double alpha, const double *in, double beta, double *out) {
    int out_size = H.root_t->size;
    int lin = H.root_s->offset, lout = H.root_t->offset;
    char trans_sym = 'T';
    bool tr        = trans != 'N';
    if (tr) {
        out_size  = H.root_s->size;
        lin       = H.root_t->offset;
        lout      = H.root_s->offset;
        trans_sym = 'N';
    }
    if (beta != 1.)
        for (int i = 0; i < out_size; i++)
            out[i] = beta * out[i]; // scal
    for (auto &l : H.dfs_order) {
        int io = tr ? l.b->t->offset : l.b->s->offset;
        int oo = tr ? l.b->s->offset : l.b->t->offset;
        leaf_vec(trans, alpha, *l.b, in + io - lin, 1., out + (oo - lout));
    }
    if (H.sym_for_leaves != 'N') {
        for (auto &l : H.dfs_order) {
            if (!l.mirror)
                continue;
            int io = tr ? l.b->t->offset : l.b->s->offset;
            int oo = tr ? l.b->s->offset : l.b->t->offset;
            leaf_vec(trans_sym, alpha, *l.b, in + oo - lin, 1., out + (io - lout));
        }
    }
}
// add_hmatrix_vector_product.hpp:107-170 (OpenMP): per-thread temp with alpha=1 per leaf, critical axpy
static void matvec_omp(const HMat &H, char trans, double alpha, const double *in, double beta, double *out) {
    int out_size = H.root_t->size;
    int lin = H.root_s->offset, lout = H.root_t->offset;
    char trans_sym = 'T';
    bool tr        = trans != 'N';
    if (tr) {
        out_size  = H.root_s->size;
        lin       = H.root_t->offset;
        lout      = H.root_s->offset;
        trans_sym = 'N';
    }
    if (beta != 1.)
        for (int i = 0; i < out_size; i++)
            out[i] = beta * out[i];
    std::vector<const HMat::Leaf *> mirrors;
    for (auto &l : H.dfs_order)
        if (l.mirror)
            mirrors.push_back(&l);
#pragma omp parallel
    {
        std::vector<double> temp(out_size, 0.);
#pragma omp for schedule(guided) nowait
        for (int b = 0; b < (int)H.dfs_order.size(); b++) {
            auto &l = H.dfs_order[b];
            int io  = tr ? l.b->t->offset : l.b->s->offset;
            int oo  = tr ? l.b->s->offset : l.b->t->offset;
            leaf_vec(trans, 1., *l.b, in + io - lin, 1., temp.data() + (oo - lout));
        }
        if (H.sym_for_leaves != 'N') {
#pragma omp for schedule(guided) nowait
            for (int b = 0; b < (int)mirrors.size(); b++) {
                auto &l = *mirrors[b];
                int io  = tr ? l.b->t->offset : l.b->s->offset;
                int oo  = tr ? l.b->s->offset : l.b->t->offset;
                leaf_vec(trans_sym, 1., *l.b, in + oo - lin, 1., temp.data() + (io - lout));
            }
        }
#pragma omp critical
        for (int i = 0; i < out_size; i++)
            out[i] += alpha * temp[i];
    }
}

// Row-major multi-RHS: hmatrix/linalg/add_hmatrix_matrix_product_row_major.hpp:58-109,
// matrix/linalg/add_matrix_matrix_product_row_major.hpp:23-46,87-106,
// hmatrix/lrmat/linalg/add_lrmat_matrix_product_row_major.hpp:11-27.  X[n][mu], Y[m][mu] (mu fastest).
static void leaf_mat_rm(char trans, const Block &b, const double *in, double *out, int mu) {
    int M = b.t->size, N = b.s->size;
    auto dense_rm = [&](char tr, int m, int n, const double *A, const double *X, double *Y) {
        // Y[(out idx)][mu] += op(A) X
        if (tr == 'N') {
            for (int j = 0; j < n; j++)
                for (int i = 0; i < m; i++) {
                    double a = A[i + (size_t)m * j];
                    for (int c = 0; c < mu; c++)
                        Y[(size_t)i * mu + c] += a * X[(size_t)j * mu + c];
                }
        } else {
            for (int j = 0; j < n; j++)
                for (int i = 0; i < m; i++) {
                    double a = A[i + (size_t)m * j];
                    for (int c = 0; c < mu; c++)
                        Y[(size_t)j * mu + c] += a * X[(size_t)i * mu + c];
                }
        }
    };
    if (b.kind == 1) {
        if (b.symmetry == 'N') {
            dense_rm(trans, M, N, b.dense.data(), in, out);
        } else { // symm: only the UPLO triangle referenced
            for (int j = 0; j < N; j++)
                for (int i = 0; i < M; i++) {
                    bool stored = b.uplo == 'L' ? i >= j : i <= j;
                    double a    = stored ? b.dense[i + (size_t)M * j] : b.dense[j + (size_t)M * i];
                    for (int c = 0; c < mu; c++)
                        out[(size_t)i * mu + c] += a * in[(size_t)j * mu + c];
                }
        }
    } else if (b.kind == 2 && b.lr.rank > 0) {
        int r = b.lr.rank;
        std::vector<double> a((size_t)r * mu, 0.);
        if (trans == 'N') {
            dense_rm('N', r, N, b.lr.V.data(), in, a.data());
            dense_rm('N', M, r, b.lr.U.data(), a.data(), out);
        } else {
            dense_rm('T', M, r, b.lr.U.data(), in, a.data());
            dense_rm('T', r, N, b.lr.V.data(), a.data(), out);
        }
    }
}
static void matmat_rm_seq(const HMat &H, char trans, double alpha, const double *in, double beta, double *out, int mu) {
    int out_size = H.root_t->size;
    int lin = H.root_s->offset, lout = H.root_t->offset;
    char trans_sym = 'T';
    bool tr        = trans != 'N';
    if (tr) {
        out_size  = H.root_s->size;
        lin       = H.root_t->offset;
        lout      = H.root_s->offset;
        trans_sym = 'N';
    }
    size_t tot = (size_t)out_size * mu;
    if (beta != 1.)
        for (size_t i = 0; i < tot; i++)
            out[i] = beta * out[i];
    std::vector<double> temp(tot, 0.);
    for (auto &l : H.dfs_order) {
        int io = tr ? l.b->t->offset : l.b->s->offset;
        int oo = tr ? l.b->s->offset : l.b->t->offset;
        leaf_mat_rm(trans, *l.b, in + (size_t)(io - lin) * mu, temp.data() + (size_t)(oo - lout) * mu, mu);
    }
    if (H.sym_for_leaves != 'N')
        for (auto &l : H.dfs_order) {
            if (!l.mirror)
                continue;
            int io = tr ? l.b->t->offset : l.b->s->offset;
            int oo = tr ? l.b->s->offset : l.b->t->offset;
            leaf_mat_rm(trans_sym, *l.b, in + (size_t)(oo - lin) * mu, temp.data() + (size_t)(io - lout) * mu, mu);
        }
    for (size_t i = 0; i < tot; i++)
        out[i] += alpha * temp[i];
}

} // namespace orc

// =============================================================================================
// C ABI (ctypes)
// =============================================================================================
using namespace orc;
extern "C" {

void orc_geometry(const char *name, int n, double z, double *out) {
    std::string g(name);
    if (g == "disk2d")
        rotated_ellipse(2, 1., 1., 0., z, n, out);
    else if (g == "ellipse")
        rotated_ellipse(3, 4., 1., 0., z, n, out);
    else if (g == "disk")
        rotated_ellipse(3, 1., 1., 0., z, n, out);
    else if (g == "ball")
        ball(n, out);
}

void *orc_cluster_create(int n, int dim, const double *coords, int leaf, int children, int partitions, int direction, int splitting, int partition_n) {
    Strategy st{direction, splitting, partition_n};
    return create_cluster_tree(n, dim, coords, leaf, children, partitions, st).release();
}
void orc_cluster_destroy(void *h) { delete static_cast<ClusterTree *>(h); }
static void preorder_nodes(const Cluster &c, const std::function<void(const Cluster &)> &f) {
    f(c);
    for (auto &ch : c.children)
        preorder_nodes(*ch, f);
}
int orc_cluster_num_nodes(void *h) {
    int n = 0;
    preorder_nodes(*static_cast<ClusterTree *>(h)->root, [&](const Cluster &) { n++; });
    return n;
}
int orc_cluster_num_partitions(void *h) { return static_cast<ClusterTree *>(h)->td.on_partition.size(); }
void orc_cluster_get(void *h, int *perm, int *nodes_int, double *nodes_real, int *partition) {
    auto *T = static_cast<ClusterTree *>(h);
    std::copy(T->td.perm.begin(), T->td.perm.end(), perm);
    int i = 0;
    preorder_nodes(*T->root, [&](const Cluster &c) {
        int *p = nodes_int + 6 * i;
        p[0]   = c.depth;
        p[1]   = c.offset;
        p[2]   = c.size;
        p[3]   = c.rank;
        p[4]   = c.counter;
        p[5]   = c.children.size();
        double *r = nodes_real + 4 * i;
        r[0]      = c.radius;
        for (int q = 0; q < 3; q++)
            r[1 + q] = q < (int)c.center.size() ? c.center[q] : 0.;
        i++;
    });
    for (size_t k = 0; k < T->td.on_partition.size(); k++) {
        partition[2 * k]     = T->td.on_partition[k]->offset;
        partition[2 * k + 1] = T->td.on_partition[k]->size;
    }
}

struct OracleH {
    std::unique_ptr<HMat> H;
    Generator gen;
};

void *orc_hmatrix_build(void *tct, void *sct, int dim, const double *xt, const double *xs, double delta, double scale, double eps, double eta, char sym, char uplo, int reqrank, int compressor, int mint, int mins, int target_partition, int partition_for_symmetry, int consistent, int parallel) {
    auto *T = static_cast<ClusterTree *>(tct);
    auto *S = static_cast<ClusterTree *>(sct);
    auto *o = new OracleH();
    o->gen  = Generator{dim, xt, xs, T->td.perm.data(), S->td.perm.data(), delta, scale};
    o->H    = build_hmatrix(*T, *S, o->gen, eps, eta, sym, uplo, reqrank, compressor, mint, mins, target_partition, partition_for_symmetry, consistent != 0, parallel != 0);
    return o;
}
void orc_hmatrix_destroy(void *h) { delete static_cast<OracleH *>(h); }
int orc_hmatrix_num_leaves(void *h) { return static_cast<OracleH *>(h)->H->preorder.size(); }
// n x 6: t_off t_size s_off s_size rank(-1 dense) mirror   (preorder; order=1 => get_leaves_from order)
void orc_hmatrix_leaves(void *h, int order, int *out) {
    auto &H  = *static_cast<OracleH *>(h)->H;
    auto &ls = order ? H.dfs_order : H.preorder;
    int i    = 0;
    for (auto &l : ls) {
        int *p = out + 6 * i++;
        p[0]   = l.b->t->offset;
        p[1]   = l.b->t->size;
        p[2]   = l.b->s->offset;
        p[3]   = l.b->s->size;
        p[4]   = l.b->kind == 2 ? l.b->lr.rank : -1;
        p[5]   = l.mirror ? 1 : 0;
    }
}
void orc_hmatrix_rootinfo(void *h, int *out) {
    auto &H = *static_cast<OracleH *>(h)->H;
    out[0]  = H.root_t->offset;
    out[1]  = H.root_t->size;
    out[2]  = H.root_s->offset;
    out[3]  = H.root_s->size;
    out[4]  = H.false_positive;
    out[5]  = H.sym_for_leaves;
    out[6]  = H.uplo_for_leaves;
}
// copy payload of preorder leaf b: dense -> D (M*N), low rank -> U (M*r), V (r*N); returns rank or -1
int orc_hmatrix_block(void *h, int b, double *U, double *V, double *D, int *pivots) {
    auto &H = *static_cast<OracleH *>(h)->H;
    Block *B = H.preorder[b].b;
    if (B->kind == 2) {
        if (U)
            std::copy(B->lr.U.begin(), B->lr.U.end(), U);
        if (V)
            std::copy(B->lr.V.begin(), B->lr.V.end(), V);
        if (pivots)
            std::copy(B->lr.pivots.begin(), B->lr.pivots.end(), pivots);
        return B->lr.rank;
    }
    if (D)
        std::copy(B->dense.begin(), B->dense.end(), D);
    return -1;
}
void orc_hmatrix_matvec(void *h, int policy, char trans, double alpha, const double *in, double beta, double *out) {
    auto &H = *static_cast<OracleH *>(h)->H;
    if (policy == 0)
        matvec_seq(H, trans, alpha, in, beta, out);
    else
        matvec_omp(H, trans, alpha, in, beta, out);
}
void orc_hmatrix_matmat_row_major(void *h, char trans, double alpha, const double *in, double beta, double *out, int mu) {
    matmat_rm_seq(*static_cast<OracleH *>(h)->H, trans, alpha, in, beta, out, mu);
}

// Stand-alone compression of one block (cluster numbering offsets); returns rank (0 = failure)
int orc_compress_block(void *tct, void *sct, int dim, const double *xt, const double *xs, double delta, double scale, int compressor, int M, int N, int row_off, int col_off, double eps, int reqrank, double *U, double *V, int *pivots, double *sing) {
    auto *T = static_cast<ClusterTree *>(tct);
    auto *S = static_cast<ClusterTree *>(sct);
    Generator g{dim, xt, xs, T->td.perm.data(), S->td.perm.data(), delta, scale};
    LowRank lr;
    bool ok;
    if (compressor == SVD) {
        std::vector<double> s;
        ok = svd_compress(g, M, N, row_off, col_off, eps, reqrank, lr, &s);
        if (sing)
            std::copy(s.begin(), s.end(), sing);
    } else {
        ok = compress(compressor, g, M, N, row_off, col_off, eps, reqrank, lr);
    }
    if (!ok)
        return 0;
    std::copy(lr.U.begin(), lr.U.end(), U);
    std::copy(lr.V.begin(), lr.V.end(), V);
    if (pivots)
        std::copy(lr.pivots.begin(), lr.pivots.end(), pivots);
    return lr.rank;
}

// Dense generator block in cluster numbering (column-major), for dense references in tests
void orc_generate_block(void *tct, void *sct, int dim, const double *xt, const double *xs, double delta, double scale, int M, int N, int row_off, int col_off, double *out) {
    auto *T = static_cast<ClusterTree *>(tct);
    auto *S = static_cast<ClusterTree *>(sct);
    Generator g{dim, xt, xs, T->td.perm.data(), S->td.perm.data(), delta, scale};
    g.copy_submatrix(M, N, row_off, col_off, out);
}

// Build an oracle H-matrix from a flat leaf list + payloads (e.g. blocks compressed by the HIP engine),
// so the reference's leaf loop can multiply with exactly the engine's data (SURVEY.md 8c, (b) get/set).
// desc: nb x 6 ints (t_off t_size s_off s_size rank(-1 dense) mirror); payload offsets in doubles into
// `data` (nb x 2: U or D offset, V offset); root = (rt_off, rt_size, rs_off, rs_size); leaves are used in
// the given order.
void *orc_hmatrix_from_blocks(int nb, const int *desc, const int64_t *offs, const double *data, const int *root, char sym_for_leaves, char uplo) {
    auto *o = new OracleH();
    o->H    = std::make_unique<HMat>();
    auto &H = *o->H;
    auto mk = [&](int off, int size) {
        auto c    = std::make_unique<Cluster>();
        c->offset = off;
        c->size   = size;
        c->rank   = 0;
        H.owned_clusters.push_back(std::move(c));
        return H.owned_clusters.back().get();
    };
    H.root_t          = mk(root[0], root[1]);
    H.root_s          = mk(root[2], root[3]);
    H.sym_for_leaves  = sym_for_leaves;
    H.uplo_for_leaves = uplo;
    for (int b = 0; b < nb; b++) {
        const int *d = desc + 6 * b;
        auto B       = std::make_unique<Block>();
        B->t         = mk(d[0], d[1]);
        B->s         = mk(d[2], d[3]);
        int M = d[1], N = d[3];
        if (sym_for_leaves != 'N' && d[0] == d[2] && d[1] == d[3]) {
            B->symmetry = sym_for_leaves;
            B->uplo     = uplo;
        }
        if (d[4] < 0) {
            B->kind = 1;
            B->dense.assign(data + offs[2 * b], data + offs[2 * b] + (size_t)M * N);
        } else {
            B->kind    = 2;
            B->lr.M    = M;
            B->lr.N    = N;
            B->lr.rank = d[4];
            B->lr.U.assign(data + offs[2 * b], data + offs[2 * b] + (size_t)M * d[4]);
            B->lr.V.assign(data + offs[2 * b + 1], data + offs[2 * b + 1] + (size_t)N * d[4]);
        }
        H.dfs_order.push_back({B.get(), d[5] != 0});
        H.preorder.push_back({B.get(), d[5] != 0});
        H.owned_blocks.push_back(std::move(B));
    }
    return o;
}

int orc_num_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
} // extern "C"

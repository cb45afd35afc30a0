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
#include <complex>
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
// partition / partition_kind: user-given partition (tree_builder.hpp:87-123): 1 = global (one part number per point),
// 2 = local ((offset, size) per part); is_complete: set_is_complete (tree_builder.hpp:176-192)
static std::unique_ptr<ClusterTree> create_cluster_tree(int n, int dim, const double *x, int leaf_size, int nchildren, int size_partition, const Strategy &st, const int *partition = nullptr, int partition_kind = 0, bool is_complete = false) {
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
    const bool given = partition && partition_kind != 0;
    if (given) {
        depth_of_partition = 1;
        stack.pop();
        std::vector<int> offsets(size_partition), sizes(size_partition);
        if (partition_kind == 2) {
            tree->td.perm_local = true;
            for (int p = 0; p < size_partition; p++) {
                offsets[p] = partition[2 * p];
                sizes[p]   = partition[2 * p + 1];
            }
        } else {
            int cpt    = 0;
            bool local = true;
            for (int p = 0; p < size_partition; p++) {
                offsets[p] = cpt;
                sizes[p]   = 0;
                int prev   = -1;
                for (int i = 0; i < n; i++)
                    if (partition[i] == p) {
                        perm[cpt] = i;
                        sizes[p]++;
                        cpt++;
                        local = local && (prev < 0 || prev == i - 1);
                        prev  = i;
                    }
            }
            tree->td.perm_local = local;
        }
        for (int p = 0; p < size_partition; p++) {
            center = compute_center(dim, x, weights.data(), offsets[p], sizes[p], perm.data());
            radius = compute_radius(dim, x, radii.data(), center, offsets[p], sizes[p], perm.data());
            stack.push(root.add_child(radius, center, p, offsets[p], sizes[p], p, true));
        }
    }
    if (size_partition == 1)
        tree->td.perm_local = true;

    while (!stack.empty()) {
        Cluster *cur = stack.top();
        stack.pop();
        bool on_level = !given && (cur->depth == depth_of_partition - 1);
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
            if (is_complete) {
                if (std::any_of(children.begin(), children.end(), [&](Cluster *a) { return a->size > leaf_size; }))
                    for (auto *ch : children)
                        stack.push(ch);
            } else {
                for (auto *ch : children)
                    if (ch->size > leaf_size)
                        stack.push(ch);
            }
        }
    }
    return tree;
}

} // namespace orc

namespace orc {
// kernel family of the generators built from now on (orc_set_kernel_family; include/hmx.h hmx_kernel) and the sin / cos of the Helmholtz
// phase: the device kernel's documented IEEE sequence restated (three-constant Cody-Waite reduction, minimax polynomials with fdlibm's
// coefficients; htool_amd/csrc/kernels_common.hpp hmx_sincos describes it) -- same operations in the same order give the same bits
static int g_family         = 0;
static double g_wavenumber = 0;
static inline int orc_kernel_family() { return g_family; }
static inline double orc_kernel_wavenumber() { return g_wavenumber; }
static inline void orc_sincos(double x, double &sn, double &cs) {
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
#define ORC_COMPLEX 0
namespace f64 {
using real   = double;
using scalar = double;
#include "hmx_oracle_body.hpp"
} // namespace f64
namespace f32 {
using real   = float;
using scalar = float;
#include "hmx_oracle_body.hpp"
} // namespace f32
#undef ORC_COMPLEX
#define ORC_COMPLEX 1
namespace z64 { // htool's HMatrix<std::complex<double>, double>
using real   = double;
using scalar = std::complex<double>;
#include "hmx_oracle_body.hpp"
} // namespace z64
namespace c32 { // HMatrix<std::complex<float>, double>
using real   = float;
using scalar = std::complex<float>;
#include "hmx_oracle_body.hpp"
} // namespace c32
#undef ORC_COMPLEX
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
void *orc_cluster_create_ex(int n, int dim, const double *coords, int leaf, int children, int partitions, int direction, int splitting, int partition_n, const int *partition, int partition_kind, int is_complete) {
    Strategy st{direction, splitting, partition_n};
    return create_cluster_tree(n, dim, coords, leaf, children, partitions, st, partition, partition_kind, is_complete != 0).release();
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
// (a rank whose cluster was never created -- a branch that became a leaf above the partition level, tree_builder.hpp:185-192 -- has no entry)
int orc_cluster_num_partitions(void *h) {
    int n = 0;
    for (const Cluster *c : static_cast<ClusterTree *>(h)->td.on_partition)
        n += c != nullptr;
    return n;
}
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
    size_t w = 0;
    for (size_t k = 0; k < T->td.on_partition.size(); k++)
        if (T->td.on_partition[k]) {
            partition[2 * w]     = T->td.on_partition[k]->offset;
            partition[2 * w + 1] = T->td.on_partition[k]->size;
            w++;
        }
}

struct OracleH {
    bool f32 = false;
    std::unique_ptr<orc::f64::HMat> d;
    std::unique_ptr<orc::f32::HMat> s;
    orc::f64::Generator gd;
    orc::f32::Generator gs;
};
extern "C++" {
template <typename F>
static auto with_h(void *h, F &&f) {
    auto *o = static_cast<OracleH *>(h);
    return o->f32 ? f(*o->s) : f(*o->d);
}
}

void *orc_hmatrix_build(void *tct, void *sct, int dim, const double *xt, const double *xs, double delta, double scale, double eps, double eta, char sym, char uplo, int reqrank, int compressor, int mint, int mins, int target_partition, int partition_for_symmetry, int consistent, int parallel, int f32, int root_partition) {
    auto *T = static_cast<ClusterTree *>(tct);
    auto *S = static_cast<ClusterTree *>(sct);
    auto *o = new OracleH();
    o->f32  = f32 != 0;
    if (o->f32) {
        o->gs = orc::f32::Generator{dim, xt, xs, T->td.perm.data(), S->td.perm.data(), delta, scale};
        o->s  = orc::f32::build_hmatrix(*T, *S, o->gs, (float)eps, eta, sym, uplo, reqrank, compressor, mint, mins, target_partition, partition_for_symmetry, consistent != 0, parallel != 0, root_partition);
    } else {
        o->gd = orc::f64::Generator{dim, xt, xs, T->td.perm.data(), S->td.perm.data(), delta, scale};
        o->d  = orc::f64::build_hmatrix(*T, *S, o->gd, eps, eta, sym, uplo, reqrank, compressor, mint, mins, target_partition, partition_for_symmetry, consistent != 0, parallel != 0, root_partition);
    }
    return o;
}
void orc_hmatrix_destroy(void *h) { delete static_cast<OracleH *>(h); }
int orc_hmatrix_num_leaves(void *h) {
    return with_h(h, [](auto &H) { return (int)H.preorder.size(); });
}
// n x 6: t_off t_size s_off s_size rank(-1 dense) mirror   (preorder; order=1 => get_leaves_from order)
void orc_hmatrix_leaves(void *h, int order, int *out) {
    with_h(h, [&](auto &H) {
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
        return 0;
    });
}
void orc_hmatrix_rootinfo(void *h, int *out) {
    with_h(h, [&](auto &H) {
        out[0] = H.root_t->offset;
        out[1] = H.root_t->size;
        out[2] = H.root_s->offset;
        out[3] = H.root_s->size;
        out[4] = H.false_positive;
        out[5] = H.sym_for_leaves;
        out[6] = H.uplo_for_leaves;
        return 0;
    });
}
// copy payload of preorder leaf b (as double): dense -> D (M*N), low rank -> U (M*r), V (r*N); returns rank or -1
int orc_hmatrix_block(void *h, int b, double *U, double *V, double *D, int *pivots) {
    return with_h(h, [&](auto &H) {
        auto *B = H.preorder[b].b;
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
    });
}
// recompression(hmatrix) (hmatrix/utils/recompression.hpp:8-13)
void orc_hmatrix_recompress(void *h, double epsilon) {
    with_h(h, [&](auto &H) {
        using R = typename std::decay<decltype(H.preorder[0].b->dense[0])>::type;
        for (auto &l : H.preorder)
            if (l.b->kind == 2)
                svd_recompression(l.b->lr, (R)epsilon);
        return 0;
    });
}
void orc_hmatrix_matvec(void *h, int policy, char trans, double alpha, const double *in, double beta, double *out) {
    with_h(h, [&](auto &H) {
        using R        = typename std::decay<decltype(H.preorder[0].b->dense[0])>::type;
        const int nin  = trans == 'N' ? H.root_s->size : H.root_t->size;
        const int nout = trans == 'N' ? H.root_t->size : H.root_s->size;
        std::vector<R> x(in, in + nin), y(out, out + nout);
        if (policy == 0)
            matvec_seq(H, trans, (R)alpha, x.data(), (R)beta, y.data());
        else
            matvec_omp(H, trans, (R)alpha, x.data(), (R)beta, y.data());
        std::copy(y.begin(), y.end(), out);
        return 0;
    });
}
void orc_hmatrix_matmat_row_major(void *h, char trans, double alpha, const double *in, double beta, double *out, int mu) {
    with_h(h, [&](auto &H) {
        using R        = typename std::decay<decltype(H.preorder[0].b->dense[0])>::type;
        const size_t nin  = (size_t)(trans == 'N' ? H.root_s->size : H.root_t->size) * mu;
        const size_t nout = (size_t)(trans == 'N' ? H.root_t->size : H.root_s->size) * mu;
        std::vector<R> x(in, in + nin), y(out, out + nout);
        matmat_rm_seq(H, trans, (R)alpha, x.data(), (R)beta, y.data(), mu);
        std::copy(y.begin(), y.end(), out);
        return 0;
    });
}

// Stand-alone compression of one block (cluster numbering offsets); returns rank (0 = failure)
extern "C++" {
template <typename NS_Gen, typename LR, typename CompressFn, typename SvdFn>
static int compress_block_impl(NS_Gen &g, int compressor, int M, int N, int row_off, int col_off, double eps, int reqrank, double *U, double *V, int *pivots, double *sing, CompressFn compress_fn, SvdFn svd_fn, LR lr) {
    bool ok;
    using R = typename std::decay<decltype(lr.U[0])>::type;
    if (compressor == 3) {
        std::vector<R> s;
        ok = svd_fn(g, M, N, row_off, col_off, (R)eps, reqrank, lr, &s);
        if (sing)
            std::copy(s.begin(), s.end(), sing);
    } else {
        ok = compress_fn(compressor, g, M, N, row_off, col_off, (R)eps, reqrank, lr);
    }
    if (!ok)
        return 0;
    std::copy(lr.U.begin(), lr.U.end(), U);
    std::copy(lr.V.begin(), lr.V.end(), V);
    if (pivots)
        std::copy(lr.pivots.begin(), lr.pivots.end(), pivots);
    return lr.rank;
}
}
int orc_compress_block(void *tct, void *sct, int dim, const double *xt, const double *xs, double delta, double scale, int compressor, int M, int N, int row_off, int col_off, double eps, int reqrank, double *U, double *V, int *pivots, double *sing, int f32) {
    auto *T = static_cast<ClusterTree *>(tct);
    auto *S = static_cast<ClusterTree *>(sct);
    if (f32) {
        orc::f32::Generator g{dim, xt, xs, T->td.perm.data(), S->td.perm.data(), delta, scale};
        return compress_block_impl(g, compressor, M, N, row_off, col_off, eps, reqrank, U, V, pivots, sing,
                                   [](int k, const orc::f32::Generator &A, int M_, int N_, int ro, int co, float e, int rq, orc::f32::LowRank &l) { return orc::f32::compress(k, A, M_, N_, ro, co, e, rq, l); },
                                   [](const orc::f32::Generator &A, int M_, int N_, int ro, int co, float e, int rq, orc::f32::LowRank &l, std::vector<float> *s) { return orc::f32::svd_compress(A, M_, N_, ro, co, e, rq, l, s); }, orc::f32::LowRank());
    }
    orc::f64::Generator g{dim, xt, xs, T->td.perm.data(), S->td.perm.data(), delta, scale};
    return compress_block_impl(g, compressor, M, N, row_off, col_off, eps, reqrank, U, V, pivots, sing,
                               [](int k, const orc::f64::Generator &A, int M_, int N_, int ro, int co, double e, int rq, orc::f64::LowRank &l) { return orc::f64::compress(k, A, M_, N_, ro, co, e, rq, l); },
                               [](const orc::f64::Generator &A, int M_, int N_, int ro, int co, double e, int rq, orc::f64::LowRank &l, std::vector<double> *s) { return orc::f64::svd_compress(A, M_, N_, ro, co, e, rq, l, s); }, orc::f64::LowRank());
}

// Dense generator block in cluster numbering (column-major), for dense references in tests
// kernel family of every generator constructed afterwards: 0 inverse distance, 1 Helmholtz (wavenumber), 2 Laplace single layer
void orc_set_kernel_family(int family, double wavenumber) {
    orc::g_family     = family;
    orc::g_wavenumber = wavenumber;
}
void orc_generate_block(void *tct, void *sct, int dim, const double *xt, const double *xs, double delta, double scale, int M, int N, int row_off, int col_off, double *out) {
    auto *T = static_cast<ClusterTree *>(tct);
    auto *S = static_cast<ClusterTree *>(sct);
    orc::f64::Generator g{dim, xt, xs, T->td.perm.data(), S->td.perm.data(), delta, scale};
    g.copy_submatrix(M, N, row_off, col_off, out);
}

// Build an oracle H-matrix from a flat leaf list + payloads (e.g. blocks compressed by the HIP engine),
// so the reference's leaf loop can multiply with exactly the engine's data (SURVEY.md 8c, (b) get/set).
// desc: nb x 6 ints (t_off t_size s_off s_size rank(-1 dense) mirror); payload offsets in doubles into
// `data` (nb x 2: U or D offset, V offset); root = (rt_off, rt_size, rs_off, rs_size); leaves are used in
// the given order.  f32 != 0: payloads are rounded to float and the leaf loop runs in float.
extern "C++" {
template <typename HM, typename BL>
static void from_blocks_impl(HM &H, BL *, int nb, const int *desc, const int64_t *offs, const double *data, const int *root, char sym_for_leaves, char uplo) {
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
        auto B       = std::make_unique<BL>();
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
}
}
void *orc_hmatrix_from_blocks(int nb, const int *desc, const int64_t *offs, const double *data, const int *root, char sym_for_leaves, char uplo, int f32) {
    auto *o = new OracleH();
    o->f32  = f32 != 0;
    if (o->f32) {
        o->s = std::make_unique<orc::f32::HMat>();
        from_blocks_impl(*o->s, (orc::f32::Block *)nullptr, nb, desc, offs, data, root, sym_for_leaves, uplo);
    } else {
        o->d = std::make_unique<orc::f64::HMat>();
        from_blocks_impl(*o->d, (orc::f64::Block *)nullptr, nb, desc, offs, data, root, sym_for_leaves, uplo);
    }
    return o;
}

// ---- complex coefficients: the same restatement instantiated for std::complex; vectors and payloads
// cross the ABI as interleaved (re, im) doubles -----------------------------------------------------------------
struct OracleZ {
    bool c32 = false;
    std::unique_ptr<orc::z64::HMat> z;
    std::unique_ptr<orc::c32::HMat> c;
    orc::z64::Generator gz;
    orc::c32::Generator gc;
};
extern "C++" {
template <typename F>
static auto with_z(void *h, F &&f) {
    auto *o = static_cast<OracleZ *>(h);
    return o->c32 ? f(*o->c) : f(*o->z);
}
template <typename C>
static std::vector<C> z_in(const double *p, size_t n) {
    std::vector<C> v(n);
    for (size_t i = 0; i < n; i++)
        v[i] = C((typename C::value_type)p[2 * i], (typename C::value_type)p[2 * i + 1]);
    return v;
}
template <typename It>
static void z_out(It b, It e, double *p) {
    for (size_t i = 0; b != e; ++b, ++i) {
        p[2 * i]     = b->real();
        p[2 * i + 1] = b->imag();
    }
}
}
void *orc_zhmatrix_build(void *tct, void *sct, int dim, const double *xt, const double *xs, double delta, double scale, double cre, double cim, double eps, double eta, char sym, char uplo, int reqrank, int compressor, int mint, int mins, int target_partition, int partition_for_symmetry, int consistent, int parallel, int c32, int root_partition) {
    auto *T = static_cast<ClusterTree *>(tct);
    auto *S = static_cast<ClusterTree *>(sct);
    auto *o = new OracleZ();
    o->c32  = c32 != 0;
    if (o->c32) {
        o->gc = orc::c32::Generator{dim, xt, xs, T->td.perm.data(), S->td.perm.data(), delta, scale, cre, cim, sym == 'H'};
        o->c  = orc::c32::build_hmatrix(*T, *S, o->gc, (float)eps, eta, sym, uplo, reqrank, compressor, mint, mins, target_partition, partition_for_symmetry, consistent != 0, parallel != 0, root_partition);
    } else {
        o->gz = orc::z64::Generator{dim, xt, xs, T->td.perm.data(), S->td.perm.data(), delta, scale, cre, cim, sym == 'H'};
        o->z  = orc::z64::build_hmatrix(*T, *S, o->gz, eps, eta, sym, uplo, reqrank, compressor, mint, mins, target_partition, partition_for_symmetry, consistent != 0, parallel != 0, root_partition);
    }
    return o;
}
void orc_zhmatrix_destroy(void *h) { delete static_cast<OracleZ *>(h); }
int orc_zhmatrix_num_leaves(void *h) {
    return with_z(h, [](auto &H) { return (int)H.preorder.size(); });
}
void orc_zhmatrix_leaves(void *h, int order, int *out) {
    with_z(h, [&](auto &H) {
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
        return 0;
    });
}
void orc_zhmatrix_rootinfo(void *h, int *out) {
    with_z(h, [&](auto &H) {
        out[0] = H.root_t->offset;
        out[1] = H.root_t->size;
        out[2] = H.root_s->offset;
        out[3] = H.root_s->size;
        out[4] = H.false_positive;
        out[5] = H.sym_for_leaves;
        out[6] = H.uplo_for_leaves;
        return 0;
    });
}
int orc_zhmatrix_block(void *h, int b, double *U, double *V, double *D) {
    return with_z(h, [&](auto &H) {
        auto *B = H.preorder[b].b;
        if (B->kind == 2) {
            if (U)
                z_out(B->lr.U.begin(), B->lr.U.end(), U);
            if (V)
                z_out(B->lr.V.begin(), B->lr.V.end(), V);
            return B->lr.rank;
        }
        if (D)
            z_out(B->dense.begin(), B->dense.end(), D);
        return -1;
    });
}
void orc_zhmatrix_recompress(void *h, double epsilon) {
    with_z(h, [&](auto &H) {
        using C = typename std::decay<decltype(H.preorder[0].b->dense[0])>::type;
        for (auto &l : H.preorder)
            if (l.b->kind == 2)
                svd_recompression(l.b->lr, (typename C::value_type)epsilon);
        return 0;
    });
}
void orc_zhmatrix_matvec(void *h, int policy, char trans, const double *alpha, const double *in, const double *beta, double *out) {
    with_z(h, [&](auto &H) {
        using C        = typename std::decay<decltype(H.preorder[0].b->dense[0])>::type;
        const int nin  = trans == 'N' ? H.root_s->size : H.root_t->size;
        const int nout = trans == 'N' ? H.root_t->size : H.root_s->size;
        std::vector<C> x = z_in<C>(in, nin), y = z_in<C>(out, nout);
        const C al = z_in<C>(alpha, 1)[0], be = z_in<C>(beta, 1)[0];
        if (policy == 0)
            matvec_seq(H, trans, al, x.data(), be, y.data());
        else
            matvec_omp(H, trans, al, x.data(), be, y.data());
        z_out(y.begin(), y.end(), out);
        return 0;
    });
}
void orc_zhmatrix_matmat_row_major(void *h, char trans, const double *alpha, const double *in, const double *beta, double *out, int mu) {
    with_z(h, [&](auto &H) {
        using C           = typename std::decay<decltype(H.preorder[0].b->dense[0])>::type;
        const size_t nin  = (size_t)(trans == 'N' ? H.root_s->size : H.root_t->size) * mu;
        const size_t nout = (size_t)(trans == 'N' ? H.root_t->size : H.root_s->size) * mu;
        std::vector<C> x = z_in<C>(in, nin), y = z_in<C>(out, nout);
        matmat_rm_seq(H, trans, z_in<C>(alpha, 1)[0], x.data(), z_in<C>(beta, 1)[0], y.data(), mu);
        z_out(y.begin(), y.end(), out);
        return 0;
    });
}
void orc_zgenerate_block(void *tct, void *sct, int dim, const double *xt, const double *xs, double delta, double scale, double cre, double cim, int hermitian, int M, int N, int row_off, int col_off, double *out) {
    auto *T = static_cast<ClusterTree *>(tct);
    auto *S = static_cast<ClusterTree *>(sct);
    orc::z64::Generator g{dim, xt, xs, T->td.perm.data(), S->td.perm.data(), delta, scale, cre, cim, hermitian};
    std::vector<std::complex<double>> tmp((size_t)M * N);
    g.copy_submatrix(M, N, row_off, col_off, tmp.data());
    z_out(tmp.begin(), tmp.end(), out);
}

int orc_num_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
// threads of the OpenMP leaf loops from now on (bench.py: the cores the process really has under a cgroup quota, not the hardware threads)
void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0)
        omp_set_num_threads(n);
#else
    (void)n;
#endif
}
} // extern "C"

// cluster_tree.cpp -- geometric cluster tree on the host, bit-exact with htool's
// ClusterTreeBuilder::create_cluster_tree (clustering/tree_builder/tree_builder.hpp:52-207) for the
// "simple" partition type, all of Partitioning / Partitioning_N x ComputeLargestExtent /
// ComputeBoundingBox x RegularSplitting / GeometricSplitting (clustering/implementations/partitioning.hpp).
//
// Not a transcription: the tree is a flat node array built level by level, every node of a level is
// split concurrently (disjoint permutation slices), and the projection keys are computed once per
// point instead of twice per comparison.  What IS kept identical is every floating-point expression's
// operation order (this file is compiled with -ffp-contract=off) and the use of libstdc++'s
// std::sort on the same comparison outcomes, so permutation, radii and centres match bit for bit.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <memory>
#include <mutex>
#include <numeric>
#include <thread>
#include <utility>

#include "hmx_host.hpp"

namespace hmx {
namespace {

using Parts = std::vector<std::pair<int, int>>; // (offset, size)

struct Frame {      // direction matrix (column-major dim x dim) + per-direction extents
    double axis[9]; // axis[p + dim*c] = component p of direction c
    double extent[3];
};

struct Builder {
    int n, dim;
    const double *x, *radii, *w;
    ClusterTreeOptions opt;
    std::vector<int32_t> &perm;
    std::vector<double> key; // projection of point id on the current direction (only filled when a slice falls back to the reference's sort)
    // The points' coordinates in the order of the CURRENT permutation, xs[dim * position + p]: every pass over a cluster (centre, radius,
    // covariance, projections) then streams contiguous memory instead of gathering x[perm[...]] from all over the array -- below the top
    // levels that gather was the whole cost of the tree (a slice of 31 250 points touches 2 MB of cache lines spread over 24 MB).
    // The values and the order they are accumulated in are unchanged, so every node is bit-identical to the gathered form.
    std::unique_ptr<double[]> xs;
    // scratch of order_along, one slot per POSITION (slices of one level are disjoint, so the nodes of a level share the arrays): allocated
    // once, not initialised -- fresh (zeroed) buffers per call cost the top levels more in page faults than the sort itself
    struct KeyPos {
        double first;
        int second;
    };
    std::unique_ptr<KeyPos[]> sk_a, sk_b;
    std::unique_ptr<int32_t[]> sk_ids;
    std::unique_ptr<double[]> sk_x;

    Builder(int n_, int dim_, const double *x_, const double *r_, const double *w_, const ClusterTreeOptions &o, std::vector<int32_t> &p)
        : n(n_), dim(dim_), x(x_), radii(r_), w(w_), opt(o), perm(p) {}
    void gather_xs(int off, int size) { // after the permutation of a slice was set from outside
        if (!xs) {
            xs.reset(new double[(size_t)n * dim]);
            sk_a.reset(new KeyPos[n]);
            sk_b.reset(new KeyPos[n]);
            sk_ids.reset(new int32_t[n]);
            sk_x.reset(new double[(size_t)n * dim]);
        }
        auto copy = [&](int lo, int hi, int) {
            for (int j = off + lo; j < off + hi; j++)
                for (int p = 0; p < dim; p++)
                    xs[(size_t)dim * j + p] = x[(size_t)dim * perm[j] + p];
        };
        if (size >= psort_min && psort_threads > 1)
            in_chunks(size, copy);
        else
            copy(0, size, 0);
    }

    // tree_builder.hpp:210-233 -- weighted mean, accumulate j then p, multiply by 1/total
    void centroid(int off, int size, double *c) const {
        double total = 0;
        for (int j = 0; j < size; j++)
            total += w ? w[perm[off + j]] : 1.0;
        for (int p = 0; p < dim; p++)
            c[p] = 0;
        const double *xo = xs.get() + (size_t)dim * off;
        for (int j = 0; j < size; j++) {
            const double wid = w ? w[perm[off + j]] : 1.0;
            for (int p = 0; p < dim; p++)
                c[p] += wid * xo[dim * j + p];
        }
        const double inv = 1.0 / total;
        for (int p = 0; p < dim; p++)
            c[p] = c[p] * inv;
    }
    // tree_builder.hpp:236-253 -- max_j( sqrt(|sum u^2|) + radii_j )
    double bounding_radius(int off, int size, const double *c) const {
        if (size >= psort_min && psort_threads > 1) { // a maximum does not depend on the order it is taken in
            std::vector<double> part(psort_threads, 0.0);
            in_chunks(size, [&](int lo, int hi, int t) { part[t] = radius_of(off + lo, hi - lo, c); });
            return *std::max_element(part.begin(), part.end());
        }
        return radius_of(off, size, c);
    }
    template <typename F>
    void in_chunks(int size, F &&body) const { // chunks of at least half the smallest wide slice: a thread costs as much as ~10^4 points
        const int nt = chunks_of(size);
        std::vector<std::thread> th;
        for (int t = 1; t < nt; t++)
            th.emplace_back([&, t] { body((int)((int64_t)size * t / nt), (int)((int64_t)size * (t + 1) / nt), t); });
        body(0, (int)((int64_t)size / nt), 0);
        for (auto &x_ : th)
            x_.join();
    }
    int chunks_of(int size) const { return std::max(1, std::min(psort_threads, size / std::max(1, psort_min / 2))); }
    double radius_of(int off, int size, const double *c) const {
        double r         = 0;
        const double *xo = xs.get() + (size_t)dim * off;
        for (int j = 0; j < size; j++) {
            double s = 0;
            for (int p = 0; p < dim; p++) {
                const double u = xo[dim * j + p] - c[p];
                s              = s + u * u;
            }
            r = std::max(r, std::sqrt(std::fabs(s)) + (radii ? radii[perm[off + j]] : 0.0));
        }
        return r;
    }

    // ---- direction policies -----------------------------------------------------------------
    // ComputeLargestExtent (partitioning.hpp:160-193) + solve_EVP_2/3 (misc/evp.hpp:13-159)
    Frame principal_axes(const ClusterNode &c) const {
        double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        const double *xo = xs.get() + (size_t)dim * c.offset;
        for (int j = 0; j < c.size; j++) {
            const double wid = w ? w[perm[c.offset + j]] : 1.0;
            double u[3];
            for (int p = 0; p < dim; p++)
                u[p] = xo[dim * j + p] - c.center[p];
            for (int p = 0; p < dim; p++)
                for (int q = 0; q < dim; q++)
                    cov[p + dim * q] += wid * u[p] * u[q];
        }
        Frame f;
        std::fill(f.axis, f.axis + 9, 0.0);
        std::fill(f.extent, f.extent + 3, 0.0);
        if (dim == 2)
            eig2(cov, f);
        else
            eig3(cov, f);
        for (int k = 0; k < dim; k++)
            f.extent[k] = f.extent[k] > 0 ? std::sqrt(f.extent[k]) : 0.0;
        return f;
    }
    static void eig2(const double *cov, Frame &f) {
        const double eps   = std::numeric_limits<double>::epsilon();
        const double trace = cov[0] + cov[3];
        const double det   = cov[0] * cov[3] - cov[2] * cov[1]; // cov(0,1)=cov[2], cov(1,0)=cov[1]
        f.extent[0]        = trace / 2.0 + std::sqrt((trace * trace / 4.0 - det));
        f.extent[1]        = trace / 2.0 - std::sqrt((trace * trace / 4.0 - det));
        if (std::fabs(f.extent[0]) > eps) {
            for (int index = 0; index < 2; index++) {
                const double lam = f.extent[(index + 1) % 2];
                double m[4];
                for (int i = 0; i < 2; i++)
                    for (int j = 0; j < 2; j++)
                        m[i + 2 * j] = cov[i + 2 * j] - (i == j ? 1.0 : 0.0) * lam;
                int col     = 0;
                double norm = 0;
                do {
                    f.axis[0 + 2 * index] = m[0 + 2 * col];
                    f.axis[1 + 2 * index] = m[1 + 2 * col];
                    norm                  = std::sqrt(f.axis[0 + 2 * index] * f.axis[0 + 2 * index] + f.axis[1 + 2 * index] * f.axis[1 + 2 * index]);
                    col++;
                } while (norm < eps && col < 2);
                if (norm < eps) {
                    f.axis[0 + 2 * index] = 1;
                    f.axis[1 + 2 * index] = 0;
                } else {
                    f.axis[0 + 2 * index] /= norm;
                    f.axis[1 + 2 * index] /= norm;
                }
            }
        } else {
            f.axis[0] = 1;
            f.axis[3] = 1;
        }
    }
    static double dot3(const double *a, const double *b) { return ((0.0 + a[0] * b[0]) + a[1] * b[1]) + a[2] * b[2]; }
    static void cross(const double *a, const double *b, double *o) {
        o[0] = a[1] * b[2] - a[2] * b[1];
        o[1] = a[2] * b[0] - a[0] * b[2];
        o[2] = a[0] * b[1] - a[1] * b[0];
    }
    static void eig3(const double *cov, Frame &f) {
        const double eps = std::numeric_limits<double>::epsilon();
        auto C           = [&](int i, int j) { return cov[i + 3 * j]; };
        const double p1  = C(0, 1) * C(0, 1) + C(0, 2) * C(0, 2) + C(1, 2) * C(1, 2);
        if (p1 < eps) { // diagonal covariance: axes sorted by decreasing eigenvalue (evp.hpp:61-75)
            const double e[3] = {C(0, 0), C(1, 1), C(2, 2)};
            int idx[3]        = {0, 1, 2};
            std::sort(idx, idx + 3, [&e](int a, int b) { return e[a] < e[b]; });
            f.axis[idx[2] + 3 * 0] = 1;
            f.axis[idx[1] + 3 * 1] = 1;
            f.axis[idx[0] + 3 * 2] = 1;
            f.extent[0]            = e[idx[2]];
            f.extent[1]            = e[idx[1]];
            f.extent[2]            = e[idx[0]];
            return;
        }
        const double q    = (C(0, 0) + C(1, 1) + C(2, 2)) / 3.0;
        const double d0   = C(0, 0) - q, d1 = C(1, 1) - q, d2 = C(2, 2) - q;
        const double p2   = d0 * d0 + d1 * d1 + d2 * d2 + 2.0 * p1;
        const double p    = std::sqrt(p2 / 6.0);
        const double invp = 1.0 / p;
        double B[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                B[i + 3 * j] = (C(i, j) - (i == j ? 1.0 : 0.0) * q) * invp;
        auto Bm           = [&](int i, int j) { return B[i + 3 * j]; };
        const double detB = Bm(0, 0) * (Bm(1, 1) * Bm(2, 2) - Bm(1, 2) * Bm(2, 1)) - Bm(0, 1) * (Bm(1, 0) * Bm(2, 2) - Bm(1, 2) * Bm(2, 0)) + Bm(0, 2) * (Bm(1, 0) * Bm(2, 1) - Bm(1, 1) * Bm(2, 0));
        const double r    = detB / 2.0;
        double phi;
        if (r <= -1)
            phi = 1.047197551196598;
        else if (r >= 1)
            phi = 0;
        else
            phi = std::acos(r) / 3.0;
        f.extent[0] = q + 2.0 * p * std::cos(phi);
        f.extent[2] = q + 2.0 * p * std::cos(phi + 2.094395102393195);
        f.extent[1] = 3.0 * q - f.extent[0] - f.extent[2];
        if (!(std::fabs(f.extent[0]) > eps)) {
            f.axis[0] = f.axis[4] = f.axis[8] = 1;
            return;
        }
        for (int index = 0; index < 3; index++) {
            double m[9];
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++)
                    m[i + 3 * j] = C(i, j) - (i == j ? 1.0 : 0.0) * f.extent[index];
            double c01[3], c02[3], c12[3];
            cross(m, m + 3, c01);
            cross(m, m + 6, c02);
            cross(m + 3, m + 6, c12);
            const double n01 = dot3(c01, c01), n02 = dot3(c02, c02), n12 = dot3(c12, c12);
            double best   = n01;
            int which     = 0;
            if (n02 > best) {
                best  = n02;
                which = 1;
            }
            if (n12 > best)
                which = 2;
            const double *v = which == 0 ? c01 : (which == 1 ? c02 : c12);
            const double s  = std::sqrt(which == 0 ? n01 : (which == 1 ? n02 : n12));
            for (int i = 0; i < 3; i++)
                f.axis[i + 3 * index] = v[i] / s;
        }
    }
    // ComputeBoundingBox (partitioning.hpp:196-231); the max starts at numeric_limits::min() as there
    Frame bbox_axes(const ClusterNode &c) const {
        double lo[3], hi[3];
        for (int p = 0; p < dim; p++) {
            lo[p] = std::numeric_limits<double>::max();
            hi[p] = std::numeric_limits<double>::min();
        }
        const double *xo = xs.get() + (size_t)dim * c.offset;
        for (int j = 0; j < c.size; j++) {
            for (int p = 0; p < dim; p++) {
                const double v = xo[dim * j + p];
                if (lo[p] > v)
                    lo[p] = v;
                if (hi[p] < v)
                    hi[p] = v;
            }
        }
        int idx[3] = {0, 1, 2};
        std::sort(idx, idx + dim, [&](int a, int b) { return (hi[a] - lo[a]) < (hi[b] - lo[b]); });
        Frame f;
        std::fill(f.axis, f.axis + 9, 0.0);
        std::fill(f.extent, f.extent + 3, 0.0);
        for (int k = 0; k < dim; k++) {
            const int a         = idx[dim - 1 - k];
            f.axis[a + dim * k] = 1;
            f.extent[k]         = hi[a] - lo[a];
        }
        return f;
    }

    // ---- ordering + splitting -----------------------------------------------------------------
    // partitioning.hpp:27-31: sort the slice by projection on `dir` with the unstable std::sort.
    //
    // htool sorts the point numbers with a comparator that looks the keys up (`key[a] < key[b]`).  When all keys of the slice are DISTINCT
    // the ascending order is unique, whatever algorithm produces it: here (key, position) pairs are sorted -- contiguous 16-byte records, in
    // chunks on several threads + merges for the large slices of the top levels, where few nodes leave most cores idle -- and the permutation
    // and the ordered coordinates are rewritten from the result.  If two keys compare equal (duplicate points, a degenerate direction) the
    // outcome of an unstable sort depends on its comparison sequence, and the slice is sorted again exactly as the reference does it.
    void order_along(int off, int size, const double *dir) {
        if (size <= 0)
            return;
        const bool wide = size >= psort_min && psort_threads > 1;
        KeyPos *a       = sk_a.get() + off;
        const double *xo = xs.get() + (size_t)dim * off;
        auto keys        = [&](int lo, int hi, int) {
            for (int j = lo; j < hi; j++) {
                double c = 0.0;
                for (int p = 0; p < dim; p++)
                    c = c + xo[dim * j + p] * dir[p];
                a[j].first  = c;
                a[j].second = j;
            }
        };
        if (wide)
            in_chunks(size, keys);
        else
            keys(0, size, 0);
        auto by_key = [](const KeyPos &u, const KeyPos &v) { return u.first < v.first; };
        if (wide) {
            const int nt = std::max(2, chunks_of(size));
            std::vector<int> cut(nt + 1);
            for (int t = 0; t <= nt; t++)
                cut[t] = (int)((int64_t)size * t / nt);
            {
                std::vector<std::thread> th;
                for (int t = 0; t < nt; t++)
                    th.emplace_back([&, t] { std::sort(a + cut[t], a + cut[t + 1], by_key); });
                for (auto &x_ : th)
                    x_.join();
            }
            KeyPos *src = a, *dst = sk_b.get() + off;
            for (int width = 1; width < nt; width *= 2) { // rounds of pairwise merges into the other buffer, the pairs of a round in parallel
                std::vector<std::thread> th;
                for (int t = 0; t < nt; t += 2 * width)
                    th.emplace_back([&, t, src, dst] {
                        const int lo = cut[t], mid = cut[std::min(t + width, nt)], hi = cut[std::min(t + 2 * width, nt)];
                        std::merge(src + lo, src + mid, src + mid, src + hi, dst + lo, by_key);
                    });
                for (auto &x_ : th)
                    x_.join();
                std::swap(src, dst);
            }
            a = src;
        } else {
            std::sort(a, a + size, by_key);
        }
        bool distinct = true;
        for (int j = 1; j < size && distinct; j++)
            distinct = a[j - 1].first < a[j].first; // false for equal keys and for unordered ones (NaN)
        if (!distinct) {                            // the reference's own call decides the order of equal keys
            std::call_once(key_once, [&] { key.resize(n); });
            for (int j = 0; j < size; j++) {
                double c = 0.0;
                for (int p = 0; p < dim; p++)
                    c = c + xo[dim * j + p] * dir[p];
                key[perm[off + j]] = c;
            }
            const double *k = key.data();
            std::sort(perm.begin() + off, perm.begin() + off + size, [k](int u, int v) { return k[u] < k[v]; });
            gather_xs(off, size);
            return;
        }
        int32_t *ids = sk_ids.get() + off;
        double *xc   = sk_x.get() + (size_t)dim * off, *xw = xs.get() + (size_t)dim * off;
        auto save    = [&](int lo, int hi, int) {
            std::copy(perm.begin() + off + lo, perm.begin() + off + hi, ids + lo);
            std::copy(xw + (size_t)dim * lo, xw + (size_t)dim * hi, xc + (size_t)dim * lo);
        };
        auto rewrite = [&](int lo, int hi, int) {
            for (int j = lo; j < hi; j++) {
                const int from = a[j].second;
                perm[off + j]  = ids[from];
                for (int p = 0; p < dim; p++)
                    xw[dim * j + p] = xc[(size_t)dim * from + p];
            }
        };
        if (wide) {
            in_chunks(size, save);
            in_chunks(size, rewrite);
        } else {
            save(0, size, 0);
            rewrite(0, size, 0);
        }
    }
    int psort_min = 1 << 16, psort_threads = 1;
    std::once_flag key_once;
    // RegularSplitting (partitioning.hpp:234-249)
    static Parts even_parts(int off, int size, int k) {
        Parts parts(k);
        const int each = size / k;
        for (int p = 0; p < k - 1; p++)
            parts[p] = {off + each * p, each};
        parts[k - 1] = {off + each * (k - 1), size - each * (k - 1)};
        return parts;
    }
    // GeometricSplitting (partitioning.hpp:253-296), including its end-of-array test
    Parts geometric_parts(int off, int size, const double *dir, int k) const {
        Parts parts;
        if (size <= k)
            return parts;
        parts.assign(k, {0, 0});
        auto proj_from = [&](int pos, const double *origin) { // pos: position in the permutation
            double s = 0.0;
            for (int p = 0; p < dim; p++)
                s = s + dir[p] * (xs[(size_t)dim * pos + p] - origin[p]);
            return s;
        };
        double origin[3];
        for (int p = 0; p < dim; p++)
            origin[p] = xs[(size_t)dim * off + p];
        const double span = proj_from(off + size - 1, origin);
        const double step = span / k;
        int cursor        = off;
        std::vector<int> offs(k, 0), sizes(k, 0);
        for (int p = 0; p < k - 1; p++) {
            int hit = cursor;
            while (hit < off + size && !(proj_from(hit, origin) > step))
                hit++;
            if (hit != n) {
                offs[p]  = cursor;
                sizes[p] = hit - cursor;
                cursor   = hit;
                for (int q = 0; q < dim; q++)
                    origin[q] = xs[(size_t)dim * hit + q];
            } else {
                break;
            }
        }
        offs[k - 1]  = cursor;
        sizes[k - 1] = size - std::accumulate(sizes.begin(), sizes.end() - 1, 0);
        for (int p = 0; p < k; p++)
            parts[p] = {offs[p], sizes[p]};
        return parts;
    }
    Parts cut(int off, int size, const double *dir, int k) const {
        return opt.splitting == HMX_SPLIT_REGULAR ? even_parts(off, size, k) : geometric_parts(off, size, dir, k);
    }

    // Partitioning_N helpers (partitioning.hpp:43-86): ordered factorisations of k over ndir directions,
    // pick the one with the best extent/count aspect ratio (first best wins).
    static void factorisations(int rest, int slots, int cap, std::vector<int> &cur, std::vector<std::vector<int>> &all) {
        if (slots == 1) {
            if (rest <= cap && rest >= 1) {
                cur.push_back(rest);
                all.push_back(cur);
                cur.pop_back();
            }
            return;
        }
        for (int f = cap; f >= 1; f--)
            if (rest % f == 0) {
                cur.push_back(f);
                factorisations(rest / f, slots - 1, f, cur, all);
                cur.pop_back();
            }
    }
    static std::vector<int> best_factorisation(int ndir, int k, const double *extent) {
        std::vector<std::vector<int>> all;
        std::vector<int> cur;
        factorisations(k, ndir, k, cur, all);
        double best = std::numeric_limits<double>::max();
        size_t pick = 0;
        for (size_t t = 0; t < all.size(); t++) {
            double hi = -std::numeric_limits<double>::infinity(), lo = std::numeric_limits<double>::infinity();
            for (int d = 0; d < ndir; d++) {
                const double a = extent[d] / double(all[t][d]);
                if (d == 0) {
                    hi = lo = a;
                } else {
                    if (hi < a)
                        hi = a; // std::max_element: first maximum
                    if (a < lo)
                        lo = a; // std::min_element: first minimum
                }
            }
            const double cost = hi / lo;
            if (cost < best) {
                best = cost;
                pick = t;
            }
        }
        return all[pick];
    }

    // compute_partitioning of Partitioning (partitioning.hpp:15-35) / Partitioning_N (:89-156)
    Parts split(const ClusterNode &c, int k) {
        const Frame f = opt.direction == HMX_DIR_LARGEST_EXTENT ? principal_axes(c) : bbox_axes(c);
        if (opt.partitioning_n) {
            int relevant = 0;
            for (int d = 0; d < dim; d++)
                if (f.extent[d] > std::numeric_limits<double>::epsilon() * 10)
                    relevant++;
            relevant                     = std::max(1, relevant);
            const std::vector<int> count = best_factorisation(relevant, k, f.extent);
            relevant                     = (int)count.size();
            struct Item {
                int off, size, d;
            };
            std::vector<Item> todo{{c.offset, c.size, 0}};
            Parts result;
            while (!todo.empty()) {
                const Item it = todo.back();
                todo.pop_back();
                const double *dir = f.axis + dim * it.d;
                order_along(it.off, it.size, dir);
                const Parts sub = cut(it.off, it.size, dir, count[it.d]);
                if ((int)sub.size() != count[it.d])
                    break;
                if (it.d < relevant - 1) {
                    for (int p = count[it.d] - 1; p >= 0; p--)
                        todo.push_back({sub[p].first, sub[p].second, it.d + 1});
                } else {
                    result.insert(result.end(), sub.begin(), sub.end());
                }
            }
            if ((int)result.size() == k) {
                std::sort(result.begin(), result.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) { return a.first < b.first; });
                return result;
            }
        }
        order_along(c.offset, c.size, f.axis);
        return cut(c.offset, c.size, f.axis, k);
    }
};

template <typename F>
void parallel_for(int count, int max_threads, F &&body) {
    const int nt = std::max(1, std::min({max_threads, count, 64})); // starting a thread costs ~30 us: 256 of them per level were most of the lower levels' time
    if (nt == 1) {
        for (int i = 0; i < count; i++)
            body(i);
        return;
    }
    std::atomic<int> next(0);
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; t++)
        pool.emplace_back([&]() {
            for (int i = next.fetch_add(1); i < count; i = next.fetch_add(1))
                body(i);
        });
    for (auto &th : pool)
        th.join();
}

} // namespace

int build_cluster_tree(int n, int dim, const double *coords, const double *radii, const double *weights,
                       const ClusterTreeOptions &opt, hmx_cluster_tree &T, const int32_t *partition, int partition_kind) {
    if (n <= 0 || (dim != 2 && dim != 3) || !coords || opt.number_of_children < 2 || opt.size_of_partition < 1 || opt.maximal_leaf_size < 1) {
        set_error("hmx_cluster_tree_create: invalid arguments (dim must be 2 or 3, children >= 2)");
        return HMX_ERR_INVALID;
    }
    if (partition_kind < 0 || partition_kind > 2 || (partition_kind != 0 && !partition)) {
        set_error("hmx_cluster_tree_create: invalid partition arguments");
        return HMX_ERR_INVALID;
    }
    const auto t_start = std::chrono::steady_clock::now();
    T.n                = n;
    T.dim              = dim;
    T.opt              = opt;
    T.perm.resize(n);
    std::iota(T.perm.begin(), T.perm.end(), 0);
    T.nodes.clear();
    T.on_partition.clear();
    Builder B(n, dim, coords, radii, weights, opt, T.perm);
    B.psort_threads = std::min(32, hmx::host_cores());
    if (const char *e = std::getenv("HMX_TREE_PSORT_MIN")) // smallest slice handled in chunks on several threads (tests lower it; a huge value disables it)
        B.psort_min = std::max(2, std::atoi(e));
    B.gather_xs(0, n);

    ClusterNode root;
    root.size = n;
    B.centroid(0, n, root.center);
    root.radius = B.bounding_radius(0, n, root.center);
    T.nodes.push_back(root);

    // "simple" partition (tree_builder.hpp:125-141): which depth carries the MPI partition, and how many
    // children the nodes just above it get.
    const int nc = opt.number_of_children, sp = opt.size_of_partition;
    int partition_depth, children_on_partition_level = sp, extra_on_last = 0;
    if (sp >= nc) {
        partition_depth             = static_cast<int>(std::floor(std::log(sp) / std::log(nc)));
        children_on_partition_level = nc;
        if (sp != std::pow(nc, partition_depth))
            extra_on_last = sp - static_cast<int>(std::pow(nc, partition_depth));
    } else {
        partition_depth = 1;
    }
    T.permutation_is_local = (sp == 1);

    const int hw = hmx::host_cores();
    std::vector<int> level{0};
    const bool given = partition_kind != 0;
    if (given) {
        // user-given partition (tree_builder.hpp:87-123): the root's children ARE the parts; every part is then split by the
        // strategy like any other cluster (no node is "above the partition level" any more)
        partition_depth = -1000;
        std::vector<int> offs(sp), sizes(sp);
        if (partition_kind == 2) {
            for (int p = 0; p < sp; p++) {
                offs[p]  = partition[2 * p];
                sizes[p] = partition[2 * p + 1];
                if (offs[p] < 0 || sizes[p] < 0 || (int64_t)offs[p] + sizes[p] > n) {
                    set_error("hmx_cluster_tree_create: local partition out of range");
                    return HMX_ERR_INVALID;
                }
            }
            T.permutation_is_local = true;
        } else {
            int cpt    = 0;
            bool local = true;
            for (int p = 0; p < sp; p++) {
                offs[p]  = cpt;
                sizes[p] = 0;
                int prev = -1;
                for (int i = 0; i < n; i++)
                    if (partition[i] == p) {
                        T.perm[cpt++] = i;
                        sizes[p]++;
                        local = local && (prev < 0 || prev == i - 1);
                        prev  = i;
                    }
            }
            if (cpt != n) {
                set_error("hmx_cluster_tree_create: global partition must map every point to a part in [0, size_of_partition)");
                return HMX_ERR_INVALID;
            }
            T.permutation_is_local = local;
            B.gather_xs(0, n); // the permutation now groups the points by part
        }
        level.clear();
        T.nodes[0].first_child = 1;
        T.nodes[0].n_children  = sp;
        for (int p = 0; p < sp; p++) {
            ClusterNode ch;
            ch.parent = 0;
            ch.depth  = 1;
            ch.offset = offs[p];
            ch.size   = sizes[p];
            B.centroid(ch.offset, ch.size, ch.center);
            ch.radius  = B.bounding_radius(ch.offset, ch.size, ch.center);
            ch.rank    = p;
            ch.counter = p;
            T.nodes.push_back(ch);
            T.on_partition.push_back(p + 1);
            level.push_back(p + 1);
        }
        if (sp == 1)
            T.permutation_is_local = true; // tree_builder.hpp:143-145
    }
    const bool level_timing = std::getenv("HMX_BUILD_TIMING") && std::atoi(std::getenv("HMX_BUILD_TIMING"));
    if (level_timing)
        fprintf(stderr, "[hmx tree] root, partition: %.1f ms\n", 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
    auto t_level = std::chrono::steady_clock::now();
    while (!level.empty()) {
        // split every node of this level concurrently: slices of the permutation are disjoint
        std::vector<Parts> parts(level.size());
        std::vector<std::vector<ClusterNode>> kids(level.size());
        B.psort_threads = std::min(32, hw / (int)std::min<size_t>(level.size(), (size_t)hw)); // the cores the level's nodes leave idle
        parallel_for((int)level.size(), hw, [&](int li) {
            const ClusterNode c = T.nodes[level[li]];
            const bool above    = (c.depth == partition_depth - 1);
            int k               = above ? children_on_partition_level : nc;
            if (above && c.counter == std::pow(nc, c.depth) - 1)
                k += extra_on_last;
            Parts p = B.split(c, k);
            bool ok = (int)p.size() == k;
            for (auto &e : p)
                ok = ok && e.second > 0;
            if (!ok)
                return; // node stays a leaf (tree_builder.hpp:193-197)
            kids[li].resize(k);
            for (int q = 0; q < k; q++) {
                ClusterNode &ch = kids[li][q];
                ch.parent       = level[li];
                ch.depth        = c.depth + 1;
                ch.offset       = p[q].first;
                ch.size         = p[q].second;
                B.centroid(ch.offset, ch.size, ch.center);
                ch.radius  = B.bounding_radius(ch.offset, ch.size, ch.center);
                ch.rank    = c.rank;
                ch.counter = c.counter * k + q;
                if (above) {
                    ch.rank    = c.counter * children_on_partition_level + q;
                    ch.counter = ch.rank;
                }
            }
        });
        if (level_timing) {
            const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_level).count();
            fprintf(stderr, "[hmx tree] level of %zu nodes (%d rows in the first): %.1f ms\n", level.size(), T.nodes[level[0]].size, 1e3 * t);
            t_level = std::chrono::steady_clock::now();
        }
        std::vector<int> next;
        for (size_t li = 0; li < level.size(); li++) {
            if (kids[li].empty())
                continue;
            const int first              = (int)T.nodes.size();
            T.nodes[level[li]].first_child = first;
            T.nodes[level[li]].n_children  = (int)kids[li].size();
            const bool above             = (T.nodes[level[li]].depth == partition_depth - 1);
            for (auto &ch : kids[li]) {
                const int id = (int)T.nodes.size();
                T.nodes.push_back(ch);
                if (above) {
                    if (ch.rank + 1 > (int)T.on_partition.size())
                        T.on_partition.resize(ch.rank + 1, -1);
                    T.on_partition[ch.rank] = id;
                }
                if (!opt.is_complete && ch.size > opt.maximal_leaf_size)
                    next.push_back(id);
            }
            if (opt.is_complete) { // complete tree: all children are split, or none (tree_builder.hpp:176-182)
                bool any = false;
                for (auto &ch : kids[li])
                    any = any || ch.size > opt.maximal_leaf_size;
                if (any)
                    for (int q = 0; q < (int)kids[li].size(); q++)
                        next.push_back(first + q);
            }
        }
        level.swap(next);
    }
    // A branch that became a leaf above the partition level (maximal_leaf_size too large for this many points and parts) leaves ranks without
    // a cluster: the reference then keeps null entries in its list of partition clusters (child constructor, cluster_node.hpp:35-42) and
    // whatever walks the partition dereferences them.  Refused here.
    for (int id : T.on_partition)
        if (id < 0) {
            set_error("hmx_cluster_tree_create: the tree stops above the partition level (a cluster of at most maximal_leaf_size points before size_of_partition parts exist): fewer parts or a smaller leaf size");
            return HMX_ERR_INVALID;
        }
    return HMX_OK;
}

} // namespace hmx

std::vector<int> hmx_cluster_tree::preorder() const {
    std::vector<int> order;
    order.reserve(nodes.size());
    std::vector<int> stack{0};
    while (!stack.empty()) {
        const int v = stack.back();
        stack.pop_back();
        order.push_back(v);
        for (int c = nodes[v].n_children - 1; c >= 0; c--)
            stack.push_back(nodes[v].first_child + c);
    }
    return order;
}

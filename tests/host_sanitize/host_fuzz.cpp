// host-side code under ASan/UBSan: cluster trees (every strategy, partitions, user partitions, complete), block trees
// (symmetry, row partitions, local roots, consistency), on-disk formats incl. malformed input
#include "hmx.h"
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#define CHECK(x) do { int rc_ = (x); if (rc_ != 0) { std::printf("FAIL %s -> %d (%s)\n", #x, rc_, hmx_last_error()); return 1; } } while (0)
int main(int argc, char **argv) {
    const bool quick = argc > 1 && std::string(argv[1]) == "quick"; // the pytest run; the full sweep takes ~15 min under ASan
    const std::vector<const char *> geoms = quick ? std::vector<const char *>{"ellipse", "disk2d"} : std::vector<const char *>{"ellipse", "disk", "ball", "disk2d"};
    const std::vector<int> sizes = quick ? std::vector<int>{1, 7, 500} : std::vector<int>{1, 2, 7, 64, 500, 3001};
    for (const char *geom : geoms) {
        const int dim = std::string(geom) == "disk2d" ? 2 : 3;
        for (int n : sizes) {
            std::vector<double> x((size_t)n * dim);
            CHECK(hmx_geometry(geom, n, 0.0, x.data()));
            for (int dir = 0; dir < 2; dir++)
                for (int split = 0; split < 2; split++)
                    for (int pn = 0; pn < 2; pn++)
                        for (int children : (quick ? std::vector<int>{2, 3} : std::vector<int>{2, 3, 4, 8}))
                            for (int parts : (quick ? std::vector<int>{1, 2, 4} : std::vector<int>{1, 2, 3, 4, 8}))
                                for (int complete = 0; complete < 2; complete++) {
                                    hmx_cluster_tree *T = nullptr;
                                    int rc = hmx_cluster_tree_create_ex(n, dim, x.data(), nullptr, nullptr, 10, children, parts, dir, split, pn, complete, nullptr, 0, &T);
                                    if (rc != 0)
                                        continue;
                                    std::vector<hmx_cluster_node> nodes(hmx_cluster_tree_num_nodes(T));
                                    hmx_cluster_tree_nodes(T, nodes.data());
                                    const int np = hmx_cluster_tree_num_partitions(T);
                                    for (char sym : {'N', 'S', 'H'})
                                        for (int tp = -1; tp < np; tp += (np > 2 ? 2 : 1)) {
                                            hmx_block_tree *B = nullptr;
                                            rc = hmx_block_tree_create(T, T, 3.0, sym, sym == 'N' ? 'N' : 'L', 0, 0, tp, tp, 1, &B);
                                            if (rc == 0) {
                                                std::vector<hmx_leaf> lv(hmx_block_tree_num_leaves(B));
                                                hmx_block_tree_leaves(B, lv.data());
                                                hmx_block_tree_save_leaves_with_rank(B, nullptr, "/tmp/hmx_asan_leaves");
                                                hmx_block_tree_destroy(B);
                                            }
                                            if (tp >= 0 && hmx_block_tree_create_local(T, T, 3.0, sym, sym == 'N' ? 'N' : 'U', 0, 0, tp, tp, 1, &B) == 0)
                                                hmx_block_tree_destroy(B);
                                        }
                                    if (n == 500 && children == 2) {
                                        CHECK(hmx_cluster_tree_save(T, "/tmp/hmx_asan_t"));
                                        hmx_cluster_tree *L = nullptr;
                                        CHECK(hmx_cluster_tree_load("/tmp/hmx_asan_t_cluster_tree_properties.csv", "/tmp/hmx_asan_t_cluster_tree.csv", &L));
                                        hmx_cluster_tree_destroy(L);
                                    }
                                    hmx_cluster_tree_destroy(T);
                                }
            // user partitions
            if (n >= 64) {
                std::vector<int32_t> gp(n), lp{0, n / 3, n / 3, n - n / 3};
                for (int i = 0; i < n; i++)
                    gp[i] = (i * 7) % 3;
                hmx_cluster_tree *T = nullptr;
                CHECK(hmx_cluster_tree_create_ex(n, dim, x.data(), nullptr, nullptr, 10, 2, 3, 0, 0, 0, 0, gp.data(), 1, &T));
                hmx_cluster_tree_destroy(T);
                CHECK(hmx_cluster_tree_create_ex(n, dim, x.data(), nullptr, nullptr, 10, 2, 2, 0, 0, 0, 0, lp.data(), 2, &T));
                hmx_cluster_tree_destroy(T);
                gp[5] = 99; // out of range part: must be refused, not crash
                if (hmx_cluster_tree_create_ex(n, dim, x.data(), nullptr, nullptr, 10, 2, 3, 0, 0, 0, 0, gp.data(), 1, &T) == 0) {
                    std::printf("FAIL: bad partition accepted\n");
                    return 1;
                }
            }
        }
    }
    // malformed files
    for (const char *txt : {"", "1|2|3\n", "2|-1|0|10|1|0|0|0|0|0\n", "2|-1|0|10|1|0|0|0|0|0\n2|0|0|5|1|0|1|0|0|0\n", "x|y|z|a|b|c|d|e|f|g\n"}) {
        FILE *f = std::fopen("/tmp/hmx_asan_bad.csv", "w");
        std::fputs(txt, f);
        std::fclose(f);
        f = std::fopen("/tmp/hmx_asan_bad_p.csv", "w");
        std::fputs("maximal leaf size: 10\nmaximal depth: 1\nminimal depth: 1\npermutation: 0,1,2\nlocal permutation: 0\n", f);
        std::fclose(f);
        hmx_cluster_tree *L = nullptr;
        if (hmx_cluster_tree_load("/tmp/hmx_asan_bad_p.csv", "/tmp/hmx_asan_bad.csv", &L) == 0)
            hmx_cluster_tree_destroy(L);
    }
    std::printf("host fuzz ok\n");
    return 0;
}

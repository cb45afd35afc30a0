/* hmx.h -- C ABI of libhmx: MI355X-native H-matrix block compression + H-matvec engine.
 *
 * This is the drop-in boundary.  Every entry point states the htool interface it
 * replaces (paths relative to htool's include/htool/).  Plain pointers and sizes only; no C++ or torch
 * types.  All functions return 0 on success and a negative hmx_status otherwise; nothing throws across
 * the boundary (htool's convention is "log and continue / return false", misc/logger.hpp:74-76 -- the
 * C++ adaptor in htool_amd/include/hmx/htool_adaptor.hpp maps non-zero to that).
 *
 * Numbering: "cluster numbering" = htool's internal numbering (position i holds user point perm[i],
 * clustering/cluster_tree_data.hpp:21); "user numbering" = the caller's.
 * Vectors may live in host or device memory (hmx_mem).  Device pointers are what a one-process-per-GPU
 * caller (torch tensor .data_ptr()) passes; `stream` is a hipStream_t cast to void* (NULL = default).
 */
#ifndef HMX_H
#define HMX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    HMX_OK                = 0,
    HMX_ERR_INVALID       = -1, /* bad argument / unsupported combination (htool logs ERROR, e.g. add_hmatrix_vector_product.hpp:59-62) */
    HMX_ERR_NO_DEVICE     = -2, /* HIP device missing: the engine never falls back to a CPU path */
    HMX_ERR_HIP           = -3, /* a HIP runtime call failed; see hmx_last_error() */
    HMX_ERR_STATE         = -4, /* call order violated (e.g. matvec before compress) */
    HMX_ERR_UNSUPPORTED   = -5
} hmx_status;

typedef enum { HMX_MEM_HOST = 0, HMX_MEM_DEVICE = 1 } hmx_mem;

/* clustering/implementations/partitioning.hpp: direction policy x splitting policy x (Partitioning | Partitioning_N) */
typedef enum { HMX_DIR_LARGEST_EXTENT = 0, HMX_DIR_BOUNDING_BOX = 1 } hmx_direction;
typedef enum { HMX_SPLIT_REGULAR = 0, HMX_SPLIT_GEOMETRIC = 1 } hmx_splitting;

/* hmatrix/lrmat/{partialACA,sympartialACA,fullACA,SVD}.hpp */
typedef enum { HMX_PARTIAL_ACA = 0, HMX_SYMPARTIAL_ACA = 1, HMX_FULL_ACA = 2, HMX_SVD = 3 } hmx_compressor;

/* Device-evaluable generators (the user's VirtualGenerator::copy_submatrix, hmatrix/interfaces/virtual_generator.hpp:24, for BEM-style
 * kernels: the ones the reference ships -- examples/use_hmatrix.cpp:33, testing/generator_test.hpp:159,185 -- and the single-layer kernels
 * its users bring; r = |x - y|, squared differences summed in coordinate order).  hmx_hmatrix_set_kernel(H, kernel, params, nparams, ...):
 *   HMX_KERNEL_INV_DIST   : K = 1 / (params[0] + params[1] * r).  With complex coefficients (hmx_hmatrix_create_z / _c)
 *                           K = (params[2] + i * params[3] * sgn) / (params[0] + params[1] * r), sgn = 1, or, when params[4] != 0,
 *                           sign(x_target[0] - x_source[0]): the complex symmetric and Hermitian test generators of
 *                           testing/generator_test.hpp:163-205 (GeneratorTestComplex, ...ComplexSymmetric, ...ComplexHermitian).
 *   HMX_KERNEL_HELMHOLTZ  : K = exp(i * k * r) / (params[0] + params[1] * r), k = params[2] -- the Helmholtz single layer
 *                           exp(i k r) / (4 pi r) for params = {0, 4 pi, k} (distinct target and source clouds; a small params[0] regularises
 *                           coincident points as the reference's test generators do).  Complex symmetric ('S' storage).  Real coefficient
 *                           types take the real part cos(k r) / (...).  cos / sin are evaluated by a fixed, documented sequence of IEEE
 *                           operations (Cody-Waite reduction + minimax polynomials, about one ulp for |k r| < 1.6e6: csrc/kernels_common.hpp
 *                           hmx_sincos), so that a host generator restating it produces the same bits.
 *   HMX_KERNEL_LAPLACE_SL : K = (params[1] + i * params[2]) / (4 pi * (params[0] + r)), defaults params[1] = 1, params[2] = 0 (the imaginary
 *                           part only with complex coefficients): the Laplace single layer 1 / (4 pi r) for params[0] = 0.
 * Anything else goes through hmx_hmatrix_set_callback (the generator on the host's cores, everything else on the device). */
typedef enum { HMX_KERNEL_INV_DIST = 0, HMX_KERNEL_HELMHOLTZ = 1, HMX_KERNEL_LAPLACE_SL = 2 } hmx_kernel;

/* coefficient type of an hmx_hmatrix: htool's HMatrix<double>, <float>, <std::complex<double>>, <std::complex<float>>
 * (coordinates are fp64 in all four).  Complex values cross this ABI as interleaved (re, im) pairs, the layout of
 * std::complex<T> and C99 T _Complex; alpha / beta are pointers to one such pair. */
typedef enum { HMX_PREC_F64 = 0, HMX_PREC_F32 = 1, HMX_PREC_Z64 = 2, HMX_PREC_C32 = 3 } hmx_precision;

typedef struct hmx_cluster_tree hmx_cluster_tree;
typedef struct hmx_block_tree hmx_block_tree;
typedef struct hmx_hmatrix hmx_hmatrix;

/* one row of the cluster table, preorder (clustering/cluster_output.hpp:60-71 fields) */
typedef struct {
    int32_t depth, offset, size, rank, counter, n_children;
    double radius;
    double center[3];
} hmx_cluster_node;

/* one leaf of the block tree, in htool's build order (hmatrix/hmatrix_output.hpp:43-54 fields + flags) */
typedef struct {
    int32_t t_offset, t_size, s_offset, s_size;
    int32_t admissible; /* 1: low-rank candidate (m_admissible_tasks), 0: dense task                     */
    int32_t mirror;     /* 1: leaf is in leaves_for_symmetry (hmatrix/hmatrix.hpp:262-264)                 */
    int32_t symmetric;  /* 1: diagonal leaf carrying symmetry/UPLO (tree_builder.hpp:125-132)              */
    int32_t rank;       /* after compression: rank, or -1 for dense (HMatrix::get_rank, hmatrix.hpp:137)   */
} hmx_leaf;

typedef struct {
    int64_t n_dense, n_lowrank, n_false_positive;
    int64_t cgen_dense, cgen_lowrank; /* number_of_generated_coefficient, hmatrix/hmatrix_output.hpp:153-175 */
    int32_t rank_min, rank_max;
    double rank_mean;
    int64_t stream_bytes;        /* bytes resident in HBM for the matvec streams                            */
    int64_t expand_coeffs;       /* coefficients streamed by the expand kernel (dense + U panels)           */
    int64_t reduce_coeffs;       /* coefficients streamed by the reduce kernel (V panels)                   */
    int64_t a_total;             /* sum of ranks (length of the intermediate vector a = V x)                */
    double t_compress_s, t_assemble_s, t_pack_s; /* hipEvent timings of the build phases                   */
    int64_t transposed_bytes;    /* what 'T' / 'C' products hold besides the operator: the tables of the product on the stored data (~3 %) and / or the transposed stream layout; 0 = nothing built */
    int64_t expanded_bytes;      /* expanded view of a compact symmetric operator (multi-RHS products), 0 = not built             */
    /* HMX_OPT_PLACE_WRITTEN, the array the reduce stage writes (a = V x): GB/s of the placement probe -- the R-stream read alone, with the
       array where first fit put it, where it stays -- and the places tried (0: nothing tried: no reserved slab, small operator, option off) */
    double placed_read_gbps, placed_first_gbps, placed_gbps;
    int64_t placed_tried;
} hmx_stats;

const char *hmx_last_error(void);
int hmx_device_count(void);
/* One-time initialisation of the device side (HIP context, load of the library's code object: ~0.2 s), otherwise paid by the
 * first hmx_hmatrix_create / compress.  Optional; for callers that time operator builds. */
int hmx_device_init(int device_id);
/* Host cores the library will use for its host-side work (cluster tree, layout, generator threads): the hardware threads capped by the
 * cgroup CPU quota of the process (a container limited to 16 cores' worth of time gains nothing from 256 threads); HMX_HOST_CORES overrides. */
int hmx_host_cores(void);

/* ---- test geometries (testing/geometry.hpp:11-61), seeded mt19937(0) --------------------------------- */
int hmx_geometry(const char *name /* "ellipse" | "disk" | "ball" (n*3 doubles) | "disk2d" (n*2) */, int n, double z, double *coords);

/* ---- cluster tree: ClusterTreeBuilder::create_cluster_tree (clustering/tree_builder/tree_builder.hpp:52-207) */
int hmx_cluster_tree_create(int n, int dim, const double *coords /* n*dim, AoS */, const double *radii /* n or NULL */,
                            const double *weights /* n or NULL */, int maximal_leaf_size, int number_of_children,
                            int size_of_partition, int direction, int splitting, int partitioning_n,
                            hmx_cluster_tree **out);
/* The full signature of ClusterTreeBuilder::create_cluster_tree (tree_builder.hpp:42,52-207): is_complete
 * (set_is_complete, :39,176-192) and a user-given partition -- partition_kind 1 = create_cluster_tree_from_global_partition
 * (partition[i] = part of point i, :46), 2 = create_cluster_tree_from_local_partition (partition[2p], [2p+1] = offset, size of
 * part p, :48), 0 = none. */
int hmx_cluster_tree_create_ex(int n, int dim, const double *coords, const double *radii, const double *weights, int maximal_leaf_size,
                               int number_of_children, int size_of_partition, int direction, int splitting, int partitioning_n,
                               int is_complete, const int32_t *partition, int partition_kind, hmx_cluster_tree **out);
/* An EXISTING cluster tree instead of a built one: whatever made it (a user's VirtualPartitioning,
 * clustering/interfaces/virtual_partitioning.hpp:9-14 through ClusterTreeBuilder::set_partitioning_strategy, tree_builder.hpp:40; a tree
 * read from disk; another library).  `nodes` is the preorder walk of htool's Cluster (preorder_tree_traversal: a node, then its children's
 * subtrees in order) with the fields of Cluster (get_depth / get_offset / get_size / get_rank / get_counter / get_children().size() /
 * get_radius / get_center), `permutation` Cluster::get_permutation() of the root, partition_nodes[k] the index (in `nodes`) of
 * get_clusters_on_partition()[k].  Checked: the permutation is one, children tile their parent in order, partitions tile the points. */
int hmx_cluster_tree_from_nodes(int n, int dim, const int32_t *permutation, int num_nodes, const hmx_cluster_node *nodes, int num_partitions,
                                const int32_t *partition_nodes, int maximal_leaf_size, int permutation_is_local, hmx_cluster_tree **out);
void hmx_cluster_tree_destroy(hmx_cluster_tree *);
int hmx_cluster_tree_size(const hmx_cluster_tree *);            /* number of points                         */
int hmx_cluster_tree_num_nodes(const hmx_cluster_tree *);
int hmx_cluster_tree_num_partitions(const hmx_cluster_tree *);
const int32_t *hmx_cluster_tree_permutation(const hmx_cluster_tree *); /* Cluster::get_permutation           */
int hmx_cluster_tree_nodes(const hmx_cluster_tree *, hmx_cluster_node *out /* num_nodes, preorder */);
int hmx_cluster_tree_partition(const hmx_cluster_tree *, int32_t *offset_size /* 2*num_partitions */);
/* get_maximal_depth, get_minimal_depth, get_maximal_leaf_size, is_permutation_local (clustering/cluster_node.hpp:63-64) */
int hmx_cluster_tree_depths(const hmx_cluster_tree *, int32_t *max_min_leafsize_local /* 4 */);
/* save_cluster_tree / read_cluster_tree (clustering/cluster_output.hpp:33-84,87-179): writes / reads
 * <prefix>_cluster_tree_properties.csv and <prefix>_cluster_tree.csv, byte-compatible with htool's files
 * (radius and centers carry 6 significant digits there, as in htool). */
int hmx_cluster_tree_save(const hmx_cluster_tree *, const char *prefix);
int hmx_cluster_tree_load(const char *properties_file, const char *tree_file, hmx_cluster_tree **out);

/* ---- block tree: HMatrixTreeBuilder::build_block_tree + reset_root (hmatrix/tree_builder/tree_builder.hpp:417-566) */
int hmx_block_tree_create(const hmx_cluster_tree *target, const hmx_cluster_tree *source, double eta, char symmetry,
                          char uplo, int min_target_depth, int min_source_depth, int target_partition_number,
                          int partition_number_for_symmetry, int block_tree_consistency, hmx_block_tree **out);
/* A user-defined admissibility condition (VirtualAdmissibilityCondition::ComputeAdmissibility,
 * hmatrix/interfaces/virtual_admissibility_condition.hpp:12; HMatrixTreeBuilder::set_admissibility_condition,
 * tree_builder.hpp:243-246) instead of the default Rjasanow-Steinbach one (:20-23): called on the host for every pair of clusters
 * the recursion visits; returns non-zero when the block (target, source) may be compressed.  NULL restores the default.
 * The `_adm` variants of the two constructors take it. */
typedef int (*hmx_admissibility_fn)(void *user, const hmx_cluster_node *target, const hmx_cluster_node *source, double eta);
int hmx_block_tree_create_adm(const hmx_cluster_tree *target, const hmx_cluster_tree *source, double eta, char symmetry,
                              char uplo, int min_target_depth, int min_source_depth, int target_partition_number,
                              int partition_number_for_symmetry, int block_tree_consistency, hmx_admissibility_fn fn, void *user,
                              hmx_block_tree **out);
/* Block tree rooted at a pair of PARTITION clusters (block-diagonal / local-to-local operator):
 * DefaultLocalApproximationBuilder, distributed_operator/utility.hpp:64-88 */
int hmx_block_tree_create_local(const hmx_cluster_tree *target, const hmx_cluster_tree *source, double eta, char symmetry,
                                char uplo, int min_target_depth, int min_source_depth, int target_partition,
                                int source_partition, int block_tree_consistency, hmx_block_tree **out);
int hmx_block_tree_create_local_adm(const hmx_cluster_tree *target, const hmx_cluster_tree *source, double eta, char symmetry,
                                char uplo, int min_target_depth, int min_source_depth, int target_partition,
                                int source_partition, int block_tree_consistency, hmx_admissibility_fn fn, void *user, hmx_block_tree **out); /* with a user admissibility condition (NULL: default) */
void hmx_block_tree_destroy(hmx_block_tree *);
int64_t hmx_block_tree_num_leaves(const hmx_block_tree *);
int hmx_block_tree_leaves(const hmx_block_tree *, hmx_leaf *out);
int hmx_block_tree_root(const hmx_block_tree *, int32_t *t_off_size_s_off_size /* 4 */, char *symmetry_for_leaves, char *uplo_for_leaves);
/* save_leaves_with_rank (hmatrix/hmatrix_output.hpp:39-55): writes <name>.csv = "nt,ns" then
 * "t_off,t_size,s_off,s_size,rank" per leaf (offsets relative to the root block, rank -1 = dense).
 * rank = hmx_hmatrix_leaf_ranks() output, or NULL for the bare block tree. */
int hmx_block_tree_save_leaves_with_rank(const hmx_block_tree *, const int32_t *rank, const char *name);

/* ---- H-matrix on the device ---------------------------------------------------------------------------- */
int hmx_hmatrix_create(const hmx_block_tree *bt, int device_id, hmx_hmatrix **out);   /* HMatrix<double,double> */
int hmx_hmatrix_create_s(const hmx_block_tree *bt, int device_id, hmx_hmatrix **out); /* HMatrix<float,double>: fp32 coefficients, fp64 coordinates */
int hmx_hmatrix_create_z(const hmx_block_tree *bt, int device_id, hmx_hmatrix **out); /* HMatrix<std::complex<double>,double> */
int hmx_hmatrix_create_c(const hmx_block_tree *bt, int device_id, hmx_hmatrix **out); /* HMatrix<std::complex<float>,double> */
int hmx_hmatrix_is_f32(const hmx_hmatrix *);
int hmx_hmatrix_precision(const hmx_hmatrix *); /* hmx_precision, -1 for NULL */
void hmx_hmatrix_destroy(hmx_hmatrix *);

/* generator = built-in kernel on (target coords, source coords), both in USER numbering (AoS, dim <= 3) */
int hmx_hmatrix_set_kernel(hmx_hmatrix *, int kernel, const double *params, int nparams, int dim,
                           const double *target_coords, const double *source_coords);

/* generator = the user's VirtualGenerator (hmatrix/interfaces/virtual_generator.hpp:24): a host callback with
 * copy_submatrix semantics -- M x N entries of rows[0..M) x cols[0..N) (USER numbering) written column-major into out.
 * Compression then runs in lock step over batches of blocks: the callback produces one cross row / column per active block and
 * iteration, all ACA arithmetic (residual updates, pivot search, error estimate) and every later product stay on the device.
 * Dense leaves and fullACA/SVD blocks are assembled through the callback (panels of whole columns) and uploaded.
 * Like htool's own build loop (HMatrixTreeBuilder::openmp_compute_blocks, hmatrix/tree_builder/tree_builder.hpp:603-648: an OpenMP
 * parallel for over the blocks) the callback is invoked CONCURRENTLY from several host threads, each on blocks of its own, while the
 * uploads and kernels of the other threads' batches are in flight.  hmx_hmatrix_set_callback_threads(H, n): n = 1 keeps every call on
 * the calling thread (what htool does under HTOOL_WITH_PYTHON_INTERFACE or without OpenMP, tree_builder.hpp:606; for generators that are
 * not thread-safe), n = 0 (default) uses all cores (at most 64; the environment variable HMX_CALLBACK_THREADS gives another count for
 * n = 0 only: an explicit n >= 1 always wins).
 * CONTRACT: with the default the generator MUST be re-entrant -- it is called from threads the library created, several at a time,
 * exactly as htool's OpenMP build calls it.  A generator that is not (one that holds an interpreter lock, caches into shared state, or
 * was written for a non-OpenMP htool build) needs hmx_hmatrix_set_callback_threads(H, 1) before hmx_hmatrix_compress.  The C++ adaptor
 * (htool_adaptor.hpp, compress_with_generator) chooses by htool's own compile-time rule; htool_amd.api pins Python generators to 1. */
typedef void (*hmx_generator_fn)(void *user, int M, int N, const int32_t *rows, const int32_t *cols, double *out);
typedef void (*hmx_generator_fn_s)(void *user, int M, int N, const int32_t *rows, const int32_t *cols, float *out);
int hmx_hmatrix_set_callback(hmx_hmatrix *, hmx_generator_fn fn, void *user);
int hmx_hmatrix_set_callback_s(hmx_hmatrix *, hmx_generator_fn_s fn, void *user);
int hmx_hmatrix_set_callback_threads(hmx_hmatrix *, int threads);

/* Per-operator options: the counterpart of HMatrixTreeBuilder's setters (hmatrix/tree_builder/tree_builder.hpp:239-264) for what is specific
 * to this engine -- layout, kernel selection, compression tuning.  Every operator carries its own values: two operators of one process may
 * differ, and no product or build ever reads the environment.  The defaults are what bench.py measures; each option also has an environment
 * variable of the same meaning that is read ONCE, when the operator is created (hmx_hmatrix_create* / hmx_hmatrix_load), as the initial value
 * -- for A/B runs of unmodified programs.  Values are integers passed as double (HMX_OPT_POOL_RANK_GUESS is fractional).
 *   when        "layout": before hmx_hmatrix_compress / hmx_hmatrix_finalize (afterwards: HMX_ERR_STATE);
 *               "build":  before hmx_hmatrix_compress;  "product": any time, takes effect at the next product. */
typedef enum {
    /* ---- layout ---- */
    HMX_OPT_R_PIECE_ROWS      = 1,  /* layout  512   rows per R-stream piece (>= 64)                                         HMX_SR_MAX           */
    HMX_OPT_R_TREE_PIECES     = 2,  /* layout  1     pieces of large source clusters follow the cluster tree (0: fixed steps) HMX_R_TREE_PIECES    */
    HMX_OPT_LAYOUT_THREADS    = 3,  /* layout  0     host threads of the pair-list construction (0: automatic, <= 16)        HMX_LAYOUT_THREADS   */
    HMX_OPT_TASK_ORDER        = 4,  /* layout  1     launch order: 1 heaviest first, 0 address order, 2 weight classes,
                                                     3 heaviest unit first, a unit's tasks kept on one XCD (they gather the
                                                     same operand rows: found in that XCD's L2 after the first fetch)        HMX_SORT_TASKS       */
    HMX_OPT_XCD_UNIT_ROWS     = 7,  /* layout  512   rows per unit of launch order 3 (>= 64)                                 HMX_XCD_UNIT_ROWS    */
    HMX_OPT_SYM_STORAGE       = 5,  /* layout  0     symmetric / Hermitian operators: 0 stored triangle + fused product,
                                                     1 mirrored leaves laid out explicitly (twice the memory)                HMX_SYM_EXPANDED     */
    HMX_OPT_SYM_GROUP         = 8,  /* layout  4     stored-data mirrored sweeps: consecutive row ranges one workgroup takes in turn, folding in LDS
                                        the column sums that belong together (a leaf's partial a', dense leaves with one source
                                        cluster); 1: every range writes all its sums (rounds 2-5)                              HMX_SYM_GROUP        */
    HMX_OPT_SYM_GROUP_SLOTS   = 9,  /* layout  -1    LDS accumulators per group (x 16 right-hand sides x coefficient size bytes); -1: what keeps
                                                     two (8-byte reals: 320) / three (4-byte: 512) workgroups of the 16-RHS sweep on a CU;
                                                     at most 896 / 1024 (one workgroup's LDS)                                  HMX_SYM_GROUP_SLOTS  */
    HMX_OPT_BUILD_TIMING      = 6,  /* build   0     per-phase build times on stderr                                         HMX_BUILD_TIMING     */
    /* ---- products ---- */
    HMX_OPT_REDUCE_WAVES      = 10, /* product 0     waves per workgroup of the single-vector reduce stage (0: automatic)    HMX_REDUCE_WAVES     */
    HMX_OPT_EXPAND_WAVES      = 11, /* product 0     ... of the expand stage (0: automatic: 4, or 8 for <= 4096 row ranges)  HMX_EXPAND_WAVES     */
    HMX_OPT_MULTI_RHS_FUSED   = 12, /* product 1     several right-hand sides in one sweep over the streams (0: one product
                                                     per right-hand side)                                                    HMX_NO_FUSED_MU (inverted) */
    HMX_OPT_MATRIX_CORES      = 13, /* product 1     groups of 16 / 32 (complex: 8 / 16) right-hand sides on the matrix cores
                                                     (0: VALU kernels throughout)                                            HMX_NO_MFMA (inverted) */
    HMX_OPT_MATRIX_CORES_F32  = 14, /* product 1     ... for 4-byte real coefficients too                                    HMX_MFMA_F32         */
    HMX_OPT_WIDE_SWEEPS       = 15, /* product 1     more than 16 (complex: 8) right-hand sides: sweeps of up to 32 (16)     HMX_MFMA_WIDE        */
    HMX_OPT_SCALAR_OPERANDS   = 16, /* product -1    VALU multi-RHS reduce stage with operands in scalar registers
                                                     (-1: automatic = 4-byte coefficients only)                              HMX_MU_SCALAR        */
    HMX_OPT_SYM_MULTI_RHS     = 17, /* product -1    several right-hand sides on a symmetric / Hermitian operator: 1 or -1 on the stored
                                                     triangle (nothing else is built), 0 on an expanded view of the operator
                                                     (twice the memory; the default for real coefficients until round 5)      HMX_SYM_MU_FUSED     */
    HMX_OPT_SYM_NO_VIEW       = 18, /* product 0     never build the expanded view                                            HMX_SYM_NO_VIEW      */
    HMX_OPT_TRANSPOSED_LAYOUT = 19, /* product -1    transposed stream layout: -1 for several right-hand sides only (HBM
                                                     permitting), 1 also for single vectors, 0 never (stored data only)      HMX_TRANS_STREAMS    */
    /* ---- compression ---- */
    HMX_OPT_CALLBACK_THREADS  = 30, /* build   0     = hmx_hmatrix_set_callback_threads                                      HMX_CALLBACK_THREADS */
    HMX_OPT_CALLBACK_DRIVERS  = 31, /* build   8     generator threads that also drive a HIP stream                          HMX_CALLBACK_DRIVERS */
    HMX_OPT_POOL_SAMPLE       = 32, /* build   1     size the cross pool from a sample run over every K-th block             HMX_POOL_SAMPLE      */
    HMX_OPT_POOL_RANK_GUESS   = 33, /* build   0     a-priori rank the pool is sized for (0: 8 + 3 log10(1/eps), >= 16)      HMX_POOL_RANK_GUESS  */
    HMX_OPT_ACA_TEAMS         = 34, /* build   1     large high-rank blocks continue in teams of workgroups                  HMX_ACA_TEAM         */
    HMX_OPT_ACA_TEAM_MIN      = 35, /* build   4096  ... from this many rows + columns                                       HMX_ACA_TEAM_MIN     */
    HMX_OPT_ACA_TEAM_AFTER    = 36, /* build   48    ... after this many iterations                                          HMX_ACA_TEAM_Q       */
    HMX_OPT_ACA_TEAM_SLICE    = 37, /* build   0     entries of a line per workgroup of a team (0: adaptive 1024 / 256)      HMX_ACA_TEAM_SLICE   */
    HMX_OPT_ACA_WAVE_MAX      = 38, /* build   256   admissible blocks with both sides <= this many points are compressed by
                                                     one wave each (0: one workgroup per block throughout; <= 256)           HMX_ACA_WAVE_MAX     */
    HMX_OPT_PLACE_WRITTEN     = 39  /* product 1     the small arrays the sweeps WRITE (reduced coefficients, partial sums) are
                                                     tried at a few places of the reserved slab (hmx_device_reserve) against the
                                                     stream read meanwhile and stay where the pair runs fastest: on MI355X a write
                                                     stream costs a streaming read 16-23 % in the same third of the physical memory,
                                                     7-10 % elsewhere (0: wherever first fit puts them)                       HMX_PLACE_WRITTEN    */
} hmx_option;
int hmx_hmatrix_set_option(hmx_hmatrix *, int option /* hmx_option */, double value);
int hmx_hmatrix_get_option(const hmx_hmatrix *, int option, double *value);

/* HMatrixTreeBuilder::{sequential,openmp}_compute_blocks (tree_builder.hpp:568-666): compress every
 * admissible leaf (fallback to dense when the compressor reports failure), assemble every dense leaf
 * (HMatrix::compute_dense_data, hmatrix.hpp:222-226), then lay the result out as matvec streams. */
int hmx_hmatrix_compress(hmx_hmatrix *, int compressor, double epsilon, int reqrank);

/* recompression(hmatrix) / RecompressedLowRankGenerator (hmatrix/utils/recompression.hpp:8-31,
 * hmatrix/lrmat/utils/SVD_recompression.hpp:19-181): SVD recompression of every low-rank leaf; epsilon <= 0 uses the
 * accuracy of hmx_hmatrix_compress.  Ranks can only decrease; the matvec streams are rebuilt. */
int hmx_hmatrix_recompress(hmx_hmatrix *, double epsilon);

/* Upload path: blocks compressed elsewhere (e.g. by htool's own CPU compressors behind a user
 * VirtualGenerator).  Low rank: U is M x r column-major, V is r x N column-major (LowRankMatrix,
 * hmatrix/lrmat/lrmat.hpp:15-45); dense: M x N column-major (matrix/matrix.hpp:100).
 * Call hmx_hmatrix_finalize() after the last block. */
int hmx_hmatrix_set_block_lowrank(hmx_hmatrix *, int64_t leaf, int rank, const double *U, const double *V);
int hmx_hmatrix_set_block_dense(hmx_hmatrix *, int64_t leaf, const double *D);
int hmx_hmatrix_finalize(hmx_hmatrix *);

/* Download path (so the reference's CPU leaf loop can multiply the engine's blocks): */
int hmx_hmatrix_leaf_ranks(const hmx_hmatrix *, int32_t *rank /* num_leaves, -1 dense */);
int hmx_hmatrix_get_block(const hmx_hmatrix *, int64_t leaf, double *U_or_D, double *V);
/* The same for MANY leaves at once: the blocks are gathered on the device into htool's layouts, cross PCIe in pieces of 256 MiB and are handed
 * out to U_or_D[k] / V[k] (V[k] may be NULL for a dense leaf; V itself may be NULL when no leaf is low rank) by the host's cores while the next
 * piece is in flight -- what a plug-in that fills htool's own HMatrix needs (DeviceLowRankGenerator / DeviceDenseBlocksGenerator of
 * htool_adaptor.hpp: HMatrix::compute_low_rank_data / compute_dense_data, hmatrix/hmatrix.hpp:222-237, for every leaf), instead of one blocking
 * copy per cross.  _s / _z / _c: the other coefficient types (complex: interleaved). */
int hmx_hmatrix_get_blocks(const hmx_hmatrix *, int64_t count, const int64_t *leaves, double *const *U_or_D, double *const *V);
int hmx_hmatrix_get_blocks_s(const hmx_hmatrix *, int64_t count, const int64_t *leaves, float *const *U_or_D, float *const *V);
int hmx_hmatrix_get_blocks_z(const hmx_hmatrix *, int64_t count, const int64_t *leaves, double *const *U_or_D, double *const *V);
int hmx_hmatrix_get_blocks_c(const hmx_hmatrix *, int64_t count, const int64_t *leaves, float *const *U_or_D, float *const *V);
/* hmx_stats may grow at its END in later versions of this header.  hmx_hmatrix_stats_sized writes at most `struct_size` bytes, so a caller
 * compiled against an older (shorter) hmx_stats is never overrun; hmx_hmatrix_stats(H, out) in source code is that call with
 * sizeof(hmx_stats) of the header it was compiled with.  (The exported function of the same name, kept for binaries built before the macro
 * existed, fills the fields hmx_stats had in the last header without the macro -- everything before placed_read_gbps.)  hmx_abi_version() returns
 * HMX_ABI_VERSION of the library: it changes when a struct grows or an entry point is added, never for existing signatures. */
#define HMX_ABI_VERSION 7
int hmx_abi_version(void);
int hmx_hmatrix_stats_sized(const hmx_hmatrix *, hmx_stats *out, size_t struct_size);
int hmx_hmatrix_stats(const hmx_hmatrix *, hmx_stats *out);
#define hmx_hmatrix_stats(H, out) hmx_hmatrix_stats_sized((H), (out), sizeof(hmx_stats))
/* Complex coefficients: the entry points above that carry coefficients, for HMatrix<std::complex<double>> (_z) and
 * HMatrix<std::complex<float>> (_c).  Symmetry 'S' = complex symmetric (mirror pass with trans 'T'), 'H' = Hermitian (mirror pass
 * with trans 'C', hemv on the diagonal leaves): hmatrix/linalg/add_hmatrix_vector_product.hpp:36-54,70.  trans in {'N','T','C'};
 * as in the reference ('T' with 'H' leaves) and ('C' with 'S' leaves) are refused (:59-62).  All compressors and the
 * recompression are available (conjugated dots as htool's Blas<T>::dot, wrappers/wrapper_blas.hpp:152-157). */
int hmx_hmatrix_set_callback_z(hmx_hmatrix *, hmx_generator_fn fn, void *user);   /* out: M*N interleaved complex doubles */
int hmx_hmatrix_set_callback_c(hmx_hmatrix *, hmx_generator_fn_s fn, void *user); /* out: M*N interleaved complex floats */
int hmx_hmatrix_set_block_lowrank_z(hmx_hmatrix *, int64_t leaf, int rank, const double *U, const double *V);
int hmx_hmatrix_set_block_dense_z(hmx_hmatrix *, int64_t leaf, const double *D);
int hmx_hmatrix_get_block_z(const hmx_hmatrix *, int64_t leaf, double *U_or_D, double *V);
int hmx_hmatrix_matvec_z(hmx_hmatrix *, char trans, const double *alpha, const double *in, const double *beta, double *out, int mem, void *stream);
int hmx_hmatrix_matvec_user_z(hmx_hmatrix *, char trans, const double *alpha, const double *in, const double *beta, double *out, int mem, void *stream);
int hmx_hmatrix_matmat_row_major_z(hmx_hmatrix *, char trans, const double *alpha, const double *in, const double *beta, double *out, int mu, int mem, void *stream);
int hmx_hmatrix_set_block_lowrank_c(hmx_hmatrix *, int64_t leaf, int rank, const float *U, const float *V);
int hmx_hmatrix_set_block_dense_c(hmx_hmatrix *, int64_t leaf, const float *D);
int hmx_hmatrix_get_block_c(const hmx_hmatrix *, int64_t leaf, float *U_or_D, float *V);
int hmx_hmatrix_matvec_c(hmx_hmatrix *, char trans, const float *alpha, const float *in, const float *beta, float *out, int mem, void *stream);
int hmx_hmatrix_matvec_user_c(hmx_hmatrix *, char trans, const float *alpha, const float *in, const float *beta, float *out, int mem, void *stream);
int hmx_hmatrix_matmat_row_major_c(hmx_hmatrix *, char trans, const float *alpha, const float *in, const float *beta, float *out, int mu, int mem, void *stream);

/* Memory: after compression the device holds the streams the products read AND the pool they were packed from (kept for
 * get_block / save / recompress / the transposed layout; 14 GB next to 18.5 GB of streams at N=1e6).  This gives the pool back:
 * only products remain possible (transposed single-vector products run on the stored data and do not need the pool; bit 0 of
 * with_transposed builds the transposed stream layout first, which the fused multi-RHS 'T' products prefer; bit 1 builds the expanded
 * view of a compact symmetric operator first -- without them such products run on the stored data / the stored triangle, which needs
 * no second layout). */
int hmx_hmatrix_release_factors(hmx_hmatrix *, int with_transposed);
/* Transposed products ('T' / 'C').  One vector: on the STORED data, as the reference does (it swaps the cluster roles on the same leaves,
 * hmatrix/linalg/add_hmatrix_vector_product.hpp:74-81) -- column sums of the E-streams per row range, then an owner-computes sweep over the
 * R-streams; every coefficient is read once, no atomics, bit-reproducible; index tables of about 3 % of the operator are built on first use.
 * Several right-hand sides: fused multi-RHS kernels on a transposed stream layout (a second copy of the streams; needs 1.15 x the operator
 * free in HBM), otherwise the same stored-data form for 16 real / 8 complex right-hand sides per sweep on the matrix cores (1.4-1.5 x slower).
 * HMX_OPT_TRANSPOSED_LAYOUT = 1 also runs single vectors on the transposed layout (N = 1e6: 2.8 instead of 3.1 ms), 0 never builds it.  A
 * row-restricted symmetric / Hermitian operator (mirrored leaves among ordinary ones) multiplies transposed on its transposed view only:
 * without it (option 0, no room, factors released without bit 0 of with_transposed) the call fails with HMX_ERR_UNSUPPORTED and says why.
 * Multi-RHS products on symmetric / Hermitian storage: HMX_OPT_SYM_MULTI_RHS (stored triangle, or an expanded view of the operator).
 * All of these -- plus work vectors and, for the user-numbering front ends, permutation and staging buffers -- are otherwise built inside
 * the FIRST product that needs them.  hmx_hmatrix_prepare(H, trans, mu) builds and allocates NOW everything products with this `trans` and
 * this many right-hand sides (1: the vector products) need; afterwards such products allocate nothing (hmx_device_alloc_count does not
 * move): no latency cliff or out-of-memory condition in the middle of a Krylov solve.  hmx_stats.transposed_bytes / expanded_bytes say
 * what the extra layouts cost.  Optional. */
int hmx_hmatrix_prepare(hmx_hmatrix *, char trans, int mu);

/* Device memory for a vector (or a row-major block of right-hand sides) that products of this operator WRITE -- y of
 * add_hmatrix_vector_product / Y of add_hmatrix_matrix_product_row_major (hmatrix/linalg/add_hmatrix_vector_product.hpp:17-32; the reference's
 * callers own plain std::vector's).  Any device pointer is a valid output; one from here lies where this operator's sweeps write fastest.  On
 * MI355X a streaming read loses 16-23 % to a write stream of ~1 % of its bytes when both lie in the same third of the physical memory and
 * 7-10 % when they do not; for the arrays the library owns that is settled at build time, the OUTPUT belongs to the caller -- the two speeds
 * of the 16-RHS expand kernel round 5 could not explain (N = 1e6 fp64: 2.03 / 2.30 ms) are nothing but where Y lies relative to the E-streams
 * (profiles/r6_modes_output_place.log).  When `bytes` is the size of an output of this operator (its rows -- trans 'T' / 'C': columns -- times a
 * number of right-hand sides) and hmx_device_reserve gave the library a slab to choose from, the operator's own product of that shape is timed
 * on zero operands with the vector at ten places, and the vector stays where it ran fastest (about 25 products, once, here -- never inside a
 * product call of the caller); otherwise the placement probe of the build decides, or the driver.  Zero-filled.  Freed by
 * hmx_hmatrix_free_vector or with the operator. */
int hmx_hmatrix_alloc_vector(hmx_hmatrix *, char trans, int64_t bytes, void **device_ptr);
int hmx_hmatrix_free_vector(hmx_hmatrix *, void *device_ptr);

/* Binary dump of the compressed operator (no counterpart in the reference): header, leaf table with
 * ranks, then per leaf U (M x r) and V (r x N) or the dense M x N block, all column-major as in htool's
 * LowRankMatrix / Matrix.  Load needs the block tree the file was written for and picks the coefficient type from the file. */
int hmx_hmatrix_save(const hmx_hmatrix *, const char *path);
int hmx_hmatrix_load(const hmx_block_tree *bt, int device_id, const char *path, hmx_hmatrix **out);

/* openmp_internal_add_hmatrix_vector_product (hmatrix/linalg/add_hmatrix_vector_product.hpp:107-170):
 * out = alpha * op(H) * in + beta * out, cluster numbering, vectors local to the H-matrix' root clusters.
 * trans in {'N','T','C'} ('C' = 'T' for real coefficients; refused on 'S' leaves as in the reference, :59-62).  mu = 1. */
int hmx_hmatrix_matvec(hmx_hmatrix *, char trans, double alpha, const double *in, double beta, double *out,
                       int mem /* hmx_mem */, void *stream);
/* add_hmatrix_vector_product (same file :173-197): user numbering; permutations done on the device.
 * Only valid when the block tree root is the cluster-tree root (or a partition with local permutation). */
int hmx_hmatrix_matvec_user(hmx_hmatrix *, char trans, double alpha, const double *in, double beta, double *out,
                            int mem, void *stream);
/* openmp_internal_add_hmatrix_matrix_product_row_major (hmatrix/linalg/add_hmatrix_matrix_product_row_major.hpp:112-178):
 * in/out row-major (mu fastest), cluster numbering. */
int hmx_hmatrix_matmat_row_major(hmx_hmatrix *, char trans, double alpha, const double *in, double beta, double *out,
                                 int mu, int mem, void *stream);

/* add_hmatrix_matrix_product (hmatrix/linalg/add_hmatrix_matrix_product.hpp:176-205): column-major in (n x mu) and out (m x mu),
 * USER numbering; permutation + transposition to the row-major kernels' layout on the device (same file :26-77). */
int hmx_hmatrix_matmat_user(hmx_hmatrix *, char trans, double alpha, const double *in, double beta, double *out,
                            int mu, int mem, void *stream);
int hmx_hmatrix_matmat_user_s(hmx_hmatrix *, char trans, float alpha, const float *in, float beta, float *out, int mu, int mem, void *stream);
int hmx_hmatrix_matmat_user_z(hmx_hmatrix *, char trans, const double *alpha, const double *in, const double *beta, double *out, int mu, int mem, void *stream);
int hmx_hmatrix_matmat_user_c(hmx_hmatrix *, char trans, const float *alpha, const float *in, const float *beta, float *out, int mu, int mem, void *stream);

/* fp32-coefficient variants (handles created with hmx_hmatrix_create_s); same semantics, float data.
 * All arithmetic of compression and product is then done in float, as htool does for CoefficientPrecision=float. */
int hmx_hmatrix_set_block_lowrank_s(hmx_hmatrix *, int64_t leaf, int rank, const float *U, const float *V);
int hmx_hmatrix_set_block_dense_s(hmx_hmatrix *, int64_t leaf, const float *D);
int hmx_hmatrix_get_block_s(const hmx_hmatrix *, int64_t leaf, float *U_or_D, float *V);
int hmx_hmatrix_matvec_s(hmx_hmatrix *, char trans, float alpha, const float *in, float beta, float *out, int mem, void *stream);
int hmx_hmatrix_matvec_user_s(hmx_hmatrix *, char trans, float alpha, const float *in, float beta, float *out, int mem, void *stream);
int hmx_hmatrix_matmat_row_major_s(hmx_hmatrix *, char trans, float alpha, const float *in, float beta, float *out, int mu, int mem, void *stream);

/* ---- DistributedOperator over RCCL (distributed_operator/distributed_operator.hpp:20-61 with one global-to-local operator, the
 * rank's row-restricted H-matrix; products of distributed_operator/linalg/add_distributed_operator_vector_product_
 * {global_to_global,local_to_local}.hpp:18-85,19-89).  One process per GPU; the communicator and the stream are the caller's.
 * The collectives are called through a table of function pointers so that the caller decides WHICH RCCL (the one it linked, or
 * the one its framework bundles); a NULL table makes libhmx dlopen("librccl.so").  Vectors are device pointers in PARTITION
 * (= cluster) numbering.  MPI_Allgatherv -> ncclAllGather for equal parts, otherwise one grouped ncclBroadcast per part;
 * MPI_Allreduce -> ncclAllReduce(sum). */
typedef struct {
    int (*all_gather)(const void *send, void *recv, size_t sendcount, int datatype, void *comm, void *stream);
    int (*all_reduce)(const void *send, void *recv, size_t count, int datatype, int op, void *comm, void *stream);
    int (*broadcast)(const void *send, void *recv, size_t count, int datatype, int root, void *comm, void *stream);
    int (*group_start)(void);
    int (*group_end)(void);
} hmx_rccl_api; /* ncclAllGather, ncclAllReduce, ncclBroadcast, ncclGroupStart, ncclGroupEnd (rccl.h) cast to these shapes */
typedef struct hmx_dist hmx_dist;
/* `local` = the H-matrix built with target_partition_number = rank on the cluster trees whose partitions define the row
 * distribution (kept by reference: it must outlive the hmx_dist). */
int hmx_dist_create(hmx_hmatrix *local, const hmx_cluster_tree *target, const hmx_cluster_tree *source, void *nccl_comm, int rank,
                    int world_size, const hmx_rccl_api *api, hmx_dist **out);
void hmx_dist_destroy(hmx_dist *);
/* DistributedOperator::add_local_to_local_operator (distributed_operator/distributed_operator.hpp:50-53) with a LocalToLocalHMatrix
 * (implementations/local_to_local_operators/hmatrix.hpp:15-56): `diag` = the H-matrix on (target partition rank) x (source partition
 * rank), built on a block tree from hmx_block_tree_create_local -- the block-diagonal operator of DefaultLocalApproximationBuilder
 * (distributed_operator/utility.hpp:64-88).  Every product adds its contribution on the rank's slices after the global-to-local
 * operator's, beta applied once (global_to_global.hpp:63-72, local_to_local.hpp:27-59).  hmx_dist_create accepts local = NULL for
 * an operator that consists of local-to-local operators only (its local-to-local products then exchange nothing).  Kept by
 * reference; same coefficient type as `local`. */
int hmx_dist_add_local_to_local_operator(hmx_dist *, hmx_hmatrix *diag);
/* DistributedOperator::add_global_to_local_operator (distributed_operator/distributed_operator.hpp:47-49): one more operator from the whole
 * source numbering to this rank's rows.  The reference holds vectors of both kinds of operators and every product loops over them
 * (global_to_global.hpp:63-72: all global-to-local operators, then all local-to-local ones, beta applied once); so does this layer --
 * both add functions may be called several times.  Root block of `op`: (target partition rank) x (whole source cluster); same coefficient
 * type as the others; kept by reference.  With more than one operator registered the output exchange is the plain one. */
int hmx_dist_add_global_to_local_operator(hmx_dist *, hmx_hmatrix *op);
/* y = alpha * op(A) * x + beta * y, x and y whole vectors replicated on every rank (global_to_global.hpp:18-85); the coefficient
 * type is the local operator's: pointers to double / float / interleaved complex accordingly, alpha / beta as in the matvec
 * entry points of that type but always passed by pointer here. */
int hmx_dist_matvec_global_to_global(hmx_dist *, char trans, const void *alpha, const void *x, const void *beta, void *y, void *stream);
/* The same for mu right-hand sides, X (n x mu) and Y (m x mu) row-major with mu fastest
 * (distributed_operator/linalg/add_distributed_operator_matrix_product_row_major_global_to_global.hpp:18-85): the exchanged slices are
 * mu-interleaved rows. */
int hmx_dist_matmat_row_major_global_to_global(hmx_dist *, char trans, const void *alpha, const void *X, const void *beta, void *Y, int mu, void *stream);
/* local slices in and out (the Krylov-side contract, local_to_local.hpp:19-89): all-gather of x, local product ('N');
 * local product into a zeroed global vector, all-reduce, slice (transposed). */
int hmx_dist_matvec_local_to_local(hmx_dist *, char trans, const void *alpha, const void *x_local, const void *beta, void *y_local, void *stream);

/* The same contract for mu right-hand sides: X_local (n_k x mu) and Y_local (m_k x mu) row-major with mu fastest, partition numbering
 * (distributed_operator/linalg/add_distributed_operator_matrix_product_row_major_local_to_local.hpp:19-95 -- what HPDDMOperator::GMV
 * calls for mu != 1, wrappers/wrapper_hpddm.hpp:126).  'N': all-gather of the mu-interleaved rows of X (local_to_global,
 * linalg/utility.hpp:11-28), local product; transposed: local product into a zeroed global matrix, then the reference's
 * MPI_Alltoallv + p axpys (:64-93) as one reduce-scatter (ncclReduceScatter on equal partitions, all-reduce + slice otherwise). */
int hmx_dist_matmat_row_major_local_to_local(hmx_dist *, char trans, const void *alpha, const void *X_local, const void *beta, void *Y_local, int mu, void *stream);
/* Column-major front ends (device pointers): X (n x mu) and Y (m x mu) stored column by column.
 *   numbering = HMX_NUMBERING_USER:      add_distributed_operator_matrix_product_global_to_global (distributed_operator/linalg/
 *       add_distributed_operator_matrix_product_global_to_global.hpp:132-279) resp. add_distributed_operator_matrix_product_local_to_local
 *       (..._matrix_product_local_to_local.hpp:66-120): every column is permuted between user and partition numbering
 *       (global_to_partition_numbering / local_to_local_partition_numbering) and the operands are transposed to the row-major layout
 *       -- one kernel each way on the device --, then the row-major product runs.  The local form needs cluster trees whose
 *       permutation is local to the partitions (create_cluster_tree_from_local_partition, or one partition).
 *   numbering = HMX_NUMBERING_PARTITION: the internal_ variants of the same files (:18-117 resp. :20-49): transposition only.
 * mu = 1 with user numbering is add_distributed_operator_vector_product_global_to_global (..._vector_product_global_to_global.hpp:97-118)
 * resp. add_distributed_operator_vector_product_local_to_local (..._vector_product_local_to_local.hpp:99-125). */
typedef enum { HMX_NUMBERING_PARTITION = 0, HMX_NUMBERING_USER = 1 } hmx_numbering;
int hmx_dist_matmat_global_to_global(hmx_dist *, char trans, const void *alpha, const void *X, const void *beta, void *Y, int mu, int numbering, void *stream);
int hmx_dist_matmat_local_to_local(hmx_dist *, char trans, const void *alpha, const void *X_local, const void *beta, void *Y_local, int mu, int numbering, void *stream);

/* HPDDMOperator::GMV (wrappers/wrapper_hpddm.hpp:102-142), the body of the Krylov-side callback up to HPDDM's own overlap exchange:
 * `in` / `out` are DEVICE pointers, column-major with leading dimension `dof` (HPDDM's local size including the overlap that follows the
 * first local_size rows), mu columns.  The first local_size rows of every column are transposed to the row-major layout (:108-116), the
 * local-to-local product with alpha = 1, beta = 0 runs (:118-125), the result is transposed back and rows [local_size, dof) of `out`
 * are zeroed (:128-138).  What remains for the caller is GMV's last statement, this->exchange(out, mu). */
int hmx_dist_gmv(hmx_dist *, const void *in, void *out, int mu, int dof, void *stream);

/* Overlap of the output exchange with the computation ("all via RCCL on a side HIP stream; overlap by chunking the
 * output range").  chunks >= 2: the expand stage of a trans = 'N' global-to-global product runs in that many row chunks on the
 * caller's stream; an event after each chunk hands its rows to a side stream, where they are exchanged (grouped ncclBroadcast,
 * one per rank: MPI_Allgatherv of global_to_global.hpp:76 restricted to the chunk) while the next chunk computes; the caller's
 * stream waits for the last exchange.  COLLECTIVE (the ranks exchange their chunk boundaries on `stream`): every rank calls it
 * with the same `chunks`.  If any rank's operator cannot be chunked (fused symmetric storage adds to rows after the expand stage)
 * all ranks keep the single exchange after the product; hmx_dist_overlap_chunks tells.  chunks <= 1: off (the default). */
int hmx_dist_set_overlap(hmx_dist *, int chunks, void *stream);
int hmx_dist_overlap_chunks(const hmx_dist *);
/* The same for hmx_dist_matmat_row_major_global_to_global with mu > 1, trans = 'N' (BASELINE configs[4]: 16 right-hand sides on 8 GPUs
 * exchange 32 MB per rank and product): the expand kernels of ALL groups of right-hand sides run chunk by chunk, the chunk's
 * mu-interleaved rows are exchanged on the side stream.  The row chunks are those of the layout the multi-RHS product runs on (the
 * expanded view of a compact symmetric operator is chunkable where its fused single-vector product is not); the ranks agree on them
 * inside the FIRST such product after hmx_dist_set_overlap (collective, like the product itself).  hmx_dist_overlap_chunks_multi:
 * the chunks in use for multi-RHS products (0: not yet exchanged, or some rank cannot chunk). */
int hmx_dist_overlap_chunks_multi(const hmx_dist *);
/* ncclReduceScatter (same argument shapes as rccl.h) for the transposed local-to-local product (MPI_Alltoallv + axpys of
 * local_to_local.hpp:77) when the collective table was given by the caller; with a NULL table it is taken from librccl.so.
 * Without it, or with unequal partitions, that product uses all-reduce + slice. */
int hmx_dist_set_reduce_scatter(hmx_dist *, int (*reduce_scatter)(const void *send, void *recv, size_t recvcount, int datatype, int op, void *comm, void *stream));

/* Point-to-point exchange of the output slices of the trans = 'N' global-to-global products (single vector, chunked / overlapped,
 * multi-RHS) instead of ncclAllGather / grouped ncclBroadcast: every rank sends its rows straight to each peer and receives each
 * peer's rows in place, one grouped ncclSend / ncclRecv per pair (MPI_Allgatherv of global_to_global.hpp:76 as the pairwise
 * exchange it is).  An 8-GPU MI355X node is a full xGMI mesh: every pair has its own link, so a slice crosses one link once where
 * a ring forwards it seven times in turn.  `send` / `recv` have the argument shapes of ncclSend / ncclRecv (rccl.h); NULL keeps
 * the ones already known (taken from librccl.so when the operator was created with a NULL collective table).  enable != 0 on
 * EVERY rank or on none.  Results are identical either way. */
int hmx_dist_set_point_to_point(hmx_dist *, int (*send)(const void *buf, size_t count, int datatype, int peer, void *comm, void *stream),
                                int (*recv)(void *buf, size_t count, int datatype, int peer, void *comm, void *stream), int enable);

/* How the disjoint output slices of a trans = 'N' global-to-global product reach every rank: 0 (default) = exchange of the slices
 * (MPI_Allgatherv of global_to_global.hpp:76: ncclAllGather, grouped ncclBroadcast or pairwise send / recv); 1 = ncclAllReduce of the
 * zero-padded length-N output vector ("RCCL all-reduce of the output vector": p times the bytes for the same result; single exchange,
 * not chunked).  Every rank must choose the same. */
int hmx_dist_set_output_collective(hmx_dist *, int all_reduce);
/* Test / A-B switches of one DistributedOperator (every rank the same value).  The environment variables HMX_DIST_FORCE_COLLECTIVES,
 * HMX_DIST_NO_ALLGATHER, HMX_DIST_NO_REDUCE_SCATTER give the initial values, read once in hmx_dist_create. */
typedef enum {
    HMX_DIST_OPT_FORCE_COLLECTIVES = 1, /* issue the collectives even with one rank */
    HMX_DIST_OPT_NO_ALLGATHER      = 2, /* grouped broadcasts instead of ncclAllGather for equal parts */
    HMX_DIST_OPT_NO_REDUCE_SCATTER = 3  /* all-reduce + slice instead of ncclReduceScatter (transposed local-to-local product) */
} hmx_dist_option;
int hmx_dist_set_option(hmx_dist *, int option /* hmx_dist_option */, int value);
/* Exposed exchange time, measured: with profiling on, hmx_dist_matvec_global_to_global (trans = 'N') records events on the caller's
 * stream at its start, after its last local kernel and when the whole result is there.  hmx_dist_last_exchange_ms returns
 * local_ms = start -> last local kernel done and exposed_ms = from there to the end (the exchange minus whatever ran under the local
 * kernels on the side stream); it synchronises on the last event. */
int hmx_dist_set_profiling(hmx_dist *, int enabled);
int hmx_dist_last_exchange_ms(hmx_dist *, float *local_ms, float *exposed_ms);

/* Timing hooks for bench.py: average duration (ms) of the last matvec's kernels measured with HIP
 * events on the launch stream; names[i] is a static string. */
int hmx_hmatrix_last_kernel_times(const hmx_hmatrix *, int max, const char **names, float *ms);
/* Enable/disable per-kernel event timing (adds event records to the stream). */
int hmx_hmatrix_set_profiling(hmx_hmatrix *, int enabled);

/* libhmx recycles large device buffers inside the process (rebuilding operators would otherwise hit multi-second hipMalloc calls);
 * this returns every parked buffer to the driver.  HMX_CACHE_GB (default 48) bounds what is kept. */
int hmx_device_trim_cache(void);
/* Takes `bytes` of device memory from the driver ONCE; device arrays of 1 MiB and more that libhmx allocates afterwards (cross pool,
 * streams, views, work vectors) are carved out of such slabs before hipMalloc is asked, and return to them when released.  For callers
 * that build operators repeatedly or time a build: on this platform hipMalloc stalls for seconds while the driver scrubs memory that
 * was released shortly before, by this or by the previous process.  A slab is also where HMX_OPT_PLACE_WRITTEN looks for a good place for
 * the arrays the products write: a generous one (half of the free memory) offers addresses of more than one kind.  May be called
 * several times (one more slab each); hmx_device_trim_cache frees the slabs nothing lives in. */
int hmx_device_reserve(int device_id, int64_t bytes);
/* Diagnostics (tools/output_place_probe.py): a range of a reserved slab as close as possible to the fraction `frac` (0 ... 1) of its extent,
 * and its return.  What hmx_hmatrix_alloc_vector does after measuring, with the place given by hand: how a kernel's time depends on where
 * the vector it writes lies. */
int hmx_device_slab_alloc_at(int device_id, int64_t bytes, double frac, void **device_ptr);
int hmx_device_slab_free(int device_id, void *device_ptr, int64_t bytes);
/* Wall time (seconds) this process has spent inside hipMalloc on behalf of libhmx so far: large allocations sporadically take seconds
 * on this platform, callers that time builds report it separately. */
double hmx_device_malloc_seconds(void);
/* Device arrays libhmx has handed out so far in this process (from the driver, a reserved slab or its buffer cache): a product call that
 * leaves it unchanged allocated nothing. */
int64_t hmx_device_alloc_count(void);

/* Device bandwidth probe: plain 16 B/lane copy of `bytes` bytes, returns GB/s (read+write counted). */
int hmx_device_copy_bandwidth(int device_id, int64_t bytes, int reps, double *gbps);
/* ... and a read-only one (16 B/lane non-temporal loads summed in registers): what a streaming-read kernel can reach at best. */
int hmx_device_read_bandwidth(int device_id, int64_t bytes, int reps, double *gbps);

#ifdef __cplusplus
}
#endif
#endif

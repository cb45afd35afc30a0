// engine_api.hpp -- what engine.hip (C ABI, DistributedOperator layer) sees of one instantiation of the engine: the opaque operator and
// the api_* functions of engine_body.hpp.  Included inside namespace hmx::{f64,f32,z64,c32} with `scalar` defined; no include guard
// on purpose.
struct HMat;
int api_create(const hmx_block_tree *bt, int device_id, HMat **out);
int api_set_kernel(HMat *H, int kernel, const double *params, int nparams, int dim, const double *tc, const double *sc);
int api_set_callback(HMat *H, void (*fn)(void *, int, int, const int32_t *, const int32_t *, scalar *), void *user);
int api_set_callback_threads(HMat *H, int threads);
int api_set_option(HMat *H, int option, double value);
int api_get_option(const HMat *H, int option, double *value);
int api_compress(HMat *Hp, int compressor, double epsilon, int reqrank);
int api_recompress(HMat *Hp, double epsilon);
int api_set_block_lowrank(HMat *H, int64_t leaf, int rank, const scalar *U, const scalar *V);
int api_set_block_dense(HMat *H, int64_t leaf, const scalar *D);
int api_finalize(HMat *Hp);
int api_leaf_ranks(const HMat *H, int32_t *rank);
int api_get_block(const HMat *Hc, int64_t leaf, scalar *U_or_D, scalar *V);
int api_get_blocks(const HMat *Hc, int64_t count, const int64_t *leaves, scalar *const *U_or_D, scalar *const *V);
int api_save(const HMat *Hc, const char *path);
int api_load(const hmx_block_tree *bt, int device_id, FILE *f, const HmxFileHeader &hd, HMat **out);
int api_release_factors(HMat *Hp, int with_transposed);
int api_stats(const HMat *H, hmx_stats *out);
int api_matvec(HMat *Hp, char trans, scalar alpha, const scalar *in, scalar beta, scalar *out, int mem, void *stream);
int api_matvec_chunked(HMat *Hp, scalar alpha, const scalar *in, scalar beta, scalar *out, void *stream, int nchunks, after_chunk_fn after_chunk, void *user, int *used);
int api_chunk_bounds(HMat *Hp, int nchunks, int *n_out, int32_t *bounds);
int api_matmat_chunked(HMat *Hp, scalar alpha, const scalar *in, scalar beta, scalar *out, int mu, void *stream, int nchunks, after_chunk_fn after_chunk, void *user, int *used);
int api_chunk_bounds_mu(HMat *Hp, int nchunks, int *n_out, int32_t *bounds);
int api_matvec_user(HMat *Hp, char trans, scalar alpha, const scalar *in, scalar beta, scalar *out, int mem, void *stream);
int api_matmat_row_major(HMat *Hp, char trans, scalar alpha, const scalar *in, scalar beta, scalar *out, int mu, int mem, void *stream);
int api_matmat_user(HMat *Hp, char trans, scalar alpha, const scalar *in, scalar beta, scalar *out, int mu, int mem, void *stream);
int api_set_profiling(HMat *H, int enabled);
int api_last_kernel_times(const HMat *H, int max, const char **names, float *ms);
int api_prepare(HMat *Hp, char trans, int mu);
int api_alloc_vector(HMat *Hp, char trans, int64_t bytes, void **ptr);
int api_free_vector(HMat *Hp, void *ptr);
int api_device_of(const HMat *H);   // device the operator lives on
int api_root(const HMat *H, int32_t *t_off_size_s_off_size);
void api_destroy(HMat *H);
// y = w + beta * y on `stream` (the DistributedOperator layer's accumulation of exchanged slices)
void api_axpby(int64_t n, const scalar *w, scalar beta, scalar *y, hipStream_t st);

/*
 * mlsgpu_hip.h -- C-ABI of the MI355X (gfx950) per-bucket device pipeline.
 *
 * The reference (bmerry/mlsgpu) has no C plugin interface for this path: its
 * boundary is the C++ class surface consumed by
 * DeviceWorkerGroupBase::Worker::operator() (src/workers.cpp:232-286), typed on
 * cl::Context / cl::CommandQueue / cl::Buffer / cl::Image2D.  This header is
 * the C-ABI a maintainer binds instead; each entry point cites the reference
 * interface it replaces.  mlsgpu_amd/host/ holds header-only C++ classes with
 * the reference's own names (SplatTreeCL, MlsFunctor, Marching, ...) built on
 * this ABI; INTEGRATION.md shows the reference-side changes.
 *
 * Conventions
 *  - every function returns MLSGPU_OK (0) or an error code; the text of the
 *    last error of the calling thread is mlsgpu_hip_last_error();
 *  - error classes follow the reference's exceptions: MLSGPU_ERR_LENGTH ~
 *    std::length_error, MLSGPU_ERR_INVALID ~ std::invalid_argument,
 *    MLSGPU_ERR_HIP ~ cl::Error, MLSGPU_ERR_NOMEM ~ CL_MEM_OBJECT_ALLOCATION_FAILURE;
 *  - objects are NOT thread-safe; one worker thread owns one context (= one
 *    HIP stream) and the objects made from it (src/mls.h:74-77,
 *    src/workers.h:183-206);
 *  - "d" prefixed pointers are device pointers on the context's device.
 */
#ifndef MLSGPU_HIP_H
#define MLSGPU_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MLSGPU_OK 0
#define MLSGPU_ERR_INVALID 1
#define MLSGPU_ERR_LENGTH 2
#define MLSGPU_ERR_HIP 3
#define MLSGPU_ERR_NOMEM 4
#define MLSGPU_ERR_CALLBACK 5
#define MLSGPU_ERR_DENSITY 6   /* Bucket::DensityError, src/bucket.h:52-65 */
#define MLSGPU_ERR_FORMAT 7    /* FastPly::FormatError, src/fast_ply.h:60-75 */

/* struct Splat, src/splat.h:40-46 == kernels/octree.cl:32-36 (32 bytes, AoS).
 * After mlsgpu_hip_tree_build the radius slot holds 1/radius^2 (octree.cl:193). */
typedef struct mlsgpu_splat
{
    float position[3];
    float radius;
    float normal[3];
    float quality;
} mlsgpu_splat;

/* MlsShape, src/mls.h:47-51 */
#define MLSGPU_SHAPE_SPHERE 0
#define MLSGPU_SHAPE_PLANE 1

/* Marching::Swathe (+ImageParams), src/marching.h:173-198.  The distance
 * field lives in a linear float buffer that is addressed exactly like the
 * reference's 2-D image: corner (x, y, z) is field[(y + z*zStride + zBias) * pitch + x]. */
typedef struct mlsgpu_swathe
{
    uint32_t width, height;
    uint32_t zStride;
    int32_t zBias;
    uint32_t zFirst, zLast;   /* closed interval of corner slices */
} mlsgpu_swathe;

/* DeviceKeyMesh, src/mesh.h:101-123: packed float xyz vertices, uint32 index
 * triplets, uint64 keys for all vertices (meaningful for external ones only);
 * internal vertices first. */
typedef struct mlsgpu_mesh
{
    float *dVertices;
    uint32_t *dTriangles;
    uint64_t *dVertexKeys;
    uint64_t numVertices;
    uint64_t numTriangles;
    uint64_t numInternalVertices;
} mlsgpu_mesh;

typedef struct mlsgpu_ctx mlsgpu_ctx;
typedef struct mlsgpu_tree mlsgpu_tree;
typedef struct mlsgpu_mls mlsgpu_mls;
typedef struct mlsgpu_marching mlsgpu_marching;
typedef struct mlsgpu_worker mlsgpu_worker;

/* Marching::Generator, src/marching.h:204-253.  enqueue() must enqueue (on
 * `stream`, a hipStream_t) work that fills slices zFirst..zLast of dField. */
typedef struct mlsgpu_generator
{
    uint32_t alignment[3];
    int (*enqueue)(void *user, void *stream, float *dField, uint64_t pitch, const mlsgpu_swathe *swathe);
    void *user;
} mlsgpu_generator;

/* Marching::OutputFunctor, src/marching.h:516-546.  The mesh is valid until
 * the functor returns (it may enqueue reads on `stream` and must synchronise
 * them itself, or copy with mlsgpu_hip_mesh_read). */
typedef int (*mlsgpu_output_fn)(void *user, void *stream, const mlsgpu_mesh *mesh);

const char *mlsgpu_hip_last_error(void);

/* ---- context: one device + one in-order stream (cl::Context + cl::CommandQueue of a worker,
 *      src/workers.cpp:207-221).  stream == NULL creates a private stream; otherwise the given
 *      hipStream_t (e.g. a torch stream) is used and not destroyed. ---- */
int mlsgpu_hip_ctx_create(int device, void *stream, mlsgpu_ctx **out);
void mlsgpu_hip_ctx_destroy(mlsgpu_ctx *ctx);
void *mlsgpu_hip_ctx_stream(mlsgpu_ctx *ctx);
int mlsgpu_hip_ctx_synchronize(mlsgpu_ctx *ctx);
/* Calls that work on the context keep their device scratch from call to call (mlsgpu_hip_bucket: the member lists, the kept
 * ranges and the counters of every recursion depth -- 24 GB behind a 10^9-splat cloud; the reference's Bucket::bucket
 * holds its counters and ranges on the host only for the duration of a call, src/bucket_impl.h:439-560).  This waits for the
 * context's stream and hands that memory back; lists handed to a callback must have been consumed.  Returns the error code. */
int mlsgpu_hip_ctx_release_scratch(mlsgpu_ctx *ctx);
int mlsgpu_hip_device_count(int *count);

/* Device / pinned memory helpers (CLH::PinnedMemory, src/clh.h:334-477; cl::Buffer). */
int mlsgpu_hip_malloc(mlsgpu_ctx *ctx, size_t bytes, void **dptr);
int mlsgpu_hip_free(mlsgpu_ctx *ctx, void *dptr);
int mlsgpu_hip_host_alloc(size_t bytes, void **hptr);
int mlsgpu_hip_host_free(void *hptr);
int mlsgpu_hip_memcpy_h2d(mlsgpu_ctx *ctx, void *dst, const void *src, size_t bytes, int async);
int mlsgpu_hip_memcpy_d2h(mlsgpu_ctx *ctx, void *dst, const void *src, size_t bytes, int async);
int mlsgpu_hip_memcpy_d2d(mlsgpu_ctx *ctx, void *dst, const void *src, size_t bytes);   /* async on the stream */
int mlsgpu_hip_memset(mlsgpu_ctx *ctx, void *dst, int value, size_t bytes);

/* Per-kernel device timing, the `--statistics-cl` facility (src/statistics_cl.cpp:62-160) with the
 * reference's stat names (kernel.octree.*.time, kernel.mls.processCorners.time, kernel.marching.*.time,
 * kernel.scaleBias.time).  Timing uses hipEvent pairs on the context's stream. */
int mlsgpu_hip_ctx_set_timing(mlsgpu_ctx *ctx, int enabled);
/* Synchronises, then returns total milliseconds and launch count of `name` since the last reset. */
int mlsgpu_hip_ctx_get_stat(mlsgpu_ctx *ctx, const char *name, double *totalMs, uint64_t *launches);
int mlsgpu_hip_ctx_reset_stats(mlsgpu_ctx *ctx);
/* Writes "name total_ms launches\n" lines; returns number of bytes needed.  The registry also carries the reference's
 * non-timer statistics of Marching under their own names (src/marching.cpp:350-352, always on): marching.overflow,
 * marching.slices.nonempty, marching.shipouts -- the SUM of the samples in the total_ms column, their number in the launches
 * column (Statistics::Variable's sum and n). */
size_t mlsgpu_hip_ctx_dump_stats(mlsgpu_ctx *ctx, char *buf, size_t bufSize);

/* ---- SplatTreeCL (src/splat_tree_cl.h:216-296) ---- */
#define MLSGPU_TREE_MAX_LEVELS 10                          /* src/splat_tree_cl.h:76 */
#define MLSGPU_TREE_MAX_SPLATS (((uint64_t) 1 << 31) / 16) /* src/splat_tree_cl.h:88 */
int mlsgpu_hip_tree_create(mlsgpu_ctx *ctx, uint64_t maxLevels, uint64_t maxSplats, mlsgpu_tree **out);
void mlsgpu_hip_tree_destroy(mlsgpu_tree *tree);
/* SplatTreeCL::resourceUsage: device bytes a tree of this capacity allocates. */
uint64_t mlsgpu_hip_tree_resource_usage(uint64_t maxLevels, uint64_t maxSplats);
/* SplatTreeCL::enqueueBuild, src/splat_tree_cl.cpp:269-335.  Enqueues on the context's stream, with one stream
 * synchronisation inside (the entry count sizes the sort); the tree is complete in stream order on return.  Borrows
 * dSplats until mlsgpu_hip_tree_clear_splats and MUTATES it (radius -> 1/radius^2). */
int mlsgpu_hip_tree_build(mlsgpu_tree *tree, mlsgpu_splat *dSplats, uint64_t firstSplat, uint64_t numSplats,
                          const uint32_t size[3], const int32_t offset[3], uint32_t subsamplingShift);
/* Batches: a device worker may take several buckets -- the SubItems of one WorkItem (src/workers.h:148-181), which the
 * reference's worker walks one by one (src/workers.cpp:232-286) -- through the path in lock-step.  Every kernel of the path
 * has a bucket dimension (blockIdx.y = bucket, per-bucket arguments in the kernel-argument segment), so a batch is ONE set
 * of launches and three host decisions; per bucket the results are those of the one-bucket entry points, bit for bit. */
#define MLSGPU_MAX_BATCH 8
typedef struct mlsgpu_tree_build
{
    mlsgpu_splat *dSplats;
    uint64_t firstSplat, numSplats;
    uint32_t size[3];
    int32_t offset[3];
} mlsgpu_tree_build;
/* SplatTreeCL::enqueueBuild for `count` buckets at once: trees[k] (distinct trees of one context and one depth) is built
 * from builds[k], arguments as mlsgpu_hip_tree_build.  One read-back of all the entry counts. */
int mlsgpu_hip_tree_build_batch(mlsgpu_tree *const *trees, const mlsgpu_tree_build *builds, uint32_t count,
                                uint32_t subsamplingShift);
/* mutate = 0: builds leave dSplats as they are (no radius -> 1/radius^2, kernels/octree.cl:193); mlsgpu_hip_mls_set then
 * makes processCorners take the reciprocal itself while it stages a splat -- the same expression on the same value, so the
 * field is bit-identical.  For callers whose splats stay resident over several builds.  Default 1: the reference's
 * behaviour. */
int mlsgpu_hip_tree_set_mutate(mlsgpu_tree *tree, int mutate);
int mlsgpu_hip_tree_mutates(const mlsgpu_tree *tree);
void mlsgpu_hip_tree_clear_splats(mlsgpu_tree *tree);
/* Measurement aid: (splat, node) entries of the last build (synchronises the stream). */
int mlsgpu_hip_tree_num_entries(mlsgpu_tree *tree, uint64_t *out);
const mlsgpu_splat *mlsgpu_hip_tree_splats(const mlsgpu_tree *tree);   /* getSplats   */
const int32_t *mlsgpu_hip_tree_commands(const mlsgpu_tree *tree);      /* getCommands */
const int32_t *mlsgpu_hip_tree_start(const mlsgpu_tree *tree);         /* getStart    */
uint64_t mlsgpu_hip_tree_commands_size(const mlsgpu_tree *tree);       /* elements allocated */
uint64_t mlsgpu_hip_tree_start_size(const mlsgpu_tree *tree);
uint32_t mlsgpu_hip_tree_num_levels(const mlsgpu_tree *tree);          /* getNumLevels */

/* ---- MlsFunctor (src/mls.h:79-170) ---- */
int mlsgpu_hip_mls_create(mlsgpu_ctx *ctx, int shape, mlsgpu_mls **out);
void mlsgpu_hip_mls_destroy(mlsgpu_mls *mls);
/* MlsFunctor::set(offset, tree, subsamplingShift), src/mls.cpp:91-94 */
int mlsgpu_hip_mls_set(mlsgpu_mls *mls, const int32_t offset[3], const mlsgpu_tree *tree, uint32_t subsamplingShift);
/* the private MlsFunctor::set(offset, splats, commands, start, shift) used by test/test_mls.cpp:475 */
int mlsgpu_hip_mls_set_buffers(mlsgpu_mls *mls, const int32_t offset[3], const mlsgpu_splat *dSplats,
                               const int32_t *dCommands, const int32_t *dStart, uint32_t subsamplingShift);
int mlsgpu_hip_mls_set_boundary_limit(mlsgpu_mls *mls, float limit);   /* src/mls.cpp:137-144 */
/* With explicit buffers (mlsgpu_hip_mls_set_buffers): 1 if the splats' radius slot still holds the radius, 0 (default) if
 * it holds 1/radius^2 as after a mutating build.  mlsgpu_hip_mls_set takes it from the tree. */
int mlsgpu_hip_mls_set_raw_radius(mlsgpu_mls *mls, int rawRadius);
/* MlsFunctor::enqueue, src/mls.cpp:101-135.  fieldRows = rows allocated in dField (bounds check). */
int mlsgpu_hip_mls_enqueue(mlsgpu_mls *mls, float *dField, uint64_t pitch, uint64_t fieldRows,
                           const mlsgpu_swathe *swathe);
/* Fills *gen so the functor can be passed to mlsgpu_hip_marching_generate (alignment = wgs = {8,8,8}). */
int mlsgpu_hip_mls_generator(mlsgpu_mls *mls, mlsgpu_generator *gen);
/* Selects the kernel: 5 = the default (sub-block culling + a bf16 matrix-core prefilter of the distance test; the
 * accumulation runs under the reference's own test), 4 = round 5's default (sub-block culling + one splat stream per
 * 2x2x2 cube of corners), 1 = the reference's structure (every corner walks every listed splat); 4 and 1 are kept for
 * A/B.  Bit-identical results.  Other values are MLSGPU_ERR_INVALID (0, 2 and 3 were intermediate designs of earlier
 * rounds). */
int mlsgpu_hip_mls_set_variant(mlsgpu_mls *mls, int variant);
/* Measurement aid: with a non-NULL device array of MLSGPU_MLS_STATS_WORDS uint64 the next enqueues run an instrumented
 * kernel that adds [0] listed splats (Sigma L of SURVEY 8d), [1] (corner, splat) distance tests executed, [2] hits (H);
 * and, for the default kernel, how its accumulation ("drain") loops use the 64 lanes of a wave -- a drain call runs as many
 * iterations as the LONGEST of its 64 lanes' hit lists: [3] drain calls, [4] iterations they ran (sum of the longest
 * list), [5] the iterations if every two consecutive calls of a wave were one call, [6] if a whole round of staged splats
 * were one call, [7] if a whole block were; [8 + n], n = 0 .. 32: lanes that had n hits in a drain call.  Lane
 * utilisation of the drain = [2] / (64 x [4]).  Kernel 5: [1] = (corner, splat) pairs the matrix prefilter evaluated,
 * [5] = 0, n = CANDIDATES of a lane in a drain call, [41] candidates the prefilter handed to the drain, [42] hits of the
 * reference's test that the prefilter missed (its superset property: must be 0).  Results are unchanged; NULL switches
 * back to the production kernel. */
#define MLSGPU_MLS_STATS_WORDS 43
int mlsgpu_hip_mls_set_stats(mlsgpu_mls *mls, uint64_t *dCounters);

/* ---- Marching (src/marching.h:494-608) ---- */
#define MLSGPU_MARCHING_MAX_DIMENSION 8192      /* src/marching.h:136 */
#define MLSGPU_MARCHING_MAX_CELL_BYTES 872      /* src/marching.h:96-100 */
int mlsgpu_hip_marching_create(mlsgpu_ctx *ctx, uint32_t maxWidth, uint32_t maxHeight, uint32_t maxDepth,
                               uint32_t maxSwathe, uint64_t meshMemory, const uint32_t alignment[3],
                               mlsgpu_marching **out);
void mlsgpu_hip_marching_destroy(mlsgpu_marching *m);
uint64_t mlsgpu_hip_marching_resource_usage(uint32_t maxWidth, uint32_t maxHeight, uint32_t maxDepth,
                                            uint32_t maxSwathe, uint64_t meshMemory, const uint32_t alignment[3]);
/* Marching::generate, src/marching.cpp:745-824.  Blocks until the bucket is done (as the reference's does). */
int mlsgpu_hip_marching_generate(mlsgpu_marching *m, const mlsgpu_generator *generator,
                                 mlsgpu_output_fn output, void *outputUser,
                                 const uint32_t size[3], const uint32_t keyOffset[3]);
/* Folds a ScaleBiasFilter (src/mesh_filter.cpp:69-113) into vertex emission: with enabled != 0 every vertex
 * handed to the output functor is already fma(v, scale, bias), bit-identical to running mlsgpu_hip_scale_bias
 * on the mesh afterwards, without the extra pass over HBM.  The worker uses this for its scale/bias filter. */
int mlsgpu_hip_marching_set_vertex_transform(mlsgpu_marching *m, int enabled, float scale, float bx, float by, float bz);
/* Counters marching.overflow / marching.shipouts / marching.slices.nonempty (src/marching.cpp:342-377) and
 * work totals: out[0]=overflow out[1]=shipouts out[2]=nonempty swathes out[3]=occupied cells
 * out[4]=unwelded vertices out[5]=indices out[6]=welded vertices out[7]=external vertices. */
int mlsgpu_hip_marching_counters(const mlsgpu_marching *m, uint64_t out[8]);
/* Lookup tables (src/marching.cpp:109-252): count[256][2] u8, start[257][2] u16, data[8192] u8, key[2432][3] u32 */
int mlsgpu_hip_marching_tables(const mlsgpu_marching *m, uint8_t *count, uint16_t *start, uint8_t *data, uint32_t *key);
/* Marching::copySlice (src/marching.cpp:447-498) on an arbitrary field buffer. */
int mlsgpu_hip_marching_copy_slice(mlsgpu_marching *m, float *dField, uint64_t pitch, uint32_t src, uint32_t trg,
                                   uint32_t width, uint32_t height, uint32_t zStride);
/* kernels/marching.cl:295-326 stand-alone (test/test_marching.cpp:401-479). */
int mlsgpu_hip_compact_vertices(mlsgpu_ctx *ctx, float *dOutVertices, uint64_t *dOutKeys, uint32_t *dIndexRemap,
                                uint32_t *dFirstExternal, const uint32_t *dVertexUnique, const float *dInVertices4,
                                const uint64_t *dInKeys, uint64_t minExternalKey, uint64_t keyOffset, uint64_t n);

/* ---- mesh plumbing ---- */
/* MeshSizes::getHostBytes, src/mesh.h:75-80 */
uint64_t mlsgpu_hip_mesh_host_bytes(const mlsgpu_mesh *mesh);
/* enqueueReadMesh into a HostKeyMesh blob [extKeys u64][vertices 3xf32][triangles 3xu32]
 * (src/mesh.cpp:51-102).  hostBlob must be 8-byte aligned.  async != 0: caller synchronises. */
int mlsgpu_hip_mesh_read(mlsgpu_ctx *ctx, const mlsgpu_mesh *mesh, void *hostBlob, int async);
/* Measurement aid (bench.py and the full-size tests check the meshes a timed pass produced without copying gigabytes
 * to the host): out[0..2] = for the vertex words, the triangle indices and the external keys (as 32-bit words) the
 * sum of word[i] * (2 i + 1) modulo 2^64.  Synchronises the stream. */
int mlsgpu_hip_mesh_checksum(mlsgpu_ctx *ctx, const mlsgpu_mesh *mesh, uint64_t out[3]);
/* ScaleBiasFilter, src/mesh_filter.cpp:69-113 + kernels/scale_bias.cl:33-41: in place. */
int mlsgpu_hip_scale_bias(mlsgpu_ctx *ctx, const mlsgpu_mesh *mesh, float scale, float bx, float by, float bz);

/* The MlsFunctor behind a generator made by mlsgpu_hip_mls_generator, or NULL for any other generator. */
mlsgpu_mls *mlsgpu_hip_mls_of_generator(const mlsgpu_generator *gen);
/* kernel variant, work counters and boundary limit of `src` for `dst` (same shape) */
int mlsgpu_hip_mls_copy_settings(mlsgpu_mls *dst, const mlsgpu_mls *src);
/* MlsFunctor::enqueue (mlsgpu_hip_mls_enqueue) for `count` buckets in ONE launch: functor k fills dFields[k] (pitches[k]
 * floats per row, fieldRows[k] rows -- NULL: unchecked) for swathes[k].  The functors share a context, the shape and the
 * kernel variant. */
int mlsgpu_hip_mls_enqueue_batch(mlsgpu_mls *const *mls, float *const *dFields, const uint64_t *pitches,
                                 const uint64_t *fieldRows, const mlsgpu_swathe *swathes, uint32_t count);

/* Output functor of a batch: as mlsgpu_output_fn, with the index of the bucket the mesh belongs to. */
typedef int (*mlsgpu_batch_output_fn)(void *user, uint32_t index, void *stream, const mlsgpu_mesh *mesh);
/* Marching::generate (src/marching.cpp:745-824) for `count` buckets in lock-step: marchings[k] (distinct objects of one
 * context) takes generators[k], sizes[3k..] and keyOffsets[3k..]; one set of launches with a bucket dimension, the swathe
 * totals and the welded counts of all buckets read back together.  Per bucket the meshes are those of
 * mlsgpu_hip_marching_generate, bit for bit, delivered bucket by bucket in order.  Buckets that need several swathes, or
 * whose swathe overflows the mesh memory, take the one-bucket path behind the shared launches.  Blocks. */
int mlsgpu_hip_marching_generate_batch(mlsgpu_marching *const *marchings, const mlsgpu_generator *generators, uint32_t count,
                                       mlsgpu_batch_output_fn output, void *outputUser,
                                       const uint32_t *sizes, const uint32_t *keyOffsets);

/* ---- DeviceWorkerGroupBase::Worker (src/workers.cpp:207-286): tree + MlsFunctor + Marching + ScaleBias ---- */
typedef struct mlsgpu_worker_config
{
    uint64_t maxBucketSplats;
    uint32_t maxCells;          /* bucket side in cells; Marching gets maxCells+1 corners */
    uint64_t meshMemory;        /* 0: (maxCells^2 * 2) worst-case cells as src/mlsgpu_core.cpp:359-370 */
    uint32_t levels;            /* default 6 */
    uint32_t subsampling;       /* default 3 */
    float boundaryLimit;        /* default 1.0 */
    int shape;
    uint32_t maxSwathe;         /* 0: whole bucket depth (no 8192-row image limit on HBM buffers) */
    float gridSpacing;          /* ScaleBiasFilter::setScaleBias(fullGrid): spacing and getVertex(0,0,0) */
    float gridOrigin[3];
} mlsgpu_worker_config;

int mlsgpu_hip_worker_create(mlsgpu_ctx *ctx, const mlsgpu_worker_config *cfg, mlsgpu_worker **out);
void mlsgpu_hip_worker_destroy(mlsgpu_worker *w);
/* DeviceWorkerGroup::resourceUsage (src/workers.cpp:184-205) per worker with ONE lane, without the item pool; a worker that
 * takes `lanes` buckets in lock-step (mlsgpu_hip_worker_set_batch) holds _lanes(cfg, lanes) = lanes times that. */
uint64_t mlsgpu_hip_worker_resource_usage(const mlsgpu_worker_config *cfg);
uint64_t mlsgpu_hip_worker_resource_usage_lanes(const mlsgpu_worker_config *cfg, uint32_t lanes);
/* One SubItem of a WorkItem (src/workers.cpp:235-285): lowExtent = sub.grid.getExtent(i).first,
 * numVertices = sub.grid.numVertices(i).  dSplats is the WorkItem's device splat buffer. */
int mlsgpu_hip_worker_process(mlsgpu_worker *w, mlsgpu_splat *dSplats, uint64_t firstSplat, uint64_t numSplats,
                              const int32_t lowExtent[3], const uint32_t numVertices[3],
                              mlsgpu_output_fn output, void *outputUser);
/* DeviceWorkerGroup::SubItem (src/workers.h:161-168): one bucket of a WorkItem whose splats share a device buffer */
typedef struct mlsgpu_subitem
{
    uint64_t firstSplat, numSplats;
    int32_t lowExtent[3];
    uint32_t numVertices[3];
    mlsgpu_splat *dSplats;      /* NULL: the buffer given to the call (one WorkItem); else this bucket's own buffer, so that
                                 * the buckets of several small WorkItems can share a batch */
} mlsgpu_subitem;
/* Lanes: how many buckets the worker takes through the path in lock-step (1 .. MLSGPU_MAX_BATCH; default 1).  Each lane
 * owns a tree, a distance field, a lattice and a mesh arena (mlsgpu_hip_worker_resource_usage bytes per lane). */
int mlsgpu_hip_worker_set_batch(mlsgpu_worker *w, uint32_t lanes);
uint32_t mlsgpu_hip_worker_batch(const mlsgpu_worker *w);
/* How much shares one set of processCorners / marching launches inside a batch: as many consecutive buckets as hold the
 * corners of `buckets` buckets of the worker's full size, (maxCells + 1)^3 each -- `buckets` full-size ones, or more small
 * ones (0: the whole batch; default 2).  The octree build always spans all lanes.  Results do not depend on it. */
int mlsgpu_hip_worker_set_marching_group(mlsgpu_worker *w, uint32_t buckets);
uint32_t mlsgpu_hip_worker_marching_group(const mlsgpu_worker *w);
/* The loop over the SubItems of a WorkItem (src/workers.cpp:232-286) with the buckets taken `lanes` at a time: per group
 * one set of launches (octree build, processCorners, marching, each with a bucket dimension) and three host decisions.
 * Every bucket's meshes equal mlsgpu_hip_worker_process's bit for bit; `output` receives them bucket by bucket, in order,
 * with the bucket's index in `items`. */
int mlsgpu_hip_worker_process_batch(mlsgpu_worker *w, mlsgpu_splat *dSplats, const mlsgpu_subitem *items, uint32_t numItems,
                                    mlsgpu_batch_output_fn output, void *outputUser);
/* How many leading items of the last _process_batch call had delivered all their meshes when it returned (numItems after a
 * success; after a failure in mid-batch the rest of the items were not processed): src/workers.cpp:281-284 accounts per
 * bucket, and so can a caller of the batch. */
uint32_t mlsgpu_hip_worker_batch_completed(const mlsgpu_worker *w);
mlsgpu_tree *mlsgpu_hip_worker_lane_tree(mlsgpu_worker *w, uint32_t lane);
mlsgpu_marching *mlsgpu_hip_worker_lane_marching(mlsgpu_worker *w, uint32_t lane);
/* keep = 1: the worker does not modify dSplats (non-mutating tree build + raw-radius processCorners, see
 * mlsgpu_hip_tree_set_mutate): resident splats can be processed again without being restored.  Default 0. */
int mlsgpu_hip_worker_set_keep_splats(mlsgpu_worker *w, int keep);
mlsgpu_tree *mlsgpu_hip_worker_tree(mlsgpu_worker *w);
mlsgpu_mls *mlsgpu_hip_worker_mls(mlsgpu_worker *w);
mlsgpu_marching *mlsgpu_hip_worker_marching(mlsgpu_worker *w);

/* ---- bucket farm: CopyGroup + one DeviceWorkerGroup per GPU (src/workers.h:214-438, src/workers.cpp:87-418,
 *      assembled by SlaveWorkers, src/mlsgpu_core.cpp:704-741).  The caller plays BucketLoader: it hands over
 *      host splats of one bucket at a time (already in grid coordinates, see mlsgpu_hip_transform_splats);
 *      the farm batches buckets into device items of at most maxBucketSplats splats (CopyGroupBase::Worker::
 *      operator(), :377-418), stages each batch in pinned memory (two buffers, so filling the next batch
 *      overlaps the copy of the previous one), picks the device whose group can take an item and has the most
 *      unallocated splat capacity (flush(), :315-375), copies host->device on that device's copy stream and
 *      queues the item; `workersPerDevice` threads per GPU (--device-threads) process the buckets of an item
 *      with mlsgpu_hip_worker_process after waiting for the item's copy event. ---- */
typedef struct mlsgpu_farm mlsgpu_farm;
typedef struct mlsgpu_farm_config
{
    uint32_t numDevices;
    const int32_t *devices;        /* device ordinals; NULL = 0 .. numDevices-1 */
    uint32_t workersPerDevice;     /* default 1 */
    uint32_t spare;                /* extra device items per GPU beyond one per worker; default 1 */
    mlsgpu_worker_config worker;   /* maxBucketSplats is also the capacity of a device item */
    uint32_t copyThreads;          /* host threads (per copy side) mlsgpu_hip_farm_submit copies a bucket with; 0 = 4 */
    uint32_t stagingBuffers;       /* pinned staging buffers per copy side; 0 = the side's GPUs + 2 (the reference has one,
                                    * src/workers.cpp:367-372) */
} mlsgpu_farm_config;
/* Output functor with the chunk it belongs to (OutputGenerator, src/workers.h:225).  Called on the worker's
 * thread; `ctx` is that worker's context (use it for mlsgpu_hip_mesh_read).  NULL: meshes are only counted. */
typedef int (*mlsgpu_farm_output_fn)(void *user, int device, uint64_t chunkId, mlsgpu_ctx *ctx, const mlsgpu_mesh *mesh);
int mlsgpu_hip_farm_create(const mlsgpu_farm_config *cfg, mlsgpu_farm_output_fn output, void *user, mlsgpu_farm **out);
void mlsgpu_hip_farm_destroy(mlsgpu_farm *farm);
/* CopyGroup::get + push for one bucket: copies the splats into the pinned staging buffer (flushing the
 * current batch first if they do not fit).  Blocks while every device item is in use. */
int mlsgpu_hip_farm_submit(mlsgpu_farm *farm, const mlsgpu_splat *hSplats, uint64_t numSplats,
                           const int32_t lowExtent[3], const uint32_t numVertices[3], uint64_t chunkId);
/* The two halves of submit for a loader that writes splats straight into the staging buffer, as the reference's
 * BucketLoader does (src/bucket_loader.cpp:56-128 fills the buffer CopyGroup::get returned, then pushes it):
 * acquire returns room for numSplats splats in pinned memory (flushing the current batch first if they do not
 * fit); push appends the bucket (at most the acquired number of splats) to the batch.  One acquire per push. */
int mlsgpu_hip_farm_acquire(mlsgpu_farm *farm, uint64_t numSplats, mlsgpu_splat **out);
int mlsgpu_hip_farm_push(mlsgpu_farm *farm, uint64_t numSplats, const int32_t lowExtent[3], const uint32_t numVertices[3],
                         uint64_t chunkId);
/* Flushes the last batch and waits until every queued bucket has been processed; reports the first error.  After an error
 * the farm refuses further buckets (queued ones are handed back unprocessed, their capacity restored): finish and destroy it. */
int mlsgpu_hip_farm_finish(mlsgpu_farm *farm);
/* out[0] buckets, [1] splats copied, [2] H2D bytes, [3] device items, [4] ship-outs, [5] vertices, [6] triangles,
 * [7] external vertices; per device d: out[8 + d] = buckets processed there (up to 16 devices). */
int mlsgpu_hip_farm_stats(mlsgpu_farm *farm, uint64_t out[24]);
/* ---- placement: where the farm's host threads and pinned memory live relative to the GPUs.  The reference places
 *      nothing (src/workers.cpp:320-351 chooses a device by free capacity only; its threads run where the scheduler puts
 *      them).  Here a GPU's NUMA node is read from sysfs (hipDeviceGetPCIBusId -> /sys/bus/pci/devices/<bdf>/numa_node) and
 *      the farm keeps one COPY SIDE per node that has one of its GPUs: a ring of pinned staging buffers allocated on that
 *      node, copy threads bound to it, feeding that node's GPUs; device worker threads are bound to their GPU's node, the
 *      read-back ring and the mesher thread to the first GPU's.  A batch is staged on the side of the GPU it would go to
 *      (most unallocated capacity) and sent to a GPU of that side -- to another side's only when none of its own can take
 *      an item.  On a one-node machine, or when sysfs does not say, there is one unbound side.  MLSGPU_HIP_SYSFS_ROOT
 *      (default /sys) and MLSGPU_HIP_DEVICE_NODES ("0,0,1,1") describe another machine to a test. ---- */
int mlsgpu_hip_device_node(int device);                                    /* -1: unknown */
int mlsgpu_hip_topology(uint32_t *numNodes, uint32_t cpusPerNode[16]);
/* the plan alone: deviceNodes[i] -> sideOfDevice[i] (sides numbered by first appearance), nodeOfSide[k], *numSides */
int mlsgpu_hip_plan_copy_sides(const int32_t *deviceNodes, uint32_t numDevices, uint32_t numNodes, int32_t *sideOfDevice,
                               int32_t *nodeOfSide, uint32_t *numSides);
/* test hook: copies through the farm's pool of copy threads compared with their sources (`rounds` copies of pseudo-random sizes up
 * to `bytes`; rounds = 0: ONE copy of exactly `bytes` bytes); returns the mismatches */
int mlsgpu_hip_test_copy_pool(uint32_t threads, uint32_t rounds, uint64_t bytes, int node);
int mlsgpu_hip_bind_thread_to_node(int node);                              /* the calling thread; 1 = bound, 0 = left alone */
/* out[0] copy sides, [1] node the read-back ring's memory is on, [2] NUMA nodes, [3] devices; per device d < 16:
 * out[4 + 3d] ordinal, [5 + 3d] node, [6 + 3d] side; per side k < 12: out[52 + 4k] node, [53 + 4k] node its staging memory
 * is on (queried), [54 + 4k] staging buffers, [55 + 4k] copy threads.  -1 = unknown / unused. */
int mlsgpu_hip_farm_placement(mlsgpu_farm *farm, int32_t out[100]);
/* The copy side's clock, seconds since the farm was created: out[0] filling staging (mlsgpu_hip_farm_submit's memcpy),
 * [1] waiting for a staging buffer, [2] waiting for a device item, [3] host-to-device copies (timed event pairs on the copy
 * streams), [4] first submit .. last flush, [5] copies, [6] batches sent to another side's GPU, [7] inside the runtime's
 * enqueue calls (event records + hipMemcpyAsync).  h2d_busy = [3] / [4]. */
int mlsgpu_hip_farm_copy_clock(mlsgpu_farm *farm, double out[8]);
/* The device workers' clock since the farm was created: out[0] sets of launches they ran (a batch of buckets in lock-step,
 * or one bucket), [1] buckets in them ([1] / [0] = buckets per set of launches), [2] seconds the workers waited for an item,
 * [3] seconds they spent processing, both summed over the workers. */
int mlsgpu_hip_farm_worker_clock(mlsgpu_farm *farm, double out[4]);
/* the same four figures for ONE device group (the `group`-th device of the configuration): what the greedy dispatch of
 * src/workers.cpp:320-351 gave that GPU and how long its workers sat without an item */
int mlsgpu_hip_farm_group_clock(mlsgpu_farm *farm, uint32_t group, double out[4]);
/* The most device items (DeviceWorkerGroup::WorkItem, src/workers.h:165-181) that were in flight at once since the farm
 * was created: taken from a group's pool by the copy side and not yet returned by a device worker. */
int mlsgpu_hip_farm_in_flight_max(mlsgpu_farm *farm, uint64_t *out);

/* HostKeyMesh, src/mesh.h:125-179: a ship-out in host memory -- keys of the EXTERNAL vertices only (they are the last
 * numVertices - numInternalVertices vertices), packed float xyz, uint32 index triplets. */
typedef struct mlsgpu_host_mesh
{
    const uint64_t *vertexKeys;
    const float *vertices;
    const uint32_t *triangles;
    uint64_t numVertices;
    uint64_t numTriangles;
    uint64_t numInternalVertices;
} mlsgpu_host_mesh;
/* MesherBase::InputFunctor (src/mesher.h:204-210) as the reference's device workers feed it: called on the farm's ONE
 * mesher thread (MesherGroup, src/workers.cpp:47-85), meshes in the order the ship-outs happened; `mesh` points into
 * the farm's pinned circular buffer and is valid until the functor returns. */
typedef int (*mlsgpu_farm_host_output_fn)(void *user, int device, uint64_t chunkId, const mlsgpu_host_mesh *mesh);
/* Routes every ship-out to the host as the reference does (OutputGeneratorBuilder::Functor, src/workers.h:488-509):
 * room in a pinned circular buffer of ringBytes (MesherGroup::meshBuffer, --mem-mesh; the worker blocks while it is
 * full), enqueueReadMesh (src/mesh.cpp:62-102) asynchronously on the worker's stream -- the worker goes on with the
 * next bucket -- and `fn` on the mesher thread once the reads have completed.  fn may be NULL (meshes are read back and
 * dropped).  Works for any number of GPUs: this is the path that brings buckets of different devices to ONE welder
 * (mlsgpu_hip_host_mesher_*).  Call before the first bucket is submitted; a device-side output functor given to
 * mlsgpu_hip_farm_create still runs first.  A ship-out larger than ringBytes is MLSGPU_ERR_LENGTH.  Calling it again
 * BETWEEN jobs (after mlsgpu_hip_farm_finish) hands the next job's meshes to another consumer; the ring keeps its size. */
int mlsgpu_hip_farm_set_host_output(mlsgpu_farm *farm, uint64_t ringBytes, mlsgpu_farm_host_output_fn fn, void *user);
/* The same route with the CONSUMER's memory instead of the ring: `landing(user, bytes, &ptr)` hands out room for one
 * ship-out (page-locked memory that stays the consumer's: mlsgpu_hip_host_mesher_farm_landing), the asynchronous read-back
 * lands there, and `fn` receives pointers into it which it may keep -- the reference's mesher also reads its
 * CircularBuffer allocation in place (src/workers.h:488-509, src/mesher.cpp:447-469) before it writes the block to its
 * temporary file; here the block never moves again.  No ship-out waits for room.  Same calling rules as
 * mlsgpu_hip_farm_set_host_output (before the first bucket, or between jobs); calling that one afterwards goes back to the
 * ring. */
typedef int (*mlsgpu_farm_landing_fn)(void *user, uint64_t bytes, void **out);
int mlsgpu_hip_farm_set_host_landing(mlsgpu_farm *farm, mlsgpu_farm_landing_fn landing, mlsgpu_farm_host_output_fn fn, void *user);
/* out[0] meshes read back, [1] bytes, [2] times a worker waited for ring space, [3] largest mesh in bytes */
int mlsgpu_hip_farm_host_stats(mlsgpu_farm *farm, uint64_t out[4]);
/* BucketLoader's world -> grid transform (src/bucket_loader.cpp:77-85, Grid::worldToVertex src/grid.cpp:99-106):
 * position = (position - reference) * (1/spacing) - lowExtent, radius *= 1/spacing.  Host-side, in place. */
void mlsgpu_hip_transform_splats(mlsgpu_splat *hSplats, uint64_t numSplats, const float reference[3], float spacing,
                                 const int32_t gridLowExtent[3]);

/* ---- bucketing of a cloud that is resident in HBM: Bucket::bucket, src/bucket.h:116-180 ---- */
typedef struct mlsgpu_grid          /* class Grid, src/grid.h: world = reference + spacing * vertex */
{
    float reference[3];
    float spacing;
    int32_t extents[6];             /* per axis [first, second): second - first cells, one more vertices */
} mlsgpu_grid;
typedef struct mlsgpu_bucket_params /* the arguments of Bucket::bucket, src/bucket.h:170-180 */
{
    uint64_t maxSplats;             /* --mem-bucket-splats / sizeof(Splat) */
    uint32_t maxCells;              /* side of a bucket in cells: (1 << (levels + subsampling - 1)) - 1 */
    uint32_t chunkCells;            /* output chunk alignment; 0 = none */
    uint32_t microCells;            /* requested microblock side; 0 = heuristic (chooseMicroSize) */
    uint64_t maxSplit;              /* maximum fan-out of one recursion level, >= 8 */
} mlsgpu_bucket_params;
typedef struct mlsgpu_bucket        /* what ProcessorType's callback receives, src/bucket.h:98-114 */
{
    int32_t extents[6];             /* the bucket's grid: same reference and spacing, these extents */
    uint64_t chunk[3];              /* Recursion::chunk */
    uint32_t depth;                 /* Recursion::depth */
    uint64_t numSplats;
    const uint32_t *dIds;           /* device: ids of the bucket's splats, ascending; valid during the callback -- or, for a
                                     * callback that reads them asynchronously, until the event it leaves in *consumed */
    const mlsgpu_splat *dSplats;    /* device: the array the ids index -- the caller's cloud (mlsgpu_hip_bucket) or the batch
                                     * of the streamed set that is resident during the callback (mlsgpu_hip_bucket_stream) */
    void **consumed;                /* out, optional: a hipEvent_t the callback recorded behind its last read of dIds (on any
                                     * stream of any device).  The bucketer orders whatever overwrites the list behind that
                                     * event ON THE GPU, so a callback that only ENQUEUES its gather (mlsgpu_hip_farm_submit_
                                     * device_async) need not wait for it: the leaves of a level are handed over back to back.
                                     * Left NULL: the list may be reused as soon as the callback returns.  The event must stay
                                     * alive until the next mlsgpu_hip_bucket call on this context has returned.  With
                                     * mlsgpu_hip_bucket_stream the resident batch (dSplats) is NOT covered: read it before
                                     * returning. */
} mlsgpu_bucket;
typedef int (*mlsgpu_bucket_fn)(void *user, mlsgpu_ctx *ctx, const mlsgpu_bucket *bucket);
/* Splits `region` into buckets of at most maxCells cells per side and maxSplats splats and calls fn for each non-empty
 * one, in the reference's order.  dSplats: all splats, world coordinates, on the device (non-finite ones are ignored
 * as by the reference's splat sets).  MLSGPU_ERR_DENSITY if a single cell holds more than maxSplats splats
 * (*cellSplats = how many).  The octree of counters of one level is dense here: at most 2^24 nodes. */
int mlsgpu_hip_bucket(mlsgpu_ctx *ctx, const mlsgpu_splat *dSplats, uint64_t numSplats, const mlsgpu_grid *region,
                      const mlsgpu_bucket_params *params, mlsgpu_bucket_fn fn, void *user, uint64_t *cellSplats);
/* FastBlobSet::makeBoundingGrid (src/splat_set_impl.h:770-811): the grid the reference reconstructs on -- reference 0,
 * extents floor(min(p - r) / spacing) rounded down to a multiple of bucketSize, and ceil(max(p + r) / spacing) -- from a
 * min / max reduction over the finite splats on the device. */
int mlsgpu_hip_bounding_grid(mlsgpu_ctx *ctx, const mlsgpu_splat *dSplats, uint64_t numSplats, float spacing,
                             uint32_t bucketSize, mlsgpu_grid *out);
/* BucketLoader (src/bucket_loader.cpp:77-85, Grid::worldToVertex src/grid.cpp:99-106) on the device:
 * dOut[i] = splat dIds[i] (or i if dIds is NULL) in the vertex coordinates of fullGrid:
 * (position - reference) / spacing - fullGrid.extents[first], radius / spacing.  dOut is what
 * mlsgpu_hip_worker_process takes, with lowExtent = bucket.extents[first] - fullGrid.extents[first] and
 * numVertices = bucket cells + 1 (the subGrid of src/bucket_loader.cpp:91-102). */
int mlsgpu_hip_bucket_load(mlsgpu_ctx *ctx, const mlsgpu_splat *dSplats, const uint32_t *dIds, uint64_t numSplats,
                           const mlsgpu_grid *fullGrid, mlsgpu_splat *dOut);

/* A bucket whose splats are already on GPU `device` (one of the farm's), e.g. from mlsgpu_hip_bucket's callback: the
 * device item is filled by mlsgpu_hip_bucket_load (gather of dIds + transform into fullGrid's vertex coordinates) --
 * no host copy.  Like a host bucket it goes to the device group with the most unallocated capacity
 * (src/workers.cpp:320-351), `device`'s own on a tie: for another GPU the gather runs on `device` and the item is
 * filled by a peer copy on the target's copy stream out of a ring of scratch buffers on `device` (gather -> copy -> reuse of
 * the slot are ordered by events on the GPUs), so a cloud resident on one GPU feeds all of them with several leaves in
 * flight.  The call returns when the gather has run: dIds may be reused, the peer copy and the bucket go on without the
 * caller.  lowExtent / numVertices as for mlsgpu_hip_worker_process.  */
int mlsgpu_hip_farm_submit_device(mlsgpu_farm *farm, int device, const mlsgpu_splat *dSplats, const uint32_t *dIds,
                                  uint64_t numSplats, const mlsgpu_grid *fullGrid, const int32_t lowExtent[3],
                                  const uint32_t numVertices[3], uint64_t chunkId);
/* The same without the wait: the call returns when the gather has been ENQUEUED; *consumed is a hipEvent_t (owned by the
 * farm, alive as long as it is) that fires when the gather has read dSplats and dIds -- what a bucketer's callback leaves in
 * mlsgpu_bucket::consumed.  A farm worker that finds several such buckets queued takes them through one set of launches
 * (mlsgpu_hip_farm_set_batch), which it rarely does when every leaf costs the feeder a round trip to a busy GPU. */
int mlsgpu_hip_farm_submit_device_async(mlsgpu_farm *farm, int device, const mlsgpu_splat *dSplats, const uint32_t *dIds,
                                        uint64_t numSplats, const mlsgpu_grid *fullGrid, const int32_t lowExtent[3],
                                        const uint32_t numVertices[3], uint64_t chunkId, void **consumed);

/* The buckets of a device item (the SubItems of a WorkItem) are taken through the path `lanes` at a time by
 * mlsgpu_hip_worker_process_batch instead of one by one (src/workers.cpp:232-286); 1 .. MLSGPU_MAX_BATCH, default 1.
 * Outputs are unchanged and still arrive bucket by bucket. */
int mlsgpu_hip_farm_set_batch(mlsgpu_farm *farm, uint32_t lanes);

/* ---- mesh sink for meshes that stay in HBM: OOCMesher's weld / components / prune / per-chunk output,
 *      src/mesher.h:203-330, src/mesher.cpp:220-852 ---- */
typedef struct mlsgpu_mesher mlsgpu_mesher;
int mlsgpu_hip_mesher_create(mlsgpu_ctx *ctx, mlsgpu_mesher **out);
void mlsgpu_hip_mesher_destroy(mlsgpu_mesher *mesher);
/* MesherBase::setPruneThreshold: components with fewer vertices than uint64(total * threshold) are dropped */
int mlsgpu_hip_mesher_set_prune_threshold(mlsgpu_mesher *mesher, double threshold);
/* on = 1: finalize runs beside other GPU work (the next job's buckets) and holds its union-find pass back to a quarter of
 * the wave slots -- no slower for surface-like meshes, and the other kernels keep their speed.  Default 0. */
int mlsgpu_hip_mesher_set_background(mlsgpu_mesher *mesher, int on);
/* Optional: room for this many vertices / triangles / external vertices in total (the arenas grow by reallocation
 * otherwise). */
int mlsgpu_hip_mesher_reserve(mlsgpu_mesher *mesher, uint64_t numVertices, uint64_t numTriangles, uint64_t numExternal);
/* MesherBase::InputFunctor for a DeviceKeyMesh: appends a copy of the mesh (device to device, on `from`'s stream).  On the
 * mesher's own device the call does not wait for the copy: it is ahead of whatever `from`'s stream does to the mesh next
 * (Marching reuses it), and the mesher waits for its pending appends before anything reads the arenas (finalize, boundary,
 * reset, a growing arena); a peer append synchronises.  Thread safe; `from` is the calling worker's context, on
 * the mesher's device or on ANOTHER GPU (then the append is a peer copy over the fabric: several GPUs' buckets welded
 * in one GPU's HBM).  Blocks may arrive in any order, chunk ids interleaved (OOCMesher::add indexes chunks[chunkId.gen]
 * and accepts any arrival order too); output chunks are in order of first arrival. */
int mlsgpu_hip_mesher_add(mlsgpu_mesher *mesher, mlsgpu_ctx *from, uint64_t chunkId, const mlsgpu_mesh *mesh);
/* a mlsgpu_farm_output_fn whose `user` is the mlsgpu_mesher: give it to mlsgpu_hip_farm_create and the farm's device
 * workers (of any of its GPUs) append their ship-outs to the device sink */
int mlsgpu_hip_mesher_farm_output(void *mesher, int device, uint64_t chunkId, mlsgpu_ctx *ctx, const mlsgpu_mesh *mesh);
/* Empties the mesher for the next job; arenas, scratch and outputs keep their capacity. */
int mlsgpu_hip_mesher_reset(mlsgpu_mesher *mesher);
/* Several meshers, one job (one process per GPU, every rank's meshes in its own HBM): the device sink's counterpart of
 * mlsgpu_hip_host_mesher_boundary / _boundary_read / _finalize_with below.  boundary() welds and labels what has been
 * added and exports every distinct external key with the (densely numbered) component that holds its vertex, and the
 * vertex / triangle count of every component; finalize_with() produces the output with the caller's verdict per component
 * in place of the prune rule.  No add between the two calls.  mlsgpu_amd/dist_sink.py merges the exports of all ranks in
 * one all-gather. */
int mlsgpu_hip_mesher_boundary(mlsgpu_mesher *mesher, uint64_t *numKeys, uint64_t *numRoots);
int mlsgpu_hip_mesher_boundary_read(mlsgpu_mesher *mesher, uint64_t *keys, uint32_t *keyRoot, uint64_t *rootVertices,
                                    uint64_t *rootTriangles);
int mlsgpu_hip_mesher_finalize_with(mlsgpu_mesher *mesher, const uint8_t *keepRoot, uint64_t numRoots, uint32_t *numChunks);
/* What MesherBase::write does before it writes files: weld by key, components, prune.  *numChunks = chunks that have
 * triangles (no output is produced for the others, src/mesher.cpp:820). */
int mlsgpu_hip_mesher_finalize(mlsgpu_mesher *mesher, uint32_t *numChunks);
/* Output chunk i (arrival order): packed float3 vertices and uint3 triangles on the device, indices relative to the
 * chunk's first vertex.  A vertex shared by two chunks is in both (externalRemap is per chunk, src/mesher.cpp:538-567).
 * Valid until the next finalize / destroy. */
int mlsgpu_hip_mesher_chunk(mlsgpu_mesher *mesher, uint32_t i, uint64_t *chunkId, uint64_t *numVertices, uint64_t *numTriangles,
                            const float **dVertices, const uint32_t **dTriangles);
/* getStatistics (src/mesher.cpp:491-536): out[0] welded vertices, [1] threshold, [2] components, [3] kept components,
 * [4] kept vertices (each welded vertex once), [5] kept triangles, [6] vertices added, [7] triangles added */
int mlsgpu_hip_mesher_stats(mlsgpu_mesher *mesher, uint64_t out[8]);
/* ---- host mesh sink: OOCMesher's weld as the reference runs it, on the host (src/mesher.cpp:220-469, north_star:
 *      "welding stays on host").  add() is MesherBase::InputFunctor: local components of the block by union-find over
 *      two edges per triangle (computeLocalComponents, :220-236), clumps merged across blocks through the external
 *      keys (updateClumpKeyMap, :286-311); finalize() applies the prune rule of getStatistics (:491-536) and lays out
 *      one mesh per chunk in which a key appears once (externalRemap, :538-567).  Meshes of ANY device can be added --
 *      it is the cross-GPU welder behind mlsgpu_hip_farm_set_host_output.  In memory: no temporary files, no reorder
 *      buffer; chunk ids may arrive interleaved.  add() is serialised by an internal mutex. ---- */
typedef struct mlsgpu_host_mesher mlsgpu_host_mesher;
int mlsgpu_hip_host_mesher_create(mlsgpu_host_mesher **out);
/* The welders' memory is mapped in slabs that the PROCESS keeps when a welder is destroyed (up to 16 GiB, or
 * MLSGPU_HIP_WELDER_CACHE_MB), because freshly mapped memory faults in slowly; the memory a welder gets is uninitialised.
 * _trim_cache sets that limit and returns what is held beyond it to the system at once (0: keep nothing); it returns the
 * bytes released.  The reference's mesher allocates per job (src/mesher.cpp:220-306) and has no counterpart. */
uint64_t mlsgpu_hip_host_mesher_trim_cache(uint64_t keepBytes);
void mlsgpu_hip_host_mesher_destroy(mlsgpu_host_mesher *mesher);
int mlsgpu_hip_host_mesher_set_prune_threshold(mlsgpu_host_mesher *mesher, double threshold);
/* Threads of the welder: add() copies the block and queues its work (local components, key map: OOCMesher::add,
 * src/mesher.cpp:370-469) as a task, and finalize builds the output block by block on the same pool (the reference runs
 * one mesher thread and parallelises its heavy rewrite with OpenMP, src/mesher.cpp:597-600).  0 = default:
 * MLSGPU_HIP_HOST_MESHER_THREADS, else min(32, hardware threads).  Before the first add; results do not depend on it. */
int mlsgpu_hip_host_mesher_set_threads(mlsgpu_host_mesher *mesher, uint32_t threads);
uint32_t mlsgpu_hip_host_mesher_threads(mlsgpu_host_mesher *mesher);
/* The welder's threads bound to the CPUs of ONE NUMA node -- the node its meshes arrive on (mlsgpu_hip_farm_placement
 * out[1]: the read-back ring is next to the farm's first GPU), so that a block's pieces are welded within one socket's
 * caches.  -1 (default): unbound.  Before the first add.  The reference's mesher thread is not placed. */
int mlsgpu_hip_host_mesher_set_node(mlsgpu_host_mesher *mesher, int node);
int mlsgpu_hip_host_mesher_node(mlsgpu_host_mesher *mesher);
/* Bounded-memory mode -- OOCMesher's temporary files (src/mesher.cpp:404-419 writes every block's vertices and triangles to
 * them, :763-852 reads them back): the welder's memory becomes mappings of nameless temporary files in `dir`, and once more
 * than `residentBytes` of blocks have arrived a block is written out and dropped from memory when its task is done; the
 * output passes fault it back in.  Results are the in-memory mode's, element for element.  Before the first add; an unusable
 * directory is MLSGPU_ERR_INVALID; NULL or "" = in memory (the default).  tmp_usage: out[0] = bytes of temporary files mapped,
 * out[1] = of which in memory right now, out[2] = bytes of blocks handed to the kernel to write out and drop. */
int mlsgpu_hip_host_mesher_set_tmp_dir(mlsgpu_host_mesher *mesher, const char *dir, uint64_t residentBytes);
int mlsgpu_hip_host_mesher_tmp_usage(mlsgpu_host_mesher *mesher, uint64_t out[3]);
int mlsgpu_hip_host_mesher_add(mlsgpu_host_mesher *mesher, uint64_t chunkId, const mlsgpu_host_mesh *mesh);
/* a mlsgpu_farm_host_output_fn whose `user` is the mlsgpu_host_mesher */
int mlsgpu_hip_host_mesher_farm_output(void *mesher, int device, uint64_t chunkId, const mlsgpu_host_mesh *mesh);
/* Ship-outs that arrive IN PLACE.  landing: `bytes` of the welder's own memory (page-locked where the process has a GPU:
 * landing_pinned says so), valid until the welder is destroyed; add_landed: add() for a mesh whose arrays lie in such
 * memory -- adopted, not copied (the index check that rides on add()'s copy runs first thing in the block's task: a bad mesh
 * makes finalize fail with MLSGPU_ERR_INVALID).  farm_landing / farm_output_landed: the pair for
 * mlsgpu_hip_farm_set_host_landing, `user` = the mesher. */
int mlsgpu_hip_host_mesher_landing(mlsgpu_host_mesher *mesher, uint64_t bytes, void **out);
int mlsgpu_hip_host_mesher_landing_pinned(mlsgpu_host_mesher *mesher);
int mlsgpu_hip_host_mesher_add_landed(mlsgpu_host_mesher *mesher, uint64_t chunkId, const mlsgpu_host_mesh *mesh);
int mlsgpu_hip_host_mesher_farm_landing(void *mesher, uint64_t bytes, void **out);
int mlsgpu_hip_host_mesher_farm_output_landed(void *mesher, int device, uint64_t chunkId, const mlsgpu_host_mesh *mesh);
int mlsgpu_hip_host_mesher_finalize(mlsgpu_host_mesher *mesher, uint32_t *numChunks);
/* Output chunk i (chunks in order of first arrival, those without triangles skipped): host pointers, valid until the
 * next add / finalize / destroy; indices relative to the chunk's first vertex. */
int mlsgpu_hip_host_mesher_chunk(mlsgpu_host_mesher *mesher, uint32_t i, uint64_t *chunkId, uint64_t *numVertices,
                                 uint64_t *numTriangles, const float **vertices, const uint32_t **triangles);
/* as mlsgpu_hip_mesher_stats */
int mlsgpu_hip_host_mesher_stats(mlsgpu_host_mesher *mesher, uint64_t out[8]);
/* Several meshers, one job (one process per GPU, each welding its own buckets; the reference's MPI build gathers every
 * ship-out on one rank instead, src/mlsgpu_mpi.cpp).  Components that cross rank boundaries and the prune threshold need
 * the other ranks' clumps: boundary() sizes and boundary_read() exports every external key this mesher has seen (sorted)
 * with the ROOT clump holding its vertex, and the vertex / triangle counts of every root clump (0 for the others).  The
 * caller unites clumps that share a key across meshers (a vertex seen by r meshers was counted r times), applies the
 * prune rule of getStatistics (src/mesher.cpp:491-536) to the merged counts and hands the verdict per root clump back
 * to finalize_with().  mlsgpu_amd/dist_sink.py does this over torch.distributed. */
int mlsgpu_hip_host_mesher_boundary(mlsgpu_host_mesher *mesher, uint64_t *numKeys, uint64_t *numClumps);
int mlsgpu_hip_host_mesher_boundary_read(mlsgpu_host_mesher *mesher, uint64_t *keys, uint32_t *keyClump,
                                         uint64_t *clumpVertices, uint64_t *clumpTriangles);
int mlsgpu_hip_host_mesher_finalize_with(mlsgpu_host_mesher *mesher, const uint8_t *keepClump, uint64_t numClumps,
                                         uint32_t *numChunks);

/* Output chunk i of a finalized mesher (mlsgpu_hip_mesher_chunk) written as FastPly::Writer's file STRAIGHT FROM HBM through
 * two pinned buffers of bufferBytes / 2 (0 = 64 MiB): while one piece is written the next travels, faces are packed into the
 * file's 13-byte records on the device, and the host never holds more of the mesh than the buffer -- the role of the
 * reference's asynchronous writer (src/async_io.h:95-140) for outputs larger than host memory.  Same bytes as
 * mlsgpu_hip_write_ply of the downloaded arrays. */
int mlsgpu_hip_mesher_write_ply(mlsgpu_mesher *mesher, uint32_t i, const char *path, const char *const *comments,
                                uint32_t numComments, uint64_t bufferBytes);

/* FastPly::Writer's file from host memory (src/fast_ply.cpp:443-521): binary little endian, header padded to 4 bytes,
 * float32 x y z, faces as uint8 3 + 3 x uint32 */
int mlsgpu_hip_write_ply(const char *path, const float *vertices, uint64_t numVertices, const uint32_t *triangles,
                         uint64_t numTriangles, const char *const *comments, uint32_t numComments);

/* ---- splat input: FastPly::Reader, src/fast_ply.h:77-262; host only ---- */
typedef struct mlsgpu_ply_reader mlsgpu_ply_reader;
/* Opens a binary PLY file of splats (float32 x y z nx ny nz radius among the vertex properties) and checks its header;
 * MLSGPU_ERR_FORMAT with the reference's message otherwise.  radius' = min(radius, maxRadius) * smooth,
 * quality = 1 / radius'^2 (Reader::decode, src/fast_ply.cpp:334-350). */
int mlsgpu_hip_ply_open(const char *path, float smooth, float maxRadius, mlsgpu_ply_reader **out);
void mlsgpu_hip_ply_close(mlsgpu_ply_reader *reader);
uint64_t mlsgpu_hip_ply_size(const mlsgpu_ply_reader *reader);
/* out[0] vertex size in bytes, [1] vertex count, [2] header size, [3..9] byte offsets of x y z nx ny nz radius */
int mlsgpu_hip_ply_layout(const mlsgpu_ply_reader *reader, uint64_t out[10]);
/* Reader::Handle::read: splats [first, first + count) into host memory (e.g. what mlsgpu_hip_farm_acquire returned) */
int mlsgpu_hip_ply_read(mlsgpu_ply_reader *reader, uint64_t first, uint64_t count, mlsgpu_splat *out);
/* File -> device memory, decode (hostThreads host threads, 0 = 4) overlapped with the host-to-device copies through two
 * pinned 64 MiB buffers on ctx's stream; returns when dOut[0 .. count) is complete.  What the reference's reader
 * threads + async I/O do for its out-of-core splat sets (src/splat_set.h:389-700), for inputs that fit in HBM. */
int mlsgpu_hip_ply_load(mlsgpu_ply_reader *reader, mlsgpu_ctx *ctx, uint64_t first, uint64_t count, mlsgpu_splat *dOut,
                        uint32_t hostThreads);

/* ---- SplatSet::FileSet, src/splat_set.h:383-700: several PLY files read as ONE splat sequence
 *      (file after file; the reference packs the file number into the upper bits of its splat ids, here ids are
 *      positions in the concatenation). ---- */
typedef struct mlsgpu_fileset mlsgpu_fileset;
int mlsgpu_hip_fileset_create(float smooth, float maxRadius, mlsgpu_fileset **out);
void mlsgpu_hip_fileset_destroy(mlsgpu_fileset *files);
/* FileSet::addFile: the header is parsed and checked now (MLSGPU_ERR_FORMAT as mlsgpu_hip_ply_open) */
int mlsgpu_hip_fileset_add_file(mlsgpu_fileset *files, const char *path);
uint64_t mlsgpu_hip_fileset_num_files(const mlsgpu_fileset *files);
uint64_t mlsgpu_hip_fileset_num_splats(const mlsgpu_fileset *files);     /* FileSet::maxSplats */
/* FileSet::setBufferSize (default 32 MiB): the host memory a load may hold, whatever the size of the files */
int mlsgpu_hip_fileset_set_buffer_size(mlsgpu_fileset *files, uint64_t bytes);
/* splats [first, first + count) of the sequence into host memory (e.g. what mlsgpu_hip_farm_acquire returned) */
int mlsgpu_hip_fileset_read(mlsgpu_fileset *files, uint64_t first, uint64_t count, mlsgpu_splat *out);
/* The same range into DEVICE memory with bounded host memory: `readerThreads` host threads (0 = 4; the reference's
 * ReaderThread, src/splat_set.h:560-700) fill the slots of one pinned buffer of the set's buffer size with consecutive
 * chunks while earlier chunks travel to the GPU on ctx's stream (the role of src/async_io.h:95-140 + CopyGroup for
 * clouds that fit in HBM: 10^9 splats are 32 GB of 288).  When every file's rows are whole 32-bit words no wider than a
 * splat (x y z nx ny nz radius as float32: 28 bytes) the threads only READ the rows, the rows cross the link as the file
 * holds them and a kernel decodes them (Reader::decode, src/fast_ply.cpp:374-400: same splats bit for bit); any other
 * layout, or MLSGPU_HIP_FILESET_RAW=0, is decoded by the threads.  8-16 threads on the GPU's NUMA node reach the link's
 * rate (61 GB/s of splats); the staging is kept by the set from call to call.  One load per set at a time.  Returns when
 * dOut[0 .. count) is complete. */
int mlsgpu_hip_fileset_load(mlsgpu_fileset *files, mlsgpu_ctx *ctx, uint64_t first, uint64_t count, mlsgpu_splat *dOut,
                            uint32_t readerThreads);
/* Bucket::bucket over a FileSet that need NOT fit the device -- the role of the reference's blob index and host bucketing
 * (FastBlobSet, src/splat_set.h:713-905; bucketRecurse, src/bucket_impl.h:439-560) for data beyond HBM.  The files are
 * streamed through a chunk buffer of `chunkSplats` splats: once to count the level's microblock octree (the reference's first
 * pass over the blobs), then once per BATCH of top-level regions that fit `budgetSplats` splats together (its second); a
 * batch is resident while its regions are split further and its buckets handed to `fn`, whose mlsgpu_bucket::dSplats is the
 * batch and dIds positions in it (valid during the callback: mlsgpu_hip_farm_submit_device copies the bucket out).  The
 * buckets -- extents, order, member splats in file order -- are those mlsgpu_hip_bucket makes of the same set resident.
 * stats (may be NULL): [0] passes over the files, [1] batches, [2] splats loaded into batches, [3] file chunks a pass did
 * not have to read (the first pass notes every chunk's bounding box; a batch skips the chunks that stay clear of its regions).
 * MLSGPU_ERR_LENGTH if one top-level region alone exceeds the budget. */
/* FastBlobSet::makeBoundingGrid (src/splat_set_impl.h:770-811) for such a set: one pass over the files through a chunk buffer */
int mlsgpu_hip_fileset_bounding_grid(mlsgpu_fileset *files, mlsgpu_ctx *ctx, float spacing, uint32_t bucketSize,
                                     uint64_t chunkSplats, uint32_t readerThreads, mlsgpu_grid *out);
int mlsgpu_hip_bucket_stream(mlsgpu_ctx *ctx, mlsgpu_fileset *files, const mlsgpu_grid *region,
                             const mlsgpu_bucket_params *params, uint64_t budgetSplats, uint64_t chunkSplats,
                             uint32_t readerThreads, mlsgpu_bucket_fn fn, void *user, uint64_t *cellSplats, uint64_t stats[4]);

/* DeviceWorkerGroupBase::computeMaxSwathe, src/workers.cpp:169-182 */
uint32_t mlsgpu_hip_compute_max_swathe(uint32_t yMax, uint32_t y, uint32_t yAlign, uint32_t zAlign);

/* ---- small device helpers that mirror the reference's one-work-item test kernels
 *      (kernels/octree.cl:355-372, mls.cl:439-481, marching.cl:364-371); each runs on the GPU ---- */
int mlsgpu_hip_test_make_code(mlsgpu_ctx *ctx, int x, int y, int z, uint32_t *out);
int mlsgpu_hip_test_level_shift(mlsgpu_ctx *ctx, const int32_t lo[3], const int32_t hi[3], int32_t *out);
int mlsgpu_hip_test_point_box_dist2(mlsgpu_ctx *ctx, const float p[3], const float lo[3], const float hi[3], float *out);
int mlsgpu_hip_test_solve_quadratic(mlsgpu_ctx *ctx, float a, float b, float c, float *out);
int mlsgpu_hip_test_fit_sphere(mlsgpu_ctx *ctx, const mlsgpu_splat *hSplats, uint32_t n, float out[5]);
int mlsgpu_hip_test_compute_key(mlsgpu_ctx *ctx, const uint32_t coords[3], const uint32_t top[3], uint64_t *out);
/* primitives: exclusive scan with seed (clogs::Scan) and stable radix sort on the low `bits` (clogs::Radixsort) */
int mlsgpu_hip_test_scan_u32(mlsgpu_ctx *ctx, uint32_t *dData, uint64_t n, uint32_t seed);
/* `repeats` scans of `count` (<= MLSGPU_MAX_BATCH) lanes each, dIn[k] -> dOut[k] (n[k] elements from seeds[k]): one set of
 * launches per repeat, as the buckets of a batch scan (src/marching.cpp:353,583,721 per bucket), back to back */
int mlsgpu_hip_test_scan_u32_batch(mlsgpu_ctx *ctx, const uint32_t *const *dIn, uint32_t *const *dOut, const uint64_t *n,
                                   const uint32_t *seeds, uint32_t count, uint32_t repeats);
int mlsgpu_hip_test_sort_u32(mlsgpu_ctx *ctx, uint32_t *dKeys, uint32_t *dValues, uint64_t n, uint32_t bits);
int mlsgpu_hip_test_sort_u64(mlsgpu_ctx *ctx, uint64_t *dKeys, uint32_t *dValues, uint64_t n, uint32_t bits);

#ifdef __cplusplus
}
#endif
#endif /* MLSGPU_HIP_H */

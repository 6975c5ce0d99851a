/* icp_hip.h -- C ABI of the MI355X (gfx950) registration backend.
 *
 * This is the drop-in boundary for the reference's GPU registration path:
 *
 *   std::tuple<utils::Matrix<float>, libcpu::point_list>
 *   libgpu::icp(const point_list& m, const point_list& p, size_t iterations,
 *               float threshold, uint vp_threshold);          libgpu/libgpu/icp.hh:19-21
 *
 * The reference has no C ABI; each entry point below names the reference code it replaces
 * (paths relative to the reference repository).  A ~40-line C++ shim with the signature above
 * sits on top of it (integration/libgpu_icp.cc).
 *
 * Conventions
 *   - every function returns 0 (ICP_HIP_OK) or a non-zero icp_hip_status; no exception crosses
 *     the ABI; icp_hip_last_error(ctx) returns a NUL-terminated message for the last failure on
 *     that context (icp_hip_last_global_error() for calls that have no context);
 *   - plain pointers and sizes only; caller-owned buffers stay caller-owned; a context owns its
 *     device memory, its HIP stream and (optionally) its RCCL communicator;
 *   - one context drives ONE GPU and must not be used from two threads at once; multi-GPU runs
 *     are one process (or thread) per GPU, each with its own context holding a shard of the
 *     query cloud (any subset of its rows: see icp_hip_set_queries) and a full replica of the tree;
 *   - points are the reference's Point3D layout: 3 packed floats x,y,z (libcpu/point-3d.hh:12-17);
 *   - the 4x4 transform is row-major, values[i*4+j] (libcpu/utils/matrix.hh:149).
 */
#ifndef ICP_HIP_H
#define ICP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum icp_hip_status {
    ICP_HIP_OK = 0,
    ICP_HIP_ERR_ARG = 1,        /* bad argument / call order                                  */
    ICP_HIP_ERR_DEGENERATE = 2, /* the cloud hits an undefined-behaviour case of the reference
                                   tree build (empty side: vp-tree.cc:114-115; capacity <= 2) */
    ICP_HIP_ERR_HIP = 3,        /* a HIP runtime call or kernel launch failed                 */
    ICP_HIP_ERR_COMM = 4,       /* RCCL / all-reduce callback failed                          */
    ICP_HIP_ERR_DEPTH = 5,      /* tree deeper than the traversal stack supports              */
    ICP_HIP_ERR_NOMEM = 6
} icp_hip_status;

/* ------------------------------------------------------------------------------------------
 * Flat structure-of-arrays VP-tree in HOST memory.
 * Replaces: the pointer tree libcpu::VPTree (libcpu/libcpu/vp-tree.hh:15-49) and its device
 * image GPUVPTreeNode blob (libgpu/libgpu/vp-tree.hh:69-81, vp-tree.cu:39-112).
 *
 * A "child code" c is an internal-node index when c >= 0 and the leaf ~c when c < 0.
 * A "row" is the index of a point in the cloud the tree was built from.
 * Every point of the cloud owns exactly one place: an internal node's centre or one bucket slot.
 * Buckets keep the reference's point order (ties in the leaf scan go to the lowest slot,
 * vp-tree.cc:131-141); each bucket starts at a multiple of 16 slots, padding slots hold +inf
 * coordinates and row 0xFFFFFFFF.
 * ---------------------------------------------------------------------------------------- */
typedef struct icp_flat_tree {
    uint64_t n_points; /* == n_nodes + sum(leaf_count)                                        */
    uint64_t n_slots;  /* bucket slots including padding                                      */
    uint32_t n_nodes;  /* internal nodes, numbered breadth-first (root == 0 when n_nodes > 0)  */
    uint32_t n_leaves; /* leaves, numbered left to right (inside before outside)              */
    uint32_t height;   /* edges on the longest root-to-leaf path                              */
    uint32_t capacity; /* a node is a leaf iff it holds < capacity points (vp-tree.cc:56)     */
    uint32_t max_leaf_count;
    int32_t root;      /* child code of the root                                              */
    /* internal nodes: n_nodes entries each */
    const float* node_cx;
    const float* node_cy;
    const float* node_cz;
    const float* node_radius;    /* sqrt(median squared distance), vp-tree.cc:77-78            */
    const int32_t* node_inside;  /* child code, points with d2 <  r2                            */
    const int32_t* node_outside; /* child code, points with d2 >= r2                            */
    const int32_t* node_parent;  /* internal index, -1 for the root                             */
    const uint32_t* node_row;    /* row of the centre                                           */
    /* leaves: n_leaves entries each */
    const uint32_t* leaf_begin; /* first slot                                                   */
    const uint32_t* leaf_count; /* points in the bucket (>= 1)                                  */
    /* buckets: n_slots entries each */
    const float* slot_x;
    const float* slot_y;
    const float* slot_z;
    const uint32_t* slot_row;
} icp_flat_tree;

/* Host-side tree build: replaces libcpu::VPTree::VPTree (libcpu/libcpu/vp-tree.cc:54-123, with
 * partition/select/median :7-52) as called from libgpu::icp (libgpu/libgpu/icp.cu:141-142).
 * `xyz_centered` is the reference cloud already centred on its mean (icp.cu:141 / icp.cc:108).
 * Produces exactly the reference tree (same centres, radii, bucket contents and order).
 * Returns ICP_HIP_ERR_DEGENERATE where the reference's build is undefined behaviour. */
int icp_hip_tree_build(const float* xyz_centered, uint64_t n, uint32_t capacity,
                       icp_flat_tree** out);
void icp_hip_tree_free(icp_flat_tree* tree);
/* Structural check of a caller-made flat tree (ranges, parent/child consistency, every row owned
 * once).  icp_hip_set_tree runs it before uploading. */
int icp_hip_tree_validate(const icp_flat_tree* tree);

/* libcpu::mean (libcpu/libcpu/point-3d.cc:177-192): sequential fp32, each term divided by n.
 * Host helper used for mu_m so that the centred cloud, hence the tree, is the reference's. */
void icp_hip_mean_host(const float* xyz, uint64_t n, float mu_out[3]);
/* libcpu::subtract(list, point) (point-3d.cc:194-210). */
void icp_hip_center_host(const float* xyz, uint64_t n, const float mu[3], float* out_xyz);

/* Loaders.  icp_hip_read_csv replaces libcpu::read_csv (libcpu/libcpu/point-3d.cc:8-134) as main
 * calls it (main/main/main.cc:24-27 with "Points_0","Points_1","Points_2"): header lookup by field
 * name, other columns ignored, std::stof per token; the reference's runtime_error texts come back
 * through icp_hip_last_global_error.  icp_hip_read_f32 reads raw float32 x,y,z triples (no
 * reference counterpart: text parsing would dwarf the registration at 64M points).
 * *xyz_out is malloc'ed (n x 3 floats, Point3D layout); free it with icp_hip_free_points. */
int icp_hip_read_csv(const char* path, const char* x_field, const char* y_field,
                     const char* z_field, float** xyz_out, uint64_t* n_out);
int icp_hip_read_f32(const char* path, float** xyz_out, uint64_t* n_out);
void icp_hip_free_points(float* xyz);

/* Rigid-transform solve on the host: 3x3 cross-covariance H (row-major, row = coordinate of the
 * centred moving cloud, column = coordinate of its matched reference points) -> row-major 4x4
 * [r t; 0 0 0 1] with r = V U^T (no reflection fix) and t = mu_m - r mu_p.
 * Replaces libgpu::find_rotation (libgpu/libgpu/svd.cc:7-28, Eigen JacobiSVD) and the
 * transpose/dot/subtract/to_transformation chain of libgpu/libgpu/icp.cu:35-62; semantics of
 * libcpu find_alignment + to_transformation (libcpu/libcpu/icp.cc:11-60).  icp_hip_run calls it
 * once per iteration when it drives the loop from the host; the device-resident loop runs the
 * same source (icp_amd/csrc/align3.hh) in a one-thread kernel and gives the same bits. */
void icp_hip_alignment_host(const float H[9], const float mu_p[3], const float mu_m[3],
                            float T_out[16]);

/* ------------------------------------------------------------------------------------------
 * Device context
 * ---------------------------------------------------------------------------------------- */
typedef struct icp_hip_ctx icp_hip_ctx;

/* Replaces the implicit device state of libgpu (GPUMatrix / GPUVPTree RAII,
 * libgpu/libgpu/matrix.cu:6-17, vp-tree.cu:7-16).  device_id < 0 means the current device. */
int icp_hip_create(icp_hip_ctx** out, int device_id);
void icp_hip_destroy(icp_hip_ctx* ctx);
const char* icp_hip_last_error(const icp_hip_ctx* ctx);
const char* icp_hip_last_global_error(void);

/* Upload (one bulk H2D per array) and pack the tree for the kernels.
 * Replaces GPUVPTree::from_cpu (libgpu/libgpu/vp-tree.cu:58-112).  The tree is borrowed only
 * for the duration of the call. */
int icp_hip_set_tree(icp_hip_ctx* ctx, const icp_flat_tree* tree);

/* Device-side tree build: replaces libcpu::VPTree::VPTree (libcpu/libcpu/vp-tree.cc:54-123, with
 * partition/select/median :7-52) as called from libgpu::icp (libgpu/libgpu/icp.cu:141-142) AND
 * GPUVPTree::from_cpu (vp-tree.cu:58-112) in one step: `xyz_centered` (host, Point3D layout, the
 * reference cloud centred on its mean) is uploaded once and the tree is built level by level in
 * HBM, straight into the layout the kernels read.  The tree is the one icp_hip_tree_build makes
 * (hence the reference's), bit for bit; icp_hip_get_tree returns it for inspection.  Same error
 * conventions as icp_hip_tree_build (ICP_HIP_ERR_DEGENERATE where the reference is undefined).
 * Clouds below 8192 points are built by icp_hip_tree_build and uploaded instead (the per-level
 * launches cost more than the serial build there); ICP_HIP_TREE_BUILD=device|host in the
 * environment at icp_hip_create forces one or the other. */
int icp_hip_build_tree(icp_hip_ctx* ctx, const float* xyz_centered, uint64_t n, uint32_t capacity);
/* Host copy of the context's tree in the flat layout (whichever way it got there).  No reference
 * counterpart beyond the friend access of GPUVPTree (libcpu/libcpu/vp-tree.hh:8-17); used by the
 * tests and to cache a tree.  Free with icp_hip_tree_free. */
int icp_hip_get_tree(icp_hip_ctx* ctx, icp_flat_tree** out);
/* Scalar fields of the context's tree only (counts, height, capacity, root); every array pointer
 * of *out is set to NULL.  Nothing is copied from the device. */
int icp_hip_get_tree_info(icp_hip_ctx* ctx, icp_flat_tree* out);

/* Upload this context's shard of the cloud to move ("p", icp.hh:20).
 * Replaces GPUMatrix::from_point_list (libgpu/libgpu/matrix.cu:95-103).
 * n_total is the size of the whole cloud over all shards (== n_local on one GPU).  Any partition
 * of the cloud's rows over the ranks gives the same registration; a rank searches the whole tree
 * for its shard, so spatially compact shards cost it about its share of the one-GPU pass and
 * random subsets 1.5-1.75x that.  The library stores the shard in an order
 * of its own (and, in a sharded run, orders it again once the whole cloud's mean is known);
 * rows, distances and the moved cloud always come back in the order given here. */
int icp_hip_set_queries(icp_hip_ctx* ctx, const float* xyz_aos, uint64_t n_local,
                        uint64_t n_total);

/* The same for a rank that CANNOT choose its rows ("GPU g holds query rows
 * [g n/G, (g+1) n/G)", whatever order the cloud comes in).  No counterpart in the reference
 * (single device).  Collective: every rank calls it with its rows.  The ranks exchange points so
 * that each registers one spatially compact region of about equal size -- the cloud cut by
 * planes perpendicular to x, y, z in turn, at the quantiles of a sample of 16384 rows per rank
 * that every rank sees -- which costs a rank about its share of the one-GPU pass, where an
 * arbitrary slice of rows costs 1.5-1.75x.  Afterwards icp_hip_run works as after icp_hip_set_queries, and
 * icp_hip_get_points is collective too: the moved points travel back and land in the rows the
 * caller handed over here.  icp_hip_nn's per-query outputs are not available on such a shard.
 * The points travel through the callback of icp_hip_set_alltoallv or, without one, through the
 * RCCL communicator (ncclSend / ncclRecv on staging buffers).  One rank: icp_hip_set_queries. */
int icp_hip_set_queries_exchange(icp_hip_ctx* ctx, const float* xyz_aos, uint64_t n_local,
                                 uint64_t n_total);

/* The shard this context registers right now: *n_local points of a cloud of *n_total (after
 * icp_hip_set_queries_exchange n_local is the size of the rank's region, not of the rows it handed
 * over).  Diagnostic (load balance of a sharded run); no counterpart in the reference. */
int icp_hip_get_shard_size(icp_hip_ctx* ctx, uint64_t* n_local, uint64_t* n_total);

/* All-to-all of variable-size blocks on host memory, supplied by the caller (tests: gloo; MPI):
 * `send` holds the blocks for ranks 0 .. n_ranks-1 back to back (send_bytes[r] each), `recv`
 * receives the blocks from ranks 0 .. n_ranks-1 back to back (recv_bytes[r] each; the library
 * knows them beforehand).  Returns 0 on success.  No counterpart in the reference. */
typedef int (*icp_hip_alltoallv_fn)(const void* send, const uint64_t* send_bytes, void* recv,
                                    const uint64_t* recv_bytes, int n_ranks, void* user);
int icp_hip_set_alltoallv(icp_hip_ctx* ctx, int n_ranks, int rank, icp_hip_alltoallv_fn fn,
                          void* user);

/* One batch nearest-neighbour pass over the shard: query_i = p_i - mu_p.
 * Replaces GPUVPTree::closest / closest_kernel (libgpu/libgpu/vp-tree.cu:181-261), with the
 * CPU rules of VPTree::search (libcpu/libcpu/vp-tree.cc:125-175).
 * Outputs are host arrays of n_local entries and may each be NULL (a NULL row_out / dist_out is
 * not computed on the device either: those two go back to the caller's order with scattered
 * stores, which cost as much as the search itself on a large shard):
 *   row_out   row (in the tree's cloud) of the point the reference search returns
 *   xyz_out   that point's coordinates (what VPTree::closest returns, vp-tree.cc:177-187)
 *   dist_out  the fp32 distance the reference search returns with it
 * kernel_ms (may be NULL) receives the HIP-event time of the kernel alone. */
int icp_hip_nn(icp_hip_ctx* ctx, const float mu_p[3], uint32_t* row_out, float* xyz_out,
               float* dist_out, float* kernel_ms);

/* icp_hip_nn for callers that cannot allocate (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE, whose GPU path allocates
 * and frees a dozen temporaries per iteration (libgpu/libgpu/matrix.cu:6-17).
 * Same kernel, same answers as icp_hip_nn against the context's tree, but:
 *   - inputs and outputs are DEVICE buffers of the caller: the queries as three arrays of n floats
 *     (d_px, d_py, d_pz; query_i = p_i - mu_p), and per query the row (d_row_out), the fp32
 *     distance (d_dist_out) and the matched point (d_yx, d_yy, d_yz); each output may be NULL
 *     (the three d_y* together); results are in the caller's order;
 *   - d_sums_out (may be NULL): 12 doubles, [0..8] sum q (x) y in find_covariance's layout
 *     (point-3d.cc:212-236), [9] internal nodes visited, [10] bucket points scanned, [11] stack
 *     entries spilled;
 *   - everything is enqueued on `stream` (a hipStream_t, NULL = the default stream) and nothing
 *     is waited for;
 *   - the call makes no hipMalloc / hipFree / allocating library call: the kernel's work queue,
 *     per-chunk sums and overflow stack live in a static device array of 17.4 MB that every GPU
 *     the library runs on holds until the process ends.  Hence the limits: n <= 1,048,576
 *     queries per call (ICP_HIP_ERR_ARG beyond) and tree height <= 32 (ICP_HIP_ERR_DEPTH: 2 on-chip stack entries + 30 in the array), and a
 *     grid of one workgroup per CU.  All calls on one GPU share that array: calls that overlap
 *     on one GPU (two streams, two contexts, two threads) race; serialise them;
 *   - the kernel reads the context's tree from the CALLER's stream.  The library waits for the
 *     last such launch (an event recorded behind it) before it releases a tree block
 *     (icp_hip_set_tree / icp_hip_build_tree replacing the tree, icp_hip_destroy), so those
 *     calls block until the caller's stream has got that far; `stream` must belong to the
 *     context's device (ICP_HIP_ERR_ARG otherwise). */
int icp_hip_nn_noalloc(icp_hip_ctx* ctx, const float* d_px, const float* d_py, const float* d_pz,
                       uint32_t n, const float mu_p[3], uint32_t* d_row_out, float* d_dist_out,
                       float* d_yx, float* d_yy, float* d_yz, double* d_sums_out, void* stream);

/* All-reduce hook for runs sharded over several contexts: sum `count` doubles in place over all
 * shards (host memory).  Return 0 on success.  Leave unset (or set RCCL below) for one GPU.
 * icp_hip_run calls it once with 4 doubles (the centroid sums of the cloud as uploaded) and then
 * ONCE PER ITERATION with 16: [0..8] sum q (x) y and [9..11] the work counters of this
 * iteration's NN pass, [12] the error and [13..15] the centroid sums of the apply pass of the
 * iteration before (zeros in the first call).  The loop condition of an iteration is therefore
 * known one call later: a run of k iterations makes k + 1 calls of 16. */
typedef int (*icp_hip_allreduce_fn)(double* values, int count, void* user);
int icp_hip_set_allreduce(icp_hip_ctx* ctx, icp_hip_allreduce_fn fn, void* user);

/* RCCL communicator over xGMI, one rank per context.  unique_id is the 128-byte ncclUniqueId
 * made by icp_hip_comm_unique_id on rank 0 and distributed by the caller.  When set, the same
 * 16 doubles are all-reduced on the device, on the context's stream, one ncclAllReduce per
 * iteration, and the loop stays on the device on every rank (iterations enqueued in blocks of
 * 1, 2, 4, 8, 8, .. like the single-GPU loop; every rank reads the same state after the
 * all-reduce and so enqueues the same collectives; a rank whose shard is empty takes part with
 * zeros).  What would make one rank choose another loop (ICP_HIP_LOOP=host, ICP_HIP_NN_KERNEL=deep,
 * ICP_HIP_TILE_STATS, icp_hip_set_allreduce) is rejected with ICP_HIP_ERR_ARG by icp_hip_run.
 * Verified with one rank on one GPU (bit-equal to the run without a communicator) and, for the
 * protocol, with two gloo ranks on the CPU; more than one RCCL rank has not run on hardware. */
int icp_hip_comm_unique_id(void* unique_id_out, size_t size /* >= 128 */);
int icp_hip_comm_init_rank(icp_hip_ctx* ctx, int n_ranks, int rank, const void* unique_id,
                           size_t size);

/* The registration loop: replaces libgpu::icp's loop (libgpu/libgpu/icp.cu:144-158) with the
 * semantics of libcpu::icp (libcpu/libcpu/icp.cc:110-125):
 *   for (i = 0; i < max_iters && error > threshold; ++i) { ... }      error starts at +inf
 * mu_m is mean(m) as icp_hip_mean_host computes it.  progress (may be NULL) is called once per
 * iteration, in order, with the iteration's error (the reference prints "Starting iter" /
 * "Error:" to stderr there, icp.cc:112-113,124).  T_out is the accumulated 4x4 (icp.cc:121).
 * Where libgpu copies nine floats to the host and back for the SVD in every iteration
 * (icp.cu:57) and one float for the error (:157), the iteration here has no host step: solve,
 * error and loop condition run in one-thread kernels, iterations are enqueued in blocks of
 * 1, 2, 4, 8, 8, .. and the state is read back once per block (the calls to progress for a block
 * follow its read-back); what was enqueued past the end of the loop does nothing.  With an
 * all-reduce callback (icp_hip_set_allreduce) the host drives every iteration instead; both
 * loops give the same bits (ICP_HIP_LOOP=host|device at icp_hip_create forces one). */
typedef void (*icp_hip_progress_fn)(uint64_t iteration /*1-based*/, uint64_t max_iters,
                                    float error, void* user);
int icp_hip_run(icp_hip_ctx* ctx, const float mu_m[3], uint64_t max_iters, float threshold,
                float T_out[16], uint64_t* iters_done, float* final_error,
                icp_hip_progress_fn progress, void* user);

/* Moved shard, same order as given to icp_hip_set_queries (icp.cu:160 to_point_list). */
int icp_hip_get_points(icp_hip_ctx* ctx, float* xyz_aos);

/* The whole registration in one call on one GPU: replaces libgpu::icp itself
 * (libgpu/libgpu/icp.hh:19-21, icp.cu:128-161) as main and bench call it (main/main/main.cc:31-34,
 * bench/bench/benchmark.cc:78) -- mu_m = mean(m), tree over m - mu_m (vp_threshold = leaf capacity),
 * the loop of icp_hip_run over the whole of p, the moved cloud back in input order.  m_xyz and
 * p_xyz are borrowed and never written; T_out receives the 4x4, new_p_xyz (n_p x 3 floats, may be
 * NULL) the moved cloud (icp.cu:160).  `ctx` is the context to run on: its tree, shard and
 * buffers are replaced, and a context that is reused for the next call keeps its device buffers
 * when they are large enough (the reference allocates and frees everything inside each call,
 * matrix.cu:6-17; a caller that wants exactly that creates and destroys the context around the
 * call).  Errors: status + icp_hip_last_error(ctx), e.g. ICP_HIP_ERR_DEGENERATE where the
 * reference's tree build is undefined behaviour. */
int icp_hip_icp(icp_hip_ctx* ctx, const float* m_xyz, uint64_t n_m, const float* p_xyz,
                uint64_t n_p, uint64_t iterations, float threshold, uint32_t vp_threshold,
                float T_out[16], float* new_p_xyz, uint64_t* iters_done, float* final_error,
                icp_hip_progress_fn progress, void* user);

/* Phase timers accumulated since the last icp_hip_set_queries (HIP events on the context's
 * stream; the reference has only gettimeofday around icp(), main.cc:29-40).  The loop of
 * icp_hip_run stays on the device between read-backs and an event costs about 5 us of stream
 * time, four per iteration: nn_ms, apply_ms and the NN trace are only collected after
 * icp_hip_set_phase_timers(ctx, 1) (off in a new context; bench.py and the tests turn them on).
 * The launch count and the work counters are always kept; icp_hip_nn always times its pass. */
int icp_hip_set_phase_timers(icp_hip_ctx* ctx, int enabled);
typedef struct icp_hip_timing {
    uint64_t nn_launches;
    double nn_ms;      /* NN + covariance kernel                       */
    double apply_ms;   /* apply + error + centroid kernel              */
    double reduce_ms;  /* final reductions of the per-block partials   */
    double host_ms;    /* wall time of icp_hip_run                     */
    /* deterministic work counters of the reference traversal, summed over the NN passes above
     * (over ALL shards for icp_hip_run, this shard for icp_hip_nn): they price the algorithmic
     * bytes of the NN kernel, 16 B per query + 16 B per internal node visited + 12 B per bucket
     * point scanned. */
    double queries;
    double nodes_visited;
    double points_scanned;
    double stack_spills; /* traversal-stack entries that left LDS for HBM (far-side nesting
                            deeper than the kernel's 2 on-chip entries); diagnostic */
} icp_hip_timing;
int icp_hip_get_timing(icp_hip_ctx* ctx, icp_hip_timing* out);
/* NN-kernel milliseconds of each iteration of the last icp_hip_run (first `capacity` entries
 * copied to ms_out when it is not NULL; *count receives the number of iterations recorded). */
int icp_hip_get_nn_trace(icp_hip_ctx* ctx, float* ms_out, uint64_t capacity, uint64_t* count);

#ifdef __cplusplus
}
#endif
#endif /* ICP_HIP_H */

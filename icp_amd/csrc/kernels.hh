// Device-side data layout and launch wrappers (see kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace icp_dev
{
    // One internal node = one 32-byte record = two 16-byte loads from one 32-byte aligned segment:
    // a lane visiting a node needs all eight words, so the node side of the tree is packed per
    // node; the bucket side, which 64 lanes sweep together, is SoA (sx/sy/sz).
    struct __attribute__((aligned(32))) DevNode
    {
        float cx, cy, cz, r;
        int32_t inside, outside, parent; // child codes: >= 0 internal index, < 0 ~leaf
        float thr; // smallest float whose correctly rounded sqrt is >= r:
                   // sqrtf(d2) < r  <=>  d2 < thr, so a descent needs no sqrt (vp-tree.cc:148)
    };

    struct DevTree
    {
        const DevNode* nodes;
        const uint2* leaf_info; // {first slot, count}
        const float* sx;        // bucket slots (+kSlotSlack slots of +inf behind n_slots)
        const float* sy;
        const float* sz;
        const uint32_t* srow;
        const uint32_t* nrow; // row of each internal node's centre
        // The "ring order" of every bucket, or null (no internal node, or a bucket > 256 points):
        // word k of a bucket's slot range = its point with the k-th smallest distance rho to the
        // centre of the bucket's PARENT node, as (bits of rho with the low 8 mantissa bits
        // cleared) | (slot of the point inside the bucket); +inf behind the bucket's points.  A
        // query at distance d from that centre is at least |rho - d| from the point (triangle
        // inequality), so a scan that starts at rho = d and walks outwards can stop as soon as the
        // gap exceeds the best distance found (nn_tile.hip, ring_walk).
        const uint32_t* ring;
        uint32_t n_slots;
        uint32_t n_nodes;
        int32_t root;
    };

    struct Mat34
    {
        float m[12]; // rows 0..2 of the row-major 4x4
    };

    constexpr int kCovSums = 12;  // [0..8] sum q (x) y (row = P coordinate, col = Y coordinate);
                                  // [9] internal nodes visited, [10] bucket points scanned
                                  // (the reference traversal's deterministic work counters);
                                  // [11] stack entries nn_tile_kernel spilled from LDS to HBM
    constexpr int kApplySums = 4; // error, sum x', sum y', sum z'
    // The sums of one iteration as the loop keeps (and all-reduces) them, 16 doubles:
    // [0..11] the NN pass' kCovSums, [12..15] the kApplySums of the apply pass before it (the
    // centroid sums of the initial cloud for the first iteration).
    constexpr int kIterSums = kCovSums + kApplySums;
    constexpr int kApplyAt = kCovSums;
    constexpr uint32_t kNnMaxBlocks = 256 * 8;
    constexpr uint32_t kSlotSlack = 256; // a staged bucket window may read this far past a bucket
    constexpr uint32_t kStreamMaxBlocks = 256 * 8;

    // State of a registration loop that stays on the device (loop.hip): the host enqueues
    // iterations ahead and reads this back once per block; every kernel of an iteration does
    // nothing once `done` is set, so iterations enqueued past the end of the loop
    // (libcpu/libcpu/icp.cc:110: i < iterations && error > threshold) change nothing.
    constexpr int kLoopBlockMax = 8; // iterations enqueued between two readbacks
    struct LoopState
    {
        float T[16];     // accumulated transform (icp.cc:121)
        float Tn[12];    // rows 0..2 of this iteration's transform, read by apply_error_kernel
        float mu_p[3];   // centroid of the cloud to move (icp.cc:114), used by the solve (:57)
        float mu_q[3];   // what the NN kernel subtracts from p (:115).  One GPU: mean(p) of the
                         // cloud as it stands, every iteration, as the reference (:114-115).
                         // Sharded: mean(p) in the first iteration, mu_m afterwards (t = mu_m -
                         // r mu_p puts the moved centroid on mu_m up to fp32 rounding, so the next
                         // pass can start before the moved cloud has been summed over the ranks)
        float mu_m[3];
        float error;     // icp.cc:123
        float threshold;
        uint32_t it;     // iterations completed (error known, loop condition evaluated)
        uint32_t started; // iterations solved and applied; started - it is 0 or 1
        uint32_t max_iters;
        uint32_t done;
        uint32_t sharded; // 1: one collective per iteration (mu_q rule above, loop_advance_step)
        double n_total;
        double work[3];  // reference-traversal counters summed over the iterations (sums 9..11)
        float errors[kLoopBlockMax]; // error after iteration i at [i % kLoopBlockMax]
    };

    struct NnLaunch
    {
        DevTree tree;
        uint32_t height;
        uint32_t max_leaf_count;
        const float *px, *py, *pz;
        float mu[3];
        uint32_t n;
        float *yx, *yy, *yz; // may be null together
        uint32_t* row_out;   // may be null
        float* dist_out;     // may be null
        const uint32_t* perm; // may be null: row_out/dist_out are written at perm[i] (input order)
        double* partials;    // [grid][kCovSums] or null
        uint32_t grid;
        const LoopState* loop; // device-resident loop: mu comes from loop->mu_q, no-op once done
    };

    int nn_max_depth_supported();
    uint32_t nn_grid(uint64_t n);
    uint32_t stream_grid(uint64_t n);

    hipError_t launch_nn(const NnLaunch& a, hipStream_t s);
    // nn_tile.hip: the production kernel (a.partials is not used: sums go to q.chunk_sums)
    constexpr uint32_t kQueueWords = 288; // 8 queue heads, one per 128-byte line, + the count of
                                          // finished workgroups (word 256)
    // The diagnostic build (ICP_HIP_TILE_STATS=1) counts into kTileStatsCopies copies of its
    // kTileStatsWords counters (workgroup b adds to copy b % 64; the host adds the copies up):
    // thousands of waves flushing ~40 counters each to ONE set of words at the end of a pass held
    // up the memory accesses of the waves still working by ~0.9 ms.
    constexpr int kTileStatsWords = 24 + 64 + 64 + 18 + 6;
    constexpr int kTileStatsCopies = 64;
    struct TileLaunch
    {
        uint32_t chunk_batches; // batches per chunk (1..8)
        uint32_t batch_queries; // queries per batch = active lanes per wave: 64, 32, 16, 8, 4 (or 2).
                                // 64 for large shards; a shard too small to fill the chip is cut
                                // into narrower batches so that every SIMD gets waves, and the
                                // idle lanes join the bucket scans as extra parts
        double* chunk_sums;     // [n_chunks][kCovSums] or null
        uint32_t* queue;        // kQueueWords words, zero at launch; the last workgroup to finish
                                // leaves them zero for the next launch
        uint2* spill;           // [max(0, height - stack entries)][grid * 256] or null
        unsigned long long* stats; // kTileStatsWords x kTileStatsCopies counters of the diagnostic build
                                   // (events, cycles by phase, histograms), or null
        uint32_t cut; // leave the node phase when waiting lanes >= cut/16 x stepping lanes (0: never)
        // Stragglers (nn_tile.hip): the last <= park_max lanes at work in a batch move to the
        // wave's lot in `park` and are run later, a batch of them together; their terms go to
        // late rows (kCovSums words each: nine fp32 products, three integer counters), late_rows
        // per chunk, of which late_cnt[chunk] are in use after the launch.  park null: off.
        uint32_t* park = nullptr;     // [grid * 4 waves][kParkLotWords] or null
        uint32_t* late = nullptr;     // [n_chunks][late_rows][kCovSums]
        uint32_t* late_cnt = nullptr; // [n_chunks]
        uint32_t late_rows = 0;
        uint32_t park_max = 0;
        uint32_t park_wait = 0; // rounds (node + leaf phase) a batch goes on with <= park_max lanes
                                // at work before it parks them: short excursions end in place
    };
    constexpr uint32_t kParkLotWords = 64 * 24; // words of ONE lot; a wave has two (nn_tile.hip asserts the size)
    constexpr uint32_t kParkMaxLanes = 32;
    hipError_t launch_nn_tile(const NnLaunch& a, const TileLaunch& q, hipStream_t s);
    // How a shard of n queries is laid on a grid of at most max_blocks resident workgroups.
    struct TilePlan
    {
        uint32_t batch_queries, grid_blocks, chunk_batches, n_chunks;
    };
    // cus (0 = not known): compute units of the device; lets the plan leave workgroup slots of a
    // CU empty when the shard has too few batches to keep every resident wave busy
    TilePlan plan_tile(uint64_t n, uint32_t max_blocks, uint32_t forced_batch_queries,
                       uint32_t cus = 0);

    // ---- the pass that allocates nothing (icp_hip_nn_noalloc) ----
    // Its work queue, per-chunk sums, partial rows and overflow stack live in a file-scope
    // __device__ array of nn_tile.hip (kNoAllocScratchBytes, held by every GPU the library runs
    // on); the limits below are what that array is sized for.
    constexpr uint32_t kNoAllocMaxQueries = 1u << 20; // queries per call
    constexpr uint32_t kNoAllocMaxBlocks = 256;       // grid: one workgroup per CU
    constexpr uint32_t kNoAllocMaxChunks = 16384;     // plan_tile(n <= 2^20, 256 blocks) stays below
    constexpr uint32_t kNoAllocSpillLevels = 30;      // tree height <= 30 + the on-chip entries
    constexpr size_t kNoAllocScratchBytes =
        4096                                                        // queue heads
        + (size_t)kNoAllocMaxChunks * kCovSums * 8                    // per-chunk sums
        + (size_t)1024 * kCovSums * 8                               // partial rows
        + (size_t)kNoAllocSpillLevels * kNoAllocMaxBlocks * 256 * 8; // overflow stack
    // once per device, with that device current: resolves the array's device address
    hipError_t nn_noalloc_setup(int device);
    // a: tree, inputs and outputs (a.grid, a.partials ignored); sums_out: kCovSums doubles in
    // device memory or null.  Returns hipErrorInvalidValue beyond the limits above.
    hipError_t launch_nn_noalloc(int device, const NnLaunch& a, double* sums_out, uint32_t cut,
                                 hipStream_t s);
    hipError_t nn_tile_blocks_per_cu(int* out);
    uint32_t nn_tile_stack_entries();
    // out[b][k] = sum of rows [b*rows_per_block, ...) of rows[n_rows][K], fixed order; K <= 16
    hipError_t launch_reduce_rows(const double* rows, uint32_t n_rows, int K, uint32_t grid,
                                  double* out, hipStream_t s);
    // `loop` null: T and mu_m are the arguments; else they are loop->Tn / loop->mu_m.
    // `out` not null (grid must be 1): the kApplySums sums go straight to out, without a
    // reduce_partials launch, followed by the loop step `tail` (kTailFinish or kTailNone)
    hipError_t launch_apply_error(float* px, float* py, float* pz, const float* yx, const float* yy,
                                  const float* yz, const Mat34& T, const float mu_m[3], uint32_t n,
                                  uint32_t grid, double* partials, LoopState* loop, hipStream_t s,
                                  double* out = nullptr, int tail = 0);
    // the per-chunk sums of nn_tile_kernel -> the kCovSums sums in `out`, in a fixed order: one
    // launch up to kChunksOneStage chunks, two (through `partials`, <= 1024 rows) above
    constexpr uint32_t kChunksOneStage = 512;
    hipError_t launch_chunk_sums(const double* chunk_sums, uint32_t n_chunks, double* partials,
                                 double* out, hipStream_t s, LoopState* loop = nullptr,
                                 int tail = 0);
    // the same with the late rows of the chunks' parked queries (TileLaunch): every chunk's row
    // of chunk_sums, then its late_cnt[chunk] late rows, in a fixed order; always two launches
    hipError_t launch_chunk_sums_late(const double* chunk_sums, uint32_t n_chunks,
                                      const uint32_t* late, const uint32_t* late_cnt,
                                      uint32_t late_rows, double* partials, double* out,
                                      hipStream_t s, LoopState* loop = nullptr, int tail = 0);
    // loop.hip: single-thread steps of the device-resident loop
    hipError_t launch_loop_init(LoopState* st, const double* sums, const float mu_m[3],
                                float threshold, uint32_t max_iters, double n_total, bool sharded,
                                hipStream_t s);
    // sums = the kIterSums doubles of one iteration (all-reduced): finish the iteration whose apply
    // pass is pending, if any, then solve the next one unless the loop is over
    hipError_t launch_loop_advance(LoopState* st, const double* sums, hipStream_t s);
    hipError_t launch_sum_points(const float* px, const float* py, const float* pz, uint32_t n,
                                 uint32_t grid, double* partials, hipStream_t s);
    // `loop` + `tail`: the device-resident loop's scalar step on these sums, run by the same
    // launch (and the launch does nothing once loop->done)
    enum : int
    {
        kTailNone = 0,
        kTailSolve = 1,
        kTailFinish = 2
    };
    hipError_t launch_reduce_partials(const double* partials, uint32_t nblocks, int K, double* out,
                                      hipStream_t s, LoopState* loop = nullptr,
                                      int tail = kTailNone);
    hipError_t launch_aos_to_soa(const float* aos, uint32_t n, float* x, float* y, float* z,
                                 hipStream_t s);
    hipError_t launch_soa_to_aos(const float* x, const float* y, const float* z, uint32_t n,
                                 float* aos, hipStream_t s);
    // sort.hip: Morton ordering of the cloud to move
    hipError_t launch_minmax(const float* aos, uint32_t n, uint32_t grid, float* partials,
                             hipStream_t s);
    hipError_t launch_morton(const float* aos, uint32_t n, const float lo[3], const float scale[3],
                             uint32_t* keys, uint32_t* idx, hipStream_t s);
    // temp null (or too small): temporary storage is allocated, the stream waited for, and freed
    hipError_t sort_pairs_u32(const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* idx_in,
                              uint32_t* idx_out, uint32_t n, uint32_t bits, hipStream_t s,
                              void* temp = nullptr, size_t temp_bytes = 0);
    hipError_t sort_pairs_u32_temp_bytes(uint32_t n, uint32_t bits, size_t* bytes);
    hipError_t launch_leaf_keys(const DevTree& t, const float* px, const float* py, const float* pz,
                                const float mu[3], uint32_t n, uint32_t* keys, uint32_t* idx,
                                hipStream_t s);
    hipError_t launch_permute_soa(const float* x, const float* y, const float* z,
                                  const uint32_t* perm, const uint32_t* order, uint32_t n, float* ox,
                                  float* oy, float* oz, uint32_t* operm, hipStream_t s);
    hipError_t launch_gather_soa(const float* aos, const uint32_t* perm, uint32_t n, float* x,
                                 float* y, float* z, hipStream_t s);
    hipError_t launch_scatter_aos(const float* x, const float* y, const float* z,
                                  const uint32_t* perm, uint32_t n, float* aos, hipStream_t s);

    // ring order of the buckets (DevTree::ring): t.ring is ignored, `ring` (n_slots words) is written;
    // leaf_parent is scratch of n_leaves words.  Needs n_nodes >= 1 and every bucket <= 256 points.
    hipError_t launch_ring_build(const DevTree& t, uint32_t n_leaves, int32_t* leaf_parent,
                                 uint32_t* ring, hipStream_t s);
    hipError_t launch_pack_nodes(const float* cx, const float* cy, const float* cz, const float* r,
                                 const int32_t* inside, const int32_t* outside,
                                 const int32_t* parent, uint32_t n, DevNode* out, hipStream_t s);
} // namespace icp_dev

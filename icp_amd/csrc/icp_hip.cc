// C-ABI implementation (include/icp_hip.h): context, device memory, the registration loop.
// Host code only; kernels and their launch wrappers live in kernels.hip.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <limits>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "host_common.hh"
#include "icp_hip.h"
#include "kernels.hh"
#include "loop_steps.hh"
#include "tree_build.hh"

using namespace icp_dev;

namespace
{
    thread_local std::string g_error;

    // ---- RCCL, resolved at run time so the library loads (and single-GPU runs work) without it
    struct Rccl
    {
        typedef struct
        {
            char internal[128];
        } UniqueId;
        typedef void* Comm;
        int (*GetUniqueId)(UniqueId*) = nullptr;
        int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
        int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
        int (*Send)(const void*, size_t, int, int, Comm, hipStream_t) = nullptr;
        int (*Recv)(void*, size_t, int, int, Comm, hipStream_t) = nullptr;
        int (*GroupStart)() = nullptr;
        int (*GroupEnd)() = nullptr;
        int (*CommDestroy)(Comm) = nullptr;
        const char* (*GetErrorString)(int) = nullptr;
        void* handle = nullptr;
        bool tried = false;
        static constexpr int kDouble = 8; // ncclDouble / ncclFloat64
        static constexpr int kSum = 0;    // ncclSum
        static constexpr int kInt8 = 0;   // ncclInt8 / ncclChar

        bool load(std::string& err)
        {
            if (handle)
                return true;
            if (tried)
            {
                err = "RCCL library not available";
                return false;
            }
            tried = true;
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            {
                handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (handle)
                    break;
            }
            if (!handle)
            {
                err = std::string("cannot load librccl: ") + dlerror();
                return false;
            }
            GetUniqueId = (decltype(GetUniqueId))dlsym(handle, "ncclGetUniqueId");
            CommInitRank = (decltype(CommInitRank))dlsym(handle, "ncclCommInitRank");
            AllReduce = (decltype(AllReduce))dlsym(handle, "ncclAllReduce");
            CommDestroy = (decltype(CommDestroy))dlsym(handle, "ncclCommDestroy");
            GetErrorString = (decltype(GetErrorString))dlsym(handle, "ncclGetErrorString");
            Send = (decltype(Send))dlsym(handle, "ncclSend");
            Recv = (decltype(Recv))dlsym(handle, "ncclRecv");
            GroupStart = (decltype(GroupStart))dlsym(handle, "ncclGroupStart");
            GroupEnd = (decltype(GroupEnd))dlsym(handle, "ncclGroupEnd");
            if (!GetUniqueId || !CommInitRank || !AllReduce || !CommDestroy)
            {
                err = "librccl is missing ncclGetUniqueId/ncclCommInitRank/ncclAllReduce";
                handle = nullptr;
                return false;
            }
            return true;
        }
    } g_rccl;

    // ---- per-device pool of what a context is made of -------------------------------------
    // The reference allocates and frees everything inside each icp() call (matrix.cu:6-17), and
    // a caller of the drop-in does the same with a context per call.  hipMalloc / hipFree /
    // hipStreamCreate / hipHostMalloc cost 10-40 us each (hipFree synchronises the device), which
    // is more than a whole registration of a small cloud, so small device blocks, streams,
    // events and the pinned state buffer return to a per-device pool instead of the driver and
    // the next context takes them from there.  Blocks above kPoolBlockMax, or beyond
    // kPoolTotalMax of cached bytes per device, go back to the driver as before.
    constexpr size_t kPoolBlockMax = 8u << 20;   // largest block that is cached
    constexpr size_t kPoolTotalMax = 64u << 20;  // cached bytes per device
    constexpr int kPoolDevices = 64;

    // A shard of fewer than kSmallBelow points is not sorted, and its loop state, NN work queue
    // and coordinates share ONE device block [LoopState | queue | x | y | z] with a pinned image
    // on the host for each direction: one copy up, one copy down, one synchronisation per
    // registration (set-up, loop and download each cost a round trip of 15-25 us otherwise).
    constexpr uint64_t kSmallBelow = 8192;
    constexpr size_t kSmallStateBytes = 256;
    constexpr size_t kSmallHead = 1536; // state + kQueueWords words, rounded up to 256
    constexpr size_t kSmallBlockBytes = kSmallHead + 3 * kSmallBelow * sizeof(float);
    constexpr size_t kStateDoubles = (sizeof(LoopState) + sizeof(double) - 1) / sizeof(double);
    constexpr size_t kPinnedHead = 512; // 16 doubles of sums + the loop state
    constexpr size_t kPinnedBytes = kPinnedHead + 2 * kSmallBlockBytes;
    static_assert(sizeof(LoopState) <= kSmallStateBytes, "LoopState outgrew its slot");
    static_assert(kSmallStateBytes + kQueueWords * sizeof(uint32_t) <= kSmallHead, "queue slot");
    static_assert((16 + kStateDoubles) * sizeof(double) <= kPinnedHead, "pinned head");
    struct PooledCtx
    {
        hipStream_t stream = nullptr;
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        double* h_sums = nullptr;
        std::vector<hipEvent_t> loop_events;
    };
    struct DevicePool
    {
        std::mutex m;
        std::multimap<size_t, void*> blocks[kPoolDevices]; // capacity in bytes -> block
        size_t cached[kPoolDevices] = {};
        std::vector<PooledCtx> ctxs[kPoolDevices];
    };
    DevicePool& pool()
    {
        static DevicePool* p = new DevicePool(); // never destroyed: the HIP runtime may be gone first
        return *p;
    }
    size_t pool_class(size_t bytes)
    {
        size_t c = 256;
        while (c < bytes)
            c <<= 1;
        return c;
    }
    hipError_t pool_malloc(void** out, size_t bytes, size_t* cap_bytes, bool* pooled)
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        *pooled = bytes <= kPoolBlockMax && dev >= 0 && dev < kPoolDevices;
        if (!*pooled)
        {
            *cap_bytes = bytes;
            return hipMalloc(out, bytes);
        }
        const size_t cls = pool_class(bytes);
        *cap_bytes = cls;
        {
            DevicePool& P = pool();
            std::lock_guard<std::mutex> lock(P.m);
            auto it = P.blocks[dev].find(cls);
            if (it != P.blocks[dev].end())
            {
                *out = it->second;
                P.blocks[dev].erase(it);
                P.cached[dev] -= cls;
                return hipSuccess;
            }
        }
        return hipMalloc(out, cls);
    }
    void pool_free(void* p, size_t cap_bytes, bool pooled)
    {
        if (!p)
            return;
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (pooled && dev >= 0 && dev < kPoolDevices)
        {
            DevicePool& P = pool();
            std::lock_guard<std::mutex> lock(P.m);
            if (P.cached[dev] + cap_bytes <= kPoolTotalMax)
            {
                P.blocks[dev].emplace(cap_bytes, p);
                P.cached[dev] += cap_bytes;
                return;
            }
        }
        (void)hipFree(p);
    }

    // Device array that is kept across calls: alloc(n) reuses the allocation when it is large
    // enough (a context that is reused for the next registration allocates nothing new).  The
    // owning context's device is current whenever alloc / release run, and its stream has been
    // synchronised before a block goes back to the pool.
    template <class T>
    struct DevBuf
    {
        T* p = nullptr;
        size_t count = 0; // logical size
        size_t cap = 0;   // allocated entries
        size_t cap_bytes = 0;
        bool pooled = false;
        bool borrowed = false; // p points into another DevBuf's block (never freed from here)
        hipError_t alloc(size_t n)
        {
            if (borrowed)
                release();
            if (n != 0 && p && n <= cap)
            {
                count = n;
                return hipSuccess;
            }
            release();
            if (n == 0)
                return hipSuccess;
            hipError_t e = pool_malloc((void**)&p, n * sizeof(T), &cap_bytes, &pooled);
            if (e == hipSuccess)
            {
                count = n;
                cap = cap_bytes / sizeof(T);
            }
            else
                p = nullptr;
            return e;
        }
        void adopt(T* ptr, size_t n) // take over an allocation made elsewhere (hipMalloc)
        {
            release();
            p = ptr;
            count = cap = n;
            cap_bytes = n * sizeof(T);
            pooled = false;
        }
        void borrow(T* ptr, size_t n)
        {
            release();
            p = ptr;
            count = cap = n;
            borrowed = true;
        }
        void release()
        {
            if (!borrowed)
                pool_free(p, cap_bytes, pooled);
            p = nullptr;
            count = cap = cap_bytes = 0;
            pooled = borrowed = false;
        }
    };
} // namespace

constexpr uint64_t kParkMinPoints = 6'000'000;       // shards from which the straggler lots pay once aligned
constexpr uint64_t kParkEarlyMinTreePoints = 48'000'000; // reference clouds from which they pay in every pass

struct icp_hip_ctx
{
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // tree replica
    bool has_tree = false;
    DevTree tree{};
    uint32_t height = 0, max_leaf = 0, capacity = 0;
    uint64_t n_tree_points = 0;
    DevBuf<DevNode> nodes;
    DevBuf<uint2> leaf_info;
    DevBuf<float> sx, sy, sz;
    DevBuf<uint32_t> srow, nrow;
    DevBuf<uint32_t> ring; // DevTree::ring (kernels.hh), when every bucket has <= 256 points
    DevBuf<unsigned char> tree_blob; // a small tree: all of the above in one block, one upload
    std::vector<unsigned char> tree_blob_host; // its host image (kept until the next upload)
    uint32_t n_leaves = 0;

    // query shard
    uint64_t n_local = 0, n_total = 0;
    bool has_queries = false;
    DevBuf<float> px, py, pz, yx, yy, yz, aos, dist;
    DevBuf<uint32_t> row;
    DevBuf<uint32_t> perm; // sorted position -> input row (Morton order), empty when unsorted
    bool sorted = false;
    bool leaf_ordered = false, resorted = false; // see maybe_resort
    bool resort_decided = false; // with a communicator: the decision every rank takes alike
    float root_radius = 0.f;     // radius of the tree's root sphere (the same tree on every rank)
    double q_extent = 0.0; // largest side of the shard's bounding box at set-up
    float order_mu[3] = {0.f, 0.f, 0.f}; // the centre the shard's leaf order was computed around
    float mean_hint[3] = {0.f, 0.f, 0.f}; // the whole cloud's mean as the last sharded run saw it
    uint64_t mean_hint_n_total = 0;       // ... for a cloud of this many points (0: none)
    DevBuf<uint32_t> sk_keys, sk_keys2, sk_idx, sk_order; // the sorts' arrays, kept for maybe_resort
    DevBuf<unsigned char> sk_temp;
    // small shard (kSmallBelow): px/py/pz, queue and loop_state are views into `qblock`
    bool small = false;
    DevBuf<unsigned char> qblock;
    unsigned char* h_up = nullptr;   // pinned image of the block on its way to the device
    unsigned char* h_down = nullptr; // ... and as read back after a registration
    size_t small_bytes = 0;          // bytes of the block in use (head + 3 padded arrays)
    uint32_t small_pad = 0;          // floats per coordinate array in the block
    bool upload_pending = false;     // h_up has not been copied to the device yet
    bool host_points_valid = false;  // h_down holds the cloud as the last registration left it
    bool first_sums_valid = false;   // first_sums = centroid sums of the cloud as uploaded
    double first_sums[kApplySums] = {0.0, 0.0, 0.0, 0.0};
    DevBuf<double> partials, sums;
    double* h_sums = nullptr; // pinned, 16 doubles

    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // icp_hip_nn_noalloc runs on the CALLER's stream and reads this context's tree: the event is
    // recorded behind every such launch and waited for before a tree block is released
    hipEvent_t noalloc_ev = nullptr;
    bool noalloc_pending = false;
    uint32_t tile_blocks = 0;   // resident nn_tile_kernel workgroups on this device (multiple of 8)
    uint32_t cus = 0;           // compute units (0 when tile_blocks is not a whole number per CU)
    bool force_deep = false;    // ICP_HIP_NN_KERNEL=deep: always run nn_cov_kernel
    int tree_build_on = 0;      // ICP_HIP_TREE_BUILD: 0 auto (by size), 1 device, 2 host
    // nn_tile_kernel work queue / per-chunk sums / overflow stack (sized by prepare_tile)
    DevBuf<uint32_t> queue;
    DevBuf<double> chunk_sums;
    DevBuf<uint2> spill;
    DevBuf<unsigned long long> stats; // ICP_HIP_TILE_STATS=1: event counters, printed per pass
    uint32_t chunk_batches = 1, n_chunks = 0, tile_grid_blocks = 8, spill_levels = 0;
    uint32_t batch_queries = 64; // queries per wave-batch of nn_tile_kernel (prepare_tile)
    bool tile_ready = false;
    uint32_t tile_cut = 4; // in 16ths; ICP_HIP_TILE_CUT overrides (0 = run the node phase to the end)
    // stragglers of nn_tile_kernel (nn_tile.hip): one lot per resident wave, late rows per chunk
    DevBuf<uint32_t> park, late, late_cnt;
    // When: measured on one MI355X (scripts/park_probe.py, bench.py): once the clouds are aligned
    // the pass is 9-13 % faster from 16M points up, 3 % at 8M, and SLOWER below (every wave ends
    // with a batch of what is left in its lot, which weighs on a short pass); while the clouds
    // are still a point spacing apart it is 6-9 % faster at 64M points (14.0 -> 13.0 ms with up to
    // 16 lanes parked per batch and 8 late rows per batch, 12.4-12.8 with 12 rows; flat from there
    // to 24 rows and between 12 and 16 lanes), even at 32M and 9 % slower at
    // 16M -- and it fetches 26 GB per 64M pass instead of 9.3: the parked queries of a
    // misaligned batch come back in batches that stage buckets from all over the tree, each for
    // one lane (the fabric has the room: 2 TB/s of 8).  So the lots are used on shards of
    // kParkMinPoints and more in the passes after the second ordering pass (maybe_resort: the
    // registration's own sign that the clouds are aligned; +0.8 GB fetched and +0.13 GB
    // written per 64M pass, for 10 % of its time), and in every pass when the REFERENCE cloud
    // has kParkEarlyMinTreePoints points or more (the 64M iteration 4.81 -> 4.65 ms; a rank's
    // share of that cloud alike -- scripts/rank_replay.py, misaligned passes of an eighth of it
    // 1.79 1.74 1.51 1.15 -> 1.58 1.65 1.33 1.01 ms, of a half 7.30 7.01 6.21 4.52 -> 6.81 6.59
    // 5.75 4.10: what makes the lots pay before the clouds are aligned is a deep tree, in which
    // a far-side search is long, not a long shard).
    // ICP_HIP_PARK=always / never and ICP_HIP_PARK_MAX / ICP_HIP_LATE_PER_BATCH override (tests,
    // experiments).
    uint32_t park_max_env = 0, late_per_batch_env = 0; // 0: not set
    uint32_t park_wait = 0;       // ICP_HIP_PARK_WAIT
    int park_mode = 0;            // 0: the rule above, 1: always, 2: never
    bool park_ready = false;      // lots and late rows are allocated for the current launch shape
    bool park_now() const
    {
        return park_ready && (park_mode == 1 || resorted || n_tree_points >= kParkEarlyMinTreePoints);
    }
    uint32_t park_max() const { return park_max_env ? park_max_env : resorted ? 8u : 16u; }
    uint32_t late_per_batch() const { return late_per_batch_env ? late_per_batch_env : resorted ? 2u : 12u; }
    uint32_t late_per_batch_most() const
    {
        return late_per_batch_env ? late_per_batch_env
               : park_mode == 1 || n_tree_points >= kParkEarlyMinTreePoints ? 12u : 2u;
    }
    uint32_t late_rows() const { return late_per_batch() * chunk_batches; }

    icp_hip_allreduce_fn allreduce = nullptr;
    void* allreduce_user = nullptr;
    Rccl::Comm comm = nullptr;
    int n_ranks = 1, rank = 0; // of the RCCL communicator

    // icp_hip_set_queries_exchange: the caller's rows went to the ranks that own their region of
    // the tree; what is needed to bring the moved points back to the caller's rows
    icp_hip_alltoallv_fn alltoallv = nullptr;
    void* alltoallv_user = nullptr;
    int ex_ranks = 0, ex_rank = 0;          // of the callback (icp_hip_set_alltoallv)
    bool exchanged = false;                 // the shard in HBM is not the caller's rows
    uint64_t ex_n_caller = 0;               // rows the caller handed over
    std::vector<uint64_t> ex_sent, ex_got;  // points sent to / received from each rank
    std::vector<uint32_t> ex_rows;          // caller's row of every point sent, in send order

    icp_hip_timing timing{};
    std::vector<float> nn_trace; // NN kernel ms of each iteration of the last icp_hip_run

    // device-resident loop (loop.hip): state on the device, its pinned host copy, 4 events per
    // enqueued iteration; ICP_HIP_LOOP=host|device overrides the choice
    LoopState* loop_state = nullptr; // behind the 16 doubles of `sums` (one allocation)
    LoopState* h_loop = nullptr;     // behind the 16 doubles of `h_sums`
    std::vector<hipEvent_t> loop_events;
    int loop_mode = 0; // 0 auto, 1 host-driven, 2 device-resident
    bool phase_timers = false; // icp_hip_set_phase_timers
};

namespace
{
    int fail(icp_hip_ctx* c, int status, const std::string& msg)
    {
        if (c)
            c->err = msg;
        g_error = msg;
        return status;
    }
    int hip_fail(icp_hip_ctx* c, hipError_t e, const char* what)
    {
        return fail(c, e == hipErrorOutOfMemory ? ICP_HIP_ERR_NOMEM : ICP_HIP_ERR_HIP,
                    std::string(what) + ": " + hipGetErrorString(e));
    }
#define HIP_TRY(ctx, expr)                                                                      \
    do                                                                                          \
    {                                                                                           \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return hip_fail(ctx, e_, #expr);                                                    \
    } while (0)

    struct DeviceGuard
    {
        int prev = -1;
        explicit DeviceGuard(int dev)
        {
            if (hipGetDevice(&prev) != hipSuccess)
                prev = -1;
            if (prev != dev)
                (void)hipSetDevice(dev);
            else
                prev = -1;
        }
        ~DeviceGuard()
        {
            if (prev >= 0)
                (void)hipSetDevice(prev);
        }
    };

    // time of the kernel(s) between ev0 and ev1, after the stream has been synchronised
    float elapsed_ms(icp_hip_ctx* c)
    {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess)
            ms = 0.f;
        return ms;
    }

    // c->sums[first, first + count) -> c->h_sums (same offsets), waited for
    int sums_to_host(icp_hip_ctx* c, int first, int count)
    {
        HIP_TRY(c, hipMemcpyAsync(c->h_sums + first, c->sums.p + first, sizeof(double) * count,
                                  hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return ICP_HIP_OK;
    }

    // Sum c->sums[first, first + count) over all ranks, in place, on the context's stream (RCCL).
    int rccl_allreduce(icp_hip_ctx* c, int first, int count)
    {
        const int rc = g_rccl.AllReduce(c->sums.p + first, c->sums.p + first, (size_t)count,
                                        Rccl::kDouble, Rccl::kSum, c->comm, c->stream);
        if (rc != 0)
            return fail(c, ICP_HIP_ERR_COMM,
                        std::string("ncclAllReduce: ") +
                            (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error"));
        return ICP_HIP_OK;
    }

    // Sum c->h_sums[first, first + count) over all shards through the caller's callback.
    int host_allreduce(icp_hip_ctx* c, int first, int count)
    {
        if (c->allreduce && c->allreduce(c->h_sums + first, count, c->allreduce_user) != 0)
            return fail(c, ICP_HIP_ERR_COMM, "all-reduce callback failed");
        return ICP_HIP_OK;
    }

    // Before a tree block is released (replaced tree, destroyed context): the last
    // icp_hip_nn_noalloc launch, which reads the tree from a stream that is not the context's.
    void drain_noalloc(icp_hip_ctx* c)
    {
        if (c->noalloc_pending && c->noalloc_ev)
            (void)hipEventSynchronize(c->noalloc_ev);
        c->noalloc_pending = false;
    }

    NnLaunch make_nn(icp_hip_ctx* c, const float mu[3], bool want_y, bool want_row, bool want_dist,
                     bool want_cov)
    {
        NnLaunch a{};
        a.tree = c->tree;
        a.height = c->height;
        a.max_leaf_count = c->max_leaf;
        a.px = c->px.p;
        a.py = c->py.p;
        a.pz = c->pz.p;
        a.mu[0] = mu[0];
        a.mu[1] = mu[1];
        a.mu[2] = mu[2];
        a.n = (uint32_t)c->n_local;
        a.yx = want_y ? c->yx.p : nullptr;
        a.yy = want_y ? c->yy.p : nullptr;
        a.yz = want_y ? c->yz.p : nullptr;
        a.row_out = want_row ? c->row.p : nullptr;
        a.dist_out = want_dist ? c->dist.p : nullptr;
        a.perm = c->sorted ? c->perm.p : nullptr;
        a.partials = want_cov ? c->partials.p : nullptr;
        a.grid = nn_grid(c->n_local);
        return a;
    }


    // Size the tile kernel's launch for the current tree + shard (called when either changes;
    // allocates, so never from the iteration loop).
    int prepare_tile(icp_hip_ctx* c)
    {
        c->tile_ready = false;
        if (!c->has_tree || !c->has_queries || c->n_local == 0)
            return ICP_HIP_OK;
        const char* env = std::getenv("ICP_HIP_BATCH_QUERIES");
        const TilePlan plan = plan_tile(c->n_local, c->tile_blocks,
                                        env ? (uint32_t)std::atoi(env) : 0u, c->cus);
        c->batch_queries = plan.batch_queries;
        c->tile_grid_blocks = plan.grid_blocks;
        c->chunk_batches = plan.chunk_batches;
        c->n_chunks = plan.n_chunks;
        if (const char* cbe = std::getenv("ICP_HIP_CHUNK_BATCHES")) // experiments only
        {
            const uint32_t cb = (uint32_t)std::atoi(cbe);
            if (cb >= 1 && cb <= 64)
            {
                const uint64_t batches = (c->n_local + plan.batch_queries - 1) / plan.batch_queries;
                c->chunk_batches = cb;
                c->n_chunks = (uint32_t)((batches + cb - 1) / cb);
            }
        }
        const uint32_t k = nn_tile_stack_entries();
        c->spill_levels = c->height > k ? c->height - k : 0;
        if (!c->small) // (a small shard's queue is part of its block and arrives zeroed with it)
        {
            HIP_TRY(c, c->queue.alloc(kQueueWords));
            // zero once: every launch of nn_tile_kernel leaves the queue zeroed behind it
            HIP_TRY(c, hipMemsetAsync(c->queue.p, 0, kQueueWords * sizeof(uint32_t), c->stream));
        }
        {
            const char* env = std::getenv("ICP_HIP_TILE_STATS");
            if (env && env[0] == '1')
                HIP_TRY(c, c->stats.alloc((size_t)kTileStatsWords * kTileStatsCopies));
        }
        // (a small shard runs a handful of batches: nothing to gather in a lot)
        c->park_ready = false;
        if (!c->small && c->batch_queries == 64 && c->park_mode != 2 &&
            (c->park_mode == 1 || c->n_local >= kParkMinPoints))
        {
            HIP_TRY(c, c->park.alloc((size_t)c->tile_grid_blocks * 4 * 2 * kParkLotWords));
            // late rows: late_per_batch per batch however the batches are cut into chunks
            // (maybe_resort makes the chunks larger later; a last partial chunk rounds up)
            const uint64_t batches = (c->n_local + c->batch_queries - 1) / c->batch_queries;
            HIP_TRY(c, c->late.alloc((size_t)(batches + 64) * c->late_per_batch_most() * kCovSums));
            HIP_TRY(c, c->late_cnt.alloc(c->n_chunks));
            c->park_ready = true;
        }
        HIP_TRY(c, c->chunk_sums.alloc((size_t)c->n_chunks * kCovSums));
        HIP_TRY(c, c->spill.alloc((size_t)c->spill_levels * c->tile_grid_blocks * 256));
        c->tile_ready = true;
        return ICP_HIP_OK;
    }

    // A small shard's block is uploaded by whoever needs it first (enqueued; every caller
    // synchronises the stream before it returns, so h_up is free again by then).
    int flush_queries(icp_hip_ctx* c)
    {
        if (!c->upload_pending)
            return ICP_HIP_OK;
        HIP_TRY(c, hipMemcpyAsync(c->qblock.p, c->h_up, c->small_bytes, hipMemcpyHostToDevice,
                                  c->stream));
        c->upload_pending = false;
        return ICP_HIP_OK;
    }

    // Enqueue one NN + covariance pass over the shard and the reduction of its kCovSums sums into
    // c->sums[0 .. kCovSums) (this shard only; nothing is waited for).  `a.loop` set: the pass of
    // the device-resident loop (`tail` = the loop step that rides on the last reduction).
    // e0 / e1 (may be null) are recorded around the NN kernel.
    int enqueue_nn(icp_hip_ctx* c, NnLaunch a, int tail, hipEvent_t e0, hipEvent_t e1)
    {
        LoopState* loop = a.loop ? c->loop_state : nullptr;
        if (c->n_local == 0)
        {
            // an empty shard (more ranks than points) adds nothing
            HIP_TRY(c, hipMemsetAsync(c->sums.p, 0, sizeof(double) * kCovSums, c->stream));
            return ICP_HIP_OK;
        }
        const bool deep = c->force_deep || !c->tile_ready;
        if (deep)
        {
            a.grid = nn_grid(c->n_local);
            if (e0)
                HIP_TRY(c, hipEventRecord(e0, c->stream));
            HIP_TRY(c, launch_nn(a, c->stream));
            if (e1)
                HIP_TRY(c, hipEventRecord(e1, c->stream));
            HIP_TRY(c, launch_reduce_partials(c->partials.p, a.grid, kCovSums, c->sums.p, c->stream,
                                              loop, tail));
            return ICP_HIP_OK;
        }
        a.grid = c->tile_grid_blocks;
        TileLaunch q{c->chunk_batches, c->batch_queries, c->chunk_sums.p, c->queue.p,
                     c->spill.p, loop ? nullptr : c->stats.p, c->tile_cut};
        const bool park = c->park_now();
        if (park)
        {
            q.park = c->park.p;
            q.late = c->late.p;
            q.late_cnt = c->late_cnt.p;
            q.late_rows = c->late_rows();
            q.park_max = c->park_max();
            q.park_wait = c->park_wait;
        }
        if (q.stats)
            HIP_TRY(c, hipMemsetAsync(q.stats, 0,
                                      sizeof(unsigned long long) * kTileStatsWords * kTileStatsCopies,
                                      c->stream));
        if (e0)
            HIP_TRY(c, hipEventRecord(e0, c->stream));
        HIP_TRY(c, launch_nn_tile(a, q, c->stream));
        if (e1)
            HIP_TRY(c, hipEventRecord(e1, c->stream));
        if (q.stats)
        {
            std::vector<unsigned long long> copies((size_t)kTileStatsWords * kTileStatsCopies);
            HIP_TRY(c, hipMemcpyAsync(copies.data(), q.stats, copies.size() * sizeof(unsigned long long),
                                      hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            unsigned long long h[kTileStatsWords] = {};
            for (int k = 0; k < kTileStatsCopies; ++k)
                for (int w = 0; w < kTileStatsWords; ++w)
                {
                    const unsigned long long v = copies[(size_t)k * kTileStatsWords + w];
                    const bool is_max = (w >= 15 && w <= 18) || w == 22 || w == 23;
                    h[w] = is_max ? std::max(h[w], v) : h[w] + v;
                }
            std::fprintf(stderr,
                         "[tile stats] batches %llu node_iters %llu node_lane_steps %llu "
                         "leaf_phases %llu tile_scans %llu tile_lanes %llu coop_scans %llu "
                         "exact_scans %llu | wave cycles: total %llu head %llu node %llu leaf %llu "
                         "(tile %llu) exact+served %llu tail %llu queue pops %llu | shader clock "
                         "%.0f MHz | longest wave %llu cycles, most batches in a wave %llu\n",
                         h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11],
                         h[12], h[13], h[14], h[21],
                         h[20] ? 100.0 * (double)h[8] / (double)h[20] : 0.0, h[22], h[23]);
            // when the waves ended, how long they lived and what a batch cost: a pass ends with
            // its slowest batch, and the waves that found the queues empty idle meanwhile
            std::fprintf(stderr,
                         "[tile stats] waves started over %.1f us, ended over %.1f us; first start to "
                         "last end %.1f us\n",
                         0.01 * (double)(h[16] - ~h[15]), 0.01 * (double)(h[18] - ~h[17]),
                         0.01 * (double)(h[18] - ~h[15]));
            auto histogram = [&](const char* what, int at, unsigned shift) {
                std::fprintf(stderr, "[tile stats] %s (bins of %u cycles):", what, 1u << shift);
                int last = 63;
                while (last > 0 && h[at + last] == 0)
                    --last;
                for (int b = 0; b <= last; ++b)
                    std::fprintf(stderr, " %llu", h[at + b]);
                std::fprintf(stderr, "\n");
            };
            std::fprintf(stderr,
                         "[tile stats] ring walks %llu over %llu lanes, %llu finished by them, %llu wave cycles\n",
                         h[170], h[171], h[172], h[173]);
            std::fprintf(stderr, "[tile stats] first descent: %llu wave cycles on levels 1-12, %llu below\n",
                         h[174], h[175]);
            histogram("wave lifetimes", 24, 17);
            histogram("batch durations", 88, 13);
            for (int k = 0; k < 2; ++k)
            {
                const unsigned long long* q = h + 152 + 9 * k;
                const double nb = q[0] ? (double)q[0] : 1.0;
                std::fprintf(stderr,
                             "[tile stats] batches %s 90112 cycles: %llu, mean %.0f cycles (node phase "
                             "%.0f, tile scans %.0f, rest of the leaf phases %.0f); per batch %.1f node "
                             "steps, %.2f leaf phases, %.2f tile scans, %.2f single-lane scans\n",
                             k ? "from" : "below", q[0], (double)q[1] / nb, (double)q[2] / nb,
                             (double)q[3] / nb, (double)q[4] / nb, (double)q[5] / nb,
                             (double)q[6] / nb, (double)q[7] / nb, (double)q[8] / nb);
            }
        }
        // chunk sums (-> <= 1024 partial rows) -> the twelve sums, in a fixed order
        if (park)
            HIP_TRY(c, launch_chunk_sums_late(c->chunk_sums.p, c->n_chunks, c->late.p, c->late_cnt.p,
                                              c->late_rows(), c->partials.p, c->sums.p, c->stream,
                                              loop, tail));
        else
            HIP_TRY(c, launch_chunk_sums(c->chunk_sums.p, c->n_chunks, c->partials.p, c->sums.p,
                                         c->stream, loop, tail));
        return ICP_HIP_OK;
    }

    // One NN + covariance pass over the shard, waited for: its kCovSums sums (this shard) in
    // c->h_sums and the kernel time in *ms.
    int nn_pass(icp_hip_ctx* c, NnLaunch a, float* ms)
    {
        *ms = 0.f;
        int rc = enqueue_nn(c, a, kTailNone, c->ev0, c->ev1);
        if (rc != ICP_HIP_OK)
            return rc;
        if ((rc = sums_to_host(c, 0, kCovSums)) != ICP_HIP_OK)
            return rc;
        if (c->n_local)
            *ms = elapsed_ms(c);
        return ICP_HIP_OK;
    }

    // Order the shard by the leaf that the near-only descent of p - mu reaches (stable: inside a
    // leaf the order it had is kept).  Enqueued on the context's stream; nothing is waited for
    // when the sort's temporary storage is at hand.  Uses yx/yy/yz as the destination of the
    // permutation and swaps them with px/py/pz: call it only between iterations.
    int leaf_order_pass(icp_hip_ctx* c, uint32_t n, const float mu[3])
    {
        uint32_t bits = 1;
        while (bits < 32 && (1ull << bits) < c->n_leaves)
            ++bits;
        HIP_TRY(c, c->sk_order.alloc(n));
        HIP_TRY(c, launch_leaf_keys(c->tree, c->px.p, c->py.p, c->pz.p, mu, n, c->sk_keys.p,
                                    c->sk_idx.p, c->stream));
        HIP_TRY(c, sort_pairs_u32(c->sk_keys.p, c->sk_keys2.p, c->sk_idx.p, c->sk_order.p, n, bits,
                                  c->stream, c->sk_temp.p, c->sk_temp.count));
        // the matched-point arrays are free between iterations: permute into them and swap
        HIP_TRY(c, launch_permute_soa(c->px.p, c->py.p, c->pz.p, c->perm.p, c->sk_order.p, n,
                                      c->yx.p, c->yy.p, c->yz.p, c->sk_idx.p, c->stream));
        std::swap(c->px, c->yx);
        std::swap(c->py, c->yy);
        std::swap(c->pz, c->yz);
        std::swap(c->perm, c->sk_idx);
        return ICP_HIP_OK;
    }

    // Store the shard in Morton order of its positions (sort.hip), then in leaf order; c->aos
    // holds the input AoS.  The key / index / temporary arrays stay with the context: a long
    // registration orders the shard a second time once the clouds are aligned (maybe_resort).
    int sort_queries(icp_hip_ctx* c, uint32_t n, uint64_t n_total)
    {
        const uint32_t grid = stream_grid(n);
        DevBuf<float> mm;
        struct Cleanup
        {
            DevBuf<float>& a;
            hipStream_t s;
            ~Cleanup()
            {
                (void)hipStreamSynchronize(s); // (error paths) nothing in flight uses a pooled block
                a.release();
            }
        } cleanup{mm, c->stream};
        HIP_TRY(c, mm.alloc((size_t)grid * 6));
        HIP_TRY(c, c->sk_keys.alloc(n));
        HIP_TRY(c, c->sk_keys2.alloc(n));
        HIP_TRY(c, c->sk_idx.alloc(n));
        HIP_TRY(c, c->perm.alloc(n));
        {
            size_t b30 = 0, b32 = 0;
            HIP_TRY(c, sort_pairs_u32_temp_bytes(n, 30u, &b30));
            HIP_TRY(c, sort_pairs_u32_temp_bytes(n, 32u, &b32));
            HIP_TRY(c, c->sk_temp.alloc(std::max<size_t>(std::max(b30, b32), 256)));
        }
        HIP_TRY(c, launch_minmax(c->aos.p, n, grid, mm.p, c->stream));
        std::vector<float> h((size_t)grid * 6);
        HIP_TRY(c, hipMemcpyAsync(h.data(), mm.p, h.size() * sizeof(float), hipMemcpyDeviceToHost,
                                  c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        float lo[3], hi[3], scale[3];
        for (int k = 0; k < 3; ++k)
        {
            lo[k] = h[k];
            hi[k] = h[3 + k];
        }
        for (uint32_t b = 1; b < grid; ++b)
            for (int k = 0; k < 3; ++k)
            {
                lo[k] = std::fmin(lo[k], h[b * 6 + k]);
                hi[k] = std::fmax(hi[k], h[b * 6 + 3 + k]);
            }
        // one cubic cell size for all axes keeps the Z-curve isotropic.  The extent is taken in
        // double: finite clouds whose extent overflows fp32 are legal input (the min/max kernel
        // turns a NaN or an infinity into an infinite bound, which is what gets rejected here,
        // like icp_host::all_finite rejects it for the clouds that are not sorted)
        bool finite = true;
        double ext = 0.0;
        for (int k = 0; k < 3; ++k)
        {
            finite = finite && std::isfinite(lo[k]) && std::isfinite(hi[k]);
            ext = std::fmax(ext, (double)hi[k] - (double)lo[k]);
        }
        if (!finite)
            return fail(c, ICP_HIP_ERR_ARG,
                        std::string("icp_hip_set_queries: ") + icp_host::kNonFiniteMessage);
        c->q_extent = ext;
        for (int k = 0; k < 3; ++k)
            scale[k] = ext > 0.0 ? (float)(1024.0 / ext) : 0.f;
        HIP_TRY(c, launch_morton(c->aos.p, n, lo, scale, c->sk_keys.p, c->sk_idx.p, c->stream));
        HIP_TRY(c, sort_pairs_u32(c->sk_keys.p, c->sk_keys2.p, c->sk_idx.p, c->perm.p, n, 30u,
                                  c->stream, c->sk_temp.p, c->sk_temp.count));
        HIP_TRY(c, launch_gather_soa(c->aos.p, c->perm.p, n, c->px.p, c->py.p, c->pz.p, c->stream));
        c->sorted = true;
        const char* env = std::getenv("ICP_HIP_LEAF_ORDER");
        c->leaf_ordered = c->has_tree && c->tree.root >= 0 && !(env && env[0] == '0');
        if (c->leaf_ordered)
        {
            // second stable sort: leaf reached by the near-only descent of p - mean(p of this shard)
            HIP_TRY(c, launch_sum_points(c->px.p, c->py.p, c->pz.p, n, grid, c->partials.p,
                                         c->stream));
            HIP_TRY(c, launch_reduce_partials(c->partials.p, grid, kApplySums, c->sums.p, c->stream));
            HIP_TRY(c, hipMemcpyAsync(c->h_sums, c->sums.p, sizeof(double) * kApplySums,
                                      hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            float mu[3] = {(float)(c->h_sums[1] / n), (float)(c->h_sums[2] / n),
                           (float)(c->h_sums[3] / n)};
            // a shard of a larger cloud: the loop will search around the whole cloud's mean,
            // which this context has seen if it registered a shard of a cloud of this size
            // before (recentre_order).  Only a hint: any centre gives the same answers, and a
            // stale one is corrected after the run's first all-reduce.
            if (n_total > n && c->mean_hint_n_total == n_total)
                std::memcpy(mu, c->mean_hint, sizeof mu);
            int rc = leaf_order_pass(c, n, mu);
            if (rc != ICP_HIP_OK)
                return rc;
            std::memcpy(c->order_mu, mu, sizeof mu);
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return ICP_HIP_OK;
    }

    // A sharded run's first all-reduce gives mean(p) of the WHOLE cloud, which is what the loop
    // subtracts; the shard was put in the leaf order of p - mean(p of this shard) at set-up.  For
    // a random subset the two means differ by a few per cent of the point spacing; for a
    // spatially compact shard (a caller that partitions a scan by position) by a good part of
    // the cloud, and the set-up order then follows leaves the searches never visit (an eighth of
    // the 64M cloud cut by median planes: first NN pass 1.3x, aligned pass 2.3x its share).
    // Then the shard is ordered again around the mean the loop uses.  `sums` = the kApplySums
    // all-reduced first sums on the host.  Rank-local: no collective, and nothing that shapes
    // the enqueue blocks depends on it.  ICP_HIP_RECENTRE=0 switches it off (experiments).
    constexpr uint64_t kRecentreMinPoints = 1u << 17;
    int recentre_order(icp_hip_ctx* c, const double* sums)
    {
        const char* env = std::getenv("ICP_HIP_RECENTRE");
        if ((env && env[0] == '0') || !c->sorted || !c->leaf_ordered ||
            c->n_local < kRecentreMinPoints || c->n_total == 0 || c->n_tree_points == 0)
            return ICP_HIP_OK;
        float mu[3];
        double d2 = 0.0;
        for (int k = 0; k < 3; ++k)
        {
            mu[k] = (float)(sums[1 + k] / (double)c->n_total); // as loop_init_step forms it
            const double d = (double)mu[k] - (double)c->order_mu[k];
            d2 += d * d;
        }
        std::memcpy(c->mean_hint, mu, sizeof mu); // for the next shard of a cloud of this size
        c->mean_hint_n_total = c->n_total;
        const double spacing = 2.0 * (double)c->root_radius / std::cbrt((double)c->n_tree_points);
        const bool again = std::sqrt(d2) > 0.25 * spacing;
        if (env && env[0] == 'v') // verbose: what the decision was taken from
            std::fprintf(stderr, "[recentre] means %.3g apart, point spacing %.3g: %s\n",
                         std::sqrt(d2), spacing, again ? "shard ordered again" : "order kept");
        if (!again)
            return ICP_HIP_OK;
        std::memcpy(c->order_mu, mu, sizeof mu);
        return leaf_order_pass(c, (uint32_t)c->n_local, mu);
    }

    // A second ordering pass once the clouds are aligned.  The set-up order is computed for the
    // pose the cloud arrives in; when the registration has converged the batches of that order
    // end in 3.3 buckets instead of 1.2 (64M points, 0.25 degree: NN pass 5.2 ms against 4.3 ms
    // re-ordered, 3 ms for the pass itself).  Decided on the host where it reads the loop state
    // anyway -- after iterations 1, 3, 7, 15, 23, .. (the ends of the enqueue blocks), in both
    // loops, from state bits that are the same in both -- and at most once per registration:
    // when the last iteration applied moved no point by more than 5 % of the point spacing and
    // at least four more iterations may follow.  Which order the shard is stored in never
    // changes an answer; it changes the order of the fp64 sums, like any other change of the
    // shard's order.  With an RCCL communicator every rank reads the same state after the
    // all-reduce and takes the same decision about its own shard (the pass involves no other
    // rank); there the last iteration applied is one ahead of the last one finished, which is
    // why the test is on `started`.
    constexpr uint64_t kResortMinPoints = 1u << 19;
    constexpr uint32_t kAlignedChunkBatches = 8; // batches per chunk after the second ordering pass
    // While the pass is still to come on a very large shard (an iteration takes milliseconds, a
    // read-back 25 us) the enqueue blocks stay at 2 iterations, so that the decision is taken
    // within two iterations of the clouds being aligned: block ends 1, 3, 5, 7, ..  (From 6M
    // points: 8M points, one rank's share of the 64M cloud on 8 GPUs, 0.760 -> 0.743 ms per
    // iteration; neutral at 4M.)
    constexpr uint64_t kResortFineBlocksFrom = 6u << 20;
    // With an RCCL communicator every rank must enqueue the same number of slots (each holds a
    // collective) although it reads `done` only at its block ends: the block sizes, and with
    // them the decision about the pass, may then depend on nothing a single rank sees -- not on
    // its shard's size, extent or order.  There the decision is taken from the all-reduced loop
    // state, the tree (the same on every rank) and n_total / n_ranks; whether a rank then runs
    // the pass on its own shard is its own business.
    // (One rank sees everything: its communicator run stays bit-identical to the plain run.
    // ICP_HIP_RESORT=g applies the many-rank rule to a communicator of one rank, for the tests.)
    bool resort_rule_global(const icp_hip_ctx* c)
    {
        const char* env = std::getenv("ICP_HIP_RESORT");
        return c->comm && (c->n_ranks > 1 || (env && env[0] == 'g'));
    }
    bool resort_fine_blocks(const icp_hip_ctx* c)
    {
        const char* env = std::getenv("ICP_HIP_RESORT");
        if (env && env[0] == '0')
            return false;
        if (resort_rule_global(c))
            return !c->resort_decided &&
                   c->n_total / (uint64_t)(c->n_ranks > 0 ? c->n_ranks : 1) >= kResortFineBlocksFrom;
        return !c->resorted && c->sorted && c->leaf_ordered && !c->allreduce &&
               c->n_local >= kResortFineBlocksFrom;
    }
    uint32_t next_block(const icp_hip_ctx* c, uint32_t block)
    {
        return std::min(block * 2, resort_fine_blocks(c) ? 2u : (uint32_t)kLoopBlockMax);
    }
    bool at_block_end(const icp_hip_ctx* c, uint32_t it) // only asked while the pass is to come
    {
        uint32_t end = 0, block = 1;
        while (end < it)
        {
            end += block;
            block = next_block(c, block);
        }
        return end == it;
    }
    int maybe_resort(icp_hip_ctx* c, const LoopState& st)
    {
        {
            const char* env = std::getenv("ICP_HIP_RESORT");
            if (env && env[0] == '0')
                return ICP_HIP_OK;
        }
        const bool sharded = resort_rule_global(c);
        // what this rank's shard allows (never part of a decision that shapes the blocks)
        const bool can = c->sorted && c->leaf_ordered && c->n_local >= kResortMinPoints &&
                         c->q_extent > 0.0;
        if (sharded ? c->resort_decided : (c->resorted || !can))
            return ICP_HIP_OK;
        if (st.done || st.started == 0 || st.max_iters - st.started < 4u ||
            !at_block_end(c, st.started))
            return ICP_HIP_OK;
        // the scale the motion is compared with: the cloud's extent -- the shard's own when the
        // shard is the cloud, else the diameter of the tree's root sphere (a median distance:
        // about the extent of the reference cloud, the same on every rank)
        const double extent =
            sharded || c->n_total > c->n_local ? 2.0 * (double)c->root_radius : c->q_extent;
        if (!(extent > 0.0))
            return ICP_HIP_OK;
        // the last iteration moved x to R x + t: by at most |R - I|_F |x - mu_m| + |R mu_m + t - mu_m|
        const float* T = st.Tn;
        double fro = 0.0, v2 = 0.0;
        for (int r = 0; r < 3; ++r)
        {
            double v = (double)T[4 * r + 3] - (double)st.mu_m[r];
            for (int k = 0; k < 3; ++k)
            {
                const double d = (double)T[4 * r + k] - (r == k ? 1.0 : 0.0);
                fro += d * d;
                v += (double)T[4 * r + k] * (double)st.mu_m[k];
            }
            v2 += v * v;
        }
        const double radius = 0.5 * std::sqrt(3.0) * extent;
        const double moved = std::sqrt(fro) * radius + std::sqrt(v2);
        const double spacing = extent / std::cbrt((double)c->n_tree_points);
        if (const char* env = std::getenv("ICP_HIP_RESORT"))
            if (env[0] == 'v') // verbose: what the decision was taken from
                std::fprintf(stderr, "[resort] after iteration %u: moved <= %.3g, point spacing %.3g\n",
                             st.started, moved, spacing);
        if (!(moved < 0.05 * spacing))
            return ICP_HIP_OK;
        c->resort_decided = true;
        if (!can || c->resorted)
            return ICP_HIP_OK;
        c->resorted = true;
        if (const char* env = std::getenv("ICP_HIP_RESORT"))
            if (env[0] == 'v')
                std::fprintf(stderr, "[resort] second ordering pass after iteration %u\n", st.started);
        // From here on the clouds are aligned: larger chunks cost the aligned pass nothing (it
        // no longer depends on two batches of a bucket meeting in L2: 2.42 -> 2.34 ms at 64M
        // points with 8 batches per chunk) -- the misaligned passes before did need the small
        // ones (14.1 -> 14.8 ms).  Fewer chunks, so the per-chunk sums still fit their buffer.
        if (c->chunk_batches == 2 && !std::getenv("ICP_HIP_CHUNK_BATCHES"))
        {
            const uint64_t batches = (c->n_local + c->batch_queries - 1) / c->batch_queries;
            c->chunk_batches = kAlignedChunkBatches;
            c->n_chunks = (uint32_t)((batches + c->chunk_batches - 1) / c->chunk_batches);
        }
        return leaf_order_pass(c, (uint32_t)c->n_local, st.mu_q);
    }

    // Put a tree on the device: node records + leaf table from host arrays; the bucket slots are
    // either uploaded from the host arrays of `t` or, when d_sx is not null, adopted from device
    // arrays that tree_build.hip filled (n_slots + kSlotSlack floats, +inf behind n_slots).
    int install_tree(icp_hip_ctx* c, const icp_flat_tree* t, float* d_sx, float* d_sy, float* d_sz,
                     uint32_t* d_srow)
    {
        if ((int)t->height > nn_max_depth_supported())
            return fail(c, ICP_HIP_ERR_DEPTH,
                        "icp_hip_set_tree: tree height " + std::to_string(t->height) +
                            " exceeds the traversal stack (" +
                            std::to_string(nn_max_depth_supported()) + " levels)");
        drain_noalloc(c);
        if (c->has_tree) // a replaced tree's blocks go back to the pool: nothing may still use them
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->has_tree = false;
        const size_t nn = t->n_nodes, nl = t->n_leaves, ns = t->n_slots;
        const size_t slack = kSlotSlack; // bucket windows / chunks are loaded whole
        c->n_leaves = (uint32_t)nl;
        c->root_radius = nn ? t->node_radius[0] : 0.f;

        // A small tree is packed on the host into one block and uploaded with ONE copy: the ten
        // copies, the staging arrays and the packing kernel below cost more than a whole
        // registration of a few thousand points.
        if (!d_sx && ns + nn <= (1u << 17))
        {
            auto up8 = [](size_t v) { return (v + 255) & ~(size_t)255; };
            const size_t o_nodes = 0, o_leaf = o_nodes + up8((nn ? nn : 1) * sizeof(DevNode)),
                         o_sx = o_leaf + up8(nl * sizeof(uint2)),
                         o_sy = o_sx + up8((ns + slack) * 4), o_sz = o_sy + up8((ns + slack) * 4),
                         o_srow = o_sz + up8((ns + slack) * 4), o_nrow = o_srow + up8(ns * 4),
                         o_ring = o_nrow + up8((nn ? nn : 1) * 4),
                         total = o_ring + up8((ns + slack) * 4);
            // (a walk needs 16 lanes on one bucket: clouds of a few thousand points run in
            // batches too narrow for that, and ranking their buckets here would only cost time)
            const bool with_ring = nn > 0 && t->max_leaf_count <= 256 && t->n_points >= 16384;
            std::vector<unsigned char>& h = c->tree_blob_host;
            try
            {
                h.assign(total, 0);
            }
            catch (const std::bad_alloc&)
            {
                return fail(c, ICP_HIP_ERR_NOMEM, "icp_hip_set_tree: out of host memory");
            }
            DevNode* hn = reinterpret_cast<DevNode*>(h.data() + o_nodes);
            for (size_t i = 0; i < nn; ++i)
            {
                DevNode N;
                N.cx = t->node_cx[i];
                N.cy = t->node_cy[i];
                N.cz = t->node_cz[i];
                N.r = t->node_radius[i];
                N.inside = t->node_inside[i];
                N.outside = t->node_outside[i];
                N.parent = t->node_parent[i];
                // thr = min { s : sqrtf(s) >= r } as pack_nodes_kernel finds it (sqrtf is correctly
                // rounded on both sides)
                float th = 0.f;
                if (N.r > 0.f)
                {
                    th = N.r * N.r;
                    while (th > 0.f && std::sqrt(std::nextafter(th, 0.f)) >= N.r)
                        th = std::nextafter(th, 0.f);
                    while (std::sqrt(th) < N.r)
                        th = std::nextafter(th, std::numeric_limits<float>::infinity());
                }
                N.thr = th;
                hn[i] = N;
            }
            uint2* hl = reinterpret_cast<uint2*>(h.data() + o_leaf);
            for (size_t l = 0; l < nl; ++l)
                hl[l] = make_uint2(t->leaf_begin[l], t->leaf_count[l]);
            auto put_slots = [&](size_t off, const float* src) {
                float* d = reinterpret_cast<float*>(h.data() + off);
                std::memcpy(d, src, ns * sizeof(float));
                for (size_t k = 0; k < slack; ++k)
                    d[ns + k] = std::numeric_limits<float>::infinity();
            };
            put_slots(o_sx, t->slot_x);
            put_slots(o_sy, t->slot_y);
            put_slots(o_sz, t->slot_z);
            std::memcpy(h.data() + o_srow, t->slot_row, ns * sizeof(uint32_t));
            if (nn)
                std::memcpy(h.data() + o_nrow, t->node_row, nn * sizeof(uint32_t));
            if (with_ring)
            {
                // the ring order of the buckets (kernels.hh, DevTree::ring), as ring_build_kernel makes it
                uint32_t* hr = reinterpret_cast<uint32_t*>(h.data() + o_ring);
                std::vector<uint32_t> keys;
                auto bucket = [&](int32_t code, size_t parent) {
                    if (code >= 0)
                        return;
                    const size_t l = (size_t)~code;
                    const uint32_t beg = t->leaf_begin[l], cnt = t->leaf_count[l];
                    keys.resize(cnt);
                    for (uint32_t j = 0; j < cnt; ++j)
                    {
                        const float dx = t->slot_x[beg + j] - t->node_cx[parent],
                                    dy = t->slot_y[beg + j] - t->node_cy[parent],
                                    dz = t->slot_z[beg + j] - t->node_cz[parent];
                        const float rho = std::sqrt(dx * dx + dy * dy + dz * dz);
                        uint32_t bits;
                        std::memcpy(&bits, &rho, 4);
                        keys[j] = (bits & 0xFFFFFF00u) | j;
                    }
                    std::sort(keys.begin(), keys.end());
                    const uint32_t padded = (cnt + 15u) & ~15u;
                    for (uint32_t j = 0; j < padded; ++j)
                        hr[beg + j] = j < cnt ? keys[j] : 0x7f800000u;
                };
                for (size_t i = 0; i < nn; ++i)
                {
                    bucket(t->node_inside[i], i);
                    bucket(t->node_outside[i], i);
                }
            }
            HIP_TRY(c, c->tree_blob.alloc(total));
            HIP_TRY(c, hipMemcpyAsync(c->tree_blob.p, h.data(), total, hipMemcpyHostToDevice,
                                      c->stream));
            c->nodes.release();
            c->leaf_info.release();
            c->sx.release();
            c->sy.release();
            c->sz.release();
            c->srow.release();
            c->nrow.release();
            c->ring.release();
            unsigned char* b = c->tree_blob.p;
            c->tree.nodes = reinterpret_cast<const DevNode*>(b + o_nodes);
            c->tree.leaf_info = reinterpret_cast<const uint2*>(b + o_leaf);
            c->tree.sx = reinterpret_cast<const float*>(b + o_sx);
            c->tree.sy = reinterpret_cast<const float*>(b + o_sy);
            c->tree.sz = reinterpret_cast<const float*>(b + o_sz);
            c->tree.srow = reinterpret_cast<const uint32_t*>(b + o_srow);
            c->tree.nrow = reinterpret_cast<const uint32_t*>(b + o_nrow);
            c->tree.ring = with_ring ? reinterpret_cast<const uint32_t*>(b + o_ring) : nullptr;
            c->tree.n_slots = (uint32_t)ns;
            c->tree.n_nodes = (uint32_t)nn;
            c->tree.root = t->root;
            c->height = t->height;
            c->max_leaf = t->max_leaf_count;
            c->capacity = t->capacity;
            c->n_tree_points = t->n_points;
            c->has_tree = true;
            return prepare_tile(c);
        }
        c->tree_blob.release();

        // staging for the SoA node arrays, packed into 32-byte records on the device
        DevBuf<float> f4[4];
        DevBuf<int32_t> i3[3];
        auto cleanup = [&]() {
            (void)hipStreamSynchronize(c->stream);
            for (auto& b : f4)
                b.release();
            for (auto& b : i3)
                b.release();
        };
        hipError_t e = hipSuccess;
        auto up = [&](auto& buf, const auto* src, size_t n) -> bool {
            if ((e = buf.alloc(n)) != hipSuccess)
                return false;
            if (n == 0)
                return true;
            e = hipMemcpyAsync(buf.p, src, n * sizeof(*src), hipMemcpyHostToDevice, c->stream);
            return e == hipSuccess;
        };
        bool ok = c->nodes.alloc(nn ? nn : 1) == hipSuccess;
        ok = ok && up(f4[0], t->node_cx, nn) && up(f4[1], t->node_cy, nn) &&
             up(f4[2], t->node_cz, nn) && up(f4[3], t->node_radius, nn) &&
             up(i3[0], t->node_inside, nn) && up(i3[1], t->node_outside, nn) &&
             up(i3[2], t->node_parent, nn) && up(c->nrow, t->node_row, nn);
        if (ok && nn)
            ok = (e = launch_pack_nodes(f4[0].p, f4[1].p, f4[2].p, f4[3].p, i3[0].p, i3[1].p,
                                        i3[2].p, (uint32_t)nn, c->nodes.p, c->stream)) ==
                 hipSuccess;
        // leaves {begin,count}
        std::vector<uint2> li(nl);
        for (size_t l = 0; l < nl; ++l)
            li[l] = make_uint2(t->leaf_begin[l], t->leaf_count[l]);
        ok = ok && up(c->leaf_info, li.data(), nl);
        // bucket slots (+slack filled with +inf so stray lanes compute inf, never NaN)
        auto up_slots = [&](DevBuf<float>& buf, const float* src) -> bool {
            if ((e = buf.alloc(ns + slack)) != hipSuccess)
                return false;
            if ((e = hipMemcpyAsync(buf.p, src, ns * sizeof(float), hipMemcpyHostToDevice,
                                    c->stream)) != hipSuccess)
                return false;
            static const std::vector<float> inf(slack, std::numeric_limits<float>::infinity());
            e = hipMemcpyAsync(buf.p + ns, inf.data(), slack * sizeof(float),
                               hipMemcpyHostToDevice, c->stream);
            return e == hipSuccess;
        };
        if (d_sx)
        {
            c->sx.adopt(d_sx, ns + slack);
            c->sy.adopt(d_sy, ns + slack);
            c->sz.adopt(d_sz, ns + slack);
            c->srow.adopt(d_srow, ns);
        }
        else
            ok = ok && up_slots(c->sx, t->slot_x) && up_slots(c->sy, t->slot_y) &&
                 up_slots(c->sz, t->slot_z) && up(c->srow, t->slot_row, ns);
        c->tree.nodes = c->nodes.p;
        c->tree.leaf_info = c->leaf_info.p;
        c->tree.sx = c->sx.p;
        c->tree.sy = c->sy.p;
        c->tree.sz = c->sz.p;
        c->tree.srow = c->srow.p;
        c->tree.nrow = c->nrow.p;
        c->tree.ring = nullptr;
        c->tree.n_slots = (uint32_t)ns;
        c->tree.n_nodes = (uint32_t)nn;
        // the ring order of the buckets (kernels.hh, DevTree::ring)
        DevBuf<int32_t> leaf_parent;
        c->ring.release();
        if (ok && nn > 0 && t->max_leaf_count <= 256)
        {
            ok = (e = c->ring.alloc(ns + slack)) == hipSuccess &&
                 (e = leaf_parent.alloc(nl)) == hipSuccess &&
                 (e = hipMemsetAsync(c->ring.p + ns, 0x7f, slack * sizeof(uint32_t), c->stream)) == hipSuccess &&
                 (e = launch_ring_build(c->tree, (uint32_t)nl, leaf_parent.p, c->ring.p, c->stream)) ==
                     hipSuccess;
            if (ok)
                c->tree.ring = c->ring.p;
        }
        if (ok)
            ok = (e = hipStreamSynchronize(c->stream)) == hipSuccess;
        leaf_parent.release();
        cleanup();
        if (!ok)
        {
            c->tree.ring = nullptr;
            return hip_fail(c, e, "icp_hip_set_tree upload");
        }

        c->tree.n_nodes = (uint32_t)nn;
        c->tree.root = t->root;
        c->height = t->height;
        c->max_leaf = t->max_leaf_count;
        c->capacity = t->capacity;
        c->n_tree_points = t->n_points;
        c->has_tree = true;
        return prepare_tile(c);
    }

    // One slot of the device-resident loop, enqueued without waiting for anything.
    //   one GPU:  NN + covariance, its sums (+ solve), apply + error, its sums (+ loop condition)
    //   RCCL:     NN + covariance, its sums, ONE all-reduce of the kIterSums doubles (this pass'
    //             sums and the apply pass' sums of the iteration before), advance (loop condition
    //             of the iteration before, then this one's solve), apply + error, its sums
    int enqueue_slot(icp_hip_ctx* c, hipEvent_t* ev)
    {
        const uint32_t n = (uint32_t)c->n_local;
        const float zero[3] = {0.f, 0.f, 0.f};
        const bool sharded = c->comm != nullptr;
        NnLaunch a = make_nn(c, zero, true, false, false, true);
        a.loop = c->loop_state;
        int rc = enqueue_nn(c, a, sharded ? kTailNone : kTailSolve, ev ? ev[0] : nullptr,
                            ev ? ev[1] : nullptr);
        if (rc != ICP_HIP_OK)
            return rc;
        if (sharded)
        {
            if ((rc = rccl_allreduce(c, 0, kIterSums)) != ICP_HIP_OK)
                return rc;
            HIP_TRY(c, launch_loop_advance(c->loop_state, c->sums.p, c->stream));
        }
        if (n == 0)
        {
            HIP_TRY(c, hipMemsetAsync(c->sums.p + kApplyAt, 0, sizeof(double) * kApplySums,
                                      c->stream));
            return ICP_HIP_OK;
        }
        const uint32_t g_st = c->small ? 1u : stream_grid(n);
        const int tail = sharded ? kTailNone : kTailFinish;
        Mat34 unused{};
        if (ev)
            HIP_TRY(c, hipEventRecord(ev[2], c->stream));
        // (a small shard: one workgroup, which also sums and runs the loop step)
        HIP_TRY(c, launch_apply_error(c->px.p, c->py.p, c->pz.p, c->yx.p, c->yy.p, c->yz.p, unused,
                                      zero, n, g_st, c->partials.p, c->loop_state, c->stream,
                                      c->small ? c->sums.p + kApplyAt : nullptr, tail));
        if (ev)
            HIP_TRY(c, hipEventRecord(ev[3], c->stream));
        if (!c->small)
            HIP_TRY(c, launch_reduce_partials(c->partials.p, g_st, kApplySums,
                                              c->sums.p + kApplyAt, c->stream, c->loop_state, tail));
        return ICP_HIP_OK;
    }

    // The centroid sums of the cloud as uploaded -> c->sums[kApplyAt ..) (this shard; enqueued)
    int enqueue_first_sums(icp_hip_ctx* c)
    {
        const uint32_t n = (uint32_t)c->n_local;
        if (n == 0)
        {
            HIP_TRY(c, hipMemsetAsync(c->sums.p + kApplyAt, 0, sizeof(double) * kApplySums,
                                      c->stream));
            return ICP_HIP_OK;
        }
        if (c->small && c->first_sums_valid)
        {
            // summed on the host while the block was packed
            for (int k = 0; k < kApplySums; ++k)
                c->h_sums[kApplyAt + k] = c->first_sums[k];
            HIP_TRY(c, hipMemcpyAsync(c->sums.p + kApplyAt, c->h_sums + kApplyAt,
                                      sizeof(double) * kApplySums, hipMemcpyHostToDevice,
                                      c->stream));
            return ICP_HIP_OK;
        }
        const uint32_t g_st = stream_grid(n);
        HIP_TRY(c, launch_sum_points(c->px.p, c->py.p, c->pz.p, n, g_st, c->partials.p, c->stream));
        HIP_TRY(c, launch_reduce_partials(c->partials.p, g_st, kApplySums, c->sums.p + kApplyAt,
                                          c->stream));
        return ICP_HIP_OK;
    }

    void loop_results(icp_hip_ctx* c, const LoopState& st, float T_out[16], uint64_t* iters_done,
                      float* final_error)
    {
        std::memcpy(T_out, st.T, sizeof st.T);
        if (iters_done)
            *iters_done = st.it;
        if (final_error)
            *final_error = st.error;
        c->timing.nodes_visited += st.work[0];
        c->timing.points_scanned += st.work[1];
        c->timing.stack_spills += st.work[2];
    }

    // libcpu::icp's loop (icp.cc:110-125) with the host out of the iteration: slots are enqueued
    // in blocks of 1, 2, 4, 8, 8, .. and the state is read back once per block; what was enqueued
    // past the end of the loop finds `done` set and does nothing.  Sharded over RCCL ranks the
    // loop condition of an iteration is known one slot later (it rides on the next all-reduce),
    // so a run of k iterations takes k + 1 slots; the NN pass of the extra slot is skipped on the
    // device when k is the iteration cap and is discarded otherwise.  Every rank reads the same
    // state bits after the all-reduce, hence enqueues the same slots and the same collectives.
    int run_device_loop(icp_hip_ctx* c, const float mu_m[3], uint64_t max_iters, float threshold,
                        float T_out[16], uint64_t* iters_done, float* final_error,
                        icp_hip_progress_fn progress, void* user)
    {
        const auto t_begin = std::chrono::steady_clock::now();
        c->nn_trace.clear();
        const uint32_t cap_iters = max_iters > 0xFFFFFFFEull ? 0xFFFFFFFEu : (uint32_t)max_iters;
        const bool sharded = c->comm != nullptr;
        int rc;
        // first mean(p) (icp.cc:114); T = I, error = +inf (:101-104)
        if (c->small && c->first_sums_valid && !sharded)
        {
            // the same step on the host, into the image of the block: it goes up with the cloud
            LoopState* hs = reinterpret_cast<LoopState*>(c->h_up);
            loop_init_step(hs, c->first_sums, mu_m, threshold, cap_iters, (double)c->n_total, false);
            if (c->upload_pending)
                rc = flush_queries(c);
            else
                rc = hipMemcpyAsync(c->loop_state, hs, sizeof(LoopState), hipMemcpyHostToDevice,
                                    c->stream) == hipSuccess
                         ? ICP_HIP_OK
                         : fail(c, ICP_HIP_ERR_HIP, "icp_hip_run: upload of the loop state failed");
            if (rc != ICP_HIP_OK)
                return rc;
        }
        else
        {
            if ((rc = flush_queries(c)) != ICP_HIP_OK || (rc = enqueue_first_sums(c)) != ICP_HIP_OK)
                return rc;
            if (sharded && (rc = rccl_allreduce(c, kApplyAt, kApplySums)) != ICP_HIP_OK)
                return rc;
            if (sharded && c->n_total > c->n_local && c->n_local >= kRecentreMinPoints &&
                ((rc = sums_to_host(c, kApplyAt, kApplySums)) != ICP_HIP_OK ||
                 (rc = recentre_order(c, c->h_sums + kApplyAt)) != ICP_HIP_OK))
                return rc;
            HIP_TRY(c, launch_loop_init(c->loop_state, c->sums.p + kApplyAt, mu_m, threshold,
                                        cap_iters, (double)c->n_total, sharded, c->stream));
        }
        c->first_sums_valid = false; // the loop moves the cloud
        const uint32_t slots_cap = cap_iters == 0 ? 0u : cap_iters + (sharded ? 1u : 0u);
        uint32_t slots = 0, it_done = 0, block = 1;
        std::vector<float> slot_nn, slot_ap; // kernel times by slot (slot i holds iteration i)
        bool done = cap_iters == 0 || !(std::numeric_limits<float>::infinity() > threshold);
        while (!done && slots < slots_cap)
        {
            const uint32_t B = std::min(block, slots_cap - slots);
            while (c->phase_timers && c->loop_events.size() < (size_t)4 * B)
            {
                hipEvent_t ev = nullptr;
                HIP_TRY(c, hipEventCreate(&ev));
                c->loop_events.push_back(ev);
            }
            for (uint32_t s = 0; s < B; ++s)
                if ((rc = enqueue_slot(c, c->phase_timers ? c->loop_events.data() + 4 * s
                                                          : nullptr)) != ICP_HIP_OK)
                    return rc;
            // (a small shard: the whole block, i.e. the state and the cloud as moved so far)
            HIP_TRY(c, hipMemcpyAsync(c->h_loop, c->loop_state,
                                      c->small ? c->small_bytes : sizeof(LoopState),
                                      hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            c->host_points_valid = c->small;
            const LoopState& st = *c->h_loop;
            for (uint32_t s = 0; s < B; ++s)
            {
                float nn_ms = 0.f, ap_ms = 0.f;
                // a slot past the end of the loop still records its events: nothing ran between
                if (c->phase_timers && c->n_local)
                {
                    hipEvent_t* ev = c->loop_events.data() + 4 * s;
                    (void)hipEventElapsedTime(&nn_ms, ev[0], ev[1]);
                    (void)hipEventElapsedTime(&ap_ms, ev[2], ev[3]);
                }
                slot_nn.push_back(nn_ms);
                slot_ap.push_back(ap_ms);
            }
            slots += B;
            for (uint32_t i = it_done; i < st.it; ++i)
            {
                c->timing.nn_ms += slot_nn[i];
                c->timing.apply_ms += slot_ap[i];
                c->timing.nn_launches += 1;
                c->timing.queries += (double)c->n_total;
                if (c->phase_timers && c->nn_trace.size() < 4096)
                    c->nn_trace.push_back(slot_nn[i]);
                if (progress)
                    progress(i + 1, max_iters, st.errors[i % (uint32_t)kLoopBlockMax], user);
            }
            it_done = st.it;
            done = st.done != 0;
            if (!done && (rc = maybe_resort(c, st)) != ICP_HIP_OK)
                return rc;
            block = next_block(c, block);
        }
        if (slots == 0) // no iteration at all: read the initial state back
        {
            HIP_TRY(c, hipMemcpyAsync(c->h_loop, c->loop_state, sizeof(LoopState),
                                      hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
        loop_results(c, *c->h_loop, T_out, iters_done, final_error);
        c->timing.host_ms +=
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin)
                .count();
        return ICP_HIP_OK;
    }

    // The same loop driven from the host: every step's sums come back to the host, which runs the
    // same scalar steps (loop_steps.hh) between the kernels.  This is the path of an all-reduce
    // callback (icp_hip_set_allreduce), of the diagnostic counters and of the level-stack kernel.
    // With a callback the order is the sharded one (one all-reduce of kIterSums doubles per
    // iteration, the loop condition one slot late); both loops give the same bits.
    int run_host_loop(icp_hip_ctx* c, const float mu_m[3], uint64_t max_iters, float threshold,
                      float T_out[16], uint64_t* iters_done, float* final_error,
                      icp_hip_progress_fn progress, void* user)
    {
        const auto t_begin = std::chrono::steady_clock::now();
        const uint32_t n = (uint32_t)c->n_local;
        const uint32_t g_st = stream_grid(n);
        const uint32_t cap_iters = max_iters > 0xFFFFFFFEull ? 0xFFFFFFFEu : (uint32_t)max_iters;
        const bool sharded = c->allreduce != nullptr;
        c->nn_trace.clear();
        LoopState st{};
        int rc;
        c->host_points_valid = false;
        // first mean(p) (icp.cc:114); T = I, error = +inf (:101-104)
        if ((rc = flush_queries(c)) != ICP_HIP_OK || (rc = enqueue_first_sums(c)) != ICP_HIP_OK ||
            (rc = sums_to_host(c, kApplyAt, kApplySums)) != ICP_HIP_OK ||
            (rc = host_allreduce(c, kApplyAt, kApplySums)) != ICP_HIP_OK)
            return rc;
        if (sharded && c->n_total > c->n_local &&
            (rc = recentre_order(c, c->h_sums + kApplyAt)) != ICP_HIP_OK)
            return rc;
        loop_init_step(&st, c->h_sums + kApplyAt, mu_m, threshold, cap_iters, (double)c->n_total,
                       sharded);
        c->first_sums_valid = false; // the loop moves the cloud
        const uint32_t g_ap = c->small ? 1u : g_st;

        float nn_ms = 0.f, ap_ms = 0.f; // kernel times of the iteration whose error is pending
        auto finished = [&](uint32_t it_before) {
            if (st.it == it_before)
                return;
            c->timing.nn_ms += nn_ms;
            c->timing.apply_ms += ap_ms;
            c->timing.nn_launches += 1;
            c->timing.queries += (double)c->n_total;
            if (c->nn_trace.size() < 4096)
                c->nn_trace.push_back(nn_ms);
            if (progress)
                progress(st.it, max_iters, st.error, user);
        };
        while (!st.done)
        {
            // y = tree.closest(p - mu_q); H = sum (p - mu_q) (x) y      icp.cc:115-117, :32-33
            float ms = 0.f;
            const bool idle = loop_nn_idle(&st);
            if (!idle)
            {
                NnLaunch a = make_nn(c, st.mu_q, true, false, false, true);
                if ((rc = enqueue_nn(c, a, kTailNone, c->ev0, c->ev1)) != ICP_HIP_OK)
                    return rc;
            }
            const uint32_t it_before = st.it;
            if (sharded)
            {
                // one all-reduce: this pass' sums + the apply pass' sums of the iteration before
                if ((rc = sums_to_host(c, 0, kIterSums)) != ICP_HIP_OK ||
                    (rc = host_allreduce(c, 0, kIterSums)) != ICP_HIP_OK)
                    return rc;
                if (!idle && n)
                    ms = elapsed_ms(c);
                loop_advance_step(&st, c->h_sums); // icp.cc:123,:110 then :119,:121
                finished(it_before);
            }
            else
            {
                if ((rc = sums_to_host(c, 0, kCovSums)) != ICP_HIP_OK)
                    return rc;
                if (n)
                    ms = elapsed_ms(c);
                loop_solve_step(&st, c->h_sums); // icp.cc:119, :121
            }
            if (st.done)
                break;
            nn_ms = ms;
            // p <- Tn p ; error ; next centroid                      icp.cc:122-123
            if (n)
            {
                Mat34 M;
                std::memcpy(M.m, st.Tn, sizeof M.m);
                HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
                HIP_TRY(c, launch_apply_error(c->px.p, c->py.p, c->pz.p, c->yx.p, c->yy.p, c->yz.p,
                                              M, st.mu_m, n, g_ap, c->partials.p, nullptr,
                                              c->stream,
                                              c->small ? c->sums.p + kApplyAt : nullptr));
                HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
                if (!c->small)
                    HIP_TRY(c, launch_reduce_partials(c->partials.p, g_st, kApplySums,
                                                      c->sums.p + kApplyAt, c->stream));
            }
            else
                HIP_TRY(c, hipMemsetAsync(c->sums.p + kApplyAt, 0, sizeof(double) * kApplySums,
                                          c->stream));
            if (!sharded)
            {
                if ((rc = sums_to_host(c, kApplyAt, kApplySums)) != ICP_HIP_OK)
                    return rc;
                ap_ms = n ? elapsed_ms(c) : 0.f;
                loop_finish_step(&st, c->h_sums + kApplyAt); // icp.cc:123, :110
                finished(it_before);
                if ((rc = maybe_resort(c, st)) != ICP_HIP_OK)
                    return rc;
            }
            else
            {
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                ap_ms = n ? elapsed_ms(c) : 0.f;
                // (after the apply pass, which still needed this iteration's matches; every rank
                // decides about its own shard from the all-reduced state, as under RCCL)
                if ((rc = maybe_resort(c, st)) != ICP_HIP_OK)
                    return rc;
            }
        }
        loop_results(c, st, T_out, iters_done, final_error);
        c->timing.host_ms +=
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin)
                .count();
        return ICP_HIP_OK;
    }
    // The shard as it stands in HBM, in the order it was handed to icp_hip_set_queries.
    int get_points_local(icp_hip_ctx* c, float* xyz_aos)
    {
        if (c->n_local == 0)
            return ICP_HIP_OK;
        if (!xyz_aos)
            return fail(c, ICP_HIP_ERR_ARG, "icp_hip_get_points: xyz_aos is NULL");
        DeviceGuard g(c->device);
        const uint32_t n = (uint32_t)c->n_local;
        if (c->small && c->host_points_valid)
        {
            // the last registration brought the block back with its state: nothing to wait for
            const float* hx = reinterpret_cast<const float*>(c->h_down + kSmallHead);
            const float *hy = hx + c->small_pad, *hz = hy + c->small_pad;
            for (uint32_t i = 0; i < n; ++i)
            {
                xyz_aos[3 * (size_t)i] = hx[i];
                xyz_aos[3 * (size_t)i + 1] = hy[i];
                xyz_aos[3 * (size_t)i + 2] = hz[i];
            }
            return ICP_HIP_OK;
        }
        {
            int rc = flush_queries(c);
            if (rc != ICP_HIP_OK)
                return rc;
        }
        HIP_TRY(c, c->aos.alloc(3ull * n));
        if (c->sorted)
            HIP_TRY(c, launch_scatter_aos(c->px.p, c->py.p, c->pz.p, c->perm.p, n, c->aos.p,
                                          c->stream));
        else
            HIP_TRY(c, launch_soa_to_aos(c->px.p, c->py.p, c->pz.p, n, c->aos.p, c->stream));
        HIP_TRY(c, hipMemcpyAsync(xyz_aos, c->aos.p, 3ull * n * sizeof(float),
                                  hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return ICP_HIP_OK;
    }

    // ---- icp_hip_set_queries_exchange: a rank that cannot choose its rows ----
    // One all-to-all of variable-size blocks between the ranks, through the caller's callback
    // or, with an RCCL communicator, ncclSend / ncclRecv on staging buffers in HBM.  send / recv
    // hold the blocks for / from ranks 0 .. R-1 back to back.
    int exchange_blocks(icp_hip_ctx* c, const void* send, const uint64_t* send_bytes, void* recv,
                        const uint64_t* recv_bytes, int R)
    {
        if (c->alltoallv)
        {
            if (c->alltoallv(send, send_bytes, recv, recv_bytes, R, c->alltoallv_user) != 0)
                return fail(c, ICP_HIP_ERR_COMM, "all-to-all callback failed");
            return ICP_HIP_OK;
        }
        if (!c->comm || !g_rccl.Send || !g_rccl.Recv || !g_rccl.GroupStart || !g_rccl.GroupEnd)
            return fail(c, ICP_HIP_ERR_COMM,
                        "icp_hip_set_queries_exchange: neither an all-to-all callback "
                        "(icp_hip_set_alltoallv) nor an RCCL communicator with ncclSend / ncclRecv");
        uint64_t ns = 0, nr = 0;
        for (int r = 0; r < R; ++r)
        {
            ns += send_bytes[r];
            nr += recv_bytes[r];
        }
        DevBuf<unsigned char> ds, dr;
        HIP_TRY(c, ds.alloc(ns ? ns : 1));
        HIP_TRY(c, dr.alloc(nr ? nr : 1));
        int rc = ICP_HIP_OK;
        hipError_t e = ns ? hipMemcpyAsync(ds.p, send, ns, hipMemcpyHostToDevice, c->stream) : hipSuccess;
        int nrc = 0;
        if (e == hipSuccess)
        {
            nrc = g_rccl.GroupStart();
            uint64_t so = 0, ro = 0;
            for (int r = 0; r < R && nrc == 0; ++r)
            {
                if (send_bytes[r])
                    nrc = g_rccl.Send(ds.p + so, (size_t)send_bytes[r], Rccl::kInt8, r, c->comm, c->stream);
                if (nrc == 0 && recv_bytes[r])
                    nrc = g_rccl.Recv(dr.p + ro, (size_t)recv_bytes[r], Rccl::kInt8, r, c->comm, c->stream);
                so += send_bytes[r];
                ro += recv_bytes[r];
            }
            const int end = g_rccl.GroupEnd();
            if (nrc == 0)
                nrc = end;
        }
        if (e == hipSuccess && nrc == 0 && nr)
            e = hipMemcpyAsync(recv, dr.p, nr, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(c->stream);
        else
            (void)hipStreamSynchronize(c->stream);
        ds.release();
        dr.release();
        if (nrc != 0)
            rc = fail(c, ICP_HIP_ERR_COMM,
                      std::string("ncclSend/ncclRecv: ") +
                          (g_rccl.GetErrorString ? g_rccl.GetErrorString(nrc) : "error"));
        else if (e != hipSuccess)
            rc = hip_fail(c, e, "icp_hip_set_queries_exchange: staging copy");
        return rc;
    }

    // The moved points of an exchanged shard go back to the ranks whose rows they are: the
    // reverse of the set-up exchange (collective: every rank calls icp_hip_get_points).
    int get_points_exchanged(icp_hip_ctx* c, float* xyz_aos)
    {
        const int R = (int)c->ex_sent.size();
        if (c->ex_n_caller != 0 && !xyz_aos)
            return fail(c, ICP_HIP_ERR_ARG, "icp_hip_get_points: xyz_aos is NULL");
        std::vector<float> mine(3 * (size_t)c->n_local + 3), back(3 * (size_t)c->ex_n_caller + 3);
        int rc = get_points_local(c, mine.data());
        if (rc != ICP_HIP_OK)
            return rc;
        std::vector<uint64_t> sb(R), rb(R);
        for (int r = 0; r < R; ++r)
        {
            sb[r] = c->ex_got[r] * 12ull;  // what came from rank r goes back to it, in that order
            rb[r] = c->ex_sent[r] * 12ull;
        }
        if ((rc = exchange_blocks(c, mine.data(), sb.data(), back.data(), rb.data(), R)) != ICP_HIP_OK)
            return rc;
        for (uint64_t k = 0; k < c->ex_n_caller; ++k)
        {
            const size_t row = c->ex_rows[k];
            xyz_aos[3 * row] = back[3 * k];
            xyz_aos[3 * row + 1] = back[3 * k + 1];
            xyz_aos[3 * row + 2] = back[3 * k + 2];
        }
        return ICP_HIP_OK;
    }
} // namespace

extern "C"
{
    const char* icp_hip_last_global_error(void) { return g_error.c_str(); }
    const char* icp_hip_last_error(const icp_hip_ctx* ctx)
    {
        return ctx ? ctx->err.c_str() : g_error.c_str();
    }

    int icp_hip_tree_build(const float* xyz_centered, uint64_t n, uint32_t capacity,
                           icp_flat_tree** out)
    {
        if (!out)
            return fail(nullptr, ICP_HIP_ERR_ARG, "icp_hip_tree_build: out is NULL");
        icp_host::FlatTreeOwner* owner = nullptr;
        std::string err;
        int rc;
        try
        {
            rc = icp_host::build_flat_tree(xyz_centered, n, capacity, &owner, err);
        }
        catch (const std::bad_alloc&)
        {
            rc = ICP_HIP_ERR_NOMEM;
            err = "icp_hip_tree_build: out of host memory";
        }
        if (rc != ICP_HIP_OK)
        {
            *out = nullptr;
            return fail(nullptr, rc, err);
        }
        *out = &owner->view;
        return ICP_HIP_OK;
    }

    void icp_hip_tree_free(icp_flat_tree* tree)
    {
        // `view` is the first member of FlatTreeOwner
        delete reinterpret_cast<icp_host::FlatTreeOwner*>(tree);
    }

    int icp_hip_tree_validate(const icp_flat_tree* tree)
    {
        std::string err;
        int rc;
        try
        {
            rc = icp_host::validate_flat_tree(tree, err);
        }
        catch (const std::bad_alloc&)
        {
            rc = ICP_HIP_ERR_NOMEM;
            err = "icp_hip_tree_validate: out of host memory";
        }
        return rc == ICP_HIP_OK ? rc : fail(nullptr, rc, err);
    }

    void icp_hip_mean_host(const float* xyz, uint64_t n, float mu_out[3])
    {
        icp_host::mean_host(xyz, n, mu_out);
    }
    void icp_hip_center_host(const float* xyz, uint64_t n, const float mu[3], float* out_xyz)
    {
        icp_host::center_host(xyz, n, mu, out_xyz);
    }

    static int hand_over(std::vector<float>& v, float** xyz_out, uint64_t* n_out, const char* who)
    {
        float* p = (float*)std::malloc(v.empty() ? 1 : v.size() * sizeof(float));
        if (!p)
            return fail(nullptr, ICP_HIP_ERR_NOMEM, std::string(who) + ": out of host memory");
        if (!v.empty())
            std::memcpy(p, v.data(), v.size() * sizeof(float));
        *xyz_out = p;
        *n_out = v.size() / 3;
        return ICP_HIP_OK;
    }

    int icp_hip_read_csv(const char* path, const char* x_field, const char* y_field,
                         const char* z_field, float** xyz_out, uint64_t* n_out)
    {
        if (!path || !x_field || !y_field || !z_field || !xyz_out || !n_out)
            return fail(nullptr, ICP_HIP_ERR_ARG, "icp_hip_read_csv: NULL argument");
        *xyz_out = nullptr;
        *n_out = 0;
        std::vector<float> v;
        std::string err;
        int rc = icp_host::read_csv(path, x_field, y_field, z_field, v, err);
        if (rc != ICP_HIP_OK)
            return fail(nullptr, rc, err);
        return hand_over(v, xyz_out, n_out, "icp_hip_read_csv");
    }

    int icp_hip_read_f32(const char* path, float** xyz_out, uint64_t* n_out)
    {
        if (!path || !xyz_out || !n_out)
            return fail(nullptr, ICP_HIP_ERR_ARG, "icp_hip_read_f32: NULL argument");
        *xyz_out = nullptr;
        *n_out = 0;
        std::vector<float> v;
        std::string err;
        int rc = icp_host::read_f32(path, v, err);
        if (rc != ICP_HIP_OK)
            return fail(nullptr, rc, err);
        return hand_over(v, xyz_out, n_out, "icp_hip_read_f32");
    }

    void icp_hip_free_points(float* xyz) { std::free(xyz); }

    void icp_hip_alignment_host(const float H[9], const float mu_p[3], const float mu_m[3],
                                float T_out[16])
    {
        icp_host::alignment_from_covariance(H, mu_p, mu_m, T_out);
    }

    int icp_hip_create(icp_hip_ctx** out, int device_id)
    {
        if (!out)
            return fail(nullptr, ICP_HIP_ERR_ARG, "icp_hip_create: out is NULL");
        *out = nullptr;
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev <= 0)
            return fail(nullptr, ICP_HIP_ERR_HIP,
                        std::string("icp_hip_create: no HIP device available (") +
                            hipGetErrorString(e) + ")");
        if (device_id < 0)
        {
            if (hipGetDevice(&device_id) != hipSuccess)
                device_id = 0;
        }
        if (device_id >= ndev)
            return fail(nullptr, ICP_HIP_ERR_ARG, "icp_hip_create: device_id out of range");
        auto* c = new (std::nothrow) icp_hip_ctx();
        if (!c)
            return fail(nullptr, ICP_HIP_ERR_NOMEM, "icp_hip_create: out of host memory");
        c->device = device_id;
        DeviceGuard g(device_id);
        auto bail = [&](hipError_t err, const char* what) {
            int rc = hip_fail(nullptr, err, what);
            icp_hip_destroy(c);
            return rc;
        };
        bool from_pool = false;
        if (device_id < kPoolDevices)
        {
            DevicePool& P = pool();
            std::lock_guard<std::mutex> lock(P.m);
            if (!P.ctxs[device_id].empty())
            {
                PooledCtx r = std::move(P.ctxs[device_id].back());
                P.ctxs[device_id].pop_back();
                c->stream = r.stream;
                c->ev0 = r.ev0;
                c->ev1 = r.ev1;
                c->h_sums = r.h_sums;
                c->loop_events = std::move(r.loop_events);
                from_pool = true;
            }
        }
        if (!from_pool)
        {
            if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess)
                return bail(e, "hipStreamCreate");
            if ((e = hipEventCreate(&c->ev0)) != hipSuccess)
                return bail(e, "hipEventCreate");
            if ((e = hipEventCreate(&c->ev1)) != hipSuccess)
                return bail(e, "hipEventCreate");
            // 16 doubles of sums + the device-resident loop's state, on the device and pinned
            // (+ the two pinned images of a small shard's block)
            if ((e = hipHostMalloc((void**)&c->h_sums, kPinnedBytes, hipHostMallocDefault)) !=
                hipSuccess)
                return bail(e, "hipHostMalloc");
        }
        c->h_loop = reinterpret_cast<LoopState*>(c->h_sums + 16);
        c->h_up = reinterpret_cast<unsigned char*>(c->h_sums) + kPinnedHead;
        c->h_down = c->h_up + kSmallBlockBytes;
        if ((e = c->sums.alloc(16 + kStateDoubles)) != hipSuccess)
            return bail(e, "hipMalloc(sums)");
        c->loop_state = reinterpret_cast<LoopState*>(c->sums.p + 16);
        {
            const char* lm = std::getenv("ICP_HIP_LOOP");
            c->loop_mode = !lm ? 0 : std::strcmp(lm, "host") == 0 ? 1
                                 : std::strcmp(lm, "device") == 0 ? 2
                                                                  : 0;
        }
        {
            int per_cu = 0, cus = 0;
            if ((e = nn_tile_blocks_per_cu(&per_cu)) != hipSuccess)
                return bail(e, "hipOccupancyMaxActiveBlocksPerMultiprocessor");
            if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount,
                                           device_id)) != hipSuccess)
                return bail(e, "hipDeviceGetAttribute");
            if ((e = hipEventCreateWithFlags(&c->noalloc_ev, hipEventDisableTiming)) != hipSuccess)
                return bail(e, "hipEventCreate");
            // the static scratch array of icp_hip_nn_noalloc: its address on this device
            if ((e = nn_noalloc_setup(device_id)) != hipSuccess)
                return bail(e, "hipGetSymbolAddress(scratch)");
            if (const char* w = std::getenv("ICP_HIP_TILE_WGS_PER_CU")) // experiments only
                if (std::atoi(w) > 0 && std::atoi(w) < per_cu)
                    per_cu = std::atoi(w);
            uint64_t blocks = (uint64_t)(per_cu > 0 ? per_cu : 1) * (uint64_t)(cus > 0 ? cus : 1);
            if (blocks > kNnMaxBlocks)
                blocks = kNnMaxBlocks;
            c->tile_blocks = (uint32_t)blocks;
            c->cus = (cus > 0 && blocks == (uint64_t)per_cu * (uint64_t)cus) ? (uint32_t)cus : 0u;
            const char* env = std::getenv("ICP_HIP_NN_KERNEL");
            c->force_deep = env && std::strcmp(env, "deep") == 0;
            const char* tb = std::getenv("ICP_HIP_TREE_BUILD");
            c->tree_build_on = !tb ? 0 : std::strcmp(tb, "device") == 0 ? 1
                                       : std::strcmp(tb, "host") == 0   ? 2
                                                                        : 0;
            const char* cut = std::getenv("ICP_HIP_TILE_CUT");
            if (cut)
                c->tile_cut = (uint32_t)std::atoi(cut);
            if (const char* pk = std::getenv("ICP_HIP_PARK"))
                c->park_mode = std::strcmp(pk, "always") == 0 ? 1
                               : (std::strcmp(pk, "never") == 0 || std::strcmp(pk, "0") == 0) ? 2 : 0;
            if (const char* pm = std::getenv("ICP_HIP_PARK_MAX"))
                c->park_max_env = std::min((uint32_t)std::max(0, std::atoi(pm)), kParkMaxLanes);
            if (const char* pw = std::getenv("ICP_HIP_PARK_WAIT"))
                c->park_wait = std::min((uint32_t)std::max(0, std::atoi(pw)), 255u);
            if (const char* lb = std::getenv("ICP_HIP_LATE_PER_BATCH"))
                c->late_per_batch_env = std::min((uint32_t)std::max(1, std::atoi(lb)), 32u);
        }
        *out = c;
        return ICP_HIP_OK;
    }

    void icp_hip_destroy(icp_hip_ctx* c)
    {
        if (!c)
            return;
        DeviceGuard g(c->device);
        if (c->stream)
            (void)hipStreamSynchronize(c->stream);
        drain_noalloc(c);
        if (c->noalloc_ev)
            (void)hipEventDestroy(c->noalloc_ev);
        c->noalloc_ev = nullptr;
        if (c->comm && g_rccl.CommDestroy)
            (void)g_rccl.CommDestroy(c->comm);
        c->nodes.release();
        c->leaf_info.release();
        c->sx.release();
        c->sy.release();
        c->sz.release();
        c->srow.release();
        c->nrow.release();
        c->ring.release();
        c->px.release();
        c->py.release();
        c->pz.release();
        c->yx.release();
        c->yy.release();
        c->yz.release();
        c->aos.release();
        c->dist.release();
        c->row.release();
        c->perm.release();
        c->queue.release();
        c->chunk_sums.release();
        c->park.release();
        c->late.release();
        c->late_cnt.release();
        c->spill.release();
        c->stats.release();
        c->partials.release();
        c->sums.release();
        c->tree_blob.release();
        c->qblock.release();
        c->sk_keys.release();
        c->sk_keys2.release();
        c->sk_idx.release();
        c->sk_order.release();
        c->sk_temp.release();
        if (c->stream && c->ev0 && c->ev1 && c->h_sums && c->device < kPoolDevices)
        {
            // the stream is idle (synchronised above): the next context on this device takes it
            PooledCtx r;
            r.stream = c->stream;
            r.ev0 = c->ev0;
            r.ev1 = c->ev1;
            r.h_sums = c->h_sums;
            r.loop_events = std::move(c->loop_events);
            DevicePool& P = pool();
            std::lock_guard<std::mutex> lock(P.m);
            if (P.ctxs[c->device].size() < 8)
            {
                P.ctxs[c->device].push_back(std::move(r));
                delete c;
                return;
            }
            c->loop_events = std::move(r.loop_events);
        }
        if (c->h_sums)
            (void)hipHostFree(c->h_sums);
        for (hipEvent_t ev : c->loop_events)
            (void)hipEventDestroy(ev);
        if (c->ev0)
            (void)hipEventDestroy(c->ev0);
        if (c->ev1)
            (void)hipEventDestroy(c->ev1);
        if (c->stream)
            (void)hipStreamDestroy(c->stream);
        delete c;
    }

    int icp_hip_set_tree(icp_hip_ctx* c, const icp_flat_tree* t)
    {
        if (!c)
            return fail(nullptr, ICP_HIP_ERR_ARG, "icp_hip_set_tree: ctx is NULL");
        {
            std::string err;
            int rc = icp_host::validate_flat_tree(t, err);
            if (rc != ICP_HIP_OK)
                return fail(c, rc, err);
        }
        DeviceGuard g(c->device);
        return install_tree(c, t, nullptr, nullptr, nullptr, nullptr);
    }

    int icp_hip_build_tree(icp_hip_ctx* c, const float* xyz_centered, uint64_t n, uint32_t capacity)
    {
        if (!c)
            return fail(nullptr, ICP_HIP_ERR_ARG, "icp_hip_build_tree: ctx is NULL");
        if (!xyz_centered || n == 0)
            return fail(c, ICP_HIP_ERR_ARG, "icp_hip_build_tree: empty cloud");
        if (n >= (1ull << 31))
            return fail(c, ICP_HIP_ERR_ARG,
                        "icp_hip_build_tree: more than 2^31-1 points are not supported");
        if (capacity < 3)
            return fail(c, ICP_HIP_ERR_DEGENERATE,
                        "icp_hip_build_tree: capacity <= 2 is undefined behaviour in the reference "
                        "(vp-tree.cc:114: empty inside set on 2-point nodes)");
        DeviceGuard g(c->device);
        drain_noalloc(c);
        c->has_tree = false;
        // A small cloud is built by the host builder (same tree, bit for bit) and uploaded: the
        // level-by-level device build costs ~0.1 ms of launches per level whatever the size.
        // (the 48k-point horse at capacity 256, 8 levels: host 0.8 ms against 1.2 ms, but with 16
        // OpenMP threads woken per call and a noisy wall time; not worth it)
        constexpr uint64_t kHostBuildBelow = 8192;
        if (c->tree_build_on == 2 || (c->tree_build_on == 0 && n < kHostBuildBelow))
        {
            icp_flat_tree* t = nullptr;
            int rc = icp_hip_tree_build(xyz_centered, n, capacity, &t);
            if (rc != ICP_HIP_OK)
                return fail(c, rc, g_error);
            rc = install_tree(c, t, nullptr, nullptr, nullptr, nullptr);
            icp_hip_tree_free(t);
            return rc;
        }
        DevBuf<float> cloud;
        HIP_TRY(c, cloud.alloc(3 * n));
        struct Free
        {
            DevBuf<float>& b;
            hipStream_t s;
            ~Free()
            {
                (void)hipStreamSynchronize(s); // nothing in flight reads a block that is pooled
                b.release();
            }
        } free_cloud{cloud, c->stream};
        HIP_TRY(c, hipMemcpyAsync(cloud.p, xyz_centered, 3 * n * sizeof(float),
                                  hipMemcpyHostToDevice, c->stream));
        BuiltTree bt;
        std::string err;
        int rc;
        try
        {
            rc = build_tree_device(cloud.p, (uint32_t)n, capacity, c->stream, bt, err);
        }
        catch (const std::bad_alloc&)
        {
            rc = ICP_HIP_ERR_NOMEM;
            err = "icp_hip_build_tree: out of host memory";
        }
        if (rc != ICP_HIP_OK)
            return fail(c, rc, "icp_hip_build_tree: " + err);
        icp_flat_tree v{};
        v.n_points = bt.n_points;
        v.n_slots = bt.n_slots;
        v.n_nodes = (uint32_t)bt.node_cx.size();
        v.n_leaves = (uint32_t)bt.leaf_begin.size();
        v.height = bt.height;
        v.capacity = bt.capacity;
        v.max_leaf_count = bt.max_leaf_count;
        v.root = bt.root;
        v.node_cx = bt.node_cx.data();
        v.node_cy = bt.node_cy.data();
        v.node_cz = bt.node_cz.data();
        v.node_radius = bt.node_radius.data();
        v.node_inside = bt.node_inside.data();
        v.node_outside = bt.node_outside.data();
        v.node_parent = bt.node_parent.data();
        v.node_row = bt.node_row.data();
        v.leaf_begin = bt.leaf_begin.data();
        v.leaf_count = bt.leaf_count.data();
        if ((int)v.height > nn_max_depth_supported())
        {
            (void)hipFree(bt.d_sx);
            (void)hipFree(bt.d_sy);
            (void)hipFree(bt.d_sz);
            (void)hipFree(bt.d_srow);
            return fail(c, ICP_HIP_ERR_DEPTH, "icp_hip_build_tree: tree too deep");
        }
        return install_tree(c, &v, bt.d_sx, bt.d_sy, bt.d_sz, bt.d_srow);
    }

    int icp_hip_get_tree_info(icp_hip_ctx* c, icp_flat_tree* out)
    {
        if (!c || !out)
            return fail(c, ICP_HIP_ERR_ARG, "icp_hip_get_tree_info: NULL argument");
        if (!c->has_tree)
            return fail(c, ICP_HIP_ERR_ARG, "icp_hip_get_tree_info: no tree on the device");
        *out = icp_flat_tree{};
        out->n_points = c->n_tree_points;
        out->n_slots = c->tree.n_slots;
        out->n_nodes = c->tree.n_nodes;
        out->n_leaves = c->n_leaves;
        out->height = c->height;
        out->capacity = c->capacity;
        out->max_leaf_count = c->max_leaf;
        out->root = c->tree.root;
        return ICP_HIP_OK;
    }

    int icp_hip_get_tree(icp_hip_ctx* c, icp_flat_tree** out)
    {
        if (!c || !out)
            return fail(c, ICP_HIP_ERR_ARG, "icp_hip_get_tree: NULL argument");
        *out = nullptr;
        if (!c->has_tree)
            return fail(c, ICP_HIP_ERR_ARG, "icp_hip_get_tree: no tree on the device");
        DeviceGuard g(c->device);
        icp_host::FlatTreeOwner* T = nullptr;
        try
        {
            T = new icp_host::FlatTreeOwner();
            const size_t nn = c->tree.n_nodes, nl = c->n_leaves, ns = c->tree.n_slots;
            std::vector<DevNode> nodes(nn);
            std::vector<uint2> li(nl);
            T->node_row.resize(nn);
            T->slot_x.resize(ns);
            T->slot_y.resize(ns);
            T->slot_z.resize(ns);
            T->slot_row.resize(ns);
            auto down = [&](void* dst, const void* src, size_t bytes) {
                return bytes ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream)
                             : hipSuccess;
            };
            hipError_t e = down(nodes.data(), c->tree.nodes, nn * sizeof(DevNode));
            if (e == hipSuccess)
                e = down(T->node_row.data(), c->tree.nrow, nn * sizeof(uint32_t));
            if (e == hipSuccess)
                e = down(li.data(), c->tree.leaf_info, nl * sizeof(uint2));
            if (e == hipSuccess)
                e = down(T->slot_x.data(), c->tree.sx, ns * sizeof(float));
            if (e == hipSuccess)
                e = down(T->slot_y.data(), c->tree.sy, ns * sizeof(float));
            if (e == hipSuccess)
                e = down(T->slot_z.data(), c->tree.sz, ns * sizeof(float));
            if (e == hipSuccess)
                e = down(T->slot_row.data(), c->tree.srow, ns * sizeof(uint32_t));
            if (e == hipSuccess)
                e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess)
            {
                delete T;
                return hip_fail(c, e, "icp_hip_get_tree download");
            }
            T->node_cx.resize(nn);
            T->node_cy.resize(nn);
            T->node_cz.resize(nn);
            T->node_radius.resize(nn);
            T->node_inside.resize(nn);
            T->node_outside.resize(nn);
            T->node_parent.resize(nn);
            for (size_t i = 0; i < nn; ++i)
            {
                T->node_cx[i] = nodes[i].cx;
                T->node_cy[i] = nodes[i].cy;
                T->node_cz[i] = nodes[i].cz;
                T->node_radius[i] = nodes[i].r;
                T->node_inside[i] = nodes[i].inside;
                T->node_outside[i] = nodes[i].outside;
                T->node_parent[i] = nodes[i].parent;
            }
            T->leaf_begin.resize(nl);
            T->leaf_count.resize(nl);
            for (size_t l = 0; l < nl; ++l)
            {
                T->leaf_begin[l] = li[l].x;
                T->leaf_count[l] = li[l].y;
            }
            icp_flat_tree& v = T->view;
            v.n_points = c->n_tree_points;
            v.n_slots = ns;
            v.n_nodes = (uint32_t)nn;
            v.n_leaves = (uint32_t)nl;
            v.height = c->height;
            v.capacity = c->capacity;
            v.max_leaf_count = c->max_leaf;
            v.root = c->tree.root;
            T->sync_view();
        }
        catch (const std::bad_alloc&)
        {
            delete T;
            return fail(c, ICP_HIP_ERR_NOMEM, "icp_hip_get_tree: out of host memory");
        }
        *out = &T->view;
        return ICP_HIP_OK;
    }

    int icp_hip_set_queries(icp_hip_ctx* c, const float* xyz_aos, uint64_t n_local,
                            uint64_t n_total)
    {
        if (!c)
            return fail(nullptr, ICP_HIP_ERR_ARG, "icp_hip_set_queries: ctx is NULL");
        if (n_local > 0 && !xyz_aos)
            return fail(c, ICP_HIP_ERR_ARG, "icp_hip_set_queries: xyz_aos is NULL");
        if (n_total < n_local || n_total == 0)
            return fail(c, ICP_HIP_ERR_ARG, "icp_hip_set_queries: n_total must be >= n_local, > 0");
        if (n_local >= (1ull << 31))
            return fail(c, ICP_HIP_ERR_ARG, "icp_hip_set_queries: shard larger than 2^31-1 points");
        DeviceGuard g(c->device);
        c->has_queries = false;
        c->exchanged = false;
        c->upload_pending = c->host_points_valid = c->first_sums_valid = false;
        c->leaf_ordered = c->resorted = c->resort_decided = false;
        const size_t n = n_local ? n_local : 1;
        HIP_TRY(c, c->yx.alloc(n));
        HIP_TRY(c, c->yy.alloc(n));
        HIP_TRY(c, c->yz.alloc(n));
        HIP_TRY(c, c->row.alloc(n));
        HIP_TRY(c, c->dist.alloc(n));
        HIP_TRY(c, c->partials.alloc((size_t)kNnMaxBlocks * kCovSums));
        c->sorted = false;
        c->small = n_local >= 1 && n_local < kSmallBelow;
        if (c->small)
        {
            // [LoopState | queue | x | y | z]: packed on the host (with the finiteness check and
            // the first centroid sums, in double, in input order), uploaded by the first user
            const uint32_t pad = ((uint32_t)n_local + 63u) & ~63u;
            HIP_TRY(c, c->qblock.alloc(kSmallBlockBytes));
            float* d = reinterpret_cast<float*>(c->qblock.p + kSmallHead);
            c->px.borrow(d, n_local);
            c->py.borrow(d + pad, n_local);
            c->pz.borrow(d + 2 * (size_t)pad, n_local);
            c->queue.borrow(reinterpret_cast<uint32_t*>(c->qblock.p + kSmallStateBytes),
                            kQueueWords);
            c->loop_state = reinterpret_cast<LoopState*>(c->qblock.p);
            c->h_loop = reinterpret_cast<LoopState*>(c->h_down);
            c->small_pad = pad;
            c->small_bytes = kSmallHead + 3 * (size_t)pad * sizeof(float);
            std::memset(c->h_up, 0, kSmallHead);
            float* hx = reinterpret_cast<float*>(c->h_up + kSmallHead);
            float *hy = hx + pad, *hz = hy + pad;
            double sx = 0.0, sy = 0.0, sz = 0.0;
            for (size_t i = 0; i < n_local; ++i)
            {
                const float x = xyz_aos[3 * i], y = xyz_aos[3 * i + 1], z = xyz_aos[3 * i + 2];
                hx[i] = x;
                hy[i] = y;
                hz[i] = z;
                sx += (double)x;
                sy += (double)y;
                sz += (double)z;
            }
            for (size_t i = n_local; i < pad; ++i)
                hx[i] = hy[i] = hz[i] = 0.f;
            // (a NaN or an infinity anywhere makes a sum non-finite; finite sums of fp32 values
            // cannot overflow a double)
            if (!std::isfinite(sx) || !std::isfinite(sy) || !std::isfinite(sz))
                return fail(c, ICP_HIP_ERR_ARG,
                            std::string("icp_hip_set_queries: ") + icp_host::kNonFiniteMessage);
            c->first_sums[0] = 0.0;
            c->first_sums[1] = sx;
            c->first_sums[2] = sy;
            c->first_sums[3] = sz;
            c->first_sums_valid = true;
            c->upload_pending = true;
            c->n_local = n_local;
            c->n_total = n_total;
            c->has_queries = true;
            c->timing = icp_hip_timing{};
            return prepare_tile(c);
        }
        c->loop_state = reinterpret_cast<LoopState*>(c->sums.p + 16);
        c->h_loop = reinterpret_cast<LoopState*>(c->h_sums + 16);
        c->qblock.release();
        HIP_TRY(c, c->px.alloc(n));
        HIP_TRY(c, c->py.alloc(n));
        HIP_TRY(c, c->pz.alloc(n));
        HIP_TRY(c, c->aos.alloc(3 * n));
        if (c->queue.borrowed)
            c->queue.release();
        if (n_local)
        {
            HIP_TRY(c, hipMemcpyAsync(c->aos.p, xyz_aos, 3 * n_local * sizeof(float),
                                      hipMemcpyHostToDevice, c->stream));
            const char* env = std::getenv("ICP_HIP_SORT_QUERIES");
            // (below ~8k points the two sorts of the set-up cost more than they save)
            const bool want_sort = !(env && env[0] == '0') && n_local >= kSmallBelow;
            if (!want_sort && !icp_host::all_finite(xyz_aos, 3 * n_local))
                return fail(c, ICP_HIP_ERR_ARG,
                            std::string("icp_hip_set_queries: ") + icp_host::kNonFiniteMessage);
            if (want_sort)
            {
                int rc = sort_queries(c, (uint32_t)n_local, n_total);
                if (rc != ICP_HIP_OK)
                    return rc;
            }
            else
                HIP_TRY(c, launch_aos_to_soa(c->aos.p, (uint32_t)n_local, c->px.p, c->py.p,
                                             c->pz.p, c->stream));
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->n_local = n_local;
        c->n_total = n_total;
        c->has_queries = true;
        c->timing = icp_hip_timing{};
        return prepare_tile(c);
    }

    int icp_hip_get_shard_size(icp_hip_ctx* c, uint64_t* n_local, uint64_t* n_total)
    {
        if (!c || !c->has_queries)
            return fail(c, ICP_HIP_ERR_ARG, "icp_hip_get_shard_size: no query shard set");
        if (n_local)
            *n_local = c->n_local;
        if (n_total)
            *n_total = c->n_total;
        return ICP_HIP_OK;
    }

    int icp_hip_set_alltoallv(icp_hip_ctx* c, int n_ranks, int rank, icp_hip_alltoallv_fn fn,
                              void* user)
    {
        if (!c)
            return fail(nullptr, ICP_HIP_ERR_ARG, "icp_hip_set_alltoallv: ctx is NULL");
        if (fn && (n_ranks < 1 || rank < 0 || rank >= n_ranks))
            return fail(c, ICP_HIP_ERR_ARG, "icp_hip_set_alltoallv: bad n_ranks / rank");
        c->alltoallv = fn;
        c->alltoallv_user = user;
        c->ex_ranks = fn ? n_ranks : 0;
        c->ex_rank = fn ? rank : 0;
        return ICP_HIP_OK;
    }

    int icp_hip_set_queries_exchange(icp_hip_ctx* c, const float* xyz_aos, uint64_t n_local,
                                     uint64_t n_total)
    {
        if (!c)
            return fail(nullptr, ICP_HIP_ERR_ARG, "icp_hip_set_queries_exchange: ctx is NULL");
        if (n_local > 0 && !xyz_aos)
            return fail(c, ICP_HIP_ERR_ARG, "icp_hip_set_queries_exchange: xyz_aos is NULL");
        if (n_total < n_local || n_total == 0 || n_local >= (1ull << 31))
            return fail(c, ICP_HIP_ERR_ARG, "icp_hip_set_queries_exchange: bad n_local / n_total");
        const int R = c->alltoallv ? c->ex_ranks : c->comm ? c->n_ranks : 1;
        // (ICP_HIP_EXCHANGE_SELF=1: a single rank goes through the exchange too, sending to itself
        // -- the only way to run the ncclSend / ncclRecv leg on a box with one GPU; tests)
        const char* self_env = std::getenv("ICP_HIP_EXCHANGE_SELF");
        if (R <= 1 && !(self_env && self_env[0] == '1' && (c->alltoallv || c->comm)))
            return icp_hip_set_queries(c, xyz_aos, n_local, n_total);
        if (!icp_host::all_finite(xyz_aos, 3 * n_local))
            return fail(c, ICP_HIP_ERR_ARG,
                        std::string("icp_hip_set_queries_exchange: ") + icp_host::kNonFiniteMessage);
        DeviceGuard g(c->device);
        int rc;
        // 1. the ranks' row counts (and coordinate sums), all to all: n_total is checked
        std::vector<uint64_t> b32(R, 32);
        std::vector<double> mine4(4 * (size_t)R), all4(4 * (size_t)R);
        {
            double s[4] = {(double)n_local, 0.0, 0.0, 0.0};
            for (uint64_t i = 0; i < n_local; ++i)
            {
                s[1] += (double)xyz_aos[3 * i];
                s[2] += (double)xyz_aos[3 * i + 1];
                s[3] += (double)xyz_aos[3 * i + 2];
            }
            for (int r = 0; r < R; ++r)
                std::memcpy(&mine4[4 * (size_t)r], s, sizeof s);
        }
        if ((rc = exchange_blocks(c, mine4.data(), b32.data(), all4.data(), b32.data(), R)) != ICP_HIP_OK)
            return rc;
        double tot[4] = {0.0, 0.0, 0.0, 0.0};
        for (int r = 0; r < R; ++r)
            for (int k = 0; k < 4; ++k)
                tot[k] += all4[4 * (size_t)r + k];
        if ((uint64_t)tot[0] != n_total)
            return fail(c, ICP_HIP_ERR_ARG,
                        "icp_hip_set_queries_exchange: the ranks' n_local add up to " +
                            std::to_string((uint64_t)tot[0]) + ", not to n_total = " +
                            std::to_string(n_total));
        // 2. a sample of the whole cloud on every rank: each rank sends the same evenly spaced
        //    kSample of its rows to everybody (a rank with fewer sends them all)
        constexpr uint64_t kSample = 16384;
        const uint64_t my_s = std::min<uint64_t>(n_local, kSample);
        std::vector<uint64_t> s_cnt(R, my_s), s_got(R, 0), b8s(R, 8);
        if ((rc = exchange_blocks(c, s_cnt.data(), b8s.data(), s_got.data(), b8s.data(), R)) != ICP_HIP_OK)
            return rc;
        uint64_t n_samp = 0;
        std::vector<uint64_t> ssb(R, my_s * 12ull), srb(R);
        for (int r = 0; r < R; ++r)
        {
            if (s_got[r] > kSample)
                return fail(c, ICP_HIP_ERR_COMM, "icp_hip_set_queries_exchange: bad sample size from a rank");
            srb[r] = s_got[r] * 12ull;
            n_samp += s_got[r];
        }
        std::vector<float> samp_send(3 * (size_t)my_s * R + 3), samp(3 * (size_t)n_samp + 3);
        for (uint64_t k = 0; k < my_s; ++k)
        {
            const uint64_t i = k * n_local / my_s;
            for (int r = 0; r < R; ++r)
                std::memcpy(&samp_send[3 * ((size_t)r * my_s + k)], xyz_aos + 3 * i, 12);
        }
        if ((rc = exchange_blocks(c, samp_send.data(), ssb.data(), samp.data(), srb.data(), R)) != ICP_HIP_OK)
            return rc;
        // 3. R spatially compact regions of about equal size: planes perpendicular to x, y, z in
        //    turn, each at the quantile of the sample that leaves both sides their share of the
        //    ranks (every rank computes the same planes from the same sample); a rank searches
        //    the whole tree for its region, and a compact region costs it about its share of
        //    the one-GPU pass (boxes: 1.0-1.15x; the tree's own top-level regions, balls and
        //    shells with more surface: 1.1-1.3x; a slice of unordered rows: 1.5-1.75x)
        struct Cut
        {
            int axis;
            float at;
            int below, above; // >= 0: index of the next cut; < 0: ~rank
        };
        std::vector<Cut> cuts;
        {
            std::vector<uint32_t> idx(n_samp);
            for (uint64_t k = 0; k < n_samp; ++k)
                idx[k] = (uint32_t)k;
            // (iterative: ranges of the index array with their rank interval and axis)
            struct Job
            {
                size_t b, e;
                int lo, hi, axis, slot; // slot: where to write this subtree's id (-1: root)
                bool above;
            };
            std::vector<Job> jobs{{0, (size_t)n_samp, 0, R, 0, -1, false}};
            while (!jobs.empty())
            {
                const Job j = jobs.back();
                jobs.pop_back();
                int id;
                if (j.hi - j.lo <= 1)
                    id = ~j.lo;
                else
                {
                    const int mid = j.lo + (j.hi - j.lo) / 2;
                    const size_t len = j.e - j.b;
                    size_t k = len * (size_t)(mid - j.lo) / (size_t)(j.hi - j.lo);
                    float at = 0.f;
                    if (len)
                    {
                        k = std::min(k, len - 1);
                        std::nth_element(idx.begin() + j.b, idx.begin() + j.b + k, idx.begin() + j.e,
                                         [&](uint32_t a, uint32_t b2) {
                                             const float va = samp[3 * (size_t)a + j.axis],
                                                         vb = samp[3 * (size_t)b2 + j.axis];
                                             return va < vb || (va == vb && a < b2);
                                         });
                        at = samp[3 * (size_t)idx[j.b + k] + j.axis];
                    }
                    id = (int)cuts.size();
                    cuts.push_back({j.axis, at, 0, 0});
                    // strictly below the cut value goes below (ties go above, like the sample's k-th)
                    size_t split = j.b;
                    if (len)
                        split = std::partition(idx.begin() + j.b, idx.begin() + j.e,
                                               [&](uint32_t a) { return samp[3 * (size_t)a + j.axis] < at; }) -
                                idx.begin();
                    jobs.push_back({j.b, split, j.lo, mid, (j.axis + 1) % 3, id, false});
                    jobs.push_back({split, j.e, mid, j.hi, (j.axis + 1) % 3, id, true});
                }
                if (j.slot >= 0)
                    (j.above ? cuts[j.slot].above : cuts[j.slot].below) = id;
            }
        }
        std::vector<uint32_t> owner(n_local ? n_local : 1);
        std::vector<uint64_t> cnt(R, 0);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < (int64_t)n_local; ++i)
        {
            int id = cuts.empty() ? ~0 : 0;
            while (id >= 0)
            {
                const Cut& q = cuts[id];
                id = xyz_aos[3 * i + q.axis] < q.at ? q.below : q.above;
            }
            owner[i] = (uint32_t)~id;
        }
        for (uint64_t i = 0; i < n_local; ++i)
            cnt[owner[i]] += 1;
        // 4. how many points every rank gets from every rank, then the points themselves
        //    (x, y, z; the caller's row stays here, in send order)
        std::vector<uint64_t> b8(R, 8), got(R, 0);
        if ((rc = exchange_blocks(c, cnt.data(), b8.data(), got.data(), b8.data(), R)) != ICP_HIP_OK)
            return rc;
        std::vector<uint64_t> at(R, 0), sb(R), rb(R);
        uint64_t n_recv = 0;
        for (int r = 0; r < R; ++r)
        {
            at[r] = r ? at[r - 1] + cnt[r - 1] : 0;
            sb[r] = cnt[r] * 12ull;
            rb[r] = got[r] * 12ull;
            n_recv += got[r];
        }
        if (n_recv >= (1ull << 31))
            return fail(c, ICP_HIP_ERR_ARG, "icp_hip_set_queries_exchange: region larger than 2^31-1 points");
        std::vector<float> send(3 * (size_t)n_local + 3), recv(3 * (size_t)n_recv + 3);
        std::vector<uint32_t> rows(n_local ? n_local : 1);
        for (uint64_t i = 0; i < n_local; ++i)
        {
            const uint64_t k = at[owner[i]]++;
            send[3 * k] = xyz_aos[3 * i];
            send[3 * k + 1] = xyz_aos[3 * i + 1];
            send[3 * k + 2] = xyz_aos[3 * i + 2];
            rows[k] = (uint32_t)i;
        }
        if ((rc = exchange_blocks(c, send.data(), sb.data(), recv.data(), rb.data(), R)) != ICP_HIP_OK)
            return rc;
        // 5. from here on this rank registers the points of its region
        if ((rc = icp_hip_set_queries(c, recv.data(), n_recv, n_total)) != ICP_HIP_OK)
            return rc;
        c->exchanged = true;
        c->ex_n_caller = n_local;
        c->ex_sent = cnt;
        c->ex_got = got;
        rows.resize(n_local);
        c->ex_rows = std::move(rows);
        return ICP_HIP_OK;
    }

    int icp_hip_nn(icp_hip_ctx* c, const float mu_p[3], uint32_t* row_out, float* xyz_out,
                   float* dist_out, float* kernel_ms)
    {
        if (!c || !mu_p)
            return fail(c, ICP_HIP_ERR_ARG, "icp_hip_nn: NULL argument");
        if (c->exchanged && (row_out || xyz_out || dist_out))
            return fail(c, ICP_HIP_ERR_ARG,
                        "icp_hip_nn: per-query outputs are not available for a shard that was "
                        "exchanged between the ranks (icp_hip_set_queries_exchange)");
        if (!c->has_tree || !c->has_queries)
            return fail(c, ICP_HIP_ERR_ARG, "icp_hip_nn: set_tree and set_queries must come first");
        DeviceGuard g(c->device);
        const uint32_t n = (uint32_t)c->n_local;
        if (kernel_ms)
            *kernel_ms = 0.f;
        if (n == 0)
            return ICP_HIP_OK;
        // rows and distances go back to the caller's order element by element (scattered 4-byte
        // stores: they double the pass at 64M points), so only when they are asked for
        NnLaunch a = make_nn(c, mu_p, true, row_out != nullptr, dist_out != nullptr, true);
        float ms = 0.f;
        {
            int rc = flush_queries(c);
            if (rc == ICP_HIP_OK)
                rc = nn_pass(c, a, &ms);
            if (rc != ICP_HIP_OK)
                return rc;
        }
        if (xyz_out)
        {
            HIP_TRY(c, c->aos.alloc(3ull * n));
            if (c->sorted)
                HIP_TRY(c, launch_scatter_aos(c->yx.p, c->yy.p, c->yz.p, c->perm.p, n, c->aos.p,
                                              c->stream));
            else
                HIP_TRY(c, launch_soa_to_aos(c->yx.p, c->yy.p, c->yz.p, n, c->aos.p, c->stream));
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->timing.nn_launches += 1;
        c->timing.nn_ms += ms;
        c->timing.nodes_visited += c->h_sums[9];   // this shard only (no all-reduce here)
        c->timing.points_scanned += c->h_sums[10];
        c->timing.stack_spills += c->h_sums[11];
        c->timing.queries += (double)n;
        if (kernel_ms)
            *kernel_ms = ms;
        if (row_out)
            HIP_TRY(c, hipMemcpy(row_out, c->row.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (dist_out)
            HIP_TRY(c, hipMemcpy(dist_out, c->dist.p, n * sizeof(float), hipMemcpyDeviceToHost));
        if (xyz_out)
            HIP_TRY(c, hipMemcpy(xyz_out, c->aos.p, 3ull * n * sizeof(float),
                                 hipMemcpyDeviceToHost));
        return ICP_HIP_OK;
    }

    int icp_hip_nn_noalloc(icp_hip_ctx* c, const float* d_px, const float* d_py, const float* d_pz,
                           uint32_t n, const float mu_p[3], uint32_t* d_row_out, float* d_dist_out,
                           float* d_yx, float* d_yy, float* d_yz, double* d_sums_out, void* stream)
    {
        if (!c || !mu_p || !d_px || !d_py || !d_pz)
            return fail(c, ICP_HIP_ERR_ARG, "icp_hip_nn_noalloc: NULL argument");
        if (!c->has_tree)
            return fail(c, ICP_HIP_ERR_ARG, "icp_hip_nn_noalloc: the context has no tree");
        if (stream)
        {
            int sdev = -1;
            if (hipStreamGetDevice((hipStream_t)stream, &sdev) == hipSuccess && sdev != c->device)
                return fail(c, ICP_HIP_ERR_ARG,
                            "icp_hip_nn_noalloc: the stream belongs to device " + std::to_string(sdev) +
                                ", the context to device " + std::to_string(c->device));
        }
        if ((d_yx || d_yy || d_yz) && !(d_yx && d_yy && d_yz))
            return fail(c, ICP_HIP_ERR_ARG, "icp_hip_nn_noalloc: d_yx, d_yy, d_yz go together");
        if (n == 0)
            return ICP_HIP_OK;
        if (n > kNoAllocMaxQueries)
            return fail(c, ICP_HIP_ERR_ARG,
                        "icp_hip_nn_noalloc: " + std::to_string(n) + " queries exceed the " +
                            std::to_string(kNoAllocMaxQueries) +
                            " the static scratch array is sized for; use icp_hip_nn");
        if (c->height > nn_tile_stack_entries() + kNoAllocSpillLevels)
            return fail(c, ICP_HIP_ERR_DEPTH,
                        "icp_hip_nn_noalloc: tree height " + std::to_string(c->height) +
                            " exceeds the static overflow stack (" +
                            std::to_string(nn_tile_stack_entries() + kNoAllocSpillLevels) +
                            " levels); use icp_hip_nn");
        DeviceGuard g(c->device);
        NnLaunch a{};
        a.tree = c->tree;
        a.height = c->height;
        a.max_leaf_count = c->max_leaf;
        a.px = d_px;
        a.py = d_py;
        a.pz = d_pz;
        a.mu[0] = mu_p[0];
        a.mu[1] = mu_p[1];
        a.mu[2] = mu_p[2];
        a.n = n;
        a.yx = d_yx;
        a.yy = d_yy;
        a.yz = d_yz;
        a.row_out = d_row_out;
        a.dist_out = d_dist_out;
        HIP_TRY(c, launch_nn_noalloc(c->device, a, d_sums_out, c->tile_cut, (hipStream_t)stream));
        // (recording an event allocates nothing; the event itself was made by icp_hip_create)
        HIP_TRY(c, hipEventRecord(c->noalloc_ev, (hipStream_t)stream));
        c->noalloc_pending = true;
        return ICP_HIP_OK;
    }

    int icp_hip_set_allreduce(icp_hip_ctx* c, icp_hip_allreduce_fn fn, void* user)
    {
        if (!c)
            return fail(nullptr, ICP_HIP_ERR_ARG, "icp_hip_set_allreduce: ctx is NULL");
        c->allreduce = fn;
        c->allreduce_user = user;
        return ICP_HIP_OK;
    }

    int icp_hip_comm_unique_id(void* unique_id_out, size_t size)
    {
        if (!unique_id_out || size < sizeof(Rccl::UniqueId))
            return fail(nullptr, ICP_HIP_ERR_ARG, "icp_hip_comm_unique_id: need a 128-byte buffer");
        std::string err;
        if (!g_rccl.load(err))
            return fail(nullptr, ICP_HIP_ERR_COMM, err);
        Rccl::UniqueId id;
        int rc = g_rccl.GetUniqueId(&id);
        if (rc != 0)
            return fail(nullptr, ICP_HIP_ERR_COMM, "ncclGetUniqueId failed");
        std::memcpy(unique_id_out, &id, sizeof id);
        return ICP_HIP_OK;
    }

    int icp_hip_comm_init_rank(icp_hip_ctx* c, int n_ranks, int rank, const void* unique_id,
                               size_t size)
    {
        if (!c || !unique_id || size < sizeof(Rccl::UniqueId) || n_ranks < 1 || rank < 0 ||
            rank >= n_ranks)
            return fail(c, ICP_HIP_ERR_ARG, "icp_hip_comm_init_rank: bad argument");
        std::string err;
        if (!g_rccl.load(err))
            return fail(c, ICP_HIP_ERR_COMM, err);
        DeviceGuard g(c->device);
        Rccl::UniqueId id;
        std::memcpy(&id, unique_id, sizeof id);
        if (c->comm)
        {
            (void)g_rccl.CommDestroy(c->comm);
            c->comm = nullptr;
        }
        int rc = g_rccl.CommInitRank(&c->comm, n_ranks, id, rank);
        if (rc != 0)
        {
            c->comm = nullptr;
            return fail(c, ICP_HIP_ERR_COMM,
                        std::string("ncclCommInitRank: ") +
                            (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error"));
        }
        c->n_ranks = n_ranks;
        c->rank = rank;
        return ICP_HIP_OK;
    }

    int icp_hip_run(icp_hip_ctx* c, const float mu_m[3], uint64_t max_iters, float threshold,
                    float T_out[16], uint64_t* iters_done, float* final_error,
                    icp_hip_progress_fn progress, void* user)
    {
        if (!c || !mu_m || !T_out)
            return fail(c, ICP_HIP_ERR_ARG, "icp_hip_run: NULL argument");
        if (!c->has_tree || !c->has_queries)
            return fail(c, ICP_HIP_ERR_ARG, "icp_hip_run: set_tree and set_queries must come first");
        DeviceGuard g(c->device);
        // The loop stays on the device unless something needs the host between the kernels: the
        // all-reduce callback, the diagnostic counters, the level-stack kernel.  Both loops give
        // the same bits (tests/test_gpu_icp_parity.py).  With an RCCL communicator every rank
        // must issue the same collectives, so the choice may not depend on anything a single rank
        // sees (its environment, an empty shard): the loop is the device-resident one or the
        // call fails.
        const bool diagnostic = c->force_deep || c->stats.p || c->loop_mode == 1;
        if (c->comm)
        {
            if (diagnostic || c->allreduce)
                return fail(c, ICP_HIP_ERR_ARG,
                            "icp_hip_run: with an RCCL communicator the loop runs on the device on "
                            "every rank; ICP_HIP_NN_KERNEL=deep, ICP_HIP_TILE_STATS, "
                            "ICP_HIP_LOOP=host and icp_hip_set_allreduce cannot be combined with it");
            if (!c->tile_ready && c->n_local > 0)
                return fail(c, ICP_HIP_ERR_ARG, "icp_hip_run: the NN kernel is not set up");
            return run_device_loop(c, mu_m, max_iters, threshold, T_out, iters_done, final_error,
                                   progress, user);
        }
        if (!diagnostic && !c->allreduce && c->tile_ready && c->n_local > 0)
            return run_device_loop(c, mu_m, max_iters, threshold, T_out, iters_done, final_error,
                                   progress, user);
        return run_host_loop(c, mu_m, max_iters, threshold, T_out, iters_done, final_error,
                             progress, user);
    }

    int icp_hip_get_points(icp_hip_ctx* c, float* xyz_aos)
    {
        if (!c || !c->has_queries)
            return fail(c, ICP_HIP_ERR_ARG, "icp_hip_get_points: no query shard set");
        if (c->exchanged)
            return get_points_exchanged(c, xyz_aos);
        return get_points_local(c, xyz_aos);
    }

    int icp_hip_icp(icp_hip_ctx* c, const float* m_xyz, uint64_t n_m, const float* p_xyz,
                    uint64_t n_p, uint64_t iterations, float threshold, uint32_t vp_threshold,
                    float T_out[16], float* new_p_xyz, uint64_t* iters_done, float* final_error,
                    icp_hip_progress_fn progress, void* user)
    {
        if (!c)
            return fail(nullptr, ICP_HIP_ERR_ARG, "icp_hip_icp: ctx is NULL");
        if (!m_xyz || !p_xyz || !T_out)
            return fail(c, ICP_HIP_ERR_ARG, "icp_hip_icp: NULL argument");
        if (n_m == 0 || n_p == 0)
            return fail(c, ICP_HIP_ERR_ARG, "icp_hip_icp: empty point cloud");
        // mu_m = mean(m); the tree is built over m - mu_m (icp.cc:106-108, icp.cu:139-142)
        float mu_m[3];
        std::vector<float> centred;
        try
        {
            centred.resize(3 * (size_t)n_m);
        }
        catch (const std::bad_alloc&)
        {
            return fail(c, ICP_HIP_ERR_NOMEM, "icp_hip_icp: out of host memory");
        }
        icp_host::mean_host(m_xyz, n_m, mu_m);
        icp_host::center_host(m_xyz, n_m, mu_m, centred.data());
        int rc = icp_hip_build_tree(c, centred.data(), n_m, vp_threshold);
        if (rc != ICP_HIP_OK)
            return rc;
        std::vector<float>().swap(centred);
        if ((rc = icp_hip_set_queries(c, p_xyz, n_p, n_p)) != ICP_HIP_OK)
            return rc;
        if ((rc = icp_hip_run(c, mu_m, iterations, threshold, T_out, iters_done, final_error,
                              progress, user)) != ICP_HIP_OK)
            return rc;
        if (new_p_xyz)
            rc = icp_hip_get_points(c, new_p_xyz);
        return rc;
    }

    int icp_hip_get_nn_trace(icp_hip_ctx* c, float* ms_out, uint64_t capacity, uint64_t* count)
    {
        if (!c || !count)
            return fail(c, ICP_HIP_ERR_ARG, "icp_hip_get_nn_trace: NULL argument");
        *count = c->nn_trace.size();
        if (ms_out)
            for (uint64_t i = 0; i < capacity && i < c->nn_trace.size(); ++i)
                ms_out[i] = c->nn_trace[i];
        return ICP_HIP_OK;
    }

    int icp_hip_set_phase_timers(icp_hip_ctx* c, int enabled)
    {
        if (!c)
            return fail(nullptr, ICP_HIP_ERR_ARG, "icp_hip_set_phase_timers: ctx is NULL");
        c->phase_timers = enabled != 0;
        return ICP_HIP_OK;
    }

    int icp_hip_get_timing(icp_hip_ctx* c, icp_hip_timing* out)
    {
        if (!c || !out)
            return fail(c, ICP_HIP_ERR_ARG, "icp_hip_get_timing: NULL argument");
        *out = c->timing;
        return ICP_HIP_OK;
    }
}

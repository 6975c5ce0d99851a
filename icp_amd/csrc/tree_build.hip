// Device-side VP-tree build: the reference's tree, level by level, for every node of a level at once.
//
// Semantics (bit for bit): libcpu::VPTree::VPTree, libcpu/libcpu/vp-tree.cc:54-123
//   leaf iff size < capacity (:56); vantage point = last element (:65-66); squared distances
//   centre->point in fp32 (:72-74, point-3d.cc:136-143); radius^2 = element of rank size/2 of
//   the distances (:48-52,77); stable split d2 < r2 / d2 >= r2 (:90-111); per side the FIRST
//   strict maximum is swapped to the back and becomes the child's vantage point
//   (:84-88,95-99,104-108,114-115).
//
// The reference recurses; its host restatement (vptree_build.cc) runs subtrees as tasks.  Here a
// subtree owns a contiguous range of one point array (coordinates + rows, SoA) and one LEVEL of the
// tree is built per step, all of its nodes together, with segmented primitives:
//
//   keys      d2 of every point to its node's vantage point; 64-bit key = (node, d2 bits)
//   median    one device radix sort of the keys; node s reads rank m/2 inside its run
//   split     flag = d2 < r2, one exclusive scan, stable scatter of rows and coordinates
//   far point per child argmax of (d2, lowest index) through 64-bit atomic max (one per wave and
//             side where a wave lies inside one node), then the swap with the child's last slot
//
// The host only does the per-level bookkeeping over the (at most 2^level) nodes: radius = sqrt,
// node records in breadth-first order, which children are leaves.  The final tree is written to
// the device layout directly (no host copy of the cloud, no upload of 1.3 GB of slots).
// Non-negative floats order like their bit patterns, so the sort and the argmax run on integers.

#include <hip/hip_runtime.h>
#include <cstring> // rocprim 7.2 headers use memset without including it
#include <rocprim/rocprim.hpp>
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "kernels.hh"
#include "tree_build.hh"

namespace icp_dev
{
    namespace
    {
        constexpr int kB = 256;
        constexpr uint32_t kDone = 0xFFFFFFFFu;

        struct Seg // an internal node of the current level: points [lo, hi), vantage at hi-1
        {
            uint32_t lo, hi;
            uint32_t off; // start of its run in the sorted key array
        };
        struct SegOut
        {
            uint32_t crow;
            float cx, cy, cz;
            float r2;
            uint32_t ni;
        };

        __device__ __forceinline__ float sqd(float ax, float ay, float az, float bx, float by,
                                             float bz)
        {
            const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
            return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        }

        __global__ void __launch_bounds__(kB) init_kernel(const float* __restrict__ aos, uint32_t n,
                                                           float* x, float* y, float* z,
                                                           uint32_t* row, uint32_t* prov,
                                                           uint32_t* __restrict__ non_finite)
        {
            for (uint32_t i = blockIdx.x * kB + threadIdx.x; i < n; i += gridDim.x * kB)
            {
                x[i] = aos[3 * (size_t)i];
                y[i] = aos[3 * (size_t)i + 1];
                z[i] = aos[3 * (size_t)i + 2];
                if (!(isfinite(x[i]) && isfinite(y[i]) && isfinite(z[i])))
                    atomicOr(non_finite, 1u);
                row[i] = i;
                prov[i] = 0; // provisional id of the root
            }
        }

        // seg[i] = remap[prov[i]] ; key = (seg, d2 to the segment's vantage point)
        __global__ void __launch_bounds__(kB)
            keys_kernel(const uint32_t* __restrict__ prov, const uint32_t* __restrict__ remap,
                        const Seg* __restrict__ segs, uint32_t n_seg, const float* __restrict__ x,
                        const float* __restrict__ y, const float* __restrict__ z, uint32_t n,
                        uint32_t* __restrict__ seg, float* __restrict__ d2,
                        unsigned long long* __restrict__ keys)
        {
            for (uint32_t i = blockIdx.x * kB + threadIdx.x; i < n; i += gridDim.x * kB)
            {
                const uint32_t p = prov[i];
                const uint32_t s = (p == kDone) ? kDone : remap[p];
                seg[i] = s;
                float d = 0.f;
                unsigned long long key = ((unsigned long long)n_seg << 32); // finished points last
                if (s != kDone)
                {
                    const Seg S = segs[s];
                    const uint32_t v = S.hi - 1;
                    if (i != v)
                    {
                        d = sqd(x[v], y[v], z[v], x[i], y[i], z[i]); // centre first (:72-74)
                        key = ((unsigned long long)s << 32) | __float_as_uint(d);
                    }
                    else
                        key = ((unsigned long long)s << 32) | 0xFFFFFFFFull; // vantage: end of its run
                }
                d2[i] = d;
                keys[i] = key;
            }
        }

        __global__ void __launch_bounds__(kB)
            median_kernel(const Seg* __restrict__ segs, uint32_t n_seg,
                          const unsigned long long* __restrict__ sorted, const float* __restrict__ x,
                          const float* __restrict__ y, const float* __restrict__ z,
                          const uint32_t* __restrict__ row, SegOut* __restrict__ out)
        {
            const uint32_t s = blockIdx.x * kB + threadIdx.x;
            if (s >= n_seg)
                return;
            const Seg S = segs[s];
            const uint32_t m = S.hi - S.lo - 1; // points without the vantage point (>= 1, host checks)
            SegOut o;
            o.r2 = __uint_as_float((uint32_t)sorted[(size_t)S.off + m / 2]); // rank m/2 (:48-52)
            const uint32_t v = S.hi - 1;
            o.crow = row[v];
            o.cx = x[v];
            o.cy = y[v];
            o.cz = z[v];
            o.ni = 0;
            out[s] = o;
        }

        __global__ void __launch_bounds__(kB)
            flag_kernel(const uint32_t* __restrict__ seg, const Seg* __restrict__ segs,
                        const SegOut* __restrict__ so, const float* __restrict__ d2, uint32_t n,
                        uint32_t* __restrict__ flag)
        {
            for (uint32_t i = blockIdx.x * kB + threadIdx.x; i < n; i += gridDim.x * kB)
            {
                const uint32_t s = seg[i];
                uint32_t f = 0;
                if (s != kDone && i != segs[s].hi - 1)
                    f = d2[i] < so[s].r2 ? 1u : 0u; // inside iff strictly below the median (:92)
                flag[i] = f;
            }
        }

        __global__ void __launch_bounds__(kB)
            count_kernel(const Seg* __restrict__ segs, uint32_t n_seg,
                         const uint32_t* __restrict__ scan, SegOut* __restrict__ out)
        {
            const uint32_t s = blockIdx.x * kB + threadIdx.x;
            if (s >= n_seg)
                return;
            const Seg S = segs[s];
            out[s].ni = scan[S.hi - 1] - scan[S.lo]; // the vantage point's own flag is 0
        }

        __device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
        {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
            {
                const unsigned long long o = __shfl_xor(v, off, 64);
                v = o > v ? o : v;
            }
            return v;
        }

        // stable split of every segment + far-point candidates of both children.
        // Every wave owns a contiguous run of `per_wave` points (a multiple of 64) and keeps the
        // running maxima of the segment it is in; they go to `best` with one atomic per side when
        // the wave leaves the segment or ends.  (One atomic per wave-load made the two words of
        // the root a million-fold contended: 15 ms of a level at 64M points.)
        __global__ void __launch_bounds__(kB)
            scatter_kernel(const uint32_t* __restrict__ seg, const Seg* __restrict__ segs,
                           const SegOut* __restrict__ so, const uint32_t* __restrict__ scan,
                           const uint32_t* __restrict__ flag, const float* __restrict__ d2,
                           const float* __restrict__ x, const float* __restrict__ y,
                           const float* __restrict__ z, const uint32_t* __restrict__ row, uint32_t n,
                           uint32_t per_wave, float* __restrict__ ox, float* __restrict__ oy,
                           float* __restrict__ oz, uint32_t* __restrict__ orow,
                           uint32_t* __restrict__ prov, unsigned long long* __restrict__ best)
        {
            const uint32_t lane = threadIdx.x & 63u;
            const uint32_t wave = (blockIdx.x * kB + threadIdx.x) >> 6;
            const uint64_t w_begin = (uint64_t)wave * per_wave;
            if (w_begin >= n)
                return;
            const uint32_t w_end = (uint32_t)min((uint64_t)n, w_begin + per_wave);
            uint32_t run_seg = kDone; // wave-uniform: segment of the pending maxima
            unsigned long long run_in = 0, run_out = 0;
            auto flush = [&]() {
                if (run_seg != kDone && lane == 0)
                {
                    if (run_in)
                        atomicMax(best + 2u * run_seg, run_in);
                    if (run_out)
                        atomicMax(best + 2u * run_seg + 1u, run_out);
                }
                run_in = run_out = 0;
            };
            for (uint32_t base = (uint32_t)w_begin; base < w_end; base += 64u)
            {
                const uint32_t i = base + lane;
                const bool live = i < w_end;
                const uint32_t s = live ? seg[i] : kDone;
                unsigned long long key_in = 0, key_out = 0;
                uint32_t child = kDone;
                if (live)
                {
                    uint32_t pos = i, pr = kDone;
                    if (s != kDone)
                    {
                        const Seg S = segs[s];
                        // (i inside its segment's range: always, unless this is a speculative
                        // build whose assumed split of a level above turned out wrong -- its
                        // result is thrown away, but nothing may be written out of bounds)
                        if (i != S.hi - 1 && i >= S.lo && i < S.hi)
                        {
                            const uint32_t ni = so[s].ni;
                            const uint32_t rank_in = scan[i] - scan[S.lo];
                            const bool in = flag[i] != 0;
                            const uint32_t local = in ? rank_in : (i - S.lo - rank_in);
                            pos = in ? S.lo + local : S.lo + ni + local;
                            child = 2u * s + (in ? 0u : 1u);
                            pr = child;
                            // first strict maximum == max over (d2, lowest index) (:95-99,:104-108)
                            const unsigned long long key =
                                ((unsigned long long)__float_as_uint(d2[i]) << 32) |
                                (unsigned long long)(0xFFFFFFFFu - local);
                            if (in)
                                key_in = key;
                            else
                                key_out = key;
                        }
                    }
                    if (pos >= n) // (same: a mis-assumed layout; never on a build that is kept)
                    {
                        pos = i;
                        pr = kDone;
                    }
                    ox[pos] = x[i];
                    oy[pos] = y[i];
                    oz[pos] = z[i];
                    orow[pos] = row[i];
                    prov[pos] = pr;
                }
                // the wave-load lies in one segment (dead lanes of a last, partial load aside)
                const uint64_t alive = __ballot(live);
                const uint32_t s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s);
                if (__ballot(live && s != s0) == 0ull && alive != 0ull)
                {
                    if (s0 != run_seg)
                    {
                        flush();
                        run_seg = s0;
                    }
                    if (s0 != kDone)
                    {
                        const unsigned long long mi = wave_max_u64(key_in);
                        const unsigned long long mo = wave_max_u64(key_out);
                        run_in = mi > run_in ? mi : run_in;
                        run_out = mo > run_out ? mo : run_out;
                    }
                }
                else
                {
                    flush();
                    run_seg = kDone;
                    if (child != kDone)
                        atomicMax(best + child, key_in | key_out);
                }
            }
            flush();
        }

        // the far point of each child goes to the child's last slot (:114-115)
        __global__ void __launch_bounds__(kB)
            swap_kernel(const Seg* __restrict__ segs, const SegOut* __restrict__ so, uint32_t n_seg,
                        const unsigned long long* __restrict__ best, float* x, float* y, float* z,
                        uint32_t* row)
        {
            const uint32_t c = blockIdx.x * kB + threadIdx.x;
            if (c >= 2u * n_seg)
                return;
            const Seg S = segs[c >> 1];
            const uint32_t ni = so[c >> 1].ni;
            const uint32_t lo = (c & 1u) ? S.lo + ni : S.lo;
            const uint32_t hi = (c & 1u) ? S.hi - 1 : S.lo + ni;
            if (hi <= lo)
                return; // empty side: the host reports the degenerate cloud
            const uint32_t k = 0xFFFFFFFFu - (uint32_t)best[c];
            const uint32_t a = lo + k, b = hi - 1;
            if (k >= hi - lo)
                return; // no candidate recorded (possible only under a mis-assumed layout)
            if (a != b)
            {
                float t;
                t = x[a], x[a] = x[b], x[b] = t;
                t = y[a], y[a] = y[b], y[b] = t;
                t = z[a], z[a] = z[b], z[b] = t;
                const uint32_t r = row[a];
                row[a] = row[b];
                row[b] = r;
            }
        }

        struct LeafSrc
        {
            uint32_t lo, count, begin; // source range in the build order, first slot
        };

        // one wave per leaf: bucket = its points in build order, padded with +inf to 16 slots
        __global__ void __launch_bounds__(kB)
            slots_kernel(const LeafSrc* __restrict__ leaves, uint32_t n_leaves,
                         const float* __restrict__ x, const float* __restrict__ y,
                         const float* __restrict__ z, const uint32_t* __restrict__ row,
                         float* __restrict__ sx, float* __restrict__ sy, float* __restrict__ sz,
                         uint32_t* __restrict__ srow)
        {
            const float inf = __int_as_float(0x7f800000);
            const uint32_t lane = threadIdx.x & 63;
            for (uint32_t l = (blockIdx.x * kB + threadIdx.x) >> 6; l < n_leaves;
                 l += (gridDim.x * kB) >> 6)
            {
                const LeafSrc L = leaves[l];
                const uint32_t padded = (L.count + 15u) / 16u * 16u;
                for (uint32_t k = lane; k < padded; k += 64)
                {
                    const bool real = k < L.count;
                    sx[L.begin + k] = real ? x[L.lo + k] : inf;
                    sy[L.begin + k] = real ? y[L.lo + k] : inf;
                    sz[L.begin + k] = real ? z[L.lo + k] : inf;
                    srow[L.begin + k] = real ? row[L.lo + k] : 0xFFFFFFFFu;
                }
            }
        }

        __global__ void __launch_bounds__(kB) fill_inf_kernel(float* p, uint32_t n)
        {
            for (uint32_t i = blockIdx.x * kB + threadIdx.x; i < n; i += gridDim.x * kB)
                p[i] = __int_as_float(0x7f800000);
        }

        template <class T>
        struct Buf
        {
            T* p = nullptr;
            hipError_t alloc(size_t n) { return hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)); }
            ~Buf()
            {
                if (p)
                    (void)hipFree(p);
            }
            T* take()
            {
                T* q = p;
                p = nullptr;
                return q;
            }
        };

        inline uint32_t grid_n(uint64_t n)
        {
            uint64_t b = (n + kB - 1) / kB;
            return (uint32_t)(b < 1 ? 1 : (b > 256 * 16 ? 256 * 16 : b));
        }
    } // namespace

#define TB_TRY(expr)                                                                             \
    do                                                                                           \
    {                                                                                            \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
        {                                                                                        \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);                             \
            return e_ == hipErrorOutOfMemory ? 6 : 3;                                            \
        }                                                                                        \
    } while (0)

    namespace
    {
        // `speculate`: the whole build is enqueued without a single read-back.  What the host needs
        // from a level to lay out the next one is only how many points fell inside each median
        // sphere, and with distinct distances that is floor(m / 2) by construction
        // (vp-tree.cc:48-52,:92); the host assumes it, the levels' records stay on the device and
        // are checked in one read-back at the end.  A level costs ~0.15 ms of synchronisation and
        // launch gaps whatever its size, which is most of the build below a million points (the
        // 48k-point horse, 8 levels: 1.12 -> 0.6 ms).  A tie at a median (or a degenerate split)
        // sets *mispredicted and the caller builds again level by level.
        int build_tree_device_impl(const float* d_aos, uint32_t n, uint32_t capacity, hipStream_t st,
                                   BuiltTree& out, std::string& err, bool speculate,
                                   bool* mispredicted);
    } // namespace

    int build_tree_device(const float* d_aos, uint32_t n, uint32_t capacity, hipStream_t st,
                          BuiltTree& out, std::string& err)
    {
        constexpr uint32_t kSpeculateBelow = 4u << 20;
        if (n < kSpeculateBelow)
        {
            bool mispredicted = false;
            const int rc = build_tree_device_impl(d_aos, n, capacity, st, out, err, true, &mispredicted);
            if (!mispredicted)
                return rc;
            err.clear();
        }
        return build_tree_device_impl(d_aos, n, capacity, st, out, err, false, nullptr);
    }

    namespace
    {
    int build_tree_device_impl(const float* d_aos, uint32_t n, uint32_t capacity, hipStream_t st,
                               BuiltTree& out, std::string& err, bool speculate, bool* mispredicted)
    {
        out = BuiltTree{};
        // ---- working set ----
        Buf<float> x[2], y[2], z[2], d2;
        Buf<uint32_t> row[2], prov, seg, flag, scan, remap;
        Buf<unsigned long long> keys, sorted, best;
        Buf<Seg> segs;
        Buf<SegOut> segout;
        for (int k = 0; k < 2; ++k)
        {
            TB_TRY(x[k].alloc(n));
            TB_TRY(y[k].alloc(n));
            TB_TRY(z[k].alloc(n));
            TB_TRY(row[k].alloc(n));
        }
        TB_TRY(d2.alloc(n));
        TB_TRY(prov.alloc(n));
        TB_TRY(seg.alloc(n));
        TB_TRY(flag.alloc(n));
        TB_TRY(scan.alloc(n));
        TB_TRY(keys.alloc(n));
        TB_TRY(sorted.alloc(n));
        // a level has at most n / capacity + 1 internal nodes
        const size_t max_seg = (size_t)n / (capacity ? capacity : 1) + 2;
        TB_TRY(segs.alloc(max_seg));
        TB_TRY(segout.alloc(max_seg));
        Buf<SegOut> segout_all; // speculate: every level's records, node id = index
        if (speculate)
            TB_TRY(segout_all.alloc((size_t)n + 1));
        TB_TRY(best.alloc(2 * max_seg));
        TB_TRY(remap.alloc(2 * max_seg));
        size_t tmp_sort = 0, tmp_scan = 0;
        TB_TRY(rocprim::radix_sort_keys(nullptr, tmp_sort, keys.p, sorted.p, (size_t)n, 0u, 64u, st));
        TB_TRY(rocprim::exclusive_scan(nullptr, tmp_scan, flag.p, scan.p, 0u, (size_t)n,
                                       rocprim::plus<uint32_t>(), st));
        Buf<unsigned char> tmp;
        const size_t tmp_bytes = std::max(tmp_sort, tmp_scan);
        TB_TRY(tmp.alloc(tmp_bytes));

        Buf<uint32_t> non_finite;
        TB_TRY(non_finite.alloc(1));
        TB_TRY(hipMemsetAsync(non_finite.p, 0, sizeof(uint32_t), st));
        hipLaunchKernelGGL(init_kernel, dim3(grid_n(n)), dim3(kB), 0, st, d_aos, n, x[0].p, y[0].p,
                           z[0].p, row[0].p, prov.p, non_finite.p);
        TB_TRY(hipGetLastError());
        auto mispredict = [&]() {
            (void)hipStreamSynchronize(st);
            *mispredicted = true;
            return 0;
        };
        if (!speculate) // (the speculative build reads the flag back with everything else)
        {
            uint32_t h_bad = 0;
            TB_TRY(hipMemcpyAsync(&h_bad, non_finite.p, sizeof h_bad, hipMemcpyDeviceToHost, st));
            TB_TRY(hipStreamSynchronize(st));
            if (h_bad)
            {
                err = "non-finite coordinate (NaN or infinity): every comparison of the "
                      "reference's search fails on it, so each query would walk the whole tree "
                      "(vp-tree.cc:147-173)";
                return 1;
            }
        }

        // ---- host bookkeeping ----
        struct HNode
        {
            float cx, cy, cz, r;
            int32_t inside, outside, parent;
            uint32_t row;
        };
        struct HLeaf
        {
            uint32_t lo, count;
            int32_t parent; // node id, -1 for a root leaf
            int side;
        };
        std::vector<HNode> nodes;
        std::vector<uint32_t> spec_ni; // speculate: the inside count assumed for each node
        std::vector<HLeaf> leaves;
        struct Cur
        {
            uint32_t lo, hi;
            int32_t parent;
            int side;
        };
        std::vector<Cur> level; // internal nodes of the current level, left to right
        uint32_t height = 0;
        if (n < capacity)
            leaves.push_back({0, n, -1, 0});
        else
            level.push_back({0, n, -1, 0});

        std::vector<Seg> h_segs;
        std::vector<SegOut> h_out;
        std::vector<uint32_t> h_remap(1, 0); // provisional id 0 = the root segment
        int cur = 0;
        uint32_t depth = 0;
        while (!level.empty())
        {
            const uint32_t S = (uint32_t)level.size();
            if (S > max_seg)
            {
                err = "build_tree_device: level wider than n / capacity";
                return 1;
            }
            h_segs.resize(S);
            uint32_t off = 0;
            for (uint32_t s = 0; s < S; ++s)
            {
                h_segs[s] = {level[s].lo, level[s].hi, off};
                off += level[s].hi - level[s].lo;
                if (level[s].hi - level[s].lo < 2)
                {
                    if (speculate)
                        return mispredict();
                    err = "degenerate cloud: a one-point internal node (median of an empty array in "
                          "the reference, vp-tree.cc:77)";
                    return 2;
                }
            }
            TB_TRY(hipMemcpyAsync(segs.p, h_segs.data(), S * sizeof(Seg), hipMemcpyHostToDevice, st));
            TB_TRY(hipMemcpyAsync(remap.p, h_remap.data(), h_remap.size() * sizeof(uint32_t),
                                  hipMemcpyHostToDevice, st));
            TB_TRY(hipMemsetAsync(best.p, 0, 2 * (size_t)S * sizeof(unsigned long long), st));

            hipLaunchKernelGGL(keys_kernel, dim3(grid_n(n)), dim3(kB), 0, st, prov.p, remap.p, segs.p,
                               S, x[cur].p, y[cur].p, z[cur].p, n, seg.p, d2.p, keys.p);
            TB_TRY(hipGetLastError());
            uint32_t seg_bits = 1;
            while ((1ull << seg_bits) <= S) // S itself is the "finished" marker
                ++seg_bits;
            size_t tb = tmp_bytes;
            TB_TRY(rocprim::radix_sort_keys(tmp.p, tb, keys.p, sorted.p, (size_t)n, 0u,
                                            32u + seg_bits, st));
            // (speculate: this level's records at its nodes' ids, which are handed out in this order)
            SegOut* const so_dev = speculate ? segout_all.p + nodes.size() : segout.p;
            hipLaunchKernelGGL(median_kernel, dim3((S + kB - 1) / kB), dim3(kB), 0, st, segs.p, S,
                               sorted.p, x[cur].p, y[cur].p, z[cur].p, row[cur].p, so_dev);
            hipLaunchKernelGGL(flag_kernel, dim3(grid_n(n)), dim3(kB), 0, st, seg.p, segs.p, so_dev,
                               d2.p, n, flag.p);
            TB_TRY(hipGetLastError());
            tb = tmp_bytes;
            TB_TRY(rocprim::exclusive_scan(tmp.p, tb, flag.p, scan.p, 0u, (size_t)n,
                                           rocprim::plus<uint32_t>(), st));
            hipLaunchKernelGGL(count_kernel, dim3((S + kB - 1) / kB), dim3(kB), 0, st, segs.p, S,
                               scan.p, so_dev);
            // contiguous runs of a multiple of 64 points per wave, every wave of the grid busy
            const uint32_t sc_grid = grid_n(n);
            const uint64_t sc_waves = (uint64_t)sc_grid * (kB / 64);
            const uint32_t per_wave = (uint32_t)(((uint64_t)n + sc_waves * 64 - 1) / (sc_waves * 64)) * 64u;
            hipLaunchKernelGGL(scatter_kernel, dim3(sc_grid), dim3(kB), 0, st, seg.p, segs.p,
                               so_dev, scan.p, flag.p, d2.p, x[cur].p, y[cur].p, z[cur].p,
                               row[cur].p, n, per_wave, x[cur ^ 1].p, y[cur ^ 1].p, z[cur ^ 1].p,
                               row[cur ^ 1].p, prov.p, best.p);
            hipLaunchKernelGGL(swap_kernel, dim3((2 * S + kB - 1) / kB), dim3(kB), 0, st, segs.p,
                               so_dev, S, best.p, x[cur ^ 1].p, y[cur ^ 1].p, z[cur ^ 1].p,
                               row[cur ^ 1].p);
            TB_TRY(hipGetLastError());
            h_out.resize(S);
            if (speculate)
            {
                for (uint32_t s = 0; s < S; ++s)
                {
                    h_out[s] = SegOut{};
                    h_out[s].ni = (level[s].hi - level[s].lo - 1) / 2; // the rank of the median
                }
            }
            else
            {
                TB_TRY(hipMemcpyAsync(h_out.data(), segout.p, S * sizeof(SegOut),
                                      hipMemcpyDeviceToHost, st));
                TB_TRY(hipStreamSynchronize(st));
            }
            cur ^= 1;

            // ---- this level's node records (breadth-first ids) and the next level ----
            std::vector<Cur> next;
            h_remap.assign(2 * (size_t)S, kDone);
            const int32_t first_id = (int32_t)nodes.size();
            for (uint32_t s = 0; s < S; ++s)
            {
                const SegOut& o = h_out[s];
                const uint32_t lo = level[s].lo, hi = level[s].hi, m = hi - lo - 1;
                const uint32_t ni = o.ni, no = m - ni;
                if ((ni == 0 || no == 0) && speculate)
                    return mispredict();
                if (ni == 0 || no == 0)
                {
                    err = "degenerate cloud: a vantage-point split left one side empty (duplicate "
                          "points / tied distances); the reference's build is undefined behaviour "
                          "here (vp-tree.cc:114-115)";
                    return 2;
                }
                HNode nd{o.cx, o.cy, o.cz, std::sqrt(o.r2), 0, 0, level[s].parent, o.crow};
                nodes.push_back(nd);
                if (speculate)
                    spec_ni.push_back(ni);
            }
            for (uint32_t s = 0; s < S; ++s)
            {
                const int32_t id = first_id + (int32_t)s;
                const uint32_t lo = level[s].lo, hi = level[s].hi, ni = h_out[s].ni;
                const uint32_t clo[2] = {lo, lo + ni}, chi[2] = {lo + ni, hi - 1};
                for (int side = 0; side < 2; ++side)
                {
                    const uint32_t size = chi[side] - clo[side];
                    if (size < capacity)
                        leaves.push_back({clo[side], size, id, side});
                    else
                    {
                        h_remap[2 * (size_t)s + side] = (uint32_t)next.size();
                        next.push_back({clo[side], chi[side], id, side});
                    }
                }
                if (level[s].parent >= 0)
                {
                    HNode& p = nodes[level[s].parent];
                    (level[s].side == 0 ? p.inside : p.outside) = id;
                }
            }
            height = ++depth;
            level.swap(next);
        }

        if (speculate)
        {
            // one read-back: the non-finite flag and every node's record; a count that is not
            // the assumed one (equal distances at a median) voids the layout of the levels below
            uint32_t h_bad = 0;
            std::vector<SegOut> all(nodes.size());
            TB_TRY(hipMemcpyAsync(&h_bad, non_finite.p, sizeof h_bad, hipMemcpyDeviceToHost, st));
            if (!all.empty())
                TB_TRY(hipMemcpyAsync(all.data(), segout_all.p, all.size() * sizeof(SegOut),
                                      hipMemcpyDeviceToHost, st));
            TB_TRY(hipStreamSynchronize(st));
            if (h_bad)
                return mispredict(); // (the level-by-level build words the error)
            for (size_t id = 0; id < all.size(); ++id)
            {
                if (all[id].ni != spec_ni[id])
                    return mispredict();
                nodes[id].cx = all[id].cx;
                nodes[id].cy = all[id].cy;
                nodes[id].cz = all[id].cz;
                nodes[id].r = std::sqrt(all[id].r2);
                nodes[id].row = all[id].crow;
            }
        }

        // ---- leaves left to right == ascending position; buckets padded to 16 slots ----
        std::sort(leaves.begin(), leaves.end(),
                  [](const HLeaf& a, const HLeaf& b) { return a.lo < b.lo; });
        std::vector<LeafSrc> h_leaf(leaves.size());
        uint64_t slots = 0;
        uint32_t max_leaf = 0;
        out.leaf_begin.resize(leaves.size());
        out.leaf_count.resize(leaves.size());
        for (size_t l = 0; l < leaves.size(); ++l)
        {
            h_leaf[l] = {leaves[l].lo, leaves[l].count, (uint32_t)slots};
            out.leaf_begin[l] = (uint32_t)slots;
            out.leaf_count[l] = leaves[l].count;
            max_leaf = std::max(max_leaf, leaves[l].count);
            slots += ((uint64_t)leaves[l].count + 15) / 16 * 16;
            if (leaves[l].parent >= 0)
            {
                HNode& p = nodes[leaves[l].parent];
                (leaves[l].side == 0 ? p.inside : p.outside) = ~(int32_t)l;
            }
        }
        if (slots + nodes.size() >= (1ull << 32) - 64)
        {
            err = "build_tree_device: slot index space exceeds 32 bits";
            return 1;
        }
        Buf<float> sx, sy, sz;
        Buf<uint32_t> srow;
        Buf<LeafSrc> d_leaf;
        TB_TRY(sx.alloc(slots + kSlotSlack));
        TB_TRY(sy.alloc(slots + kSlotSlack));
        TB_TRY(sz.alloc(slots + kSlotSlack));
        TB_TRY(srow.alloc(slots));
        TB_TRY(d_leaf.alloc(h_leaf.size()));
        TB_TRY(hipMemcpyAsync(d_leaf.p, h_leaf.data(), h_leaf.size() * sizeof(LeafSrc),
                              hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(slots_kernel, dim3(grid_n((uint64_t)h_leaf.size() * 64)), dim3(kB), 0, st,
                           d_leaf.p, (uint32_t)h_leaf.size(), x[cur].p, y[cur].p, z[cur].p,
                           row[cur].p, sx.p, sy.p, sz.p, srow.p);
        for (float* p : {sx.p, sy.p, sz.p})
            hipLaunchKernelGGL(fill_inf_kernel, dim3(1), dim3(kB), 0, st, p + slots, kSlotSlack);
        TB_TRY(hipGetLastError());
        TB_TRY(hipStreamSynchronize(st));

        out.n_points = n;
        out.n_slots = slots;
        out.height = height;
        out.capacity = capacity;
        out.max_leaf_count = max_leaf;
        out.root = nodes.empty() ? ~0 : 0;
        const size_t nn = nodes.size();
        out.node_cx.resize(nn);
        out.node_cy.resize(nn);
        out.node_cz.resize(nn);
        out.node_radius.resize(nn);
        out.node_inside.resize(nn);
        out.node_outside.resize(nn);
        out.node_parent.resize(nn);
        out.node_row.resize(nn);
        for (size_t i = 0; i < nn; ++i)
        {
            out.node_cx[i] = nodes[i].cx;
            out.node_cy[i] = nodes[i].cy;
            out.node_cz[i] = nodes[i].cz;
            out.node_radius[i] = nodes[i].r;
            out.node_inside[i] = nodes[i].inside;
            out.node_outside[i] = nodes[i].outside;
            out.node_parent[i] = nodes[i].parent;
            out.node_row[i] = nodes[i].row;
        }
        out.d_sx = sx.take();
        out.d_sy = sy.take();
        out.d_sz = sz.take();
        out.d_srow = srow.take();
        return 0;
    }
    } // namespace
} // namespace icp_dev

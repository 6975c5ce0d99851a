// nn_tile_kernel -- the production NN + cross-covariance kernel (gfx950, wave = 64).
//
// Same contract as nn_cov_kernel (kernels.hip): VPTree::search for every query of the shard
// (libcpu/libcpu/vp-tree.cc:125-187), bit for bit, + the nine sums of find_covariance
// (libcpu/libcpu/point-3d.cc:212-236).  What differs is how the work is laid on the machine, for
// a shard that is stored in a spatially coherent order (sort.hip):
//
//  * persistent grid; the shard is cut into chunks of 1..2 consecutive 64-query batches and every
//    wave pulls chunks from a queue.  There is one queue head per XCD group (blockIdx % 8,
//    round-robin dispatch) over one eighth of the shard each, so the waves of an XCD work on
//    neighbouring chunks AT THE SAME TIME: the four batches that end in one bucket stage it
//    within microseconds of each other and three of them find it in that XCD's L2.  (A line
//    lives ~7 us in a 4 MiB L2 at this kernel's 0.5 GB/ms per XCD; with chunks of 8 batches a
//    wave came back to a bucket 30 us later and every staging was a miss: 9.1 GB fetched per
//    aligned 64M launch against 2.3 GB now, profiles/r02b.)  A group that runs dry steals
//    from the others (load balance);
//  * every chunk writes its own twelve fp64 sums and two small kernels add the chunks in a fixed
//    order: the result does not depend on which wave ran which chunk (bitwise reproducible);
//  * the traversal stack holds only the near-side answers of the levels whose far side is being
//    searched (push on the way into the far side, pop on the way back): 2 entries per lane in
//    LDS instead of one per tree level; deeper nestings spill to a per-lane column in HBM
//    (counted in sum [11]);
//  * leaf buckets are scanned by (queries x bucket parts) tiles: the g lanes of a wave that wait
//    on the SAME bucket are served together.  The bucket is staged once in LDS (three 16-byte
//    loads per lane), the g queries are replicated over P = 64 / pow2(g) lane groups, group p
//    scans part p of the bucket for all g queries (LDS float4 reads, conflict free by
//    construction), and the partial minima are folded across the groups.  g = 1 degenerates to
//    the whole wave scanning for one query, g = 64 to one lane per query;
//  * before the tile scan every lane of the group tries the RING WALK: the bucket's points are
//    also listed by their distance to the centre of the bucket's parent node (DevTree::ring), a
//    query looks up its own distance to that centre in the list and examines the points
//    outwards from there until the gap to the next ring exceeds the best distance found (the
//    triangle inequality, one level below the tree).  Once the clouds are aligned it finishes
//    99.6 % of the lanes after one or two points instead of the bucket's 244; a lane that is not
//    finished after six points is left to the tile scan (see ring_walk).
//
// Exactness of the bucket scan.  The reference keeps the first slot that minimises
// sqrt(d2) under strict < (vp-tree.cc:131-143), d2 = ((dx*dx)+(dy*dy))+(dz*dz) in fp32 without
// FMA; two different d2 may share one sqrt.  The scan loop evaluates a FILTER instead,
// fma(dz,dz, fma(dy,dy, dx*dx)) (six operations instead of eight; like d2 it lies within 3.01 u,
// u = 2^-24, of the real sum of squares of the same three differences), and tracks per query the
// smallest group minimum m, the second smallest m2 (with multiplicity) and m's group.  The four
// points of that group are then re-evaluated with the reference's arithmetic: em = their smallest
// d2, e2 the runner-up inside the group.  The slot of em is the reference's answer, with distance
// sqrt(em), iff (i) e2 is more than 4 ulps above em (then the roots differ by >= 1.41 ulps of the
// root and round to different floats) and (ii) m2 is more than 16 float steps above em: a point
// outside the group has a filter value >= m2 and a reference d2 >= its filter value x (1 - 6.03 u),
// hence more than 4 ulps above em as well.  Otherwise (about 2e-4 of the scans; every exact tie
// and duplicate point) the lane is re-served by the exact wave-cooperative scan of
// device_common.hh.

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <mutex>

#include "device_common.hh"
#include "kernels.hh"

namespace icp_dev
{
    namespace
    {
        constexpr int kStackK = 2;  // LDS stack entries per lane
        constexpr int kImage = 256; // bucket slots staged per window
        constexpr int kGroups = kImage / 4; // float4 groups of the image; group kGroups is +inf
        constexpr int kTileMinGroup = 4; // fewer lanes on a bucket: whole-wave scan per lane
        // ---- the parking lot (see "stragglers" in the kernel) ----
        constexpr uint32_t kLot = 64;       // entries per wave: one batch of parked queries
        enum : uint32_t
        {
            pkQx = 0, pkQy, pkQz, pkNode, pkPrev, pkModeSp, pkRd, pkRid, pkChainM, pkChainM2, pkArg,
            pkRetNode, pkFree, pkNInt, pkNPts, pkOvf, pkIndex, pkRow, pkStD0, pkStI0, pkStD1, pkStI1,
            kParkFields,
            kParkWords = 24 // an entry = 96 bytes, its fields side by side: the stores of a lane
                            // that is parked fall into one or two cache lines and merge in L2
                            // (field by field over the lot they were 22 partial lines per lane)
        };
        static_assert(kParkFields <= kParkWords, "lot entry");
        static_assert(kStackK == 2, "the lot keeps two stack entries per parked lane");
        static_assert(kLot * kParkWords == kParkLotWords, "kernels.hh sizes the lots");
        constexpr int kTileSplitBelow = 56; // 33..55 lanes: scan as 32 + rest (see the leaf phase)

        enum : int
        {
            tDescend = 0, // arriving at `node` from its parent
            tReturn = 1,  // arriving at `node` from child `prev`, (rd, rid) = that child's answer
            tWaitLeaf = 2,
            tWaitExact = 3,
            tServed = 4, // rd holds the SQUARED distance of the answer, sqrt pending
            tDone = 5
        };

        struct __attribute__((aligned(16))) WaveLds
        {
            float st_d[kStackK * kWave];
            uint32_t st_i[kStackK * kWave];
            float bx[kImage + 4], by[kImage + 4], bz[kImage + 4]; // + the all-inf float4 group
            float4 qb[kWave]; // queries by rank in the group, then their results
            double sums[kCovSums + 4]; // this wave's running totals (fixed summation order)
        };
        static_assert(sizeof(WaveLds) <= 5296, "LDS budget: 7 workgroups of 4 waves on a CU (7 x 4 x 5296 B = 148 KB)");
        static_assert(offsetof(WaveLds, by) - offsetof(WaveLds, bx) == (kImage + 4) * sizeof(float) &&
                          offsetof(WaveLds, bz) - offsetof(WaveLds, by) == (kImage + 4) * sizeof(float),
                      "tile_scan addresses by and bz relative to bx");

        __device__ __forceinline__ void wave_lds_fence()
        {
            // LDS operations of one wave execute in order; this only pins the compiler
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }

        // second smallest (with multiplicity) and smallest of a stream: m <= m2 always
        __device__ __forceinline__ void track2(float v, float& m, float& m2)
        {
            m2 = __builtin_amdgcn_fmed3f(m, m2, v);
            m = fminf(m, v);
        }

        // One value per level of a descent: m = min(|radius - d|, d) - slack, a lower bound of both
        // distances that decide whether the level can change an answer on the way back up (an
        // answer below it passes the prune test, vp-tree.cc:152, and is not beaten by the centre,
        // :154).  The chain keeps the smallest, the second smallest (with multiplicity) and the
        // node of the smallest -- on the bit patterns, as signed integers (integer min / max
        // instead of a canonicalising move + float min each).  m may be slightly negative (the
        // bound minus its slack): any negative float is below every non-negative one as an
        // integer too, so the bound turns negative and stays so, which is all that matters (no
        // distance is below it: the lane climbs the long way); a NaN (a +inf centre distance
        // against a +inf radius) reads as a large positive integer and never lowers a bound, as
        // with fminf.
        __device__ __forceinline__ void chain_track(float& chain_m, float& chain_m2, int& arg, float m,
                                                    int node)
        {
            const int a = __float_as_int(chain_m), b = __float_as_int(chain_m2), v = __float_as_int(m);
            arg = v < a ? node : arg;
            chain_m2 = __int_as_float(min(b, max(a, v)));
            chain_m = __int_as_float(min(a, v));
        }
        constexpr int kNearMark = (int)0x80000000; // `prev` of a lane that jumped to the one level
                                                   // of its chain that matters (no child code: there
                                                   // are fewer than 2^31 leaves)
        constexpr int kFarMark = (int)0x80000001;  // `prev` of a lane that jumped back to the node
                                                   // its chain hangs from: it arrives from that
                                                   // node's far side, whichever child that is (the
                                                   // way up only asks whether `prev` is the near
                                                   // child; no child code either)

        // ---- fixed-order wave sums of doubles without LDS traffic (gfx950 permlane swaps) ----
        // [a | b] : lanes 0..31 get a[l] + a[l+32], lanes 32..63 get b[l-32] + b[l]
        __device__ __forceinline__ double swap_add_halves(double a, double b)
        {
            const uint64_t ua = (uint64_t)__double_as_longlong(a), ub = (uint64_t)__double_as_longlong(b);
            const auto lo = __builtin_amdgcn_permlane32_swap((uint32_t)ua, (uint32_t)ub, false, false);
            const auto hi = __builtin_amdgcn_permlane32_swap((uint32_t)(ua >> 32), (uint32_t)(ub >> 32),
                                                             false, false);
            const double x = __longlong_as_double((long long)(((uint64_t)hi[0] << 32) | lo[0]));
            const double y = __longlong_as_double((long long)(((uint64_t)hi[1] << 32) | lo[1]));
            return x + y;
        }
        // a, b hold two 32-lane halves each: rows of 16 lanes become a.row0+a.row1, b.row0+b.row1,
        // a.row2+a.row3, b.row2+b.row3
        __device__ __forceinline__ double swap_add_rows(double a, double b)
        {
            const uint64_t ua = (uint64_t)__double_as_longlong(a), ub = (uint64_t)__double_as_longlong(b);
            const auto lo = __builtin_amdgcn_permlane16_swap((uint32_t)ua, (uint32_t)ub, false, false);
            const auto hi = __builtin_amdgcn_permlane16_swap((uint32_t)(ua >> 32), (uint32_t)(ub >> 32),
                                                             false, false);
            const double x = __longlong_as_double((long long)(((uint64_t)hi[0] << 32) | lo[0]));
            const double y = __longlong_as_double((long long)(((uint64_t)hi[1] << 32) | lo[1]));
            return x + y;
        }
        template <int CTRL>
        __device__ __forceinline__ double dpp_shr_zero(double v)
        {
            const uint64_t u = (uint64_t)__double_as_longlong(v);
            // bound_ctrl: lanes without a source read 0 = +0.0
            const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, CTRL, 0xf, 0xf, true);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), CTRL, 0xf, 0xf, true);
            return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
        }
        // lane 15 of every 16-lane row ends with the sum of its row
        __device__ __forceinline__ double row_sum_f64(double v)
        {
            v += dpp_shr_zero<0x111>(v); // row_shr:1
            v += dpp_shr_zero<0x112>(v); // row_shr:2
            v += dpp_shr_zero<0x114>(v); // row_shr:4
            v += dpp_shr_zero<0x118>(v); // row_shr:8
            return v;
        }

        // Stage the window [first, first + cw) of a bucket in the image: x is +inf in every slot
        // that is not the bucket's.
        __device__ __forceinline__ void stage_window(const DevTree& t, WaveLds& L, uint32_t first,
                                                     uint32_t cw, int lane)
        {
            const float kInf = __int_as_float(0x7f800000);
            const uint32_t s0 = 4u * (uint32_t)lane;
            float4 vx = make_float4(kInf, kInf, kInf, kInf);
            float4 vy = make_float4(0.f, 0.f, 0.f, 0.f), vz = vy;
            if (s0 < cw)
            {
                const size_t o = (size_t)first + s0;
                vx = *reinterpret_cast<const float4*>(t.sx + o);
                vy = *reinterpret_cast<const float4*>(t.sy + o);
                vz = *reinterpret_cast<const float4*>(t.sz + o);
                vx.y = (s0 + 1u < cw) ? vx.y : kInf;
                vx.z = (s0 + 2u < cw) ? vx.z : kInf;
                vx.w = (s0 + 3u < cw) ? vx.w : kInf;
            }
            *reinterpret_cast<float4*>(L.bx + s0) = vx;
            *reinterpret_cast<float4*>(L.by + s0) = vy;
            *reinterpret_cast<float4*>(L.bz + s0) = vz;
        }

        // ---- the ring walk: a bucket scan that starts where the answer must be ----
        // DevTree::ring (kernels.hh) lists a bucket's points by their distance rho to the centre
        // of the bucket's parent node.  A query at distance dq from that centre is at least
        // |rho - dq| from the point, so the lane looks up dq in the list and examines the points
        // outwards from there, nearest ring first, until the gap to the next ring on either side
        // exceeds the best distance found: nothing beyond can be as close.  Once the clouds are
        // aligned that is after one or two points instead of the bucket's 244.  Every lane walks
        // for its own query (the image and the ring words of the bucket are staged in LDS; the
        // ring words lie in the query table of the tile scan, which is not in use meanwhile); a
        // lane that is not finished after kWalkMax points leaves its query to the tile scan.
        //   Bounds.  Stored: rho_lo = the computed distance with its low 8 mantissa bits cleared,
        // so the point's real distance lies in [rho_lo (1 - 4e-7), rho_lo (1 + 2^-15) (1 + 4e-7)];
        // dq is the raw v_sqrt_f32 of the reference's d2 (within 4e-7 of the real distance).  The
        // gap to a ring above is rho_lo - dq, to a ring below dq - rho_lo (1 + 2^-15), each minus
        // a slack of 2^-20 (rho + dq), more than twice those errors.  The walk stops when
        // gap^2 > m1 (1 + 1e-5), m1 the smallest reference d2 found: every point not examined
        // then has a real squared distance above m1 (1 + 1e-5) and a reference d2 (within 4e-7 of
        // it) far more than 4 ulps above m1.  Among the points examined the answer is the one of
        // m1 if the runner-up m2 is more than 4 ulps above it (their roots then round to different
        // floats, top of the file); otherwise the lane goes to the exact scan, as after a tile scan.
        constexpr int kWalkMax = 6;
        constexpr int kWalkMinGroup = 16; // lanes on a bucket from which the walk is tried
        __device__ __forceinline__ bool ring_walk(const WaveLds& L, uint32_t cnt, bool active, float qx,
                                                  float qy, float qz, float dq, float& out_m,
                                                  uint32_t& out_idx, bool& out_ok)
        {
            const float kInf = __int_as_float(0x7f800000);
            const uint32_t* rw = reinterpret_cast<const uint32_t*>(L.qb); // 256 words, +inf behind cnt
            // how many rings lie below dq (the list is ascending, +inf padded to 256 words)
            uint32_t pos = 0;
#pragma unroll
            for (uint32_t step = 128u; step >= 1u; step >>= 1)
            {
                const float rho = __uint_as_float(rw[pos + step - 1u] & 0xFFFFFF00u);
                pos += rho < dq ? step : 0u;
            }
            int up = (int)pos, dn = (int)pos - 1;
            float m1 = kInf, m2 = kInf;
            uint32_t s1 = 0;
            bool done = !active;
#pragma nounroll
            for (int it = 0; it <= kWalkMax; ++it)
            {
                if (__ballot(!done) == 0ull)
                    break;
                if (!done)
                {
                    const uint32_t wu = rw[min(up, 255)], wd = rw[max(dn, 0)];
                    const float ru = __uint_as_float(wu & 0xFFFFFF00u);
                    const float rl = __uint_as_float(wd & 0xFFFFFF00u);
                    const float gu = (uint32_t)up < cnt ? __fsub_rn(ru, dq) : kInf;
                    const float gd = dn >= 0 ? __fsub_rn(dq, __fadd_rn(rl, __fmul_rn(rl, 0x1p-15f))) : kInf;
                    const bool take_up = gu <= gd;
                    const float gap = take_up ? gu : gd;
                    const float ref = take_up ? ru : rl;
                    const float g = __fsub_rn(gap, __fmaf_rn(__fadd_rn(ref, dq), 0x1p-20f, 1e-18f));
                    if (g > 0.f && __fmul_rn(g, g) > __fmul_rn(m1, 1.00001f))
                        done = true; // (both sides exhausted: the gap is +inf)
                    else if (it < kWalkMax)
                    {
                        const uint32_t slot = (take_up ? wu : wd) & 0xFFu;
                        up += take_up ? 1 : 0;
                        dn -= take_up ? 0 : 1;
                        const float d2 = sq_dist(qx, qy, qz, L.bx[slot], L.by[slot], L.bz[slot]);
                        m2 = d2 < m1 ? m1 : fminf(m2, d2);
                        s1 = d2 < m1 ? slot : s1;
                        m1 = fminf(m1, d2);
                    }
                }
            }
            out_m = m1;
            out_idx = s1;
            out_ok = (int)(__float_as_uint(m2) - __float_as_uint(m1)) > 4;
            return active && done;
        }

        // Tile scan of bucket [beg, beg+cnt) for the g lanes in `grp` (all of them wait on it).
        // The 64 lanes form P = 64/g parts of g lanes; lane (part p, slot s) scans the float4
        // groups p, p+P, p+2P, .. of the staged image for the query of rank s.  Inside the loop
        // only the minimum of each float4 group of the FILTER values is tracked (its step, the
        // smallest and the second smallest group minimum); the winning group is re-read once at
        // the end and evaluated with the reference's arithmetic.  On return every member lane has
        // out_m = the reference's d2 of the best point of that group, out_idx = its slot in the
        // bucket, out_ok = whether it is proven to be the reference's answer (top of the file).
        __device__ __forceinline__ void tile_scan(const DevTree& t, WaveLds& L, uint32_t beg,
                                                  uint32_t cnt, uint64_t grp, bool member, float qx,
                                                  float qy, float qz, int lane, float& out_m,
                                                  bool& out_ok, uint32_t& out_idx,
                                                  uint32_t& staged)
        {
            const float kInf = __int_as_float(0x7f800000);
            const uint32_t g = (uint32_t)__builtin_popcountll(grp); // 1..64
            const uint32_t P = 64u / g;
            // lane / g for lane < 64 (exact for every g in 1..64)
            const uint32_t part = ((uint32_t)lane * (65536u / g + 1u)) >> 16;
            const uint32_t slot = (uint32_t)lane - part * g;
            const bool idle = part >= P; // lanes past the last whole part read the +inf group

            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(
                (uint32_t)(grp >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)grp, 0u));
            if (member)
                L.qb[rank] = make_float4(qx, qy, qz, 0.f);

            // m, m2: smallest and second smallest group minimum of the FILTER values
            // fma(dz,dz, fma(dy,dy, dx*dx)); em: the reference's d2 of the best point of the
            // filter's winning group; idx: its slot, bit 31 = the runner-up inside that group is
            // more than 4 ulps above em
            float m = kInf, m2 = kInf, em = kInf;
            uint32_t idx = 0;
#pragma nounroll
            for (uint32_t w0 = 0; w0 < cnt; w0 += kImage) // one window unless the bucket is > 256
            {
                const uint32_t cw = min((uint32_t)kImage, cnt - w0);
                // ---- stage the window: x is +inf in every slot that is not the bucket's ----
                // (`staged` = first slot of the window the image holds: the batches of a chunk
                // that share a bucket, and the two halves of a split group, stage it once)
                if (staged != beg + w0)
                {
                    staged = beg + w0;
                    stage_window(t, L, beg + w0, cw, lane);
                }
                wave_lds_fence();
                const float4 q = L.qb[slot];
                const uint32_t n4 = (cw + 3u) >> 2;           // float4 groups in the window
                const uint32_t steps = (n4 + P - 1u) / P;     // wave-uniform
                // byte offset of this lane's float4 group in bx/by/bz.  Lanes past the last whole
                // part shadow part 0: nothing folds their results in.  Every group below kGroups
                // is staged (+inf behind the bucket), group kGroups is the +inf pad; only a scan
                // that can step past it (P * steps > kGroups + 1: long buckets in many parts)
                // clamps its offsets.
                // (LDS pointers, 32 bits wide: the select that keeps the best group is one
                // instruction, where a generic pointer takes a second one for its high half)
                typedef const __attribute__((address_space(3))) char* lds_ptr;
                typedef float v4f __attribute__((ext_vector_type(4)));
                typedef const __attribute__((address_space(3))) v4f* lds_f4;
                lds_ptr const img = (lds_ptr)L.bx;
                lds_ptr gp = img + (idle ? 0u : 16u * part); // this lane's group in bx
                const uint32_t gstride = 16u * P;
                const float m_before = m;
                lds_ptr bp = img; // group of the smallest minimum so far
                constexpr uint32_t kBy = (kImage + 4) * sizeof(float), kBz = 2u * kBy;
                auto step = [&](lds_ptr at) {
                    const v4f X = *(lds_f4)(at);
                    const v4f Y = *(lds_f4)(at + kBy);
                    const v4f Z = *(lds_f4)(at + kBz);
                    const float d0 = sq_dist_fma(q.x, q.y, q.z, X.x, Y.x, Z.x);
                    const float d1 = sq_dist_fma(q.x, q.y, q.z, X.y, Y.y, Z.y);
                    const float d2 = sq_dist_fma(q.x, q.y, q.z, X.z, Y.z, Z.z);
                    const float d3 = sq_dist_fma(q.x, q.y, q.z, X.w, Y.w, Z.w);
                    const float gm = fminf(fminf(fminf(d0, d1), d2), d3);
                    bp = (gm < m) ? at : bp;
                    m2 = __builtin_amdgcn_fmed3f(m, m2, gm);
                    // distances are >= 0 (or +inf): they order like their bit patterns
                    m = __uint_as_float(min(__float_as_uint(m), __float_as_uint(gm)));
                };
                if (P * steps <= (uint32_t)kGroups + 1u)
                {
#pragma unroll 2
                    for (uint32_t k = 0; k < steps; ++k)
                    {
                        step(gp);
                        gp += gstride;
                    }
                }
                else
                {
                    lds_ptr const pad = img + 16u * (uint32_t)kGroups;
#pragma nounroll
                    for (uint32_t k = 0; k < steps; ++k)
                    {
                        step(gp < pad ? gp : pad);
                        gp += gstride;
                    }
                }
                if (m < m_before)
                {
                    // the winning group again: slot of m and the runner-up inside the group
                    const v4f X = *(lds_f4)(bp);
                    const v4f Y = *(lds_f4)(bp + kBy);
                    const v4f Z = *(lds_f4)(bp + kBz);
                    const float d0 = sq_dist(q.x, q.y, q.z, X.x, Y.x, Z.x);
                    const float d1 = sq_dist(q.x, q.y, q.z, X.y, Y.y, Z.y);
                    const float d2 = sq_dist(q.x, q.y, q.z, X.z, Y.z, Z.z);
                    const float d3 = sq_dist(q.x, q.y, q.z, X.w, Y.w, Z.w);
                    float a = kInf, b2 = kInf;
                    track2(d0, a, b2);
                    track2(d1, a, b2);
                    track2(d2, a, b2);
                    track2(d3, a, b2);
                    const uint32_t c = (d0 == a) ? 0u : (d1 == a) ? 1u : (d2 == a) ? 2u : 3u;
                    idx = w0 + ((uint32_t)(bp - img) >> 2) + c;
                    if ((__float_as_uint(b2) - __float_as_uint(a)) > 4u)
                        idx |= 0x80000000u;
                    em = a;
                }
                wave_lds_fence(); // the next window overwrites the image
            }
            // ---- fold the parts: part p takes part p + half, halving until one part is left ----
#pragma nounroll
            for (uint32_t h = P; h > 1u;)
            {
                const uint32_t half = (h + 1u) >> 1;
                const int src = lane + (int)(half * g);
                const float mb = __shfl(m, src, kWave);
                const float m2b = __shfl(m2, src, kWave);
                const float emb = __shfl(em, src, kWave);
                const uint32_t ib = __shfl(idx, src, kWave);
                if (part + half < h)
                {
                    // the part with the smaller filter minimum carries its exact pair on; a tie
                    // leaves m2 == m and the lane goes to the exact scan
                    idx = (mb < m) ? ib : idx;
                    em = (mb < m) ? emb : em;
                    m2 = fminf(fminf(m2, m2b), fmaxf(m, mb));
                    m = fminf(m, mb);
                }
                h = half;
            }
            if ((uint32_t)lane < g)
                L.qb[lane] = make_float4(em, m2, __uint_as_float(idx), 0.f);
            wave_lds_fence();
            const float4 r = L.qb[member ? rank : 0u];
            out_m = r.x;
            out_idx = __float_as_uint(r.z) & 0x7FFFFFFFu;
            // every point outside the winning group has a filter value >= m2; filter and
            // reference d2 are both within 3.01 u of the real sum, so m2 more than 16 float steps
            // above em puts every such point's reference d2 more than 4 ulps above em
            out_ok = (__float_as_uint(r.z) >> 31) != 0u &&
                     (int)(__float_as_uint(r.y) - __float_as_uint(r.x)) > 16;
            wave_lds_fence();
        }

        // Whole-wave scan of one bucket of <= 256 slots for ONE query (arguments wave-uniform):
        // the cheaper form when only a few lanes wait on the bucket.  Lane l holds the float4
        // group l of the bucket (slots 4l .. 4l+3: three 16-byte loads per lane; a bucket starts
        // at a multiple of 16 slots and its padding is +inf, so a group never reaches into the
        // next bucket and padding slots evaluate to +inf).  Returns the SQUARED distance of the
        // reference's answer (its sqrt is the reference's distance) and the answer's slot.
        __device__ __forceinline__ void coop_scan(const DevTree& t, float bx, float by, float bz,
                                                  uint32_t beg, uint32_t cnt, int lane, float& m_out,
                                                  uint32_t& win_out)
        {
            const float kInf = __int_as_float(0x7f800000);
            const uint32_t s0 = 4u * (uint32_t)lane;
            float s[4] = {kInf, kInf, kInf, kInf};
            if (s0 < cnt)
            {
                const size_t o = (size_t)beg + s0;
                const float4 X = *reinterpret_cast<const float4*>(t.sx + o);
                const float4 Y = *reinterpret_cast<const float4*>(t.sy + o);
                const float4 Z = *reinterpret_cast<const float4*>(t.sz + o);
                s[0] = sq_dist(bx, by, bz, X.x, Y.x, Z.x);
                s[1] = sq_dist(bx, by, bz, X.y, Y.y, Z.y);
                s[2] = sq_dist(bx, by, bz, X.z, Y.z, Z.z);
                s[3] = sq_dist(bx, by, bz, X.w, Y.w, Z.w);
            }
            const float m = wave_min_nonneg(fminf(fminf(s[0], s[1]), fminf(s[2], s[3])));
            // candidates: every slot within 4 ulps of the minimum (a superset of the slots whose
            // sqrt equals sqrt(m)); exactly one candidate is the common case
            const float near = __uint_as_float(__float_as_uint(m) + 4u);
            uint64_t h[4];
            uint32_t total = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c)
            {
                h[c] = __ballot(s[c] <= near);
                total += (uint32_t)__builtin_popcountll(h[c]);
            }
            if (total != 1u)
            {
                // exact rule: first slot whose d2 rounds to the same sqrt as the minimum
                const float hi = sq_ceiling_for_root(sqrtf(m));
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    h[c] = __ballot(s[c] <= hi);
            }
            // first slot among the hits: slot = 4 * lane + c (scalar arithmetic)
            uint32_t first = 0xFFFFFFFFu; // a NaN-poisoned bucket keeps slot 0, like the reference's scan
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (h[c])
                    first = min(first, 4u * (uint32_t)__builtin_ctzll(h[c]) + (uint32_t)c);
            m_out = m;
            win_out = beg + (first == 0xFFFFFFFFu ? 0u : first);
        }

    } // namespace

        // The kernel's arguments as ONE structure.  What the batches use (the tree, the clouds,
        // the matched points, the centroid, the sizes of a batch) is read once and lives in
        // SGPRs; what is needed once per chunk or only when a query is parked, spilled or
        // reported (the queue, the chunk sums, the lots and late rows, the overflow stack, rows
        // and distances) is read from the kernel-argument segment where it is used, through a
        // pointer the compiler cannot see through: kept in registers across the kernel they
        // pushed the scalar registers past what the wave has (89 spilled, into the hot loops).
        // (Reading EVERYTHING there, the clouds and outputs of every batch included, removed 52
        // of the 62 spills of the kernel before the lots and was 0.7 % slower: a v_readlane at
        // the head of a batch is cheaper than a scalar load and its wait.)
        struct TileArgs
        {
            DevTree t;
            const float *px, *py, *pz;
            float mux, muy, muz;
            uint32_t n;
            float *yx, *yy, *yz;
            uint32_t cut, batch_queries, park_max, want_rows; // want_rows: row_out or dist_out set
            // ---- read where they are used ----
            uint32_t* row_out;
            float* dist_out;
            const uint32_t* perm;
            double* chunk_sums;
            uint32_t* queue;
            uint2* spill;
            uint32_t n_batches, chunk_batches, n_chunks, late_rows;
            uint32_t *park, *late, *late_cnt;
            unsigned long long* stats;
            const LoopState* loop;
        };
        typedef const __attribute__((address_space(4))) TileArgs* tile_args_ptr;

        template <bool STATS>
        __global__ void __launch_bounds__(kBlock, 7) // 7 workgroups per CU (LDS 7 x 20.7 KiB, <= 72 VGPRs)
            nn_tile_kernel(const TileArgs args)
        {
            __shared__ WaveLds s_w[kWavesPerBlock];
            // (explicit arguments start at offset 0 of the kernel-argument segment)
            auto cold = [&]() -> tile_args_ptr {
                tile_args_ptr p = (tile_args_ptr)__builtin_amdgcn_kernarg_segment_ptr();
                asm volatile("" : "+s"(p));
                return p;
            };
            const DevTree t = {args.t.nodes, args.t.leaf_info, args.t.sx, args.t.sy, args.t.sz, nullptr,
                               nullptr, args.t.ring, args.t.n_slots, args.t.n_nodes, args.t.root};
            const float* __restrict__ px = args.px;
            const float* __restrict__ py = args.py;
            const float* __restrict__ pz = args.pz;
            float* __restrict__ yx = args.yx;
            float* __restrict__ yy = args.yy;
            float* __restrict__ yz = args.yz;
            float mux = args.mux, muy = args.muy, muz = args.muz;
            const uint32_t n = args.n, cut = args.cut, batch_queries = args.batch_queries;
            const uint32_t park_max = args.park_max & 0xFFu; // 0: no lots
            const uint32_t park_wait = args.park_max >> 8;   // rounds a batch goes on before it parks
            const LoopState* loop = cold()->loop;
            unsigned long long* stats = STATS ? cold()->stats : nullptr;
            if (loop)
            {
                // device-resident loop (loop.hip): the centroid comes from the previous
                // iteration's sums; an iteration enqueued past the end of the loop does nothing
                // (or has started its last iteration and only waits for that one's error)
                if (loop->done || loop->started >= loop->max_iters)
                    return;
                // (wave-uniform: kept in SGPRs)
                mux = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(loop->mu_q[0])));
                muy = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(loop->mu_q[1])));
                muz = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(loop->mu_q[2])));
            }
            // wave-level event counts of the diagnostic build (ICP_HIP_TILE_STATS=1)
            uint32_t st_node = 0, st_tile = 0, st_tile_g = 0, st_tile_len = 0, st_coop = 0,
                     st_exact = 0, st_leafph = 0, st_batch = 0, st_node_lanes = 0, st_walk = 0,
                     st_walk_done = 0, st_walk_lanes = 0;
            // shader cycles by phase (diagnostic build only)
            unsigned long long cy_node = 0, cy_tile = 0, cy_coop = 0, cy_exact = 0, cy_tail = 0,
                               cy_head = 0, cy_total = 0, cy_pop = 0, cy_walk = 0, cy_dtop = 0,
                               cy_dbot = 0, cy_lv = 0;
            const unsigned long long rt_begin = STATS ? __builtin_amdgcn_s_memrealtime() : 0ull;
            if (STATS)
                stats += (size_t)(blockIdx.x % (uint32_t)kTileStatsCopies) * kTileStatsWords;
            auto now = [&]() -> unsigned long long {
                return STATS ? (unsigned long long)__builtin_amdgcn_s_memtime() : 0ull;
            };
            const unsigned long long cy_begin = now();

            const int lane = threadIdx.x & (kWave - 1);
            const int w = threadIdx.x >> 6;
            WaveLds& L = s_w[w];
            float* st_d = L.st_d + lane;
            uint32_t* st_i = L.st_i + lane;
            if (lane < 4)
            {
                L.bx[kImage + lane] = __int_as_float(0x7f800000);
                L.by[kImage + lane] = 0.f;
                L.bz[kImage + lane] = 0.f;
            }
            uint32_t staged = 0xFFFFFFFFu; // wave-uniform: window of the bucket image in LDS
            uint32_t staged_cnt = 0;       // wave-uniform: bucket slots in that window
            uint32_t ring_tag = 0xFFFFFFFFu; // wave-uniform: bucket whose ring words lie in L.qb
            uint32_t walk_skip = 0;          // wave-uniform: groups to serve without trying the walk
            uint32_t walk_backoff = 8;       // ... after the next walk that does not pay (doubles)
            int lc_code = 0;               // wave-uniform: child code of the last bucket served
            uint32_t lc_beg = 0, lc_cnt = 0; // (0 is an internal node: matches no leaf) + its entry
            // this lane's column of the overflow stack: entry e at spill_col[e * spill_stride]
            // (made where it is used, from a thread index the compiler cannot see through: kept
            // across the kernel the address was one of the registers that spilled to scratch)
            auto spill_at = [&](uint32_t e) -> uint2* {
                uint32_t tid = threadIdx.x;
                asm volatile("" : "+v"(tid));
                const size_t spill_stride = (size_t)gridDim.x * kBlock;
                return cold()->spill + ((size_t)blockIdx.x * kBlock + tid + (size_t)e * spill_stride);
            };

            // ---- stragglers: the parking lot ----
            // Once the clouds are aligned 99 % of the lanes of a batch are finished by their first
            // bucket; the few that are not (a query whose answer lies on a node's centre or on
            // its sphere: the reference searches that node's far side, vp-tree.cc:152-160) used
            // to keep the whole wave stepping through nodes for one or two lanes -- a batch with
            // one cost three times a batch without, and 29 % of the batches had one.  Now the
            // lanes that are still at work when at most `park_max` of a batch are, put their
            // whole traversal state into the wave's lot in HBM and the batch ends without them;
            // when the lot holds a batch worth of parked queries (or the queues are empty) the
            // wave runs them together as one more batch, from wherever each of them stood.  The
            // traversal of a query is untouched: it only continues in another lane, later.
            //   Its terms go to a row of their own: chunk c owns rows [c * late_rows, (c + 1) *
            // late_rows) of `late` (twelve words each: the nine fp32 products, the three counters
            // as integers), hands them out in the order its batches park queries and writes how
            // many it handed out to late_cnt[c]; the chunk-sum reduction adds row 0 (chunk_sums)
            // and the late rows of each chunk in a fixed order.  Whether a lane is parked depends
            // on its batch and its chunk only (how many lanes are left, how many rows the chunk
            // has used), never on what else the wave has in its lot -- the lot is emptied before
            // a batch that could overflow it -- so the sums stay bitwise reproducible whatever
            // wave runs a chunk.
            const bool park_on = park_max != 0u;
            // (entry e of this wave's lot; made where it is used, from an index the compiler
            // cannot see through, like spill_at below: the addresses must not be kept in
            // registers across the kernel)
            // A wave has TWO lots: queries are parked in lot `cur`; a flush takes all of its
            // entries, one per lane, and makes the other lot the current one -- so the entries
            // just taken stay intact while their batch runs, and the two words of an entry that
            // only the end of the query needs (its index and its late row) are read from there
            // instead of riding in registers through the batch.
            auto lot_at = [&](uint32_t which, uint32_t e) -> uint32_t* {
                uint32_t tid = threadIdx.x;
                asm volatile("" : "+v"(tid), "+v"(e));
                return cold()->park +
                       ((size_t)((blockIdx.x * (uint32_t)kWavesPerBlock + (tid >> 6)) * 2u + which) * (kLot * kParkWords) +
                        (size_t)e * kParkWords);
            };
            uint32_t cur = 0; // wave-uniform: the lot that takes parked queries
            uint32_t lot = 0;     // wave-uniform: entries in the lot
            uint32_t k_chunk = 0; // wave-uniform: late rows the current chunk has handed out

            // XCD group x owns chunks [x*per_group, (x+1)*per_group); its head is queue[x*32]
            uint32_t victim = 0;
            uint32_t grp_x = blockIdx.x & 7u;
            uint32_t chunk = 0, b = 0, b_end = 0;
            bool have_chunk = false;
            for (;;)
            {
                // ---- the next piece of work: a batch of the current chunk, or the lot ----
                bool flush = park_on && lot + park_max > kLot;
                if (!flush && b == b_end)
                {
                    if (have_chunk)
                    {
                        // the chunk's sums, and how many late rows it handed out
                        have_chunk = false;
                        uint32_t lane_c = (uint32_t)lane;
                        asm volatile("" : "+v"(lane_c));
                        tile_args_ptr A = cold();
                        if (A->chunk_sums && lane_c < (uint32_t)kCovSums)
                            A->chunk_sums[(size_t)chunk * kCovSums + lane_c] = L.sums[lane_c];
                        if (park_on && lane_c == 0u)
                            A->late_cnt[chunk] = k_chunk;
                    }
                    // ---- pull the next chunk ----
                    const unsigned long long cy_p0 = now();
                    const uint32_t n_chunks = cold()->n_chunks;
                    const uint32_t per_group = (n_chunks + 7u) >> 3;
                    while (victim < 8u)
                    {
                        const uint32_t lo = min(grp_x * per_group, n_chunks);
                        const uint32_t hi = min(lo + per_group, n_chunks);
                        uint32_t c = 0;
                        if (lane == 0)
                            c = __hip_atomic_fetch_add(cold()->queue + grp_x * 32u, 1u, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT);
                        c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
                        if (c < hi - lo)
                        {
                            chunk = lo + c;
                            have_chunk = true;
                            break;
                        }
                        victim += 1u; // this group is drained: steal from the next one
                        grp_x = (grp_x + 1u) & 7u;
                    }
                    cy_pop += now() - cy_p0;
                    if (have_chunk)
                    {
                        // (`lane_c` is opaque to the compiler so that the zero and the address of
                        // this lane's chunk sum are made here, not kept in registers -- they were
                        // what spilled to scratch -- across the whole kernel)
                        uint32_t lane_c = (uint32_t)lane;
                        asm volatile("" : "+v"(lane_c));
                        if (lane_c < (uint32_t)kCovSums)
                        {
                            uint32_t z;
                            asm volatile("v_mov_b32 %0, 0" : "=v"(z));
                            *reinterpret_cast<uint2*>(&L.sums[lane_c]) = make_uint2(z, z); // +0.0
                        }
                        b = chunk * cold()->chunk_batches;
                        b_end = min(b + cold()->chunk_batches, cold()->n_batches);
                        k_chunk = 0;
                    }
                    else
                    {
                        // the queues are empty: what is left is the lot, and nothing is parked again
                        if (lot == 0)
                            break;
                        flush = true;
                    }
                }
                // a flush with the queues empty is final: its lanes run to the end in place
                const bool final_flush = flush && !have_chunk;
            {
                if (STATS)
                    st_batch += 1u;
                const unsigned long long cy_b0 = now();
                // (diagnostic build: what this batch adds to the wave's counters)
                const unsigned long long cy_node_b0 = cy_node, cy_tile_b0 = cy_tile, cy_coop_b0 = cy_coop;
                const uint32_t st_node_b0 = st_node, st_tile_b0 = st_tile, st_coop_b0 = st_coop,
                               st_leafph_b0 = st_leafph;
                // `valid`: this lane holds a query that the end of this batch finishes
                bool valid = !flush && (uint32_t)lane < batch_queries && b * batch_queries + (uint32_t)lane < n;
                float qx = 0.f, qy = 0.f, qz = 0.f;
                if (valid)
                {
                    const uint32_t i = b * batch_queries + (uint32_t)lane;
                    // p_centered = p - mu_p (point-3d.cc:202-206)
                    qx = __fsub_rn(px[i], mux);
                    qy = __fsub_rn(py[i], muy);
                    qz = __fsub_rn(pz[i], muz);
                }
                int node = t.root, prev = 0;
                int mode = tDescend;
                // The current descent chain (from the root, or from the far child of `ret_node`):
                // `chain_m` is a lower bound of min(|radius - d|, d) over its nodes.  If the bucket
                // at its end answers with a distance below it, every prune test on the way back up
                // passes (vp-tree.cc:152) and no centre is closer (:154): the lane jumps straight
                // back to `ret_node` (or is done) instead of climbing level by level.
                // If the answer is below the second smallest only, exactly one level (`arg`) can
                // matter: the lane goes there, and from there back to `ret_node`.
                // (A lane that jumps to `ret_node` arrives there from the far side: `prev` =
                // kFarMark says so; which child that is, the way up does not ask.)
                int ret_node = -1, arg = -1;
                float chain_m = __int_as_float(0x7f800000), chain_m2 = chain_m;
                // stack depth (low 8 bits: a tree is at most 32 levels high) and, above it, how many
                // entries this query has put into its HBM column (one register instead of two)
                uint32_t spo = 0;
                uint32_t n_int = 0, n_pts = 0; // work counters of this query
                uint32_t low_rounds = 0; // wave-uniform: rounds with no more than park_max lanes at work
                float rd = 0.f;
                uint32_t rid = 0;
                uint32_t lbeg = 0, lcnt = 0;
                cy_head += now() - cy_b0;
                if (flush)
                {
                    // the parked queries, each where it stood (one per lane, `lot` of them)
                    valid = (uint32_t)lane < lot;
                    mode = tDone;
                    if (valid)
                    {
                        const uint32_t* e = lot_at(cur, (uint32_t)lane);
                        qx = __uint_as_float(e[pkQx]);
                        qy = __uint_as_float(e[pkQy]);
                        qz = __uint_as_float(e[pkQz]);
                        node = (int)e[pkNode];
                        prev = (int)e[pkPrev];
                        const uint32_t ms = e[pkModeSp];
                        mode = (int)(ms & 0xFFu);
                        rd = __uint_as_float(e[pkRd]);
                        rid = e[pkRid];
                        chain_m = __uint_as_float(e[pkChainM]);
                        chain_m2 = __uint_as_float(e[pkChainM2]);
                        arg = (int)e[pkArg];
                        ret_node = (int)e[pkRetNode];
                        n_int = e[pkNInt];
                        n_pts = e[pkNPts];
                        spo = (ms >> 8) | (e[pkOvf] << 8);
                        st_d[0] = __uint_as_float(e[pkStD0]);
                        st_i[0] = e[pkStI0];
                        st_d[kWave] = __uint_as_float(e[pkStD1]);
                        st_i[kWave] = e[pkStI1];
                    }
                    lot = 0;
                    cur ^= 1u; // (what this batch parks again goes to the other lot)
                }
                else if (!valid)
                    mode = tDone;
                else if (t.root < 0)
                {
                    const uint2 li = t.leaf_info[~t.root];
                    lbeg = li.x;
                    lcnt = li.y;
                    n_pts = li.y;
                    prev = t.root;
                    node = -1;
                    mode = tWaitLeaf;
                }

                // ---- the first descent, root to bucket, in a loop of its own: every lane goes
                // down one level per turn and nothing is pushed, popped or compared on the way
                // (vp-tree.cc:147-150 only); the general loop below does the same steps among
                // its other cases, at three times the instructions ----
                const unsigned long long cy_d0 = now();
                // waves that step through nodes wait on memory most of the time: when one of them
                // is ready it goes first, the scanning waves fill the issue slots in between
                __builtin_amdgcn_s_setprio(1);
                if (t.root >= 0 && !flush)
                {
                    uint32_t depth = 0; // wave-uniform: levels gone down
                    cy_lv = now();
                    for (;;)
                    {
                        const bool going = mode == tDescend;
                        const uint64_t goers = __ballot(going);
                        if (goers == 0ull)
                            break;
                        if (STATS)
                        {
                            const unsigned long long t_lv = now();
                            (depth < 12u ? cy_dtop : cy_dbot) += t_lv - cy_lv;
                            cy_lv = t_lv;
                            st_node += 1u;
                            st_node_lanes += (uint32_t)__builtin_popcountll(goers);
                        }
                        // The lanes of a batch are neighbours: most of the way down they stand
                        // on the SAME node.  Then its record comes through the scalar cache (one
                        // 32-byte read per wave, constant address space: the tree is never
                        // written while a pass runs) instead of 64 lanes x 32 bytes through the
                        // vector memory pipe, which the bucket images need.
                        const int node0 = __builtin_amdgcn_readlane(node, (int)__builtin_ctzll(goers));
                        const bool same = (__ballot(node != node0) & goers) == 0ull;
                        depth += 1u;
                        if (going)
                        {
                            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                            typedef const __attribute__((address_space(4))) u4* const_u4;
                            DevNode N;
                            if (same)
                            {
                                const_u4 rec = (const_u4)(t.nodes + node0);
                                const u4 lo = rec[0], hi = rec[1];
                                N.cx = __uint_as_float(lo.x);
                                N.cy = __uint_as_float(lo.y);
                                N.cz = __uint_as_float(lo.z);
                                N.r = __uint_as_float(lo.w);
                                N.inside = (int)hi.x;
                                N.outside = (int)hi.y;
                                N.parent = (int)hi.z;
                                N.thr = __uint_as_float(hi.w);
                            }
                            else
                                N = t.nodes[node];
                            const float d2 = sq_dist(qx, qy, qz, N.cx, N.cy, N.cz);
                            const float da = __builtin_amdgcn_sqrtf(d2); // as in the general step
                            const float lb = __fsub_rn(
                                fminf(fabsf(__fsub_rn(N.r, da)), da),
                                __fmaf_rn(__fadd_rn(da, N.r), 0x1p-20f, 1e-18f));
                            chain_track(chain_m, chain_m2, arg, lb, node);
                            const int go = d2 < N.thr ? N.inside : N.outside; // :147-148
                            if (go >= 0)
                                node = go;
                            else
                            {
                                uint2 li = make_uint2(lc_beg, lc_cnt);
                                if (go != lc_code)
                                    li = t.leaf_info[~go];
                                lbeg = li.x;
                                lcnt = li.y;
                                n_pts += li.y;
                                n_int += depth; // (every level down to here was a node visited)
                                prev = go;
                                mode = tWaitLeaf;
                            }
                        }
                    }
                }
                cy_node += now() - cy_d0;

                for (;;)
                {
                    // ---- stragglers leave for the lot (see the top of the kernel) ----
                    if (park_on && !final_flush)
                    {
                        const bool at_work = mode <= tReturn;
                        const uint64_t act = __ballot(at_work);
                        const uint32_t na = (uint32_t)__builtin_popcountll(act);
                        if (na != 0u && na <= park_max && low_rounds++ >= park_wait)
                        {
                            // a lane whose stack reaches into its HBM column stays (the column
                            // belongs to the lane), and then so does the rest: the batch goes on
                            // (the queries of a flush have their late rows already)
                            const uint32_t need = flush ? 0u : na;
                            const uint32_t late_rows = cold()->late_rows;
                            if (__ballot(at_work && (spo & 0xFFu) > (uint32_t)kStackK) == 0ull &&
                                k_chunk + need <= late_rows)
                            {
                                if (at_work)
                                {
                                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi(
                                        (uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
                                    uint32_t* e = lot_at(cur, lot + rank);
                                    uint32_t i, late_row;
                                    if (flush)
                                    {
                                        const uint32_t* from = lot_at(cur ^ 1u, (uint32_t)lane);
                                        i = from[pkIndex];
                                        late_row = from[pkRow];
                                    }
                                    else
                                    {
                                        i = b * batch_queries + (uint32_t)lane;
                                        late_row = chunk * late_rows + k_chunk + rank;
                                    }
                                    e[pkQx] = __float_as_uint(qx);
                                    e[pkQy] = __float_as_uint(qy);
                                    e[pkQz] = __float_as_uint(qz);
                                    e[pkNode] = (uint32_t)node;
                                    e[pkPrev] = (uint32_t)prev;
                                    e[pkModeSp] = (uint32_t)mode | ((spo & 0xFFu) << 8);
                                    e[pkRd] = __float_as_uint(rd);
                                    e[pkRid] = rid;
                                    e[pkChainM] = __float_as_uint(chain_m);
                                    e[pkChainM2] = __float_as_uint(chain_m2);
                                    e[pkArg] = (uint32_t)arg;
                                    e[pkRetNode] = (uint32_t)ret_node;
                                    e[pkNInt] = n_int;
                                    e[pkNPts] = n_pts;
                                    e[pkOvf] = spo >> 8;
                                    e[pkIndex] = i;
                                    e[pkRow] = late_row;
                                    e[pkStD0] = __float_as_uint(st_d[0]);
                                    e[pkStI0] = st_i[0];
                                    e[pkStD1] = __float_as_uint(st_d[kWave]);
                                    e[pkStI1] = st_i[kWave];
                                    mode = tDone;
                                    valid = false; // not this batch's to finish
                                }
                                lot += na;
                                k_chunk += need;
                                if (STATS)
                                    st_tile_len += na; // (diagnostic build: lanes parked)
                            }
                        }
                    }
                    // ---- node phase: step every lane until it waits on a bucket or is done ----
                    const unsigned long long cy_n0 = now();
                    __builtin_amdgcn_s_setprio(1);
                    for (;;)
                    {
                        const uint64_t stepping = __ballot(mode <= tReturn);
                        if (stepping == 0ull)
                            break;
                        // a few stragglers deep in far subtrees must not hold up the lanes that
                        // already wait on a bucket: serve those first once they are cut/16 of
                        // the stepping lanes (default 4/16: measured 5 % faster on a misaligned
                        // 32M cloud than 16/16, flat below)
                        if (cut != 0u &&
                            (uint32_t)__builtin_popcountll(stepping) * cut <= 16u *
                                (uint32_t)__builtin_popcountll(__ballot(mode == tWaitLeaf)))
                            break;
                        if (STATS)
                        {
                            st_node += 1u;
                            st_node_lanes += (uint32_t)__builtin_popcountll(stepping);
                        }
                        if (mode <= tReturn)
                        {
                            const DevNode N = t.nodes[node];
                            n_int += (mode == tDescend) ? 1u : 0u;
                            const float d2 = sq_dist(qx, qy, qz, N.cx, N.cy, N.cz);
                            const bool in_near = d2 < N.thr; // sqrtf(d2) < N.r (:147-148)
                            if (mode == tDescend)
                            {
                                // v_sqrt_f32 (1 ulp) instead of the rounded sqrtf: the slack
                                // covers it five times over; a bound that is too small only
                                // sends the lane the long way round
                                const float da = __builtin_amdgcn_sqrtf(d2);
                                const float lb = __fsub_rn(
                                    fminf(fabsf(__fsub_rn(N.r, da)), da),
                                    __fmaf_rn(__fadd_rn(da, N.r), 0x1p-20f, 1e-18f));
                                chain_track(chain_m, chain_m2, arg, lb, node);
                            }
                            const int nearc = in_near ? N.inside : N.outside;
                            const int farc = in_near ? N.outside : N.inside;
                            int go = nearc;
                            bool down = true;
                            if (mode == tReturn)
                            {
                                const float d = sqrtf(d2); // only the way up compares distances
                                bool up;
                                // (a lane that jumped here from its bucket came up the near side)
                                const bool marked = prev == kNearMark;
                                bool done_after = false;
                                if (prev == nearc || marked)
                                {
                                    // back from the near side (:152)
                                    if (rd < fabsf(__fsub_rn(N.r, d)))
                                        up = true;
                                    else
                                    {
                                        // the one level that matters on the chain from the root:
                                        // once its far side is searched the lane is done, noted
                                        // in the sign of the distance parked here (distances are
                                        // >= 0; -0.0 carries the sign too)
                                        const float parked =
                                            (marked && ret_node < 0) ? __uint_as_float(__float_as_uint(rd) | 0x80000000u)
                                                                     : rd;
                                        const uint32_t sp = spo & 0xFFu;
                                        if (sp < (uint32_t)kStackK)
                                        {
                                            st_d[sp * kWave] = parked;
                                            st_i[sp * kWave] = rid;
                                        }
                                        else
                                        {
                                            *spill_at(sp - kStackK) =
                                                make_uint2(__float_as_uint(parked), rid);
                                            spo += 0x100u;
                                        }
                                        spo += 1u;
                                        go = farc; // :160
                                        up = false;
                                        ret_node = node; // a new chain starts at the far child
                                        chain_m = chain_m2 = __int_as_float(0x7f800000);
                                    }
                                }
                                else
                                {
                                    // back from the far side (:163): far wins only if strictly closer
                                    spo -= 1u;
                                    const uint32_t sp = spo & 0xFFu;
                                    float nd;
                                    uint32_t ni;
                                    if (sp < (uint32_t)kStackK)
                                    {
                                        nd = st_d[sp * kWave];
                                        ni = st_i[sp * kWave];
                                    }
                                    else
                                    {
                                        const uint2 e = *spill_at(sp - kStackK);
                                        nd = __uint_as_float(e.x);
                                        ni = e.y;
                                    }
                                    done_after = (__float_as_uint(nd) >> 31) != 0u;
                                    nd = fabsf(nd);
                                    if (!(rd < nd))
                                    {
                                        rd = nd;
                                        rid = ni;
                                    }
                                    up = true;
                                }
                                down = !up;
                                if (up)
                                {
                                    if (d < rd) // :154 / :165 / :171 centre wins only if strictly closer
                                    {
                                        rd = d;
                                        rid = t.n_slots + (uint32_t)node;
                                    }
                                    // nothing between this level and where the chain began can
                                    // change the answer (marked), nor anything above it (done_after)
                                    prev = marked ? kFarMark : node;
                                    node = done_after ? -1 : marked ? ret_node : N.parent;
                                    mode = node < 0 ? tDone : tReturn;
                                }
                            }
                            if (down)
                            {
                                if (go >= 0)
                                {
                                    node = go;
                                    mode = tDescend;
                                }
                                else
                                {
                                    // the table entry of the last bucket this wave served is at
                                    // hand (neighbouring batches end in the same buckets)
                                    uint2 li = make_uint2(lc_beg, lc_cnt);
                                    if (go != lc_code)
                                        li = t.leaf_info[~go];
                                    lbeg = li.x;
                                    lcnt = li.y;
                                    n_pts += li.y;
                                    prev = go;
                                    mode = tWaitLeaf;
                                }
                            }
                        }
                    }
                    const unsigned long long cy_n1 = now();
                    cy_node += cy_n1 - cy_n0;
                    __builtin_amdgcn_s_setprio(0);
                    // ---- leaf phase: serve the waiting lanes bucket by bucket ----
                    uint64_t pend = __ballot(mode == tWaitLeaf);
                    if (pend == 0ull)
                        break;
                    if (STATS)
                        st_leafph += 1u;
                    while (pend)
                    {
                        // the bucket whose image is still in LDS first (the batch before ended
                        // in it, more often than not), then by lane
                        const uint64_t kept = __ballot(mode == tWaitLeaf && lbeg == staged) & pend;
                        const int l = (int)__builtin_ctzll(kept ? kept : pend);
                        const uint32_t beg = readlane_u(lbeg, l), cnt = readlane_u(lcnt, l);
                        lc_code = __builtin_amdgcn_readlane(prev, l);
                        lc_beg = beg;
                        lc_cnt = cnt;
                        bool member = (mode == tWaitLeaf) && (lbeg == beg);
                        uint64_t grp = __ballot(member);
                        pend &= ~grp;
                        int g_all = __builtin_popcountll(grp);
                        // ---- first the ring walk (see ring_walk): the lanes it finishes leave the
                        // group, the tile scan below serves the rest ----
                        // (a small group's tile scan is a few steps long -- 64 / g parts share the
                        // bucket -- and costs less than a walk)
                        if (g_all >= kWalkMinGroup && t.ring && cnt <= (uint32_t)kImage)
                        {
                            if (walk_skip != 0u)
                                walk_skip -= 1u;
                            else
                            {
                                const unsigned long long cy_w0 = now();
                                if (staged != beg)
                                {
                                    staged = beg;
                                    stage_window(t, L, beg, cnt, lane);
                                }
                                staged_cnt = cnt;
                                if (ring_tag != beg)
                                {
                                    ring_tag = beg;
                                    uint32_t s0 = 4u * (uint32_t)lane;
                                    asm volatile("" : "+v"(s0)); // (the address is made here, not kept)
                                    uint4 rv = make_uint4(0x7f800000u, 0x7f800000u, 0x7f800000u, 0x7f800000u);
                                    if (s0 < cnt)
                                    {
                                        rv = *reinterpret_cast<const uint4*>(t.ring + (size_t)beg + s0);
                                        rv.y = (s0 + 1u < cnt) ? rv.y : 0x7f800000u;
                                        rv.z = (s0 + 2u < cnt) ? rv.z : 0x7f800000u;
                                        rv.w = (s0 + 3u < cnt) ? rv.w : 0x7f800000u;
                                    }
                                    reinterpret_cast<uint4*>(L.qb)[lane] = rv;
                                }
                                wave_lds_fence();
                                // distance to the centre the rings are measured from: the
                                // bucket's parent, which is the node the lane stands on
                                // (one node for the whole group: its record comes through the
                                // scalar cache, like the first descent's)
                                float dq = 0.f;
                                {
                                    typedef float f4 __attribute__((ext_vector_type(4)));
                                    typedef const __attribute__((address_space(4))) f4* const_f4;
                                    const int parent = __builtin_amdgcn_readlane(node, l);
                                    const f4 pc = *(const_f4)(t.nodes + parent); // cx, cy, cz, r
                                    if (member)
                                        dq = __builtin_amdgcn_sqrtf(sq_dist(qx, qy, qz, pc.x, pc.y, pc.z));
                                }
                                float wm;
                                uint32_t widx;
                                bool wok;
                                const bool walked = ring_walk(L, cnt, member, qx, qy, qz, dq, wm, widx, wok);
                                wave_lds_fence(); // the tile scan's query table overwrites the ring words
                                if (walked)
                                {
                                    rd = wm;
                                    rid = beg + widx;
                                    mode = wok ? tServed : tWaitExact;
                                    member = false;
                                }
                                const uint64_t rest = __ballot(member);
                                if (STATS)
                                {
                                    st_walk += 1u;
                                    st_walk_done += (uint32_t)(g_all - __builtin_popcountll(rest));
                                    st_walk_lanes += (uint32_t)g_all;
                                    cy_walk += now() - cy_w0;
                                }
                                // a walk that finishes less than half of its lanes does not pay
                                // (misaligned clouds): leave it out for the next 8, 16, .. 256 groups
                                if (2 * __builtin_popcountll(rest) > g_all)
                                {
                                    walk_skip = walk_backoff;
                                    walk_backoff = min(2u * walk_backoff, 256u);
                                }
                                else
                                    walk_backoff = 8u;
                                grp = rest;
                                g_all = __builtin_popcountll(rest);
                            }
                        }
                        if (g_all == 0)
                        {
                        }
                        else if (g_all >= kTileMinGroup)
                        {
                            // 33..55 lanes would leave the tile half empty (one part only):
                            // serve the first 32 as two parts, then the rest as their own group
                            const bool split = g_all > 32 && g_all < kTileSplitBelow;
                            const uint32_t rank_all = __builtin_amdgcn_mbcnt_hi(
                                (uint32_t)(grp >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)grp, 0u));
                            uint64_t rest = grp;
                            while (rest)
                            {
                                const bool mem =
                                    member && (rest == grp ? (!split || rank_all < 32u) : rank_all >= 32u);
                                const uint64_t sub = __ballot(mem);
                                rest &= ~sub;
                                float m;
                                bool unique;
                                uint32_t idx;
                                const unsigned long long cy_t0 = now();
                                tile_scan(t, L, beg, cnt, sub, mem, qx, qy, qz, lane, m, unique, idx,
                                          staged);
                                ring_tag = 0xFFFFFFFFu; // (its query table lies where the ring words were)
                                cy_tile += now() - cy_t0;
                                staged_cnt = min((uint32_t)kImage, beg + cnt - staged); // slots of the window in LDS
                                if (STATS)
                                {
                                    st_tile += 1u;
                                    st_tile_g += (uint32_t)__builtin_popcountll(sub);
                                }
                                if (mem)
                                {
                                    rd = m;
                                    rid = beg + idx;
                                    mode = unique ? tServed : tWaitExact;
                                }
                            }
                        }
                        else if (cnt <= (uint32_t)kImage)
                        {
                            uint64_t one = grp; // one or two lanes: the whole wave serves each
                            while (one)
                            {
                                const int k = (int)__builtin_ctzll(one);
                                one &= one - 1;
                                if (STATS)
                                    st_coop += 1u;
                                float m;
                                uint32_t win;
                                coop_scan(t, readlane_f(qx, k), readlane_f(qy, k),
                                          readlane_f(qz, k), beg, cnt, lane, m, win);
                                if (lane == k)
                                {
                                    rd = m;
                                    rid = win;
                                    mode = tServed;
                                }
                            }
                        }
                        else if (member)
                            mode = tWaitExact; // large bucket, small group: the looping exact scan
                    }
                    const unsigned long long cy_l1 = now();
                    cy_coop += cy_l1 - cy_n1; // the whole leaf phase; tile scans are cy_tile of it
                    // ---- rare: the exact scan for lanes whose minimum is not isolated ----
                    uint64_t ex = __ballot(mode == tWaitExact);
                    while (ex)
                    {
                        const int l = (int)__builtin_ctzll(ex);
                        ex &= ex - 1;
                        if (STATS)
                            st_exact += 1u;
                        const float bx = readlane_f(qx, l), by = readlane_f(qy, l),
                                    bz = readlane_f(qz, l);
                        const uint32_t beg = readlane_u(lbeg, l), cnt = readlane_u(lcnt, l);
                        float r;
                        uint32_t win;
                        scan_bucket<0>(t, bx, by, bz, beg, cnt, lane, r, win);
                        if (lane == l)
                        {
                            rd = r;
                            rid = win;
                            if (r < chain_m)
                            {
                                node = ret_node;
                                prev = kFarMark;
                            }
                            else if (r < chain_m2)
                            {
                                node = arg;
                                prev = kNearMark;
                            }
                            mode = node < 0 ? tDone : tReturn;
                        }
                    }
                    if (mode == tServed)
                    {
                        const float m = rd; // squared distance of the bucket's answer
                        rd = sqrtf(m);
                        if (rd < chain_m)
                        {
                            node = ret_node;
                            prev = kFarMark;
                        }
                        else if (rd < chain_m2)
                        {
                            node = arg; // the one level that can matter
                            prev = kNearMark;
                        }
                        mode = node < 0 ? tDone : tReturn;
                    }
                    cy_exact += now() - cy_l1;
                }

                const unsigned long long cy_e0 = now();
                float ox = 0.f, oy = 0.f, oz = 0.f;
                if (valid)
                {
                    const uint32_t in_image = rid - staged; // `staged` invalid: a huge offset
                    if (in_image < staged_cnt && staged != 0xFFFFFFFFu)
                    {
                        // the answer is a slot of the bucket window that is still in LDS
                        ox = L.bx[in_image];
                        oy = L.by[in_image];
                        oz = L.bz[in_image];
                    }
                    else if (rid < t.n_slots)
                    {
                        ox = t.sx[rid];
                        oy = t.sy[rid];
                        oz = t.sz[rid];
                    }
                    else
                    {
                        const DevNode N = t.nodes[rid - t.n_slots];
                        ox = N.cx;
                        oy = N.cy;
                        oz = N.cz;
                    }
                    // the query's index: its place in the batch, or what its lot entry says
                    const uint32_t i = flush ? lot_at(cur ^ 1u, (uint32_t)lane)[pkIndex]
                                             : b * batch_queries + (uint32_t)lane;
                    if (yx)
                    {
                        yx[i] = ox;
                        yy[i] = oy;
                        yz[i] = oz;
                    }
                    if (args.want_rows)
                    {
                        tile_args_ptr A = cold();
                        const uint32_t o = A->perm ? A->perm[i] : i; // back to the caller's order
                        if (A->row_out)
                            A->row_out[o] = rid < t.n_slots ? A->t.srow[rid] : A->t.nrow[rid - t.n_slots];
                        if (A->dist_out)
                            A->dist_out[o] = rd;
                    }
                }
                // fp32 products as find_covariance forms them (point-3d.cc:220-231), summed in
                // fp64 in a fixed order: twelve per-lane values -> (permlane32 swap) six registers
                // of two 32-lane halves -> (permlane16 swap) three registers of four 16-lane rows
                // -> row sums by DPP -> this wave's running totals in LDS.
                if (flush)
                {
                    // a parked query's terms go to its own row (fp32 products, point-3d.cc:220-231)
                    if (valid)
                    {
                        // (word by word: built as 16-byte vectors the three counters became
                        // one register tuple for the whole kernel, and that tuple spilled)
                        const uint32_t late_row = lot_at(cur ^ 1u, (uint32_t)lane)[pkRow];
                        uint32_t* row = cold()->late + (size_t)late_row * kCovSums;
                        row[0] = __float_as_uint(__fmul_rn(qx, ox));
                        row[1] = __float_as_uint(__fmul_rn(qx, oy));
                        row[2] = __float_as_uint(__fmul_rn(qx, oz));
                        row[3] = __float_as_uint(__fmul_rn(qy, ox));
                        row[4] = __float_as_uint(__fmul_rn(qy, oy));
                        row[5] = __float_as_uint(__fmul_rn(qy, oz));
                        row[6] = __float_as_uint(__fmul_rn(qz, ox));
                        row[7] = __float_as_uint(__fmul_rn(qz, oy));
                        row[8] = __float_as_uint(__fmul_rn(qz, oz));
                        row[9] = n_int;
                        row[10] = n_pts;
                        row[11] = spo >> 8;
                    }
                }
                else
                {
                    // (a lane parked by this batch adds nothing here: ox = oy = oz = 0 and its
                    // counters travel with it)
                    if (!valid)
                    {
                        n_int = n_pts = 0u;
                        spo = 0u;
                    }
                    double v[kCovSums];
                    v[0] = (double)__fmul_rn(qx, ox);
                    v[1] = (double)__fmul_rn(qx, oy);
                    v[2] = (double)__fmul_rn(qx, oz);
                    v[3] = (double)__fmul_rn(qy, ox);
                    v[4] = (double)__fmul_rn(qy, oy);
                    v[5] = (double)__fmul_rn(qy, oz);
                    v[6] = (double)__fmul_rn(qz, ox);
                    v[7] = (double)__fmul_rn(qz, oy);
                    v[8] = (double)__fmul_rn(qz, oz);
                    v[9] = (double)n_int;
                    v[10] = (double)n_pts;
                    v[11] = (double)(spo >> 8);
                    double h[6], r[3];
#pragma unroll
                    for (int k = 0; k < 6; ++k)
                        h[k] = swap_add_halves(v[2 * k], v[2 * k + 1]); // [v2k | v2k+1]
#pragma unroll
                    for (int k = 0; k < 3; ++k) // rows: v4k, v4k+2, v4k+1, v4k+3
                        r[k] = row_sum_f64(swap_add_rows(h[2 * k], h[2 * k + 1]));
                    uint32_t lane_s = (uint32_t)lane; // (opaque: the address below is made here)
                    asm volatile("" : "+v"(lane_s));
                    if ((lane_s & 15u) == 15u) // the last lane of a row holds the row's sum
                    {
                        const int row = (int)(lane_s >> 4);
                        const int k0 = ((row & 1) << 1) | (row >> 1);
#pragma unroll
                        for (int k = 0; k < 3; ++k)
                            L.sums[4 * k + k0] += r[k];
                    }
                }
                cy_tail += now() - cy_e0;
                if (STATS && lane == 0)
                {
                    // histogram of batch durations (bins of 8192 cycles), and what the batches
                    // below / from 90112 cycles (11 bins) were made of
                    const unsigned long long dur = now() - cy_b0;
                    atomicAdd(stats + 88 + min(63ull, dur >> 13), 1ull);
                    unsigned long long* cls = stats + 152 + (dur >= (11ull << 13) ? 9 : 0);
                    atomicAdd(cls + 0, 1ull);
                    atomicAdd(cls + 1, dur);
                    atomicAdd(cls + 2, cy_node - cy_node_b0);
                    atomicAdd(cls + 3, cy_tile - cy_tile_b0);
                    atomicAdd(cls + 4, (cy_coop - cy_coop_b0) - (cy_tile - cy_tile_b0));
                    atomicAdd(cls + 5, (unsigned long long)(st_node - st_node_b0));
                    atomicAdd(cls + 6, (unsigned long long)(st_leafph - st_leafph_b0));
                    atomicAdd(cls + 7, (unsigned long long)(st_tile - st_tile_b0));
                    atomicAdd(cls + 8, (unsigned long long)(st_coop - st_coop_b0));
                }
                if (!flush)
                    b += 1u;
            }
            }
            // (diagnostic build: this wave's end, before the counters below queue up on their words)
            const unsigned long long rt_end = STATS ? __builtin_amdgcn_s_memrealtime() : 0ull;
            cy_total = now() - cy_begin;
            // the last workgroup to get here leaves the queue heads zero for the next launch.
            // Why relaxed atomics are enough (nothing but the queue words is ordered by them):
            //  * every access to the heads and to the count is an agent-scope atomic on one
            //    word, and atomics on one word are totally ordered (coherence), whatever their
            //    memory order;
            //  * a wave's last fetch_add on a head RETURNED before the wave left the loop above
            //    (the loop's exit depends on the value), all four waves pass the barrier below
            //    after that, and only then does thread 0 take its ticket: the ticket of a
            //    workgroup follows every head access of that workgroup in program order, and a
            //    returned atomic has been performed at L2;
            //  * so when a workgroup draws the last ticket, every head access of every
            //    workgroup has been performed, and its stores are the last writes to those
            //    words in this launch; the next launch on the stream starts after this one has
            //    ended, which makes them visible to it.
            // (The sums and outputs need no ordering here: nobody in this launch reads them.)
            __syncthreads();
            if (threadIdx.x == 0)
            {
                uint32_t* queue = cold()->queue;
                const uint32_t finished = __hip_atomic_fetch_add(queue + 256, 1u, __ATOMIC_RELAXED,
                                                                 __HIP_MEMORY_SCOPE_AGENT);
                if (finished == gridDim.x - 1u)
                {
                    for (uint32_t x = 0; x < 8u; ++x)
                        __hip_atomic_store(queue + x * 32u, 0u, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(queue + 256, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (STATS && lane == 0)
            {
                atomicAdd(stats + 0, (unsigned long long)st_batch);
                atomicAdd(stats + 1, (unsigned long long)st_node);
                atomicAdd(stats + 2, (unsigned long long)st_node_lanes);
                atomicAdd(stats + 3, (unsigned long long)st_leafph);
                atomicAdd(stats + 4, (unsigned long long)st_tile);
                atomicAdd(stats + 5, (unsigned long long)st_tile_g);
                atomicAdd(stats + 6, (unsigned long long)st_coop);
                atomicAdd(stats + 7, (unsigned long long)st_exact);
                atomicAdd(stats + 8, cy_total);
                atomicAdd(stats + 9, cy_head);
                atomicAdd(stats + 10, cy_node);
                atomicAdd(stats + 11, cy_coop);
                atomicAdd(stats + 12, cy_tile);
                atomicAdd(stats + 13, cy_exact);
                atomicAdd(stats + 14, cy_tail);
                atomicAdd(stats + 20, rt_end - rt_begin);
                atomicAdd(stats + 21, cy_pop);
                atomicAdd(stats + 170, (unsigned long long)st_walk);
                atomicAdd(stats + 171, (unsigned long long)st_walk_lanes);
                atomicAdd(stats + 172, (unsigned long long)st_walk_done);
                atomicAdd(stats + 173, cy_walk);
                atomicAdd(stats + 174, cy_dtop);
                atomicAdd(stats + 175, cy_dbot);
                atomicMax(stats + 22, cy_total);
                atomicMax(stats + 23, (unsigned long long)st_batch);
                // histogram of wave lifetimes (bins of 131072 cycles); first / last wave to start
                // and to end, on the 100 MHz clock all waves share
                atomicAdd(stats + 24 + min(63ull, cy_total >> 17), 1ull);
                atomicMax(stats + 15, ~rt_begin);
                atomicMax(stats + 16, rt_begin);
                atomicMax(stats + 17, ~rt_end);
                atomicMax(stats + 18, rt_end);
                (void)st_tile_len;
            }
        }
    hipError_t nn_tile_blocks_per_cu(int* out)
    {
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(out, nn_tile_kernel<false>, kBlock, 0);
    }

    uint32_t nn_tile_stack_entries() { return kStackK; }

    TilePlan plan_tile(uint64_t n, uint32_t max_blocks, uint32_t forced_bq, uint32_t cus)
    {
        // queries per batch: 64 (one per lane) when that fills every resident wave slot; a
        // smaller shard is cut into narrower batches (32/16/8/4 queries per wave) so that all
        // SIMDs get work and each wave's serial chain of bucket scans is shorter
        TilePlan p{};
        uint32_t bq = 64;
        if (forced_bq == 64 || forced_bq == 32 || forced_bq == 16 || forced_bq == 8 ||
            forced_bq == 4 || forced_bq == 2)
            bq = forced_bq;
        else
            while (bq > 4 && (n + bq - 1) / bq < (uint64_t)max_blocks * 4)
                bq >>= 1;
        p.batch_queries = bq;
        const uint64_t batches = (n + bq - 1) / bq;
        // every workgroup resident; at least one batch per wave
        uint64_t g = (batches + 3) / 4;
        if (g > max_blocks)
            g = max_blocks;
        // A shard that fills the chip only a few batches deep runs faster with fewer, faster
        // waves: the pass ends when the slowest wave has done its ceil(batches / waves) batches,
        // and a wave's speed falls with every other wave on its SIMD.  Measured (ms per NN pass,
        // aligned, 4 / 5 / 6 / 7 workgroups per CU): 1M points 0.138 / 0.148 / 0.158 / 0.168,
        // 2M 0.213 (5) / 0.221 / 0.230, 4M 0.346 (5) / 0.340 / 0.344, 64M: 7 is 4 % faster than 6.
        // So: the most workgroups per CU that still leave ~4 batches per wave, at least 4.
        if (cus != 0 && g == max_blocks && max_blocks % cus == 0 && max_blocks / cus > 4)
        {
            uint32_t per_cu = max_blocks / cus;
            while (per_cu > 4 && batches * 10 < (uint64_t)38 * 4 * cus * per_cu)
                --per_cu;
            g = (uint64_t)cus * per_cu;
        }
        p.grid_blocks = (uint32_t)(g < 1 ? 1 : g);
        // chunks: at most 2 batches each (measured at 64M: 1 -> 7.64, 2 -> 7.55, 4 -> 7.62,
        // 8 -> 7.87 ms per iteration; the L2 reason is at the top of the file), 1 when the shard
        // has fewer than ~16 batches per wave
        const uint64_t cb = batches / ((uint64_t)p.grid_blocks * 4 * 8);
        p.chunk_batches = (uint32_t)(cb < 1 ? 1 : (cb > 2 ? 2 : cb));
        p.n_chunks = (uint32_t)((batches + p.chunk_batches - 1) / p.chunk_batches);
        return p;
    }

    // ---------------------------------------------------------------------------------------
    // The pass that allocates nothing (icp_hip_nn_noalloc): the kernels above with their scratch
    // (work queue, per-chunk sums, partial rows, overflow stack) in a static device array.
    // In host code `scratch` is only a handle: the array's device address is resolved once per
    // device by hipGetSymbolAddress (each GPU holds its own copy at its own address) and host
    // code uses that pointer everywhere.
    // ---------------------------------------------------------------------------------------
    namespace
    {
        static __device__ __align__(256) unsigned char scratch[kNoAllocScratchBytes];

        constexpr int kMaxDevices = 64;
        unsigned char* g_scratch_ptr[kMaxDevices] = {};
        std::mutex g_scratch_mutex;

        constexpr size_t kScrQueue = 0;
        constexpr size_t kScrChunkSums = 4096;
        constexpr size_t kScrPartials = kScrChunkSums + (size_t)kNoAllocMaxChunks * kCovSums * 8;
        constexpr size_t kScrSpill = kScrPartials + (size_t)1024 * kCovSums * 8;
        static_assert(kScrSpill + (size_t)kNoAllocSpillLevels * kNoAllocMaxBlocks * 256 * 8 ==
                          kNoAllocScratchBytes,
                      "scratch layout");
        static_assert(kScrChunkSums % 256 == 0 && kScrPartials % 256 == 0 && kScrSpill % 256 == 0,
                      "scratch regions keep the array's alignment");
    } // namespace

    hipError_t nn_noalloc_setup(int device)
    {
        if (device < 0 || device >= kMaxDevices)
            return hipErrorInvalidDevice;
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (g_scratch_ptr[device])
            return hipSuccess;
        void* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)); // the current device's copy
        if (e != hipSuccess)
            return e;
        if (!ptr)
            return hipErrorInvalidSymbol;
        g_scratch_ptr[device] = static_cast<unsigned char*>(ptr);
        return hipSuccess;
    }

    hipError_t launch_nn_noalloc(int device, const NnLaunch& a_in, double* sums_out, uint32_t cut,
                                 hipStream_t s)
    {
        if (device < 0 || device >= kMaxDevices)
            return hipErrorInvalidDevice;
        unsigned char* base;
        {
            std::lock_guard<std::mutex> lock(g_scratch_mutex);
            base = g_scratch_ptr[device];
        }
        if (!base)
            return hipErrorNotInitialized;
        if (a_in.n == 0 || a_in.n > kNoAllocMaxQueries ||
            a_in.height > (uint32_t)kStackK + kNoAllocSpillLevels)
            return hipErrorInvalidValue;
        const TilePlan p = plan_tile(a_in.n, kNoAllocMaxBlocks, 0);
        if ((size_t)p.n_chunks * kCovSums * 8 > kScrPartials - kScrChunkSums)
            return hipErrorInvalidValue;
        NnLaunch a = a_in;
        a.grid = p.grid_blocks;
        a.perm = nullptr;
        a.loop = nullptr;
        a.partials = nullptr;
        uint32_t* queue = reinterpret_cast<uint32_t*>(base + kScrQueue);
        double* chunk_sums = reinterpret_cast<double*>(base + kScrChunkSums);
        double* partials = reinterpret_cast<double*>(base + kScrPartials);
        uint2* spill = reinterpret_cast<uint2*>(base + kScrSpill);
        // the array keeps the last call's contents: the queue heads must start at zero
        hipError_t e = hipMemsetAsync(queue, 0, kQueueWords * sizeof(uint32_t), s);
        if (e != hipSuccess)
            return e;
        const TileLaunch q{p.chunk_batches, p.batch_queries, chunk_sums, queue, spill, nullptr, cut};
        if ((e = launch_nn_tile(a, q, s)) != hipSuccess)
            return e;
        if (sums_out)
            e = launch_chunk_sums(chunk_sums, p.n_chunks, partials, sums_out, s);
        return e;
    }

    hipError_t launch_nn_tile(const NnLaunch& a, const TileLaunch& q, hipStream_t s)
    {
        if (a.grid == 0 || q.chunk_batches == 0 || !q.queue)
            return hipErrorInvalidValue;
        if (a.height > (uint32_t)kStackK && !q.spill)
            return hipErrorInvalidValue;
        if (q.park && (q.park_max == 0 || q.park_max > kParkMaxLanes || !q.late || !q.late_cnt ||
                       q.late_rows == 0))
            return hipErrorInvalidValue;
        const uint32_t bq = q.batch_queries;
        if (bq != 64u && bq != 32u && bq != 16u && bq != 8u && bq != 4u && bq != 2u)
            return hipErrorInvalidValue;
        const uint32_t n_batches = (a.n + bq - 1) / bq;
        const uint32_t n_chunks = (n_batches + q.chunk_batches - 1) / q.chunk_batches;
        TileArgs k{};
        k.t = a.tree;
        k.px = a.px;
        k.py = a.py;
        k.pz = a.pz;
        k.mux = a.mu[0];
        k.muy = a.mu[1];
        k.muz = a.mu[2];
        k.n = a.n;
        k.yx = a.yx;
        k.yy = a.yy;
        k.yz = a.yz;
        k.cut = q.cut;
        k.batch_queries = bq;
        k.park_max = q.park ? (q.park_max | (q.park_wait << 8)) : 0u;
        k.want_rows = (a.row_out || a.dist_out) ? 1u : 0u;
        k.row_out = a.row_out;
        k.dist_out = a.dist_out;
        k.perm = a.perm;
        k.chunk_sums = q.chunk_sums;
        k.queue = q.queue;
        k.spill = q.spill;
        k.n_batches = n_batches;
        k.chunk_batches = q.chunk_batches;
        k.n_chunks = n_chunks;
        k.late_rows = q.late_rows;
        k.park = q.park;
        k.late = q.late;
        k.late_cnt = q.late_cnt;
        k.stats = q.stats;
        k.loop = a.loop;
        if (q.stats)
            hipLaunchKernelGGL(nn_tile_kernel<true>, dim3(a.grid), dim3(kBlock), 0, s, k);
        else
            hipLaunchKernelGGL(nn_tile_kernel<false>, dim3(a.grid), dim3(kBlock), 0, s, k);
        return hipGetLastError();
    }
} // namespace icp_dev

// nn_tile_kernel -- the production NN + cross-covariance kernel (gfx950, wave = 64).
//
// Same contract as nn_cov_kernel (kernels.hip): VPTree::search for every query of the shard
// (libcpu/libcpu/vp-tree.cc:125-187), bit for bit, + the nine sums of find_covariance
// (libcpu/libcpu/point-3d.cc:212-236).  What differs is how the work is laid on the machine, for
// a shard that is stored in a spatially coherent order (sort.hip):
//
//  * persistent grid; the shard is cut into chunks of 1..8 consecutive 64-query batches and every
//    wave pulls chunks from a queue.  There is one queue head per XCD group (blockIdx % 8,
//    round-robin dispatch) over one eighth of the shard each, so the waves of an XCD work on
//    neighbouring chunks at the same time (a bucket is fetched from HBM about once and then
//    lives in that XCD's L2); a group that runs dry steals from the others (load balance);
//  * every chunk writes its own twelve fp64 sums and two small kernels add the chunks in a fixed
//    order: the result does not depend on which wave ran which chunk (bitwise reproducible);
//  * the traversal stack holds only the near-side answers of the levels whose far side is being
//    searched (push on the way into the far side, pop on the way back): 4 entries per lane in
//    LDS instead of one per tree level; deeper nestings spill to a per-lane column in HBM
//    (counted in sum [11]);
//  * leaf buckets are scanned as (bucket points x queries) tiles on the matrix cores: the g lanes
//    of a wave that wait on the SAME bucket are served together.  The bucket is staged once in
//    LDS relative to its first point o (p' = p - o, w = |p'|^2), the queries likewise
//    (q' = q - o), and v_mfma_f32_16x16x4_f32 evaluates  w - 2 q'.p'  (= |q - p|^2 - |q'|^2 up
//    to rounding) for 16 points x 16 queries per instruction; the vector ALU only keeps, per
//    query, the smallest and second smallest value and where the smallest was.  This is a
//    FILTER, not the answer: see "Exactness" below.  One or two waiting lanes are served by a
//    whole-wave scan instead (coop_scan).
//
// Exactness of the bucket scan.  The reference keeps the first slot that minimises
// sqrt(d2) under strict < (vp-tree.cc:131-143), d2 = ((dx*dx)+(dy*dy))+(dz*dz) in fp32 without
// FMA; two different d2 may share one sqrt.  The matrix-core values differ from d2 - |q'|^2 by at
// most E0 = 9.01 u L^2 (u = 2^-24, L = |q'| + max |p'|; derivation at tile_scan).  The four
// points of the winning group are re-evaluated with the reference's arithmetic; their minimum m
// is the reference's answer with distance sqrt(m) iff (i) the runner-up inside the group is
// more than 4 ulps above m (then the roots differ by >= 1.41 ulps of the root and round to
// different floats) and (ii) the second smallest filter value exceeds the smallest by more than
// 2.5 E + 32 u max(m, filter minimum) with E = 32 u (|q'|^2 + max |p'|^2) >= 1.77 E0: then
// every point outside the group has a reference d2 more than 4 ulps above m.  Otherwise (about
// 1e-3 of the scans; every exact tie and duplicate point) the lane is re-served by the exact
// wave-cooperative scan of device_common.hh.

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
        constexpr int kStackK = 3;  // LDS stack entries per lane
        constexpr int kImage = 256; // bucket slots staged per window
        constexpr int kTileMinGroup = 3; // fewer lanes on a bucket: whole-wave scan per lane
        constexpr int kPathLevels = 20;  // levels of the node-record cache (deeper nodes: L2)
        using f32x4 = __attribute__((ext_vector_type(4))) float;

        enum : int
        {
            tDescend = 0, // arriving at `node` from its parent
            tReturn = 1,  // arriving at `node` from child `prev`, (rd, rid) = that child's answer
            tWaitLeaf = 2,
            tWaitExact = 3,
            tServed = 4, // rd holds the SQUARED distance of the answer, sqrt pending
            tDone = 5
        };

        struct __attribute__((aligned(32))) WaveLds
        {
            float st_d[kStackK * kWave];
            uint32_t st_i[kStackK * kWave];
            float img[4 * kImage]; // bucket window: x[256] y[256] z[256] as stored in the tree (slots
                                   // that are not the bucket's hold the origin o) and
                                   // |p - o|^2[256] (+inf there); matrix operand A = img - (o, 0)
            float4 qb[kWave]; // queries by rank in the group (matrix operand B), then their results
            // node records of the last descent, by tree level: the queries of consecutive batches
            // take almost the same path, so most node visits are served from here instead of L2
            uint32_t pc_tag[kPathLevels];
            DevNode pc_rec[kPathLevels];
            double sums[kCovSums + 4]; // this wave's running totals (fixed summation order)
        };
        static_assert(sizeof(WaveLds) <= 8192, "LDS budget: 4 waves x 8 KiB = 32 KiB per workgroup, 5 per CU");
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

        // Largest of a non-negative float over the wave (wave-uniform result).
        __device__ __forceinline__ float wave_max_nonneg(float v)
        {
            int x = __float_as_int(v);
            x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x111, 0xf, 0xf, false)); // row_shr:1
            x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x112, 0xf, 0xf, false)); // row_shr:2
            x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x114, 0xf, 0xf, false)); // row_shr:4
            x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x118, 0xf, 0xf, false)); // row_shr:8
            x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x142, 0xa, 0xf, false)); // row_bcast:15
            x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x143, 0xc, 0xf, false)); // row_bcast:31
            return __int_as_float(__builtin_amdgcn_readlane(x, 63));
        }

        // Tile scan of bucket [beg, beg+cnt) for the g lanes in `grp` (all of them wait on it).
        //
        // Filter (matrix cores).  o = the bucket's first point.  Staged: p' = fl(p - o) and
        // w = fl-chain |p'|^2 per slot (+inf, with p' = 0, in slots that are not the bucket's),
        // B = (-2 q'_x, -2 q'_y, -2 q'_z, 1) per query with q' = fl(q - o).  One
        // v_mfma_f32_16x16x4_f32 gives D[point i][query n] = fma-chain(w_i - 2 q'_n . p'_i) for 16
        // points x 16 queries: lane l holds, for query l & 15 of the tile, the four points
        // 16 (l >> 4) + 4 r + s (r = 0..3) of a 64-slot stretch, s = the instruction's number in
        // the stretch (its A operand is component s of one ds_read_b128).  Per query tile the
        // lane keeps the smallest group minimum, the second smallest (with multiplicity) and
        // the code of the smallest's group; the four lanes l & 15 + 16 h share a query and are
        // folded at the end.
        //
        // Error of the filter.  With S = the real |q - p|^2 of the float inputs, c = |q'|^2 and
        // u = 2^-24:  q - o = q' (1 + e), p - o = p' (1 + e'), |e|, |e'| <= u, so S differs from
        // S' = |q' - p'|^2 by at most 2 sqrt(S') u L + u^2 L^2, L = |q'| + |p'|; w carries 3 u |p'|^2;
        // the four roundings of the chain are each at most u (|p'|^2 + 2 |q'| |p'|) whatever their
        // order (partial sums of a dot product are bounded by the product of the norms).  Hence
        // |D + c - S| <= (2 + 3 + 4 + small) u L^2 <= 9.01 u L^2 =: E0, and L^2 <= 2 (c + max w).
        // The reference's own d2 is within 6 u S of S.  Member lanes then decide as described at
        // the top of the file; a bound that is too large only sends more lanes to the exact scan.
        //
        // On return a member lane has: out_m = the reference's d2 of the best point of the
        // winning group, out_idx = that point's slot in the bucket, out_ok = whether it is
        // proven to be the reference's answer.
        template <bool STATS>
        __device__ __forceinline__ void tile_scan(const DevTree& t, WaveLds& L, uint32_t beg,
                                                  uint32_t cnt, uint64_t grp, bool member, float qx,
                                                  float qy, float qz, int lane, float& out_m,
                                                  bool& out_ok, uint32_t& out_idx,
                                                  uint32_t& staged, float& img_r2,
                                                  unsigned long long* cyc)
        {
            const float kInf = __int_as_float(0x7f800000);
            const unsigned long long c0 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
            const uint32_t g = (uint32_t)__builtin_popcountll(grp); // 1..64
            const uint32_t T = (g + 15u) >> 4;                      // query tiles of 16
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(
                (uint32_t)(grp >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)grp, 0u));
            const bool single = cnt <= (uint32_t)kImage;
            const bool have_image = single && staged == beg;
            // first window's slots and the origin (the bucket's first point; wave-uniform): all
            // loads are issued before the first use, one round trip
            const uint32_t s0 = 4u * (uint32_t)lane;
            float4 X = make_float4(0.f, 0.f, 0.f, 0.f), Y = X, Z = X;
            if (!have_image && s0 < cnt)
            {
                const size_t o = (size_t)beg + s0;
                X = *reinterpret_cast<const float4*>(t.sx + o);
                Y = *reinterpret_cast<const float4*>(t.sy + o);
                Z = *reinterpret_cast<const float4*>(t.sz + o);
            }
            const float ox = t.sx[beg], oy = t.sy[beg], oz = t.sz[beg];
            const float qpx = __fsub_rn(qx, ox), qpy = __fsub_rn(qy, oy), qpz = __fsub_rn(qz, oz);
            if (member)
                L.qb[rank] = make_float4(__fmul_rn(qpx, -2.f), __fmul_rn(qpy, -2.f),
                                         __fmul_rn(qpz, -2.f), 1.f);
            const uint32_t lrow = (uint32_t)lane >> 4, lcol = (uint32_t)lane & 15u;
            // operand A: component lrow of the points 4 lcol .. 4 lcol + 3 of a 64-slot stretch,
            // relative to the origin (row 3 is |p'|^2 as staged)
            const float* const a_base = L.img + lrow * (uint32_t)kImage + 4u * lcol;
            const float a_sub = lrow == 0u ? ox : lrow == 1u ? oy : lrow == 2u ? oz : 0.f;
            float r2 = have_image ? img_r2 : 0.f;
            // stage one window of the bucket (`staged` = first slot of the bucket whose only
            // window the image holds: the batches of a chunk that share a bucket stage it once)
            auto stage = [&](uint32_t w0, uint32_t cw, bool preloaded) {
                if (!preloaded)
                {
                    X = Y = Z = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (s0 < cw)
                    {
                        const size_t o = (size_t)beg + w0 + s0;
                        X = *reinterpret_cast<const float4*>(t.sx + o);
                        Y = *reinterpret_cast<const float4*>(t.sy + o);
                        Z = *reinterpret_cast<const float4*>(t.sz + o);
                    }
                }
                float4 vw;
                auto rel = [&](float& x, float& y, float& z, bool in, float& rw) {
                    x = in ? x : ox;
                    y = in ? y : oy;
                    z = in ? z : oz;
                    const float dx = __fsub_rn(x, ox), dy = __fsub_rn(y, oy), dz = __fsub_rn(z, oz);
                    const float w = __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
                    rw = in ? w : kInf;
                };
                rel(X.x, Y.x, Z.x, s0 < cw, vw.x);
                rel(X.y, Y.y, Z.y, s0 + 1u < cw, vw.y);
                rel(X.z, Y.z, Z.z, s0 + 2u < cw, vw.z);
                rel(X.w, Y.w, Z.w, s0 + 3u < cw, vw.w);
                *reinterpret_cast<float4*>(L.img + s0) = X;
                *reinterpret_cast<float4*>(L.img + kImage + s0) = Y;
                *reinterpret_cast<float4*>(L.img + 2 * kImage + s0) = Z;
                *reinterpret_cast<float4*>(L.img + 3 * kImage + s0) = vw;
                // max |p'|^2 over the window's real slots
                float wmax = 0.f;
                wmax = fmaxf(wmax, vw.x == kInf ? 0.f : vw.x);
                wmax = fmaxf(wmax, vw.y == kInf ? 0.f : vw.y);
                wmax = fmaxf(wmax, vw.z == kInf ? 0.f : vw.z);
                wmax = fmaxf(wmax, vw.w == kInf ? 0.f : vw.w);
                r2 = fmaxf(r2, wave_max_nonneg(wmax));
            };
            if (single && !have_image)
            {
                stage(0u, cnt, true);
                staged = beg;
            }
            if (!single)
                staged = 0xFFFFFFFFu;
            wave_lds_fence();
            const unsigned long long c1 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
            const float* const qf = reinterpret_cast<const float*>(L.qb);
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            // one query tile (16 queries) at a time: the state of a tile is three registers
#pragma nounroll
            for (uint32_t q = 0; q < T; ++q)
            {
                // operand B: component lrow of query 16 q + lcol
                const float bq = qf[(16u * q + lcol) * 4u + lrow];
                float m = kInf, m2 = kInf;
                uint32_t best = 0u;
#pragma nounroll
                for (uint32_t w0 = 0; w0 < cnt; w0 += kImage) // one window unless the bucket is > 256
                {
                    const uint32_t cw = min((uint32_t)kImage, cnt - w0);
                    if (!single)
                    {
                        wave_lds_fence(); // the window before has been read
                        stage(w0, cw, q == 0u && w0 == 0u);
                        wave_lds_fence();
                    }
                    const uint32_t nJ = (cw + 63u) >> 6;
                    float4 An = *reinterpret_cast<const float4*>(a_base);
#pragma nounroll
                    for (uint32_t J = 0; J < nJ; ++J)
                    {
                        float4 A4 = An;
                        // the next stretch's operand is on its way while this one is multiplied
                        // (stretch 3 wraps to 0: a harmless re-read inside the image)
                        An = *reinterpret_cast<const float4*>(a_base + 64u * ((J + 1u) & 3u));
                        A4.x = __fsub_rn(A4.x, a_sub); // p' = fl(p - o), the subtraction w was made from
                        A4.y = __fsub_rn(A4.y, a_sub);
                        A4.z = __fsub_rn(A4.z, a_sub);
                        A4.w = __fsub_rn(A4.w, a_sub);
                        const uint32_t code0 = ((w0 >> 6) + J) << 2;
                        // the four products are issued back to back, then looked at
                        const f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A4.x, bq, zero, 0, 0, 0);
                        const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A4.y, bq, zero, 0, 0, 0);
                        const f32x4 d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(A4.z, bq, zero, 0, 0, 0);
                        const f32x4 d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(A4.w, bq, zero, 0, 0, 0);
                        auto track = [&](const f32x4& d, uint32_t code) {
                            const float gm = fminf(fminf(d[0], d[1]), fminf(d[2], d[3]));
                            best = (gm < m) ? code : best;
                            m2 = __builtin_amdgcn_fmed3f(m, m2, gm);
                            m = fminf(m, gm);
                        };
                        track(d0, code0);
                        track(d1, code0 + 1u);
                        track(d2, code0 + 2u);
                        track(d3, code0 + 3u);
                    }
                }
                // fold the four lanes (rows) that share the query: rows 0|1 and 2|3 by
                // v_permlane16_swap, then the halves by v_permlane32_swap (no LDS traffic);
                // ties keep either side: m2 == m then, and the lane goes to the exact scan
                uint32_t b = best | (lrow << 28);
                auto fold = [&](auto swap) {
                    const auto sm = swap(__float_as_uint(m), __float_as_uint(m), false, false);
                    const auto s2 = swap(__float_as_uint(m2), __float_as_uint(m2), false, false);
                    const auto sb = swap(b, b, false, false);
                    const float ma = __uint_as_float(sm[0]), mb = __uint_as_float(sm[1]);
                    m2 = fminf(fminf(__uint_as_float(s2[0]), __uint_as_float(s2[1])), fmaxf(ma, mb));
                    b = (mb < ma) ? sb[1] : sb[0];
                    m = fminf(ma, mb);
                };
                fold([](uint32_t x, uint32_t y, bool a, bool c) {
                    return __builtin_amdgcn_permlane16_swap(x, y, false, false);
                });
                fold([](uint32_t x, uint32_t y, bool a, bool c) {
                    return __builtin_amdgcn_permlane32_swap(x, y, false, false);
                });
                if (lrow == 0u) // row 0 publishes by rank (this tile's queries have been read)
                    L.qb[16u * q + lcol] = make_float4(m, m2, __uint_as_float(b), 0.f);
            }
            img_r2 = r2;
            const unsigned long long c2 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
            wave_lds_fence();
            const float4 r = L.qb[member ? rank : 0u];
            wave_lds_fence();
            // ---- the winning group again, with the reference's arithmetic ----
            const float a1 = r.x, a2 = r.y;
            const uint32_t b = __float_as_uint(r.z);
            const uint32_t code = b & 0x0FFFFFFFu;
            const uint32_t base = ((code >> 2) << 6) + ((b >> 28) << 4) + (code & 3u);
            float e[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
            {
                const uint32_t off = base + 4u * (uint32_t)k;
                e[k] = kInf;
                if (member && off < cnt)
                {
                    if (single) // the image still holds the bucket as stored in the tree
                        e[k] = sq_dist(qx, qy, qz, L.img[off], L.img[kImage + off],
                                       L.img[2 * kImage + off]);
                    else
                    {
                        const size_t o = (size_t)beg + off;
                        e[k] = sq_dist(qx, qy, qz, t.sx[o], t.sy[o], t.sz[o]);
                    }
                }
            }
            float em = kInf, e2 = kInf;
            track2(e[0], em, e2);
            track2(e[1], em, e2);
            track2(e[2], em, e2);
            track2(e[3], em, e2);
            const uint32_t c = (e[0] == em) ? 0u : (e[1] == em) ? 1u : (e[2] == em) ? 2u : 3u;
            const float cq = __fadd_rn(__fadd_rn(__fmul_rn(qpx, qpx), __fmul_rn(qpy, qpy)),
                                       __fmul_rn(qpz, qpz));
            // 2.5 E + 32 u max(m, filter minimum), E = 32 u (c + max w): 80 u = 4.77e-6
            const float thr = __fadd_rn(
                __fadd_rn(__fmul_rn(__fadd_rn(cq, r2), 4.77e-6f),
                          __fmul_rn(fmaxf(em, __fadd_rn(a1, cq)), 1.91e-6f)),
                1e-30f);
            out_m = em;
            out_idx = base + 4u * c;
            out_ok = (__float_as_uint(e2) - __float_as_uint(em)) > 4u && em < kInf &&
                     __fsub_rn(a2, a1) > thr;
            if (STATS)
            {
                const unsigned long long c3 = __builtin_amdgcn_s_memtime();
                cyc[0] += c1 - c0;
                cyc[1] += c2 - c1;
                cyc[2] += c3 - c2;
                cyc[3] += have_image ? 1ull : 0ull;
            }
        }

        // Whole-wave scan of one bucket of <= 256 slots for ONE query (arguments wave-uniform):
        // the cheaper form when only one or two lanes wait on the bucket.  All chunk loads are
        // issued before the first use.  Returns the SQUARED distance of the reference's answer
        // (its sqrt is the reference's distance) and the answer's slot.
        __device__ __forceinline__ void coop_scan(const DevTree& t, float bx, float by, float bz,
                                                  uint32_t beg, uint32_t cnt, int lane, float& m_out,
                                                  uint32_t& win_out)
        {
            const float kInf = __int_as_float(0x7f800000);
            const uint32_t last = beg + cnt - 1u;
            float x[4], y[4], z[4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
            {
                x[c] = y[c] = z[c] = kInf;
                if ((uint32_t)(c * kWave) < cnt) // wave-uniform
                {
                    const uint32_t j = min(beg + (uint32_t)(c * kWave + lane), last);
                    x[c] = t.sx[j];
                    y[c] = t.sy[j];
                    z[c] = t.sz[j];
                }
            }
            float s[4];
            float smin = kInf;
#pragma unroll
            for (int c = 0; c < 4; ++c)
            {
                const float d2 = sq_dist(bx, by, bz, x[c], y[c], z[c]);
                s[c] = ((uint32_t)(c * kWave + lane) < cnt) ? d2 : kInf;
                smin = fminf(smin, s[c]);
            }
            const float m = wave_min_nonneg(smin);
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
            uint32_t win = beg; // a NaN-poisoned bucket keeps slot 0, like the reference's scan
            if (h[0])
                win = beg + (uint32_t)__builtin_ctzll(h[0]);
            else if (h[1])
                win = beg + 64u + (uint32_t)__builtin_ctzll(h[1]);
            else if (h[2])
                win = beg + 128u + (uint32_t)__builtin_ctzll(h[2]);
            else if (h[3])
                win = beg + 192u + (uint32_t)__builtin_ctzll(h[3]);
            m_out = m;
            win_out = win;
        }

    } // namespace

        template <bool STATS>
        __global__ void __launch_bounds__(kBlock, 5) // 5 workgroups per CU (LDS 5 x 32 KiB, <= 96 VGPRs)
            nn_tile_kernel(DevTree t, const float* __restrict__ px, const float* __restrict__ py,
                           const float* __restrict__ pz, float mux, float muy, float muz, uint32_t n,
                           float* __restrict__ yx, float* __restrict__ yy, float* __restrict__ yz,
                           uint32_t* __restrict__ row_out, float* __restrict__ dist_out,
                           const uint32_t* __restrict__ perm, double* __restrict__ chunk_sums,
                           uint32_t* __restrict__ queue, uint2* __restrict__ spill,
                           uint32_t n_batches, uint32_t chunk_batches, uint32_t n_chunks,
                           uint32_t cut, uint32_t batch_queries,
                           unsigned long long* __restrict__ stats,
                           const LoopState* __restrict__ loop)
        {
            __shared__ WaveLds s_w[kWavesPerBlock];
            if (loop)
            {
                // device-resident loop (loop.hip): the centroid comes from the previous
                // iteration's sums; an iteration enqueued past the end of the loop does nothing
                // (or has started its last iteration and only waits for that one's error)
                if (loop->done || loop->started >= loop->max_iters)
                    return;
                mux = loop->mu_q[0];
                muy = loop->mu_q[1];
                muz = loop->mu_q[2];
            }
            // wave-level event counts of the diagnostic build (ICP_HIP_TILE_STATS=1)
            uint32_t st_node = 0, st_tile = 0, st_tile_g = 0, st_tile_len = 0, st_coop = 0,
                     st_exact = 0, st_leafph = 0, st_batch = 0, st_node_lanes = 0;
            // shader cycles by phase (diagnostic build only)
            unsigned long long cy_node = 0, cy_tile = 0, cy_coop = 0, cy_exact = 0, cy_tail = 0,
                               cy_head = 0, cy_total = 0, cy_hits = 0, cy_pop = 0;
            auto now = [&]() -> unsigned long long {
                return STATS ? (unsigned long long)__builtin_amdgcn_s_memtime() : 0ull;
            };
            const unsigned long long cy_begin = now();
            const unsigned long long rt_begin = STATS ? __builtin_amdgcn_s_memrealtime() : 0ull;
            unsigned long long cy_ts[4] = {0ull, 0ull, 0ull, 0ull};

            const int lane = threadIdx.x & (kWave - 1);
            const int w = threadIdx.x >> 6;
            WaveLds& L = s_w[w];
            float* st_d = L.st_d + lane;
            uint32_t* st_i = L.st_i + lane;
            if (lane < kPathLevels)
                L.pc_tag[lane] = 0xFFFFFFFFu;
            uint32_t staged = 0xFFFFFFFFu; // wave-uniform: bucket whose image is in LDS
            float img_r2 = 0.f;            // wave-uniform: max |p'|^2 of that image
            uint32_t staged_cnt = 0;       // wave-uniform: slots of that bucket
            int lc_code = 0;               // wave-uniform: child code of the last bucket served
            uint32_t lc_beg = 0, lc_cnt = 0; // (0 is an internal node: matches no leaf) + its entry
            // this lane's column of the overflow stack: entry e at spill_col[e * spill_stride]
            const size_t spill_stride = (size_t)gridDim.x * kBlock;
            uint2* spill_col = spill + (size_t)blockIdx.x * kBlock + threadIdx.x;

            // XCD group x owns chunks [x*per_group, (x+1)*per_group); its head is queue[x*32]
            const uint32_t per_group = (n_chunks + 7u) >> 3;
            uint32_t victim = 0;
            uint32_t grp_x = blockIdx.x & 7u;
            for (;;)
            {
                // ---- pull the next chunk ----
                const unsigned long long cy_p0 = now();
                uint32_t chunk = 0;
                bool have = false;
                while (victim < 8u)
                {
                    const uint32_t lo = min(grp_x * per_group, n_chunks);
                    const uint32_t hi = min(lo + per_group, n_chunks);
                    uint32_t c = 0;
                    if (lane == 0)
                        c = __hip_atomic_fetch_add(queue + grp_x * 32u, 1u, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
                    c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
                    if (c < hi - lo)
                    {
                        chunk = lo + c;
                        have = true;
                        break;
                    }
                    victim += 1u; // this group is drained: steal from the next one
                    grp_x = (grp_x + 1u) & 7u;
                }
                cy_pop += now() - cy_p0;
                if (!have)
                    break;
                if (lane < kCovSums)
                    L.sums[lane] = 0.0;
                const uint32_t b_begin = chunk * chunk_batches;
                const uint32_t b_end = min(b_begin + chunk_batches, n_batches);
            for (uint32_t b = b_begin; b < b_end; ++b)
            {
                if (STATS)
                    st_batch += 1u;
                const unsigned long long cy_b0 = now();
                const uint32_t i = b * batch_queries + (uint32_t)lane;
                const bool valid = (uint32_t)lane < batch_queries && i < n;
                float qx = 0.f, qy = 0.f, qz = 0.f;
                if (valid)
                {
                    // p_centered = p - mu_p (point-3d.cc:202-206)
                    qx = __fsub_rn(px[i], mux);
                    qy = __fsub_rn(py[i], muy);
                    qz = __fsub_rn(pz[i], muz);
                }
                int node = t.root, prev = 0;
                int mode = tDescend;
                // The current descent chain (from the root, or from the far child of `ret_node`):
                // `chain_lb` is a lower bound of min |radius - d| over its nodes and `chain_d2` the
                // smallest squared centre distance.  If the bucket at its end answers with a
                // distance below chain_lb, every prune test on the way back up passes
                // (vp-tree.cc:152) and no centre is closer (:154): the lane jumps straight back to
                // `ret_node` (or is done) instead of climbing level by level.
                int ret_node = -1, top = t.root;
                float chain_lb = __int_as_float(0x7f800000), chain_d2 = chain_lb;
                uint32_t sp = 0, ovf = 0;
                uint32_t n_int = 0, n_pts = 0; // work counters of this query
                float rd = 0.f;
                uint32_t rid = 0;
                uint32_t lbeg = 0, lcnt = 0;
                cy_head += now() - cy_b0;
                if (!valid)
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
                if (t.root >= 0)
                {
                    for (;;)
                    {
                        const bool going = mode == tDescend;
                        const uint64_t goers = __ballot(going);
                        if (goers == 0ull)
                            break;
                        if (STATS)
                        {
                            st_node += 1u;
                            st_node_lanes += (uint32_t)__builtin_popcountll(goers);
                        }
                        uint32_t ci = 0;
                        bool hit = true;
                        DevNode N{};
                        if (going)
                        {
                            ci = 31u - (uint32_t)__builtin_clz((uint32_t)node + 1u);
                            const uint32_t cj = min(ci, (uint32_t)kPathLevels - 1u);
                            const uint32_t tag = L.pc_tag[cj];
                            N = L.pc_rec[cj];
                            hit = ci < (uint32_t)kPathLevels && tag == (uint32_t)node;
                        }
                        const bool all_hit = __ballot(!hit) == 0ull;
                        if (STATS && all_hit)
                            cy_hits += 1ull;
                        if (going)
                        {
                            if (!all_hit)
                            {
                                N = t.nodes[node];
                                if (lane == (int)__builtin_ctzll(goers) && ci < (uint32_t)kPathLevels)
                                {
                                    L.pc_tag[ci] = (uint32_t)node;
                                    L.pc_rec[ci] = N;
                                }
                            }
                            n_int += 1u;
                            const float d2 = sq_dist(qx, qy, qz, N.cx, N.cy, N.cz);
                            const float da = __builtin_amdgcn_sqrtf(d2); // as in the general step
                            const float lb = __fsub_rn(
                                fabsf(__fsub_rn(N.r, da)),
                                __fadd_rn(__fmul_rn(__fadd_rn(da, N.r), 0x1p-20f), 1e-18f));
                            chain_lb = fminf(chain_lb, lb);
                            chain_d2 = fminf(chain_d2, d2);
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
                                prev = go;
                                mode = tWaitLeaf;
                            }
                        }
                    }
                }
                cy_node += now() - cy_d0;

                for (;;)
                {
                    // ---- node phase: step every lane until it waits on a bucket or is done ----
                    const unsigned long long cy_n0 = now();
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
                        // the node record: from the path cache in LDS when every stepping
                        // lane finds its node there (the common case: neighbouring queries walk
                        // the same path), else from memory, and the first stepping lane leaves
                        // its record behind for the next visit of that tree level
                        uint32_t ci = 0;
                        bool hit = false;
                        DevNode N{};
                        if (mode <= tReturn)
                        {
                            // tag and record are read together: one LDS round trip
                            ci = 31u - (uint32_t)__builtin_clz((uint32_t)node + 1u);
                            const uint32_t cj = min(ci, (uint32_t)kPathLevels - 1u);
                            const uint32_t tag = L.pc_tag[cj];
                            N = L.pc_rec[cj];
                            hit = ci < (uint32_t)kPathLevels && tag == (uint32_t)node;
                        }
                        const bool all_hit = __ballot(mode <= tReturn && !hit) == 0ull;
                        if (STATS && all_hit)
                            cy_hits += 1ull;
                        if (mode <= tReturn)
                        {
                            if (!all_hit)
                            {
                                N = t.nodes[node];
                                if (lane == (int)__builtin_ctzll(stepping) &&
                                    ci < (uint32_t)kPathLevels)
                                {
                                    L.pc_tag[ci] = (uint32_t)node;
                                    L.pc_rec[ci] = N;
                                }
                            }
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
                                    fabsf(__fsub_rn(N.r, da)),
                                    __fadd_rn(__fmul_rn(__fadd_rn(da, N.r), 0x1p-20f), 1e-18f));
                                chain_lb = fminf(chain_lb, lb);
                                chain_d2 = fminf(chain_d2, d2);
                            }
                            const int nearc = in_near ? N.inside : N.outside;
                            const int farc = in_near ? N.outside : N.inside;
                            int go = nearc;
                            bool down = true;
                            if (mode == tReturn)
                            {
                                const float d = sqrtf(d2); // only the way up compares distances
                                bool up;
                                if (prev == nearc)
                                {
                                    // back from the near side (:152)
                                    if (rd < fabsf(__fsub_rn(N.r, d)))
                                        up = true;
                                    else
                                    {
                                        if (sp < (uint32_t)kStackK)
                                        {
                                            st_d[sp * kWave] = rd;
                                            st_i[sp * kWave] = rid;
                                        }
                                        else
                                        {
                                            spill_col[(sp - kStackK) * spill_stride] =
                                                make_uint2(__float_as_uint(rd), rid);
                                            ovf += 1u;
                                        }
                                        sp += 1u;
                                        go = farc; // :160
                                        up = false;
                                        ret_node = node; // a new chain starts at the far child
                                        top = farc;
                                        chain_lb = chain_d2 = __int_as_float(0x7f800000);
                                    }
                                }
                                else
                                {
                                    // back from the far side (:163): far wins only if strictly closer
                                    sp -= 1u;
                                    float nd;
                                    uint32_t ni;
                                    if (sp < (uint32_t)kStackK)
                                    {
                                        nd = st_d[sp * kWave];
                                        ni = st_i[sp * kWave];
                                    }
                                    else
                                    {
                                        const uint2 e = spill_col[(sp - kStackK) * spill_stride];
                                        nd = __uint_as_float(e.x);
                                        ni = e.y;
                                    }
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
                                    prev = node;
                                    node = N.parent;
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
                    // ---- leaf phase: serve the waiting lanes bucket by bucket ----
                    uint64_t pend = __ballot(mode == tWaitLeaf);
                    if (pend == 0ull)
                        break;
                    if (STATS)
                        st_leafph += 1u;
                    while (pend)
                    {
                        const int l = (int)__builtin_ctzll(pend);
                        const uint32_t beg = readlane_u(lbeg, l), cnt = readlane_u(lcnt, l);
                        lc_code = __builtin_amdgcn_readlane(prev, l);
                        lc_beg = beg;
                        lc_cnt = cnt;
                        const bool member = (mode == tWaitLeaf) && (lbeg == beg);
                        const uint64_t grp = __ballot(member);
                        pend &= ~grp;
                        const int g_all = __builtin_popcountll(grp);
                        if (g_all >= kTileMinGroup)
                        {
                            float m;
                            bool ok;
                            uint32_t idx;
                            const unsigned long long cy_t0 = now();
                            tile_scan<STATS>(t, L, beg, cnt, grp, member, qx, qy, qz, lane, m, ok,
                                             idx, staged, img_r2, cy_ts);
                            staged_cnt = cnt; // meaningful while `staged` is valid
                            cy_tile += now() - cy_t0;
                            if (STATS)
                            {
                                st_tile += 1u;
                                st_tile_g += (uint32_t)g_all;
                            }
                            if (member)
                            {
                                rd = m;
                                rid = beg + idx;
                                mode = ok ? tServed : tWaitExact;
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
                            // r * r is within 2 ulps of every d2 whose root is r
                            if (r < chain_lb && chain_d2 > __fmul_rn(__fmul_rn(r, r), 1.000002f))
                            {
                                node = ret_node;
                                prev = top;
                            }
                            mode = node < 0 ? tDone : tReturn;
                        }
                    }
                    if (mode == tServed)
                    {
                        const float m = rd; // squared distance of the bucket's answer
                        rd = sqrtf(m);
                        // no centre of the chain is strictly closer: d2_k > m (1 + 1e-6) gives
                        // sqrtf(d2_k) >= sqrtf(m)
                        if (rd < chain_lb && chain_d2 > __fmul_rn(m, 1.000001f))
                        {
                            node = ret_node;
                            prev = top;
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
                        // the answer is a slot of the bucket whose image is still in LDS
                        ox = L.img[in_image];
                        oy = L.img[kImage + in_image];
                        oz = L.img[2 * kImage + in_image];
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
                    if (yx)
                    {
                        yx[i] = ox;
                        yy[i] = oy;
                        yz[i] = oz;
                    }
                    if (row_out || dist_out)
                    {
                        const uint32_t o = perm ? perm[i] : i; // back to the caller's order
                        if (row_out)
                            row_out[o] = rid < t.n_slots ? t.srow[rid] : t.nrow[rid - t.n_slots];
                        if (dist_out)
                            dist_out[o] = rd;
                    }
                }
                // fp32 products as find_covariance forms them (point-3d.cc:220-231), summed in
                // fp64 in a fixed order: twelve per-lane values -> (permlane32 swap) six registers
                // of two 32-lane halves -> (permlane16 swap) three registers of four 16-lane rows
                // -> row sums by DPP -> this wave's running totals in LDS.
                {
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
                    v[11] = (double)ovf;
                    double h[6], r[3];
#pragma unroll
                    for (int k = 0; k < 6; ++k)
                        h[k] = swap_add_halves(v[2 * k], v[2 * k + 1]); // [v2k | v2k+1]
#pragma unroll
                    for (int k = 0; k < 3; ++k) // rows: v4k, v4k+2, v4k+1, v4k+3
                        r[k] = row_sum_f64(swap_add_rows(h[2 * k], h[2 * k + 1]));
                    if ((lane & 15) == 15) // the last lane of a row holds the row's sum
                    {
                        const int row = lane >> 4;
                        const int k0 = ((row & 1) << 1) | (row >> 1);
#pragma unroll
                        for (int k = 0; k < 3; ++k)
                            L.sums[4 * k + k0] += r[k];
                    }
                }
                cy_tail += now() - cy_e0;
            }
                if (chunk_sums && lane < kCovSums)
                    chunk_sums[(size_t)chunk * kCovSums + lane] = L.sums[lane];
            }
            // the last workgroup to get here leaves the queue heads zero for the next launch
            __syncthreads();
            if (threadIdx.x == 0)
            {
                const uint32_t finished = __hip_atomic_fetch_add(queue + 256, 1u, __ATOMIC_ACQ_REL,
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
                cy_total = now() - cy_begin;
                atomicAdd(stats + 8, cy_total);
                atomicAdd(stats + 9, cy_head);
                atomicAdd(stats + 10, cy_node);
                atomicAdd(stats + 11, cy_coop);
                atomicAdd(stats + 12, cy_tile);
                atomicAdd(stats + 13, cy_exact);
                atomicAdd(stats + 14, cy_tail);
                atomicAdd(stats + 15, cy_hits);
                atomicAdd(stats + 20, (unsigned long long)__builtin_amdgcn_s_memrealtime() - rt_begin);
                atomicAdd(stats + 21, cy_pop);
                atomicMax(stats + 22, cy_total);
                atomicMax(stats + 23, (unsigned long long)st_batch);
                atomicAdd(stats + 16, cy_ts[0]);
                atomicAdd(stats + 17, cy_ts[1]);
                atomicAdd(stats + 18, cy_ts[2]);
                atomicAdd(stats + 19, cy_ts[3]);
                (void)st_tile_len;
            }
        }
    hipError_t nn_tile_blocks_per_cu(int* out)
    {
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(out, nn_tile_kernel<false>, kBlock, 0);
    }

    uint32_t nn_tile_stack_entries() { return kStackK; }

    TilePlan plan_tile(uint64_t n, uint32_t max_blocks, uint32_t forced_bq)
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
        p.grid_blocks = (uint32_t)(g < 1 ? 1 : g);
        // chunks: ~8 per wave for load balance, at most 8 batches each
        const uint64_t cb = batches / ((uint64_t)p.grid_blocks * 4 * 8);
        p.chunk_batches = (uint32_t)(cb < 1 ? 1 : (cb > 8 ? 8 : cb));
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
        {
            uint32_t g1 = (p.n_chunks + 63) / 64;
            g1 = g1 > 1024 ? 1024 : g1;
            if ((e = launch_reduce_rows(chunk_sums, p.n_chunks, kCovSums, g1, partials, s)) != hipSuccess)
                return e;
            e = launch_reduce_partials(partials, g1, kCovSums, sums_out, s);
        }
        return e;
    }

    hipError_t launch_nn_tile(const NnLaunch& a, const TileLaunch& q, hipStream_t s)
    {
        if (a.grid == 0 || q.chunk_batches == 0 || !q.queue)
            return hipErrorInvalidValue;
        if (a.height > (uint32_t)kStackK && !q.spill)
            return hipErrorInvalidValue;
        const uint32_t bq = q.batch_queries;
        if (bq != 64u && bq != 32u && bq != 16u && bq != 8u && bq != 4u && bq != 2u)
            return hipErrorInvalidValue;
        const uint32_t n_batches = (a.n + bq - 1) / bq;
        const uint32_t n_chunks = (n_batches + q.chunk_batches - 1) / q.chunk_batches;
        if (q.stats)
            hipLaunchKernelGGL(nn_tile_kernel<true>, dim3(a.grid), dim3(kBlock), 0, s, a.tree, a.px,
                               a.py, a.pz, a.mu[0], a.mu[1], a.mu[2], a.n, a.yx, a.yy, a.yz,
                               a.row_out, a.dist_out, a.perm, q.chunk_sums, q.queue, q.spill,
                               n_batches, q.chunk_batches, n_chunks, q.cut, bq, q.stats, a.loop);
        else
            hipLaunchKernelGGL(nn_tile_kernel<false>, dim3(a.grid), dim3(kBlock), 0, s, a.tree, a.px,
                               a.py, a.pz, a.mu[0], a.mu[1], a.mu[2], a.n, a.yx, a.yy, a.yz,
                               a.row_out, a.dist_out, a.perm, q.chunk_sums, q.queue, q.spill,
                               n_batches, q.chunk_batches, n_chunks, q.cut, bq, nullptr, a.loop);
        return hipGetLastError();
    }
} // namespace icp_dev

// gfx950 (MI355X, CDNA4) kernels of the registration path.  wave = 64 lanes everywhere.
//
//   nn_cov_kernel        VPTree::search / closest (libcpu/libcpu/vp-tree.cc:125-187) for one shard
//                        of queries + the nine cross-covariance sums of find_covariance
//                        (libcpu/libcpu/point-3d.cc:212-236) fused behind it.
//                        Replaces closest_kernel (libgpu/libgpu/vp-tree.cu:181-261), subtract_rowwise
//                        (matrix.cu:243-270), product_kernel + sum_colwise (matrix.cu:369-421).
//   apply_error_kernel   apply_alignment (libcpu/libcpu/icp.cc:62-79) + compute_error (:81-95) + the
//                        next iteration's centroid sums.  Replaces apply_alignment_kernel /
//                        compute_error_kernel (libgpu/libgpu/icp.cu:64-125) and divide/block_sum.
//   sum_points_kernel    centroid sums of the untouched cloud (first iteration's mean, icp.cc:114).
//   reduce_partials      fixed-order second stage of every reduction (deterministic results).
//   aos_to_soa / soa_to_aos   Point3D <-> device SoA.
//
// Bit-exactness: every distance is sqrt(((dx*dx)+(dy*dy))+(dz*dz)) in fp32 with separately
// rounded multiplies and adds (__fmul_rn/__fadd_rn/__fsub_rn never contract; the file is also
// built with -ffp-contract=off) and a correctly rounded sqrtf, exactly as the reference's x86-64
// SSE2 build evaluates point-3d.cc:136-143.  Comparisons are made on the sqrt'ed values like
// vp-tree.cc:132-173 (two different squared distances can share one sqrt).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_common.hh"
#include "kernels.hh"
#include "loop_steps.hh"

namespace icp_dev
{
    enum : int
    {
        kDescend = 0, // arriving at `node` from its parent
        kReturn = 1,  // arriving at `node` from child `prev`, (rd, rid) = that child's answer
        kWaitLeaf = 2,
        kDone = 3
    };

    // One lane per query; the recursion of VPTree::search is unrolled into a per-lane state
    // machine.  Parent links replace a node stack; the only per-level state is the near side's
    // answer while the far side is searched, kept in LDS as [level][lane] (bank == lane, so it is
    // conflict free at any mix of depths).  Leaf buckets are scanned by the whole wave for one
    // query at a time with coalesced 256-byte loads.
    template <int MAXD, int NCH>
    __global__ void __launch_bounds__(kBlock)
        nn_cov_kernel(DevTree t, const float* __restrict__ px, const float* __restrict__ py,
                      const float* __restrict__ pz, float mux, float muy, float muz, uint32_t n,
                      float* __restrict__ yx, float* __restrict__ yy, float* __restrict__ yz,
                      uint32_t* __restrict__ row_out, float* __restrict__ dist_out,
                      const uint32_t* __restrict__ perm, double* __restrict__ partials)
    {
        __shared__ float s_near_d[kWavesPerBlock * MAXD * kWave];
        __shared__ uint32_t s_near_i[kWavesPerBlock * MAXD * kWave];
        __shared__ double s_red[kWavesPerBlock * kCovSums];

        const int lane = threadIdx.x & (kWave - 1);
        const int w = threadIdx.x >> 6;
        float* near_d = s_near_d + w * MAXD * kWave + lane;
        uint32_t* near_i = s_near_i + w * MAXD * kWave + lane;

        double acc[kCovSums];
#pragma unroll
        for (int k = 0; k < kCovSums; ++k)
            acc[k] = 0.0;

        const uint32_t wave_stride = gridDim.x * kWavesPerBlock * kWave;
        for (uint32_t base = (blockIdx.x * kWavesPerBlock + w) * kWave; base < n;
             base += wave_stride)
        {
            const uint32_t i = base + lane;
            const bool valid = i < n;
            float qx = 0.f, qy = 0.f, qz = 0.f;
            if (valid)
            {
                // p_centered = p - mu_p (point-3d.cc:202-206)
                qx = __fsub_rn(px[i], mux);
                qy = __fsub_rn(py[i], muy);
                qz = __fsub_rn(pz[i], muz);
            }
            int node = t.root, prev = 0, level = 0;
            int mode = kDescend;
            uint32_t n_int = 0, n_pts = 0; // work counters of this query
            float rd = 0.f;
            uint32_t rid = 0;
            uint32_t lbeg = 0, lcnt = 0;
            if (!valid)
                mode = kDone;
            else if (t.root < 0)
            {
                const uint2 li = t.leaf_info[~t.root];
                lbeg = li.x;
                lcnt = li.y;
                n_pts = li.y;
                prev = t.root;
                node = -1;
                mode = kWaitLeaf;
            }

            for (;;)
            {
                // ---- node phase: step every lane until it waits on a bucket or is done ----
                while (__ballot(mode <= kReturn) != 0ull)
                {
                    if (mode <= kReturn)
                    {
                        const DevNode N = t.nodes[node];
                        n_int += (mode == kDescend) ? 1u : 0u; // each internal node is entered once
                        const float d = sqrtf(sq_dist(qx, qy, qz, N.cx, N.cy, N.cz)); // :147
                        const bool in_near = d < N.r;                                // :148
                        const int nearc = in_near ? N.inside : N.outside;
                        const int farc = in_near ? N.outside : N.inside;
                        int go = nearc;
                        bool down = true;
                        if (mode == kReturn)
                        {
                            bool up;
                            if (prev == nearc)
                            {
                                // back from the near side (:152)
                                if (rd < fabsf(__fsub_rn(N.r, d)))
                                    up = true;
                                else
                                {
                                    near_d[level * kWave] = rd;
                                    near_i[level * kWave] = rid;
                                    go = farc; // :160
                                    up = false;
                                }
                            }
                            else
                            {
                                // back from the far side (:163): far wins only if strictly closer
                                const float nd = near_d[level * kWave];
                                const uint32_t ni = near_i[level * kWave];
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
                                if (d < rd) // :154 / :165 / :171 the centre wins only if strictly closer
                                {
                                    rd = d;
                                    rid = t.n_slots + (uint32_t)node;
                                }
                                prev = node;
                                node = N.parent;
                                level -= 1;
                                mode = node < 0 ? kDone : kReturn;
                            }
                        }
                        if (down)
                        {
                            if (go >= 0)
                            {
                                node = go;
                                level += 1;
                                mode = kDescend;
                            }
                            else
                            {
                                const uint2 li = t.leaf_info[~go];
                                lbeg = li.x;
                                lcnt = li.y;
                                n_pts += li.y;
                                prev = go;
                                mode = kWaitLeaf;
                            }
                        }
                    }
                }
                // ---- leaf phase: the wave serves its waiting lanes one bucket at a time ----
                uint64_t pend = __ballot(mode == kWaitLeaf);
                if (pend == 0ull)
                    break;
                while (pend)
                {
                    const int l = (int)__builtin_ctzll(pend);
                    pend &= pend - 1;
                    const float bx = readlane_f(qx, l), by = readlane_f(qy, l),
                                bz = readlane_f(qz, l);
                    const uint32_t beg = readlane_u(lbeg, l), cnt = readlane_u(lcnt, l);
                    float r;
                    uint32_t win;
                    scan_bucket<NCH>(t, bx, by, bz, beg, cnt, lane, r, win);
                    if (lane == l)
                    {
                        rd = r;
                        rid = win;
                        mode = node < 0 ? kDone : kReturn;
                    }
                }
            }

            if (valid)
            {
                float ox, oy, oz;
                uint32_t row;
                if (rid < t.n_slots)
                {
                    ox = t.sx[rid];
                    oy = t.sy[rid];
                    oz = t.sz[rid];
                    row = t.srow[rid];
                }
                else
                {
                    const DevNode N = t.nodes[rid - t.n_slots];
                    ox = N.cx;
                    oy = N.cy;
                    oz = N.cz;
                    row = t.nrow[rid - t.n_slots];
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
                        row_out[o] = row;
                    if (dist_out)
                        dist_out[o] = rd;
                }
                // fp32 products as find_covariance forms them (point-3d.cc:220-231), summed in fp64
                acc[0] += (double)__fmul_rn(qx, ox);
                acc[1] += (double)__fmul_rn(qx, oy);
                acc[2] += (double)__fmul_rn(qx, oz);
                acc[3] += (double)__fmul_rn(qy, ox);
                acc[4] += (double)__fmul_rn(qy, oy);
                acc[5] += (double)__fmul_rn(qy, oz);
                acc[6] += (double)__fmul_rn(qz, ox);
                acc[7] += (double)__fmul_rn(qz, oy);
                acc[8] += (double)__fmul_rn(qz, oz);
                acc[9] += (double)n_int;
                acc[10] += (double)n_pts;
            }
        }
        if (partials)
            block_store_partials<kCovSums>(acc, s_red, partials);
    }

    // p <- T p (icp.cc:66-74, left-to-right adds, no FMA); error terms (icp.cc:88-91);
    // sums: [0] error, [1..3] moved-point coordinates (next mean).
    __global__ void __launch_bounds__(kBlock)
        apply_error_kernel(float* __restrict__ px, float* __restrict__ py, float* __restrict__ pz,
                           const float* __restrict__ yx, const float* __restrict__ yy,
                           const float* __restrict__ yz, Mat34 T, float mmx, float mmy, float mmz,
                           uint32_t n, double* __restrict__ partials, LoopState* loop,
                           double* out, int tail)
    {
        __shared__ double s_red[kWavesPerBlock * kApplySums];
        if (loop)
        {
            // device-resident loop: this iteration's transform was written by loop_solve_kernel;
            // an iteration enqueued past the end of the loop moves nothing
            if (loop->done)
                return;
#pragma unroll
            for (int k = 0; k < 12; ++k)
                T.m[k] = loop->Tn[k];
            mmx = loop->mu_m[0];
            mmy = loop->mu_m[1];
            mmz = loop->mu_m[2];
        }
        double acc[kApplySums] = {0.0, 0.0, 0.0, 0.0};
        auto one = [&](float x, float y, float z, float ux, float uy, float uz, float& nx, float& ny,
                       float& nz) {
            nx = __fadd_rn(
                __fadd_rn(__fadd_rn(__fmul_rn(x, T.m[0]), __fmul_rn(y, T.m[1])), __fmul_rn(z, T.m[2])),
                T.m[3]);
            ny = __fadd_rn(
                __fadd_rn(__fadd_rn(__fmul_rn(x, T.m[4]), __fmul_rn(y, T.m[5])), __fmul_rn(z, T.m[6])),
                T.m[7]);
            nz = __fadd_rn(
                __fadd_rn(__fadd_rn(__fmul_rn(x, T.m[8]), __fmul_rn(y, T.m[9])), __fmul_rn(z, T.m[10])),
                T.m[11]);
            const float ex = __fsub_rn(__fadd_rn(ux, mmx), nx);
            const float ey = __fsub_rn(__fadd_rn(uy, mmy), ny);
            const float ez = __fsub_rn(__fadd_rn(uz, mmz), nz);
            const float e =
                __fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(ez, ez));
            acc[0] += (double)e;
            acc[1] += (double)nx;
            acc[2] += (double)ny;
            acc[3] += (double)nz;
        };
        // four points per lane and step (16-byte loads and stores; the arrays are 256-byte
        // aligned), the last n % 4 points one by one
        const uint32_t n4 = n >> 2;
        for (uint32_t g = blockIdx.x * kBlock + threadIdx.x; g < n4; g += gridDim.x * kBlock)
        {
            const float4 X = reinterpret_cast<const float4*>(px)[g];
            const float4 Y = reinterpret_cast<const float4*>(py)[g];
            const float4 Z = reinterpret_cast<const float4*>(pz)[g];
            const float4 U = reinterpret_cast<const float4*>(yx)[g];
            const float4 V = reinterpret_cast<const float4*>(yy)[g];
            const float4 W = reinterpret_cast<const float4*>(yz)[g];
            float4 A, B, C;
            one(X.x, Y.x, Z.x, U.x, V.x, W.x, A.x, B.x, C.x);
            one(X.y, Y.y, Z.y, U.y, V.y, W.y, A.y, B.y, C.y);
            one(X.z, Y.z, Z.z, U.z, V.z, W.z, A.z, B.z, C.z);
            one(X.w, Y.w, Z.w, U.w, V.w, W.w, A.w, B.w, C.w);
            reinterpret_cast<float4*>(px)[g] = A;
            reinterpret_cast<float4*>(py)[g] = B;
            reinterpret_cast<float4*>(pz)[g] = C;
        }
        if (blockIdx.x == 0 && threadIdx.x < (n & 3u))
        {
            const uint32_t i = (n4 << 2) + threadIdx.x;
            float nx, ny, nz;
            one(px[i], py[i], pz[i], yx[i], yy[i], yz[i], nx, ny, nz);
            px[i] = nx;
            py[i] = ny;
            pz[i] = nz;
        }
        if (!out)
        {
            block_store_partials<kApplySums>(acc, s_red, partials);
            return;
        }
        // a grid of ONE workgroup (small clouds): its partial sums are the sums, and the loop step
        // that consumes them runs here instead of in reduce_partials_kernel
        block_store_partials<kApplySums>(acc, s_red, out);
        __syncthreads();
        if (threadIdx.x == 0 && loop && tail == kTailFinish)
            loop_finish_step(loop, out);
    }

    __global__ void __launch_bounds__(kBlock)
        sum_points_kernel(const float* __restrict__ px, const float* __restrict__ py,
                          const float* __restrict__ pz, uint32_t n, double* __restrict__ partials)
    {
        __shared__ double s_red[kWavesPerBlock * kApplySums];
        double acc[kApplySums] = {0.0, 0.0, 0.0, 0.0};
        for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
        {
            acc[1] += (double)px[i];
            acc[2] += (double)py[i];
            acc[3] += (double)pz[i];
        }
        block_store_partials<kApplySums>(acc, s_red, partials);
    }

    // out[k] = sum over blocks of partials[b*K + k], in a fixed order (bitwise reproducible).
    __global__ void __launch_bounds__(kBlock)
        reduce_partials_kernel(const double* __restrict__ partials, uint32_t nblocks, int K,
                               double* out, LoopState* loop, int tail)
    {
        if (loop && loop->done)
            return; // an iteration enqueued past the end of the device-resident loop
        __shared__ double s[kSumRowsLds];
        block_sum_rows(partials, nblocks, K, out, s);
        // device-resident loop without a collective in between: the scalar step that consumes
        // these sums runs here instead of in a launch of its own (loop_steps.hh)
        if (threadIdx.x == 0 && loop)
        {
            if (tail == kTailSolve)
                loop_solve_step(loop, out);
            else if (tail == kTailFinish)
                loop_finish_step(loop, out);
        }
    }

    // Stage one of the chunk-sum reduction: block b adds rows [b*per, (b+1)*per) of rows[n][K]
    // in a fixed order (thread = 16 row lanes x 16 columns) -> out[b][K].
    __global__ void __launch_bounds__(kBlock)
        reduce_rows_kernel(const double* __restrict__ rows, uint32_t n_rows, int K, uint32_t per,
                           double* __restrict__ out)
    {
        __shared__ double s[16][17];
        const int col = threadIdx.x & 15, rl = threadIdx.x >> 4;
        const uint32_t begin = blockIdx.x * per;
        const uint32_t end = min(begin + per, n_rows);
        double v = 0.0;
        if (col < K)
            for (uint32_t r = begin + rl; r < end; r += 16)
                v += rows[(size_t)r * K + col];
        s[rl][col] = v;
        __syncthreads();
        if (threadIdx.x < (uint32_t)K)
        {
            double tot = 0.0;
            for (int j = 0; j < 16; ++j)
                tot += s[j][threadIdx.x];
            out[(size_t)blockIdx.x * K + threadIdx.x] = tot;
        }
    }

    // The same with late rows (nn_tile.hip, stragglers): chunk r adds its row of `rows` and then
    // its late_cnt[r] rows of `late` (nine fp32 products, three integer counters per row).
    __global__ void __launch_bounds__(kBlock)
        reduce_rows_late_kernel(const double* __restrict__ rows, uint32_t n_rows, uint32_t per,
                                const uint32_t* __restrict__ late,
                                const uint32_t* __restrict__ late_cnt, uint32_t late_rows,
                                double* __restrict__ out)
    {
        __shared__ double s[16][17];
        const int col = threadIdx.x & 15, rl = threadIdx.x >> 4;
        const uint32_t begin = blockIdx.x * per;
        const uint32_t end = min(begin + per, n_rows);
        double v = 0.0;
        if (col < kCovSums)
            for (uint32_t r = begin + rl; r < end; r += 16)
            {
                v += rows[(size_t)r * kCovSums + col];
                const uint32_t cnt = min(late_cnt[r], late_rows);
                const uint32_t* lr = late + (size_t)r * late_rows * kCovSums + col;
                for (uint32_t k = 0; k < cnt; ++k)
                {
                    const uint32_t w = lr[(size_t)k * kCovSums];
                    v += col < 9 ? (double)__uint_as_float(w) : (double)w;
                }
            }
        s[rl][col] = v;
        __syncthreads();
        if (threadIdx.x < (uint32_t)kCovSums)
        {
            double tot = 0.0;
            for (int j = 0; j < 16; ++j)
                tot += s[j][threadIdx.x];
            out[(size_t)blockIdx.x * kCovSums + threadIdx.x] = tot;
        }
    }

    __global__ void __launch_bounds__(kBlock)
        aos_to_soa_kernel(const float* __restrict__ aos, uint32_t n, float* __restrict__ x,
                          float* __restrict__ y, float* __restrict__ z)
    {
        for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
        {
            x[i] = aos[3 * (size_t)i];
            y[i] = aos[3 * (size_t)i + 1];
            z[i] = aos[3 * (size_t)i + 2];
        }
    }

    __global__ void __launch_bounds__(kBlock)
        soa_to_aos_kernel(const float* __restrict__ x, const float* __restrict__ y,
                          const float* __restrict__ z, uint32_t n, float* __restrict__ aos)
    {
        for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
        {
            aos[3 * (size_t)i] = x[i];
            aos[3 * (size_t)i + 1] = y[i];
            aos[3 * (size_t)i + 2] = z[i];
        }
    }

    // ---- ring order of the buckets (DevTree::ring) ----
    __global__ void __launch_bounds__(kBlock)
        leaf_parent_kernel(const DevNode* __restrict__ nodes, uint32_t n_nodes,
                           int32_t* __restrict__ leaf_parent)
    {
        for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_nodes; i += gridDim.x * kBlock)
        {
            const DevNode N = nodes[i];
            if (N.inside < 0)
                leaf_parent[~N.inside] = (int32_t)i;
            if (N.outside < 0)
                leaf_parent[~N.outside] = (int32_t)i;
        }
    }

    // One wave per bucket (<= 256 points, 4 per lane): key = (distance to the parent's centre, low
    // 8 mantissa bits cleared) | slot; the rank of a key among the bucket's keys (they are
    // distinct: the slot is part of them) is its place in the ring order.
    __global__ void __launch_bounds__(kBlock)
        ring_build_kernel(DevTree t, const int32_t* __restrict__ leaf_parent, uint32_t n_leaves,
                          uint32_t* __restrict__ ring)
    {
        __shared__ uint32_t s_keys[kWavesPerBlock][256];
        const int lane = threadIdx.x & (kWave - 1);
        const int w = threadIdx.x >> 6;
        for (uint32_t l = blockIdx.x * kWavesPerBlock + w; l < n_leaves; l += gridDim.x * kWavesPerBlock)
        {
            const uint2 li = t.leaf_info[l];
            const uint32_t beg = li.x, cnt = min(li.y, 256u);
            const uint32_t padded = (cnt + 15u) & ~15u;
            const DevNode N = t.nodes[leaf_parent[l]];
            uint32_t key[4], rank[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int c = 0; c < 4; ++c)
            {
                const uint32_t slot = 4u * (uint32_t)lane + (uint32_t)c;
                key[c] = 0x7f800000u;
                if (slot < cnt)
                {
                    const float rho = sqrtf(sq_dist(t.sx[beg + slot], t.sy[beg + slot], t.sz[beg + slot],
                                                    N.cx, N.cy, N.cz));
                    key[c] = (__float_as_uint(rho) & 0xFFFFFF00u) | slot;
                }
                s_keys[w][slot] = key[c];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (uint32_t j = 0; j < cnt; ++j)
            {
                const uint32_t k = s_keys[w][j];
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    rank[c] += k < key[c] ? 1u : 0u;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
            {
                const uint32_t slot = 4u * (uint32_t)lane + (uint32_t)c;
                if (slot < cnt)
                    ring[beg + rank[c]] = key[c];
                else if (slot < padded)
                    ring[beg + slot] = 0x7f800000u;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier(); // the next bucket overwrites this wave's keys
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }

    // SoA host tree arrays -> packed 32-byte node records.
    __global__ void __launch_bounds__(kBlock)
        pack_nodes_kernel(const float* cx, const float* cy, const float* cz, const float* r,
                          const int32_t* inside, const int32_t* outside, const int32_t* parent,
                          uint32_t n, DevNode* __restrict__ out)
    {
        for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
        {
            DevNode N;
            N.cx = cx[i];
            N.cy = cy[i];
            N.cz = cz[i];
            N.r = r[i];
            N.inside = inside[i];
            N.outside = outside[i];
            N.parent = parent[i];
            // thr = min { s : sqrtf(s) >= r }, found by stepping from r*r (a few ulps at most)
            float th = 0.f;
            if (N.r > 0.f)
            {
                th = __fmul_rn(N.r, N.r);
                while (th > 0.f && sqrtf(__uint_as_float(__float_as_uint(th) - 1u)) >= N.r)
                    th = __uint_as_float(__float_as_uint(th) - 1u);
                while (sqrtf(th) < N.r)
                    th = __uint_as_float(__float_as_uint(th) + 1u);
            }
            N.thr = th;
            out[i] = N;
        }
    }

    // ------------------------------------------------------------------------------------
    // launch wrappers (plain C++ so the C-ABI translation unit needs no kernel syntax)
    // ------------------------------------------------------------------------------------
    static inline uint32_t grid_for(uint64_t n, uint32_t cap_blocks)
    {
        uint64_t b = (n + kBlock - 1) / kBlock;
        if (b < 1)
            b = 1;
        return (uint32_t)(b < cap_blocks ? b : cap_blocks);
    }

    template <int MAXD>
    static hipError_t launch_nn_d(const NnLaunch& a, hipStream_t s)
    {
        const dim3 g(a.grid), b(kBlock);
#define ICP_NN_GO(NCH)                                                                          \
    hipLaunchKernelGGL((nn_cov_kernel<MAXD, NCH>), g, b, 0, s, a.tree, a.px, a.py, a.pz, a.mu[0], \
                       a.mu[1], a.mu[2], a.n, a.yx, a.yy, a.yz, a.row_out, a.dist_out, a.perm, a.partials)
        if (a.max_leaf_count <= 64)
            ICP_NN_GO(1);
        else if (a.max_leaf_count <= 128)
            ICP_NN_GO(2);
        else if (a.max_leaf_count <= 256)
            ICP_NN_GO(4);
        else
            ICP_NN_GO(0);
#undef ICP_NN_GO
        return hipGetLastError();
    }

    int nn_max_depth_supported() { return 64; }

    uint32_t nn_grid(uint64_t n) { return grid_for(n, kNnMaxBlocks); }
    uint32_t stream_grid(uint64_t n) { return grid_for(n, kStreamMaxBlocks); }

    hipError_t launch_nn(const NnLaunch& a, hipStream_t s)
    {
        // the LDS stack is indexed by the depth of an internal node: 0 .. height-1
        if (a.height <= 12)
            return launch_nn_d<12>(a, s);
        if (a.height <= 20)
            return launch_nn_d<20>(a, s);
        if (a.height <= 32)
            return launch_nn_d<32>(a, s);
        if (a.height <= 64)
            return launch_nn_d<64>(a, s);
        return hipErrorInvalidValue;
    }

    hipError_t launch_apply_error(float* px, float* py, float* pz, const float* yx, const float* yy,
                                  const float* yz, const Mat34& T, const float mu_m[3], uint32_t n,
                                  uint32_t grid, double* partials, LoopState* loop,
                                  hipStream_t s, double* out, int tail)
    {
        if (out && grid != 1)
            return hipErrorInvalidValue;
        hipLaunchKernelGGL(apply_error_kernel, dim3(grid), dim3(kBlock), 0, s, px, py, pz, yx, yy,
                           yz, T, mu_m[0], mu_m[1], mu_m[2], n, partials, loop, out, tail);
        return hipGetLastError();
    }

    hipError_t launch_chunk_sums(const double* chunk_sums, uint32_t n_chunks, double* partials,
                                 double* out, hipStream_t s, LoopState* loop, int tail)
    {
        if (n_chunks <= kChunksOneStage)
            return launch_reduce_partials(chunk_sums, n_chunks, kCovSums, out, s, loop, tail);
        uint32_t g1 = (n_chunks + 63) / 64;
        g1 = g1 > 1024 ? 1024 : g1;
        hipError_t e = launch_reduce_rows(chunk_sums, n_chunks, kCovSums, g1, partials, s);
        if (e != hipSuccess)
            return e;
        return launch_reduce_partials(partials, g1, kCovSums, out, s, loop, tail);
    }

    hipError_t launch_chunk_sums_late(const double* chunk_sums, uint32_t n_chunks,
                                      const uint32_t* late, const uint32_t* late_cnt,
                                      uint32_t late_rows, double* partials, double* out,
                                      hipStream_t s, LoopState* loop, int tail)
    {
        if (n_chunks == 0 || !late || !late_cnt)
            return hipErrorInvalidValue;
        uint32_t g1 = (n_chunks + 63) / 64;
        g1 = g1 > 1024 ? 1024 : g1;
        const uint32_t per = (n_chunks + g1 - 1) / g1;
        hipLaunchKernelGGL(reduce_rows_late_kernel, dim3(g1), dim3(kBlock), 0, s, chunk_sums, n_chunks,
                           per, late, late_cnt, late_rows, partials);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess)
            return e;
        return launch_reduce_partials(partials, g1, kCovSums, out, s, loop, tail);
    }

    hipError_t launch_sum_points(const float* px, const float* py, const float* pz, uint32_t n,
                                 uint32_t grid, double* partials, hipStream_t s)
    {
        hipLaunchKernelGGL(sum_points_kernel, dim3(grid), dim3(kBlock), 0, s, px, py, pz, n,
                           partials);
        return hipGetLastError();
    }

    hipError_t launch_reduce_partials(const double* partials, uint32_t nblocks, int K, double* out,
                                      hipStream_t s, LoopState* loop, int tail)
    {
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(kBlock), 0, s, partials, nblocks,
                           K, out, loop, tail);
        return hipGetLastError();
    }

    hipError_t launch_reduce_rows(const double* rows, uint32_t n_rows, int K, uint32_t grid,
                                  double* out, hipStream_t s)
    {
        if (K > 16 || grid == 0)
            return hipErrorInvalidValue;
        const uint32_t per = (n_rows + grid - 1) / grid;
        hipLaunchKernelGGL(reduce_rows_kernel, dim3(grid), dim3(kBlock), 0, s, rows, n_rows, K, per,
                           out);
        return hipGetLastError();
    }

    hipError_t launch_aos_to_soa(const float* aos, uint32_t n, float* x, float* y, float* z,
                                 hipStream_t s)
    {
        hipLaunchKernelGGL(aos_to_soa_kernel, dim3(stream_grid(n)), dim3(kBlock), 0, s, aos, n, x,
                           y, z);
        return hipGetLastError();
    }

    hipError_t launch_soa_to_aos(const float* x, const float* y, const float* z, uint32_t n,
                                 float* aos, hipStream_t s)
    {
        hipLaunchKernelGGL(soa_to_aos_kernel, dim3(stream_grid(n)), dim3(kBlock), 0, s, x, y, z, n,
                           aos);
        return hipGetLastError();
    }

    hipError_t launch_ring_build(const DevTree& t, uint32_t n_leaves, int32_t* leaf_parent,
                                 uint32_t* ring, hipStream_t s)
    {
        if (t.n_nodes == 0 || n_leaves == 0)
            return hipErrorInvalidValue;
        hipLaunchKernelGGL(leaf_parent_kernel, dim3(stream_grid(t.n_nodes)), dim3(kBlock), 0, s,
                           t.nodes, t.n_nodes, leaf_parent);
        const uint32_t blocks = (n_leaves + kWavesPerBlock - 1) / kWavesPerBlock;
        hipLaunchKernelGGL(ring_build_kernel, dim3(blocks < 8192u ? blocks : 8192u), dim3(kBlock), 0, s,
                           t, leaf_parent, n_leaves, ring);
        return hipGetLastError();
    }

    hipError_t launch_pack_nodes(const float* cx, const float* cy, const float* cz, const float* r,
                                 const int32_t* inside, const int32_t* outside,
                                 const int32_t* parent, uint32_t n, DevNode* out, hipStream_t s)
    {
        if (n == 0)
            return hipSuccess;
        hipLaunchKernelGGL(pack_nodes_kernel, dim3(stream_grid(n)), dim3(kBlock), 0, s, cx, cy, cz,
                           r, inside, outside, parent, n, out);
        return hipGetLastError();
    }
} // namespace icp_dev

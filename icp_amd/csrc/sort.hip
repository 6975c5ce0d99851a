// Spatial ordering of the cloud to move (set-up path, once per icp_hip_set_queries).
//
// Two stable sorts: Morton order of the positions, then (when the tree is already on the device)
// the index of the leaf a near-only descent of the centred point ends in.  Leaves are numbered
// left to right, so the final order is "by leaf region, Morton inside a region": at the first
// iteration the 64 queries of a wave wait on one or two buckets, and after the cloud has moved
// the order is still a tighter clustering than the plain Z-curve.  The keys only shape the
// order of evaluation; they need not be (and are not) bit-faithful to the search's decisions.
//
// The reference processes queries in input order (vp-tree.cc:182-184); every query is independent,
// so the order is free.  Here the shard is stored in Morton (Z-curve) order of the points'
// positions: the 64 queries of a wave are then neighbours in space, walk the same tree path and
// wait on the same one or two leaf buckets, which turns per-query bucket traffic from HBM into
// L1/L2 hits and lets a wave load a bucket once for all of its lanes.  A rigid motion keeps
// neighbours neighbours, so the order stays good over the iterations and is computed once.
// `perm[i]` = input row of the point stored at sorted position i; every output is un-permuted.
//
// The radix sort itself is rocPRIM's device-wide stable sort (deterministic order -> bitwise
// reproducible sums); keys and gathers are local kernels.

#include <hip/hip_runtime.h>
#include <cstring> // rocprim 7.2 headers use memset without including it
#include <rocprim/rocprim.hpp>
#include <stdint.h>

#include "kernels.hh"

namespace icp_dev
{
    namespace
    {
        constexpr int kBlock = 256;

        __global__ void __launch_bounds__(kBlock)
            minmax_kernel(const float* __restrict__ aos, uint32_t n, float* __restrict__ partials)
        {
            __shared__ float s[6][kBlock];
            float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
                for (int k = 0; k < 3; ++k)
                {
                    const float v = aos[3 * (size_t)i + k];
                    lo[k] = fminf(lo[k], v);
                    hi[k] = fmaxf(hi[k], v);
                    if (!isfinite(v)) // fmin/fmax skip a NaN: make it show (the host rejects it)
                    {
                        lo[k] = -INFINITY;
                        hi[k] = INFINITY;
                    }
                }
            for (int k = 0; k < 3; ++k)
            {
                s[k][threadIdx.x] = lo[k];
                s[3 + k][threadIdx.x] = hi[k];
            }
            __syncthreads();
            for (int off = kBlock / 2; off > 0; off >>= 1)
            {
                if ((int)threadIdx.x < off)
                    for (int k = 0; k < 3; ++k)
                    {
                        s[k][threadIdx.x] = fminf(s[k][threadIdx.x], s[k][threadIdx.x + off]);
                        s[3 + k][threadIdx.x] =
                            fmaxf(s[3 + k][threadIdx.x], s[3 + k][threadIdx.x + off]);
                    }
                __syncthreads();
            }
            if (threadIdx.x < 6)
                partials[blockIdx.x * 6 + threadIdx.x] = s[threadIdx.x][0];
        }

        __device__ __forceinline__ uint32_t spread10(uint32_t v) // 10 bits -> every third bit
        {
            v &= 0x3ffu;
            v = (v | (v << 16)) & 0x030000ffu;
            v = (v | (v << 8)) & 0x0300f00fu;
            v = (v | (v << 4)) & 0x030c30c3u;
            v = (v | (v << 2)) & 0x09249249u;
            return v;
        }

        __global__ void __launch_bounds__(kBlock)
            morton_kernel(const float* __restrict__ aos, uint32_t n, float lox, float loy, float loz,
                          float sx, float sy, float sz, uint32_t* __restrict__ keys,
                          uint32_t* __restrict__ idx)
        {
            for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
            {
                const float x = (aos[3 * (size_t)i] - lox) * sx;
                const float y = (aos[3 * (size_t)i + 1] - loy) * sy;
                const float z = (aos[3 * (size_t)i + 2] - loz) * sz;
                const uint32_t xi = (uint32_t)fminf(fmaxf(x, 0.f), 1023.f);
                const uint32_t yi = (uint32_t)fminf(fmaxf(y, 0.f), 1023.f);
                const uint32_t zi = (uint32_t)fminf(fmaxf(z, 0.f), 1023.f);
                keys[i] = spread10(xi) | (spread10(yi) << 1) | (spread10(zi) << 2);
                idx[i] = i;
            }
        }

        // key = index of the leaf reached by always taking the near child (d < radius)
        __global__ void __launch_bounds__(kBlock)
            leaf_key_kernel(DevTree t, const float* __restrict__ px, const float* __restrict__ py,
                            const float* __restrict__ pz, float mux, float muy, float muz, uint32_t n,
                            uint32_t* __restrict__ keys, uint32_t* __restrict__ idx)
        {
            for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
            {
                const float qx = px[i] - mux, qy = py[i] - muy, qz = pz[i] - muz;
                int node = t.root;
                while (node >= 0)
                {
                    const DevNode N = t.nodes[node];
                    const float dx = qx - N.cx, dy = qy - N.cy, dz = qz - N.cz;
                    node = (dx * dx + dy * dy + dz * dz < N.r * N.r) ? N.inside : N.outside;
                }
                keys[i] = (uint32_t)~node;
                idx[i] = i;
            }
        }

        __global__ void __launch_bounds__(kBlock)
            permute_soa_kernel(const float* __restrict__ x, const float* __restrict__ y,
                               const float* __restrict__ z, const uint32_t* __restrict__ perm,
                               const uint32_t* __restrict__ order, uint32_t n, float* __restrict__ ox,
                               float* __restrict__ oy, float* __restrict__ oz,
                               uint32_t* __restrict__ operm)
        {
            for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
            {
                const uint32_t r = order[i];
                ox[i] = x[r];
                oy[i] = y[r];
                oz[i] = z[r];
                operm[i] = perm[r];
            }
        }

        __global__ void __launch_bounds__(kBlock)
            gather_soa_kernel(const float* __restrict__ aos, const uint32_t* __restrict__ perm,
                              uint32_t n, float* __restrict__ x, float* __restrict__ y,
                              float* __restrict__ z)
        {
            for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
            {
                const size_t r = perm[i];
                x[i] = aos[3 * r];
                y[i] = aos[3 * r + 1];
                z[i] = aos[3 * r + 2];
            }
        }

        __global__ void __launch_bounds__(kBlock)
            scatter_aos_kernel(const float* __restrict__ x, const float* __restrict__ y,
                               const float* __restrict__ z, const uint32_t* __restrict__ perm,
                               uint32_t n, float* __restrict__ aos)
        {
            for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
            {
                const size_t r = perm[i];
                aos[3 * r] = x[i];
                aos[3 * r + 1] = y[i];
                aos[3 * r + 2] = z[i];
            }
        }
    } // namespace

    hipError_t launch_minmax(const float* aos, uint32_t n, uint32_t grid, float* partials,
                             hipStream_t s)
    {
        hipLaunchKernelGGL(minmax_kernel, dim3(grid), dim3(kBlock), 0, s, aos, n, partials);
        return hipGetLastError();
    }

    hipError_t launch_morton(const float* aos, uint32_t n, const float lo[3], const float scale[3],
                             uint32_t* keys, uint32_t* idx, hipStream_t s)
    {
        hipLaunchKernelGGL(morton_kernel, dim3(stream_grid(n)), dim3(kBlock), 0, s, aos, n, lo[0],
                           lo[1], lo[2], scale[0], scale[1], scale[2], keys, idx);
        return hipGetLastError();
    }

    hipError_t launch_leaf_keys(const DevTree& t, const float* px, const float* py, const float* pz,
                                const float mu[3], uint32_t n, uint32_t* keys, uint32_t* idx,
                                hipStream_t s)
    {
        hipLaunchKernelGGL(leaf_key_kernel, dim3(stream_grid(n)), dim3(kBlock), 0, s, t, px, py, pz,
                           mu[0], mu[1], mu[2], n, keys, idx);
        return hipGetLastError();
    }

    hipError_t launch_permute_soa(const float* x, const float* y, const float* z,
                                  const uint32_t* perm, const uint32_t* order, uint32_t n, float* ox,
                                  float* oy, float* oz, uint32_t* operm, hipStream_t s)
    {
        hipLaunchKernelGGL(permute_soa_kernel, dim3(stream_grid(n)), dim3(kBlock), 0, s, x, y, z,
                           perm, order, n, ox, oy, oz, operm);
        return hipGetLastError();
    }

    // Stable LSD radix sort of (key, idx) pairs on the low `bits` bits of the keys.  Set-up path: allocates and frees
    // rocPRIM's temporary storage.
    hipError_t sort_pairs_u32_temp_bytes(uint32_t n, uint32_t bits, size_t* bytes)
    {
        *bytes = 0;
        return rocprim::radix_sort_pairs(nullptr, *bytes, (const uint32_t*)nullptr,
                                         (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                         (uint32_t*)nullptr, (size_t)n, 0u, bits, (hipStream_t)0);
    }

    hipError_t sort_pairs_u32(const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* idx_in,
                              uint32_t* idx_out, uint32_t n, uint32_t bits, hipStream_t s,
                              void* temp, size_t temp_bytes)
    {
        size_t bytes = 0;
        hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys_out, idx_in, idx_out,
                                                 (size_t)n, 0u, bits, s);
        if (e != hipSuccess)
            return e;
        if (temp && temp_bytes >= bytes) // the caller's temporary storage: enqueued, nothing waited for
            return rocprim::radix_sort_pairs(temp, bytes, keys_in, keys_out, idx_in, idx_out,
                                             (size_t)n, 0u, bits, s);
        void* tmp = nullptr;
        if ((e = hipMalloc(&tmp, bytes ? bytes : 16)) != hipSuccess)
            return e;
        e = rocprim::radix_sort_pairs(tmp, bytes, keys_in, keys_out, idx_in, idx_out, (size_t)n, 0u,
                                      bits, s);
        hipError_t e2 = hipStreamSynchronize(s);
        (void)hipFree(tmp);
        return e != hipSuccess ? e : e2;
    }

    hipError_t launch_gather_soa(const float* aos, const uint32_t* perm, uint32_t n, float* x,
                                 float* y, float* z, hipStream_t s)
    {
        hipLaunchKernelGGL(gather_soa_kernel, dim3(stream_grid(n)), dim3(kBlock), 0, s, aos, perm, n,
                           x, y, z);
        return hipGetLastError();
    }

    hipError_t launch_scatter_aos(const float* x, const float* y, const float* z,
                                  const uint32_t* perm, uint32_t n, float* aos, hipStream_t s)
    {
        hipLaunchKernelGGL(scatter_aos_kernel, dim3(stream_grid(n)), dim3(kBlock), 0, s, x, y, z,
                           perm, n, aos);
        return hipGetLastError();
    }
} // namespace icp_dev

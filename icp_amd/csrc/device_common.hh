// Device-side helpers shared by the NN kernels (kernels.hip, nn_tile.hip).  gfx950, wave = 64.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hh"

namespace icp_dev
{
    constexpr int kWave = 64;
    constexpr int kBlock = 256;
    constexpr int kWavesPerBlock = kBlock / kWave;

    __device__ __forceinline__ float sq_dist(float ax, float ay, float az, float bx, float by,
                                             float bz)
    {
        const float dx = __fsub_rn(ax, bx);
        const float dy = __fsub_rn(ay, by);
        const float dz = __fsub_rn(az, bz);
        return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    }

    // The same sum with fused multiply-adds: NOT the reference's rounding (three roundings instead
    // of five, each within 2^-24 of its partial sum: like the reference's value it lies within
    // 3.01 * 2^-24 of the real sum of squares of the three fp32 differences).  Only ever a filter:
    // an answer is always re-evaluated with sq_dist.
    __device__ __forceinline__ float sq_dist_fma(float ax, float ay, float az, float bx, float by,
                                                 float bz)
    {
        const float dx = __fsub_rn(ax, bx);
        const float dy = __fsub_rn(ay, by);
        const float dz = __fsub_rn(az, bz);
        return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
    }

    __device__ __forceinline__ float readlane_f(float v, int l)
    {
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
    }
    __device__ __forceinline__ uint32_t readlane_u(uint32_t v, int l)
    {
        return (uint32_t)__builtin_amdgcn_readlane((int)v, l);
    }

    // Minimum over the 64 lanes of a NON-NEGATIVE float (incl. +inf): such floats order like their
    // bit patterns, so the reduction runs on ints through DPP row shifts and row broadcasts
    // (no LDS traffic).  Result is wave-uniform.
    __device__ __forceinline__ float wave_min_nonneg(float v)
    {
        int x = __float_as_int(v);
        // One instruction per step: the minimum with the shifted operand fused into the DPP
        // form (a lane without a source keeps its value).  Written out because the compiler
        // makes a copy, a v_mov_dpp and a v_min of each step (18 vector instructions for the six);
        // s_nop 1: a DPP operand must not be read for two cycles after the VALU wrote it.
        asm volatile("s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                     "s_nop 1"
                     : "+v"(x));
        return __int_as_float(__builtin_amdgcn_readlane(x, 63));
    }

    // Largest squared distance whose correctly rounded fp32 sqrt still equals r = sqrtf(s_min).
    // sqrt(s) rounds to r  <=>  sqrt(s) < (r + next(r))/2 =: m  <=>  s < m*m ; m has <= 25
    // significant bits so m*m is exact in fp64 and is never itself an fp32 value.
    __device__ __forceinline__ float sq_ceiling_for_root(float r)
    {
        const float r_up = __uint_as_float(__float_as_uint(r) + 1u);
        const double m = 0.5 * ((double)r + (double)r_up);
        const double m2 = m * m;
        float f = (float)m2; // round to nearest ...
        if ((double)f > m2)  // ... then step down to the largest float below m2
            f = __uint_as_float(__float_as_uint(f) - 1u);
        return f;
    }

    // Wave-cooperative scan of one leaf bucket for ONE query (all arguments wave-uniform).
    // Reference rule (vp-tree.cc:131-143): best = first slot i minimising sqrt(d2_i) under strict <.
    // Equivalent evaluation: r = sqrt(min d2); winner = first slot whose d2 <= ceiling(r).
    template <int NCH>
    __device__ __forceinline__ void scan_bucket(const DevTree& t, float bx, float by, float bz,
                                                uint32_t beg, uint32_t cnt, int lane, float& r_out,
                                                uint32_t& win_out)
    {
        const float kInf = __int_as_float(0x7f800000);
        float smin = kInf;
        uint32_t win = beg; // a NaN-poisoned bucket keeps slot 0, like the reference's scan
        if constexpr (NCH > 0)
        {
            float s[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c)
            {
                s[c] = kInf;
                if ((uint32_t)(c * kWave) < cnt) // wave-uniform
                {
                    const uint32_t j = c * kWave + lane;
                    // buckets are over-allocated by one wave of slots, so this never leaves the array
                    const float x = t.sx[beg + j], y = t.sy[beg + j], z = t.sz[beg + j];
                    const float d2 = sq_dist(bx, by, bz, x, y, z);
                    s[c] = (j < cnt) ? d2 : kInf;
                    smin = fminf(smin, s[c]);
                }
            }
            const float m = wave_min_nonneg(smin);
            const float r = sqrtf(m);
            const float hi = sq_ceiling_for_root(r);
            bool found = false;
#pragma unroll
            for (int c = 0; c < NCH; ++c)
            {
                if (!found && (uint32_t)(c * kWave) < cnt)
                {
                    const uint64_t hit = __ballot(s[c] <= hi);
                    if (hit)
                    {
                        win = beg + c * kWave + (uint32_t)__builtin_ctzll(hit);
                        found = true;
                    }
                }
            }
            r_out = r;
        }
        else
        {
#pragma nounroll
            for (uint32_t c0 = 0; c0 < cnt; c0 += kWave)
            {
                const uint32_t j = c0 + lane;
                const float x = t.sx[beg + j], y = t.sy[beg + j], z = t.sz[beg + j];
                const float d2 = sq_dist(bx, by, bz, x, y, z);
                smin = fminf(smin, (j < cnt) ? d2 : kInf);
            }
            const float m = wave_min_nonneg(smin);
            const float r = sqrtf(m);
            const float hi = sq_ceiling_for_root(r);
#pragma nounroll
            for (uint32_t c0 = 0; c0 < cnt; c0 += kWave)
            {
                const uint32_t j = c0 + lane;
                const float x = t.sx[beg + j], y = t.sy[beg + j], z = t.sz[beg + j];
                const float d2 = sq_dist(bx, by, bz, x, y, z);
                const uint64_t hit = __ballot((j < cnt) && (d2 <= hi));
                if (hit)
                {
                    win = beg + c0 + (uint32_t)__builtin_ctzll(hit);
                    break;
                }
            }
            r_out = r;
        }
        win_out = win;
    }

    __device__ __forceinline__ double wave_sum_f64(double v)
    {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            v += __shfl_down(v, off, kWave);
        return v; // lane 0 holds the sum
    }

    // Block-level fixed-order sum of K per-lane doubles -> partials[blockIdx.x * K + k].
    template <int K>
    __device__ __forceinline__ void block_store_partials(const double (&acc)[K], double* red,
                                                         double* partials)
    {
        const int lane = threadIdx.x & (kWave - 1);
        const int w = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < K; ++k)
        {
            const double s = wave_sum_f64(acc[k]);
            if (lane == 0)
                red[w * K + k] = s;
        }
        __syncthreads();
        if (threadIdx.x < K)
        {
            double s = 0.0;
#pragma unroll
            for (int ww = 0; ww < kWavesPerBlock; ++ww)
                s += red[ww * K + threadIdx.x];
            partials[(size_t)blockIdx.x * K + threadIdx.x] = s;
        }
    }

    // out[k] = sum over rows of rows[b*K + k] in a fixed order, by one workgroup of 256 threads:
    // thread = (row lane, column), 16 x 16 for K <= 16 and 64 x 4 for K <= 4; `s` is
    // kSumRowsLds doubles of LDS.
    constexpr int kSumRowsLds = 64 * 5;
    __device__ __forceinline__ void block_sum_rows(const double* rows, uint32_t n_rows, int K,
                                                   double* out, double* s)
    {
        const int cw = K <= 4 ? 4 : 16;        // columns per row lane group
        const int lanes = (int)kBlock / cw;    // row lanes
        const int col = threadIdx.x % cw, rl = threadIdx.x / cw;
        double v = 0.0;
        if (col < K)
        {
            for (uint32_t r = rl; r < n_rows; r += lanes)
                v += rows[(size_t)r * K + col];
        }
        s[rl * (cw + 1) + col] = v;
        __syncthreads();
        if (threadIdx.x < (uint32_t)K)
        {
            double tot = 0.0;
            for (int j = 0; j < lanes; ++j)
                tot += s[j * (cw + 1) + threadIdx.x];
            out[threadIdx.x] = tot;
        }
        __syncthreads();
    }

} // namespace icp_dev

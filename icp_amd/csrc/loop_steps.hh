// The scalar steps of the device-resident loop as device functions (one thread runs them):
// used by the stand-alone kernels of loop.hip and as the tail of reduce_partials_kernel.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "align3.hh"
#include "kernels.hh"

namespace icp_dev
{
    // T = I, error = +inf (icp.cc:101-104), mu_p = mean(p) (:114) from the centroid sums
    // (`sums` = kApplySums doubles)
    __host__ __device__ inline void loop_init_step(LoopState* st, const double* sums,
                                                   const float mu_m[3], float threshold,
                                                   uint32_t max_iters, double n_total,
                                                   bool sharded)
    {
        for (int i = 0; i < 16; ++i)
            st->T[i] = (i % 5 == 0) ? 1.0f : 0.0f;
        for (int i = 0; i < 12; ++i)
            st->Tn[i] = (i % 5 == 0) ? 1.0f : 0.0f;
        for (int k = 0; k < 3; ++k)
        {
            st->mu_m[k] = mu_m[k];
            st->mu_p[k] = st->mu_q[k] = (float)(sums[1 + k] / n_total);
        }
        const float inf = __builtin_huge_valf();
        st->error = inf;
        st->threshold = threshold;
        st->it = 0;
        st->started = 0;
        st->max_iters = max_iters;
        st->done = (max_iters > 0u && inf > threshold) ? 0u : 1u;
        st->sharded = sharded ? 1u : 0u;
        st->n_total = n_total;
        st->work[0] = st->work[1] = st->work[2] = 0.0;
        for (int i = 0; i < kLoopBlockMax; ++i)
            st->errors[i] = 0.0f;
    }

    // H from the covariance sums (`sums` = kCovSums doubles) -> find_alignment (icp.cc:27-60,
    // :119) -> T = Tn T (:121)
    __host__ __device__ inline void loop_solve_step(LoopState* st, const double* sums)
    {
        if (st->done)
            return;
        float H[9], mu_p[3], mu_m[3], Tn[16], T[16], Tt[16];
        for (int k = 0; k < 9; ++k)
            H[k] = (float)sums[k];
        for (int k = 0; k < 3; ++k)
        {
            mu_p[k] = st->mu_p[k];
            mu_m[k] = st->mu_m[k];
        }
        for (int k = 0; k < 16; ++k)
            T[k] = st->T[k];
        icp_align::alignment_from_covariance(H, mu_p, mu_m, Tn);
        icp_align::matmul4(Tn, T, Tt);
        for (int k = 0; k < 16; ++k)
            st->T[k] = Tt[k];
        for (int k = 0; k < 12; ++k)
            st->Tn[k] = Tn[k];
        st->work[0] += sums[9];
        st->work[1] += sums[10];
        st->work[2] += sums[11];
        // Sharded runs start the next NN pass before this iteration's apply pass has been summed
        // over the ranks: t = mu_m - r mu_p puts the centroid of the moved cloud on mu_m (up to
        // fp32 rounding), and that is what the pass subtracts there.  On one GPU the sums are at
        // hand first and loop_finish_step sets mean(new_p), as the reference does (icp.cc:114).
        if (st->sharded)
            for (int k = 0; k < 3; ++k)
                st->mu_q[k] = st->mu_m[k];
        st->started += 1u;
    }

    // error (icp.cc:123) and the next centroid from the apply pass' sums (`sums` = kApplySums
    // doubles); loop condition (:110)
    __host__ __device__ inline void loop_finish_step(LoopState* st, const double* sums)
    {
        if (st->done)
            return;
        const float error = (float)sums[0];
        for (int k = 0; k < 3; ++k)
            st->mu_p[k] = (float)(sums[1 + k] / st->n_total);
        if (!st->sharded) // mu_p = mean(new_p); p_centered = new_p - mu_p (icp.cc:114-115)
            for (int k = 0; k < 3; ++k)
                st->mu_q[k] = st->mu_p[k];
        st->error = error;
        st->errors[st->it % (uint32_t)kLoopBlockMax] = error;
        const uint32_t it = st->it + 1u;
        st->it = it;
        if (!(it < st->max_iters && error > st->threshold))
            st->done = 1u;
    }

    // One step on the kIterSums doubles of an iteration whose apply pass could not be finished
    // before the next NN pass started (sharded runs: one all-reduce per iteration carries the
    // apply pass' sums of iteration k together with the NN pass' sums of iteration k + 1).
    __host__ __device__ inline void loop_advance_step(LoopState* st, const double* sums)
    {
        if (st->done)
            return;
        if (st->started > st->it)
            loop_finish_step(st, sums + kApplyAt);
        if (!st->done)
            loop_solve_step(st, sums);
    }

    // true when an NN pass enqueued now has nothing to do: the loop is over, or every iteration
    // has been started and only the last error is outstanding
    __host__ __device__ inline bool loop_nn_idle(const LoopState* st)
    {
        return st->done || st->started >= st->max_iters;
    }
} // namespace icp_dev

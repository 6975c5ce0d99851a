// The scalar steps of the device-resident loop as device functions (one thread runs them):
// used by the stand-alone kernels of loop.hip and as the tail of reduce_partials_kernel.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "align3.hh"
#include "kernels.hh"

namespace icp_dev
{
    // H from the covariance sums -> find_alignment (icp.cc:27-60, :119) -> T = Tn T (:121)
    __device__ inline void loop_solve_step(LoopState* st, const double* sums)
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
    }

    // error (icp.cc:123) and the next centroid from the apply pass' sums; loop condition (:110)
    __device__ inline void loop_finish_step(LoopState* st, const double* sums)
    {
        if (st->done)
            return;
        const float error = (float)sums[0];
        for (int k = 0; k < 3; ++k)
            st->mu_p[k] = (float)(sums[1 + k] / st->n_total);
        st->error = error;
        st->errors[st->it % (uint32_t)kLoopBlockMax] = error;
        const uint32_t it = st->it + 1u;
        st->it = it;
        if (!(it < st->max_iters && error > st->threshold))
            st->done = 1u;
    }
} // namespace icp_dev

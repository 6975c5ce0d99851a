// The registration loop's scalar steps on the device, so that an iteration needs no host round
// trip: libcpu::icp's loop body (libcpu/libcpu/icp.cc:110-125) between the two big kernels.
//
//   loop_init    T = I, error = +inf (icp.cc:101-104), mu_p = mean(p) (:114) from the centroid sums
//   loop_solve   H from the covariance sums -> find_alignment (icp.cc:27-60; align3.hh, the same
//                source the host compiles) -> T = Tn T (:121); replaces the 9-float D2H, the host
//                SVD and the 9-float H2D of libgpu/libgpu/icp.cu:57 and its transpose / dot /
//                subtract / to_transformation kernels (:35-62)
//   loop_finish  error (:123) and the next centroid from the apply pass' sums; the loop condition
//                (:110) -> `done`
//
// One thread each: nine floats in, sixteen out.  No MFMA: nothing here is a dense contraction.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hh"
#include "loop_steps.hh"

namespace icp_dev
{
    namespace
    {
        __global__ void loop_init_kernel(LoopState* st, const double* sums, float mx, float my,
                                         float mz, float threshold, uint32_t max_iters,
                                         double n_total)
        {
            if (threadIdx.x != 0 || blockIdx.x != 0)
                return;
            for (int i = 0; i < 16; ++i)
                st->T[i] = (i % 5 == 0) ? 1.0f : 0.0f;
            for (int i = 0; i < 12; ++i)
                st->Tn[i] = (i % 5 == 0) ? 1.0f : 0.0f;
            st->mu_m[0] = mx;
            st->mu_m[1] = my;
            st->mu_m[2] = mz;
            for (int k = 0; k < 3; ++k)
                st->mu_p[k] = (float)(sums[1 + k] / n_total);
            const float inf = __int_as_float(0x7f800000);
            st->error = inf;
            st->threshold = threshold;
            st->it = 0;
            st->max_iters = max_iters;
            st->done = (max_iters > 0u && inf > threshold) ? 0u : 1u;
            st->n_total = n_total;
            st->work[0] = st->work[1] = st->work[2] = 0.0;
            for (int i = 0; i < kLoopBlockMax; ++i)
                st->errors[i] = 0.0f;
        }

        __global__ void loop_solve_kernel(LoopState* st, const double* sums)
        {
            if (threadIdx.x == 0 && blockIdx.x == 0)
                loop_solve_step(st, sums);
        }

        __global__ void loop_finish_kernel(LoopState* st, const double* sums)
        {
            if (threadIdx.x == 0 && blockIdx.x == 0)
                loop_finish_step(st, sums);
        }
    } // namespace

    hipError_t launch_loop_init(LoopState* st, const double* sums, const float mu_m[3],
                                float threshold, uint32_t max_iters, double n_total, hipStream_t s)
    {
        hipLaunchKernelGGL(loop_init_kernel, dim3(1), dim3(64), 0, s, st, sums, mu_m[0], mu_m[1],
                           mu_m[2], threshold, max_iters, n_total);
        return hipGetLastError();
    }
    hipError_t launch_loop_solve(LoopState* st, const double* sums, hipStream_t s)
    {
        hipLaunchKernelGGL(loop_solve_kernel, dim3(1), dim3(64), 0, s, st, sums);
        return hipGetLastError();
    }
    hipError_t launch_loop_finish(LoopState* st, const double* sums, hipStream_t s)
    {
        hipLaunchKernelGGL(loop_finish_kernel, dim3(1), dim3(64), 0, s, st, sums);
        return hipGetLastError();
    }
} // namespace icp_dev

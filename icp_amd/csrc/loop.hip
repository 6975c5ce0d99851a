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
//   loop_advance finish + solve on one set of sums: the step of a sharded run, whose single
//                all-reduce per iteration carries both
// The steps themselves are in loop_steps.hh (host and device: the host-driven loop runs the same
// source).  loop_solve / loop_finish run as the tail of reduce_partials_kernel (kernels.hip).
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
                                         double n_total, bool sharded)
        {
            if (threadIdx.x != 0 || blockIdx.x != 0)
                return;
            const float mu_m[3] = {mx, my, mz};
            loop_init_step(st, sums, mu_m, threshold, max_iters, n_total, sharded);
        }

        __global__ void loop_advance_kernel(LoopState* st, const double* sums)
        {
            if (threadIdx.x == 0 && blockIdx.x == 0)
                loop_advance_step(st, sums);
        }
    } // namespace

    hipError_t launch_loop_init(LoopState* st, const double* sums, const float mu_m[3],
                                float threshold, uint32_t max_iters, double n_total, bool sharded,
                                hipStream_t s)
    {
        hipLaunchKernelGGL(loop_init_kernel, dim3(1), dim3(64), 0, s, st, sums, mu_m[0], mu_m[1],
                           mu_m[2], threshold, max_iters, n_total, sharded);
        return hipGetLastError();
    }
    hipError_t launch_loop_advance(LoopState* st, const double* sums, hipStream_t s)
    {
        hipLaunchKernelGGL(loop_advance_kernel, dim3(1), dim3(64), 0, s, st, sums);
        return hipGetLastError();
    }
} // namespace icp_dev

// integration/libgpu_icp.cc -- the reference-side binding: `libgpu::icp` with the reference's own
// signature (libgpu/libgpu/icp.hh:19-21), implemented over the C ABI of include/icp_hip.h.
//
// A maintainer drops this file into libgpu/libgpu/ in place of icp.cu / vp-tree.cu / matrix.cu /
// svd.cc and links libicp_hip.so; `main` and `bench` keep calling libgpu::icp unchanged
// (main/main/main.cc:31-34, bench/bench/benchmark.cc:78).  It compiles against the reference's
// headers where they lie (checked by `make -C integration check` in the build container).
//
// Behaviour kept from libgpu::icp (libgpu/libgpu/icp.cu:128-161): inputs borrowed and never
// mutated, outputs by value, all device state created and destroyed inside the call, progress
// lines on std::cerr ("Starting iter i/N", "Error: e"), std::runtime_error on failure.

#include <iostream>
#include <stdexcept>
#include <string>

#include "icp_hip.h"
#include "libgpu/icp.hh"
#ifdef ICP_HIP_TREE_FROM_LIBCPU
// keep building the tree with libcpu::VPTree, as libgpu::icp does (libgpu/libgpu/icp.cu:141-142),
// and hand it over through the friend hook (libcpu/libcpu/vp-tree.hh:8-17)
#include "flatten_from_libcpu.hh"
#include "libcpu/vp-tree.hh"
#endif

namespace
{
    static_assert(sizeof(libcpu::Point3D) == 3 * sizeof(float), "Point3D must be 3 packed floats");

    struct Ctx
    {
        icp_hip_ctx* c = nullptr;
        ~Ctx() { icp_hip_destroy(c); }
    };
    void check(int rc, const icp_hip_ctx* c)
    {
        if (rc != ICP_HIP_OK)
            throw std::runtime_error(c ? icp_hip_last_error(c) : icp_hip_last_global_error());
    }
    void progress(uint64_t it, uint64_t max_iters, float error, void*)
    {
        std::cerr << "Starting iter " << it << "/" << max_iters << std::endl;
        std::cerr << "Error: " << error << std::endl;
    }
} // namespace

namespace libgpu
{
    std::tuple<utils::Matrix<float>, libcpu::point_list>
    icp(const libcpu::point_list& m, const libcpu::point_list& p, size_t iterations,
        float threshold, uint vp_threshold)
    {
        const float* m_xyz = reinterpret_cast<const float*>(m.data());
        const float* p_xyz = reinterpret_cast<const float*>(p.data());

        // device state lives exactly as long as the call, like the reference's RAII GPUMatrix /
        // GPUVPTree (matrix.cu:13-17, vp-tree.cu:12-16); a caller that registers many clouds can
        // keep one context alive instead (icp_hip_icp reuses its buffers)
        Ctx ctx;
        check(icp_hip_create(&ctx.c, -1), nullptr);
        float T[16];
        uint64_t iters = 0;
        float error = 0;
        libcpu::point_list new_p(p.size());
#ifdef ICP_HIP_TREE_FROM_LIBCPU
        // icp.cu:139-142 step by step: mu_m, the host tree over m - mu_m, flattened by the friend
        // class and uploaded in bulk (vp-tree.cu:58-112 copies it node by node), then the loop
        const libcpu::Point3D mu = libcpu::mean(m);
        const libcpu::point_list centred = libcpu::subtract(m, mu);
        const libcpu::VPTree cpu_tree(vp_threshold, centred);
        GPUVPTree::Flat flat;
        GPUVPTree::from_cpu(cpu_tree, vp_threshold, &centred, flat);
        check(icp_hip_set_tree(ctx.c, &flat.view), ctx.c);
        check(icp_hip_set_queries(ctx.c, p_xyz, p.size(), p.size()), ctx.c);
        const float mu_m[3] = {mu.x, mu.y, mu.z};
        check(icp_hip_run(ctx.c, mu_m, iterations, threshold, T, &iters, &error, progress, nullptr),
              ctx.c);
        check(icp_hip_get_points(ctx.c, reinterpret_cast<float*>(new_p.data())), ctx.c);
        (void)m_xyz;
#else
        // mean(m), the tree over m - mean(m) (built in HBM; the reference builds it on the host
        // and copies it node by node, icp.cu:141-142, vp-tree.cu:58-112), the loop, the download
        check(icp_hip_icp(ctx.c, m_xyz, m.size(), p_xyz, p.size(), iterations, threshold,
                          vp_threshold, T, reinterpret_cast<float*>(new_p.data()), &iters, &error,
                          progress, nullptr),
              ctx.c);
#endif

        utils::Matrix<float> transformation(4, 4);
        for (size_t i = 0; i < 4; ++i)
            for (size_t j = 0; j < 4; ++j)
                transformation.set(i, j, T[4 * i + j]);
        return {transformation, new_p};
    }
} // namespace libgpu

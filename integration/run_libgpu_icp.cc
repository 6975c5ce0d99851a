// integration/run_libgpu_icp.cc -- runs the reference-side binding for real: a caller that knows
// only the reference's interface (`libgpu/icp.hh`, `libcpu::point_list`) registers two clouds
// through libgpu::icp (libgpu/libgpu/icp.hh:19-21) as main does (main/main/main.cc:31-34), with
// integration/libgpu_icp.cc + libicp_hip.so behind it and the reference's own point-3d.cc (and,
// in the ICP_HIP_TREE_FROM_LIBCPU build, vp-tree.cc) linked in unmodified.
//
//   run_libgpu_icp <m.f32> <p.f32> <capacity> [--check-tree]
// The clouds are raw float32 x,y,z triples (the test writes the fixtures out that way; the
// reference's CSV files do not travel to the GPU box).  Prints the 4x4, the iteration count is on
// stderr like the reference prints it, then the first points of the moved cloud.
//   --check-tree (ICP_HIP_TREE_FROM_LIBCPU build): flatten a libcpu::VPTree through the friend
//   class, validate it, and compare it array by array with icp_hip_tree_build's tree.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "icp_hip.h"
#include "libgpu/icp.hh"
#ifdef ICP_HIP_TREE_FROM_LIBCPU
#include "flatten_from_libcpu.hh"
#endif

static libcpu::point_list read_f32(const char* path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f)
        throw std::runtime_error(std::string("cannot open ") + path);
    const std::streamsize bytes = f.tellg();
    f.seekg(0);
    libcpu::point_list v((size_t)bytes / sizeof(libcpu::Point3D));
    f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(v.size() * sizeof(libcpu::Point3D)));
    return v;
}

#ifdef ICP_HIP_TREE_FROM_LIBCPU
template <class T>
static bool same(const T* a, const T* b, size_t n)
{
    return n == 0 || std::memcmp(a, b, n * sizeof(T)) == 0;
}

static int check_tree(const libcpu::point_list& m, uint32_t capacity)
{
    const libcpu::Point3D mu = libcpu::mean(m);
    const libcpu::point_list centred = libcpu::subtract(m, mu);
    const libcpu::VPTree tree(capacity, centred);
    libgpu::GPUVPTree::Flat flat;
    libgpu::GPUVPTree::from_cpu(tree, capacity, &centred, flat);
    if (icp_hip_tree_validate(&flat.view) != ICP_HIP_OK)
    {
        std::cerr << "validate: " << icp_hip_last_global_error() << std::endl;
        return 2;
    }
    icp_flat_tree* own = nullptr;
    if (icp_hip_tree_build(reinterpret_cast<const float*>(centred.data()), centred.size(), capacity,
                           &own) != ICP_HIP_OK)
    {
        std::cerr << "icp_hip_tree_build: " << icp_hip_last_global_error() << std::endl;
        return 3;
    }
    const icp_flat_tree &a = flat.view, &b = *own;
    bool ok = a.n_points == b.n_points && a.n_slots == b.n_slots && a.n_nodes == b.n_nodes &&
              a.n_leaves == b.n_leaves && a.height == b.height && a.max_leaf_count == b.max_leaf_count &&
              a.root == b.root;
    ok = ok && same(a.node_cx, b.node_cx, a.n_nodes) && same(a.node_cy, b.node_cy, a.n_nodes) &&
         same(a.node_cz, b.node_cz, a.n_nodes) && same(a.node_radius, b.node_radius, a.n_nodes) &&
         same(a.node_inside, b.node_inside, a.n_nodes) && same(a.node_outside, b.node_outside, a.n_nodes) &&
         same(a.node_parent, b.node_parent, a.n_nodes) && same(a.leaf_begin, b.leaf_begin, a.n_leaves) &&
         same(a.leaf_count, b.leaf_count, a.n_leaves) && same(a.slot_x, b.slot_x, a.n_slots) &&
         same(a.slot_y, b.slot_y, a.n_slots) && same(a.slot_z, b.slot_z, a.n_slots);
    // rows may differ where points coincide; the coordinates they name may not
    for (uint64_t s = 0; ok && s < a.n_slots; ++s)
        if (a.slot_row[s] != 0xFFFFFFFFu)
            ok = std::memcmp(&centred[a.slot_row[s]], &centred[b.slot_row[s]], 12) == 0;
    for (uint32_t i = 0; ok && i < a.n_nodes; ++i)
        ok = std::memcmp(&centred[a.node_row[i]], &centred[b.node_row[i]], 12) == 0;
    icp_hip_tree_free(own);
    std::cout << "tree: " << a.n_nodes << " nodes, " << a.n_leaves << " leaves, height " << a.height
              << (ok ? ": flattened libcpu::VPTree == icp_hip_tree_build" : ": MISMATCH") << std::endl;
    return ok ? 0 : 4;
}
#endif

int main(int argc, char** argv)
{
    if (argc < 4)
    {
        std::cerr << "usage: run_libgpu_icp <m.f32> <p.f32> <capacity> [--check-tree]" << std::endl;
        return 1;
    }
    try
    {
        const libcpu::point_list m = read_f32(argv[1]), p = read_f32(argv[2]);
        const uint capacity = (uint)std::stoul(argv[3]);
        if (argc > 4 && std::strcmp(argv[4], "--check-tree") == 0)
        {
#ifdef ICP_HIP_TREE_FROM_LIBCPU
            return check_tree(m, capacity);
#else
            std::cerr << "--check-tree needs the ICP_HIP_TREE_FROM_LIBCPU build" << std::endl;
            return 1;
#endif
        }
        auto [T, new_p] = libgpu::icp(m, p, 200, 1e-5f, capacity); // main.cc:33 with its defaults
        std::printf("T");
        for (size_t i = 0; i < 4; ++i)
            for (size_t j = 0; j < 4; ++j)
                std::printf(" %.9g", T.get(i, j));
        std::printf("\n");
        for (size_t i = 0; i < new_p.size() && i < 5; ++i)
            std::printf("P %.9g %.9g %.9g\n", new_p[i].x, new_p[i].y, new_p[i].z);
        double worst = 0; // test_functionnal.cc:14-20: the moved cloud lands on the reference cloud
        for (size_t i = 0; i < new_p.size() && i < m.size(); ++i)
        {
            const double d[3] = {(double)new_p[i].x - m[i].x, (double)new_p[i].y - m[i].y,
                                 (double)new_p[i].z - m[i].z};
            for (double v : d)
                worst = std::max(worst, v < 0 ? -v : v);
        }
        std::printf("max_abs_moved_minus_reference %.6g\n", worst);
    }
    catch (const std::exception& e)
    {
        std::cerr << "error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}

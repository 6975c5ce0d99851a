// Device-side VP-tree build (tree_build.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace icp_dev
{
    // Result of build_tree_device: node / leaf tables on the host (small), bucket slots on the
    // device (the caller owns the four device arrays and frees them with hipFree).
    struct BuiltTree
    {
        uint64_t n_points = 0, n_slots = 0;
        uint32_t height = 0, capacity = 0, max_leaf_count = 0;
        int32_t root = ~0;
        std::vector<float> node_cx, node_cy, node_cz, node_radius;
        std::vector<int32_t> node_inside, node_outside, node_parent;
        std::vector<uint32_t> node_row;
        std::vector<uint32_t> leaf_begin, leaf_count;
        float *d_sx = nullptr, *d_sy = nullptr, *d_sz = nullptr; // n_slots + kSlotSlack entries
        uint32_t* d_srow = nullptr;                              // n_slots entries
    };

    // d_aos: the centred cloud on the device, Point3D layout.  Returns an icp_hip_status value
    // (0 ok, 1 argument, 2 degenerate cloud, 3 HIP error, 6 out of memory) and a message in `err`.
    int build_tree_device(const float* d_aos, uint32_t n, uint32_t capacity, hipStream_t st,
                          BuiltTree& out, std::string& err);
} // namespace icp_dev

// Host-side declarations shared by the C-ABI translation units (no HIP types here).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "icp_hip.h"

namespace icp_host
{
    // Owns the arrays an icp_flat_tree view points into.  `view` must stay the first member:
    // icp_hip_tree_build hands out &view and icp_hip_tree_free casts it back.
    struct FlatTreeOwner
    {
        icp_flat_tree view{};
        std::vector<float> node_cx, node_cy, node_cz, node_radius;
        std::vector<int32_t> node_inside, node_outside, node_parent;
        std::vector<uint32_t> node_row;
        std::vector<uint32_t> leaf_begin, leaf_count;
        std::vector<float> slot_x, slot_y, slot_z;
        std::vector<uint32_t> slot_row;
        void sync_view();
        ~FlatTreeOwner();
    };

    int build_flat_tree(const float* xyz, uint64_t n, uint32_t capacity, FlatTreeOwner** out,
                        std::string& err);
    int validate_flat_tree(const icp_flat_tree* t, std::string& err);
    // NaN / infinity anywhere in a[0..count)?  (the search degenerates on them: rejected up front)
    bool all_finite(const float* a, uint64_t count);
    extern const char* const kNonFiniteMessage;
    void mean_host(const float* a, uint64_t n, float out[3]);
    void center_host(const float* a, uint64_t n, const float mu[3], float* out);

    // loaders (io.cc): libcpu::read_csv restated (point-3d.cc:8-134) and a raw float32 reader
    int read_csv(const char* path, const char* x_field, const char* y_field, const char* z_field,
                 std::vector<float>& xyz, std::string& err);
    int read_f32(const char* path, std::vector<float>& xyz, std::string& err);

    // 3x3 cross-covariance (row = P coordinate, col = Y coordinate) -> 4x4 rigid transform.
    // libcpu/libcpu/icp.cc:27-60 (find_alignment after find_covariance) + to_transformation :11-25.
    void alignment_from_covariance(const float H[9], const float mu_p[3], const float mu_m[3],
                                   float T[16]);
    // utils::dot for 4x4, libcpu/libcpu/utils/matrix.hh:213-232
    void matmul4(const float a[16], const float b[16], float out[16]);
} // namespace icp_host

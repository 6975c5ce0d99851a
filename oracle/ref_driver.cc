// oracle/ref_driver.cc -- TEST INFRASTRUCTURE ONLY (build container only).
//
// A thin extern "C" window onto the REAL reference libcpu, whose own sources
// (/root/reference/libcpu/libcpu/{point-3d,vp-tree}.cc) are compiled unmodified next to this
// file by oracle/Makefile into oracle/_ref/libicp_ref.so.  It is used to (1) validate the
// restatement in oracle/icp_oracle.cc bit for bit and (2) generate the golden fixtures under
// tests/golden/ (tests/golden/make_golden.py).  /root/reference does not exist on the GPU
// box: the built library travels there like the product's own .so files, and the only thing on
// the box that may load it is the `cpu_baseline` leg of bench.py (kind "reference", timed beside
// the GPU path, never part of it), which falls back to the port when the file is absent; no test
// marked gpu, smoke() or product code depends on it.
//
// libcpu/libcpu/icp.cc is NOT built: it needs Eigen 3.3.7 (<Eigen/Dense>, icp.cc:3), which the
// reference fetches by URL and is absent offline.
//
// The private tree is reached through the hook the reference itself provides for its GPU
// backend: `friend class libgpu::GPUVPTree` (libcpu/libcpu/vp-tree.hh:8-17).

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "libcpu/point-3d.hh"
#include "libcpu/vp-tree.hh"
#include "matrix.hh"

namespace libgpu
{
    // The friend named by libcpu/libcpu/vp-tree.hh:17.  Read-only walker.
    class GPUVPTree
    {
    public:
        struct Dump
        {
            std::vector<uint8_t> kind;
            std::vector<float> node_f;
            std::vector<uint32_t> leaf_count;
            std::vector<float> leaf_xyz;
        };

        static void walk(const libcpu::VPTree& t, Dump& d)
        {
            if (!t.inside)
            {
                d.kind.push_back(0);
                d.leaf_count.push_back((uint32_t)t.points.size());
                for (const auto& p : t.points)
                {
                    d.leaf_xyz.push_back(p.x);
                    d.leaf_xyz.push_back(p.y);
                    d.leaf_xyz.push_back(p.z);
                }
                return;
            }
            d.kind.push_back(1);
            d.node_f.push_back(t.center.x);
            d.node_f.push_back(t.center.y);
            d.node_f.push_back(t.center.z);
            d.node_f.push_back(t.radius);
            walk(*t.inside, d);
            walk(*t.outside, d);
        }
    };
} // namespace libgpu

namespace
{
    libcpu::point_list to_list(const float* a, uint64_t n)
    {
        libcpu::point_list v(n);
        for (uint64_t i = 0; i < n; ++i)
            v[i] = libcpu::Point3D{a[3 * i], a[3 * i + 1], a[3 * i + 2]};
        return v;
    }
    void from_list(const libcpu::point_list& v, float* out)
    {
        for (size_t i = 0; i < v.size(); ++i)
        {
            out[3 * i] = v[i].x;
            out[3 * i + 1] = v[i].y;
            out[3 * i + 2] = v[i].z;
        }
    }
} // namespace

extern "C"
{
    // libcpu::read_csv (point-3d.cc:105-134).  Two-call protocol: out==NULL returns the count.
    int64_t ref_read_csv(const char* path, float* out, uint64_t cap)
    {
        try
        {
            auto v = libcpu::read_csv(path, "Points_0", "Points_1", "Points_2");
            if (out)
            {
                if (cap < v.size())
                    return -2;
                from_list(v, out);
            }
            return (int64_t)v.size();
        }
        catch (const std::exception&)
        {
            return -1;
        }
    }

    float ref_squared_distance(const float a[3], const float b[3])
    {
        return libcpu::squared_distance({a[0], a[1], a[2]}, {b[0], b[1], b[2]});
    }

    void ref_mean(const float* a, uint64_t n, float out[3])
    {
        auto m = libcpu::mean(to_list(a, n));
        out[0] = m.x;
        out[1] = m.y;
        out[2] = m.z;
    }

    void ref_subtract(const float* a, uint64_t n, const float mu[3], float* out)
    {
        from_list(libcpu::subtract(to_list(a, n), libcpu::Point3D{mu[0], mu[1], mu[2]}), out);
    }

    void ref_find_covariance(const float* p, const float* y, uint64_t n, float out[9])
    {
        auto [a, b, c, d, e, f, g, h, i] = libcpu::find_covariance(to_list(p, n), to_list(y, n));
        float s[9] = {a, b, c, d, e, f, g, h, i};
        std::memcpy(out, s, sizeof s);
    }

    // utils::dot on 4x4 (matrix.hh:213-232)
    void ref_matmul4(const float a[16], const float b[16], float out[16])
    {
        utils::Matrix<float> A(4, 4), B(4, 4);
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
            {
                A.set(i, j, a[i * 4 + j]);
                B.set(i, j, b[i * 4 + j]);
            }
        auto C = utils::dot(A, B);
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                out[i * 4 + j] = C.get(i, j);
    }

    // libcpu::dot(Matrix 3x3, Point3D) and subtract(Point3D, Point3D): t = mu_m - r*mu_p
    void ref_translation(const float r[9], const float mu_p[3], const float mu_m[3], float t[3])
    {
        utils::Matrix<float> R(3, 3);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                R.set(i, j, r[i * 3 + j]);
        auto v = libcpu::subtract(libcpu::Point3D{mu_m[0], mu_m[1], mu_m[2]},
                                  libcpu::dot(R, libcpu::Point3D{mu_p[0], mu_p[1], mu_p[2]}));
        t[0] = v.x;
        t[1] = v.y;
        t[2] = v.z;
    }

    void* ref_tree_build(uint32_t capacity, const float* pts, uint64_t n)
    {
        return new libcpu::VPTree(capacity, to_list(pts, n));
    }
    void ref_tree_free(void* t)
    {
        delete static_cast<libcpu::VPTree*>(t);
    }
    // VPTree::closest (vp-tree.cc:177-187)
    void ref_tree_closest(void* t, const float* q, uint64_t nq, float* out_xyz)
    {
        from_list(static_cast<libcpu::VPTree*>(t)->closest(to_list(q, nq)), out_xyz);
    }
    // VPTree::search distances (vp-tree.cc:125-175)
    void ref_tree_search_dist(void* t, const float* q, uint64_t nq, float* out_dist)
    {
        auto* tree = static_cast<libcpu::VPTree*>(t);
        for (uint64_t i = 0; i < nq; ++i)
            out_dist[i] =
                std::get<1>(tree->search(libcpu::Point3D{q[3 * i], q[3 * i + 1], q[3 * i + 2]}));
    }

    void ref_tree_dump_sizes(void* t, uint64_t* n_kind, uint64_t* n_internal, uint64_t* n_leaves,
                             uint64_t* n_leaf_pts)
    {
        libgpu::GPUVPTree::Dump d;
        libgpu::GPUVPTree::walk(*static_cast<libcpu::VPTree*>(t), d);
        *n_kind = d.kind.size();
        *n_internal = d.node_f.size() / 4;
        *n_leaves = d.leaf_count.size();
        *n_leaf_pts = d.leaf_xyz.size() / 3;
    }
    void ref_tree_dump(void* t, uint8_t* kind, float* node_f, uint32_t* leaf_count, float* leaf_xyz)
    {
        libgpu::GPUVPTree::Dump d;
        libgpu::GPUVPTree::walk(*static_cast<libcpu::VPTree*>(t), d);
        std::copy(d.kind.begin(), d.kind.end(), kind);
        std::copy(d.node_f.begin(), d.node_f.end(), node_f);
        std::copy(d.leaf_count.begin(), d.leaf_count.end(), leaf_count);
        std::copy(d.leaf_xyz.begin(), d.leaf_xyz.end(), leaf_xyz);
    }
}

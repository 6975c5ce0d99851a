// oracle/icp_oracle.cc -- TEST INFRASTRUCTURE ONLY.
//
// A CPU restatement of the reference registration path (alex-van-vliet/icp, libcpu).
// It exists so that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can
// check / time the HIP product against the reference algorithm.  Nothing under icp_amd/
// may include, link or call it.
//
// Parity pin: every function below except the 3x3 SVD is checked bit-for-bit against the
// reference's own libcpu sources compiled unmodified into oracle/_ref (see oracle/Makefile,
// oracle/ref_driver.cc, tests/test_oracle.py) and against the committed golden
// fixtures generated from that build (tests/golden/).  The SVD step is Eigen 3.3.7
// JacobiSVD<Matrix3f> (third-party, fetched by URL in the reference's CMakeLists.txt:27-30,
// absent offline): its 2x2-Jacobi algorithm is restated from its published description and
// pinned only through numpy.linalg.svd (polar factor V U^T) -- "parity unpinned" at the
// last-bit level for that step.
//
// Build: g++ -std=c++17 -O3 -DNDEBUG -fopenmp -ffp-contract=off (no -march=native): scalar
// SSE2 fp32 with separate multiply and add, which is the arithmetic the reference's CMake
// Release build produces.
//
// Every function cites the reference file:line it follows (paths relative to
// /root/reference).

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <vector>

#include <omp.h>

namespace
{
    struct P3
    {
        float x, y, z;
    };

    // A point that remembers which row of the input cloud it came from.
    struct TaggedPoint
    {
        P3 p;
        uint32_t row;
    };

    // libcpu/libcpu/point-3d.cc:136-143 -- (ax-bx)^2 + (ay-by)^2 + (az-bz)^2, left to right.
    inline float sqdist(const P3& a, const P3& b)
    {
        float dx = a.x - b.x;
        float dy = a.y - b.y;
        float dz = a.z - b.z;
        return dx * dx + dy * dy + dz * dz;
    }

    struct Found
    {
        P3 p;
        uint32_t row;
        float dist;
    };

    struct Counters
    {
        uint32_t internal_visited = 0;
        uint32_t leaf_points = 0;
        uint32_t leaves = 0;
    };

    // libcpu/libcpu/vp-tree.hh:15-49 -- a node is a leaf iff it has no `inside` child.
    struct VPNode
    {
        std::unique_ptr<VPNode> inside, outside;
        TaggedPoint center{};
        float radius = 0;
        std::vector<TaggedPoint> bucket;
        bool degenerate = false; // an empty side was produced (UB in the reference)

        VPNode() = default; // filled by orc_tree_from_flat

        // libcpu/libcpu/vp-tree.cc:54-123
        VPNode(uint32_t capacity, std::vector<TaggedPoint> pts)
        {
            if (pts.size() < capacity) // :56 strict <
            {
                bucket = std::move(pts);
                return;
            }
            center = pts.back(); // :65-66 vantage point = last element
            pts.pop_back();

            const size_t m = pts.size();
            std::vector<float> d2(m);
#pragma omp parallel for if (m > 65536)
            for (size_t i = 0; i < m; ++i)
                d2[i] = sqdist(center.p, pts[i].p); // :74 centre first

            // :48-52,77 -- element of rank m/2 (upper median); value is algorithm independent.
            float r2;
            {
                std::vector<float> tmp(d2);
                if (tmp.empty())
                {
                    degenerate = true;
                    return;
                }
                std::nth_element(tmp.begin(), tmp.begin() + m / 2, tmp.end());
                r2 = tmp[m / 2];
            }
            radius = std::sqrt(r2); // :78 (float overload)

            // :80-111 stable split, remembering the FIRST strict maximum of each side.
            std::vector<TaggedPoint> in, out;
            in.reserve(m / 2 + 1);
            out.reserve(m - m / 2 + 1);
            float in_far = 0, out_far = 0;
            size_t in_keep = 0, out_keep = 0;
            for (size_t i = 0; i < m; ++i)
            {
                float d = d2[i];
                if (d < r2)
                {
                    if (d > in_far)
                    {
                        in_far = d;
                        in_keep = in.size();
                    }
                    in.push_back(pts[i]);
                }
                else
                {
                    if (d > out_far)
                    {
                        out_far = d;
                        out_keep = out.size();
                    }
                    out.push_back(pts[i]);
                }
            }
            if (in.empty() || out.empty())
            {
                // :114-115 would call back() on an empty vector: undefined behaviour in the
                // reference.  Flag it instead of emulating it.
                degenerate = true;
                return;
            }
            std::swap(in.back(), in[in_keep]); // :114
            std::swap(out.back(), out[out_keep]); // :115
            pts.clear();
            pts.shrink_to_fit();
            d2.clear();
            d2.shrink_to_fit();

            inside = std::make_unique<VPNode>(capacity, std::move(in)); // :118-121
            outside = std::make_unique<VPNode>(capacity, std::move(out));
            degenerate = inside->degenerate || outside->degenerate;
        }

        // libcpu/libcpu/vp-tree.cc:125-175
        Found search(const P3& q, Counters* c) const
        {
            if (!inside)
            {
                // :131-143 linear scan on sqrt'ed distances, strict <  => lowest slot wins ties
                size_t best_i = 0;
                float best = std::sqrt(sqdist(q, bucket[0].p));
                for (size_t i = 1; i < bucket.size(); ++i)
                {
                    float d = std::sqrt(sqdist(q, bucket[i].p));
                    if (d < best)
                    {
                        best_i = i;
                        best = d;
                    }
                }
                if (c)
                {
                    c->leaf_points += (uint32_t)bucket.size();
                    c->leaves += 1;
                }
                return {bucket[best_i].p, bucket[best_i].row, best};
            }
            if (c)
                c->internal_visited += 1;

            float d = std::sqrt(sqdist(q, center.p)); // :147
            const bool near_is_inside = d < radius; // :148
            Found n = near_is_inside ? inside->search(q, c) : outside->search(q, c);

            if (n.dist < std::fabs(radius - d)) // :152
            {
                if (d < n.dist) // :154
                    return {center.p, center.row, d};
                return n;
            }
            Found o = near_is_inside ? outside->search(q, c) : inside->search(q, c); // :160
            if (o.dist < n.dist) // :163
            {
                if (d < o.dist)
                    return {center.p, center.row, d};
                return o;
            }
            if (d < n.dist) // :171
                return {center.p, center.row, d};
            return n;
        }

        void stats(uint64_t& internal, uint64_t& leaves, uint32_t depth, uint32_t& height,
                   uint64_t& min_leaf, uint64_t& max_leaf) const
        {
            if (!inside)
            {
                leaves += 1;
                height = std::max(height, depth);
                min_leaf = std::min<uint64_t>(min_leaf, bucket.size());
                max_leaf = std::max<uint64_t>(max_leaf, bucket.size());
                return;
            }
            internal += 1;
            inside->stats(internal, leaves, depth + 1, height, min_leaf, max_leaf);
            outside->stats(internal, leaves, depth + 1, height, min_leaf, max_leaf);
        }

        // Pre-order dump.  kinds: 1 internal (then inside subtree, then outside subtree), 0 leaf.
        void dump(std::vector<uint8_t>& kind, std::vector<float>& node_f /*cx,cy,cz,r*/,
                  std::vector<uint32_t>& node_row, std::vector<uint32_t>& leaf_count,
                  std::vector<float>& leaf_xyz, std::vector<uint32_t>& leaf_row) const
        {
            if (!inside)
            {
                kind.push_back(0);
                leaf_count.push_back((uint32_t)bucket.size());
                for (const auto& t : bucket)
                {
                    leaf_xyz.push_back(t.p.x);
                    leaf_xyz.push_back(t.p.y);
                    leaf_xyz.push_back(t.p.z);
                    leaf_row.push_back(t.row);
                }
                return;
            }
            kind.push_back(1);
            node_f.push_back(center.p.x);
            node_f.push_back(center.p.y);
            node_f.push_back(center.p.z);
            node_f.push_back(radius);
            node_row.push_back(center.row);
            inside->dump(kind, node_f, node_row, leaf_count, leaf_xyz, leaf_row);
            outside->dump(kind, node_f, node_row, leaf_count, leaf_xyz, leaf_row);
        }
    };

    // ---------------------------------------------------------------------------------
    // 3x3 SVD: restatement of Eigen 3.3.7 JacobiSVD<Matrix3f>(ComputeFullU|ComputeFullV),
    // square real case, no preconditioner (third-party; see header).  All fp32.
    // ---------------------------------------------------------------------------------
    struct Rot
    {
        float c, s;
    }; // Eigen JacobiRotation: [c s; -s c] applied as in applyOnTheLeft/Right

    inline Rot rot_transpose(Rot r)
    {
        return {r.c, -r.s};
    }
    // (a*b) as Eigen's JacobiRotation::operator*
    inline Rot rot_mul(Rot a, Rot b)
    {
        return {a.c * b.c - a.s * b.s, a.c * b.s + a.s * b.c};
    }

    // Eigen JacobiRotation::makeJacobi(x, y, z) for the symmetric 2x2 [[x, y],[y, z]]
    inline Rot make_jacobi(float x, float y, float z)
    {
        float deno = 2.0f * std::fabs(y);
        if (deno < FLT_MIN)
            return {1.0f, 0.0f};
        float tau = (x - z) / deno;
        float w = std::sqrt(tau * tau + 1.0f);
        float t = (tau > 0) ? 1.0f / (tau + w) : 1.0f / (tau - w);
        float sign_t = t > 0 ? 1.0f : -1.0f;
        float n = 1.0f / std::sqrt(t * t + 1.0f);
        Rot r;
        r.s = -sign_t * (y / std::fabs(y)) * std::fabs(t) * n;
        r.c = n;
        return r;
    }

    // Eigen internal::apply_rotation_in_the_plane(x, y, j): x' = c x + s y ; y' = -s x + c y
    inline void plane_rot(float& x, float& y, Rot j)
    {
        float xi = x, yi = y;
        x = j.c * xi + j.s * yi;
        y = -j.s * xi + j.c * yi;
    }
    // MatrixBase::applyOnTheLeft(p, q, j): rows p and q, rotation j as given (B = J B).
    inline void apply_left(float M[9], int p, int q, Rot j)
    {
        for (int k = 0; k < 3; ++k)
            plane_rot(M[p * 3 + k], M[q * 3 + k], j);
    }
    // MatrixBase::applyOnTheRight(p, q, j): columns p and q with j.transpose() (B = B J).
    inline void apply_right(float M[9], int p, int q, Rot j)
    {
        Rot a = rot_transpose(j);
        for (int k = 0; k < 3; ++k)
            plane_rot(M[k * 3 + p], M[k * 3 + q], a);
    }

    void jacobi_svd3(const float H[9], float U[9], float S[3], float V[9])
    {
        float W[9];
        float scale = 0;
        for (int i = 0; i < 9; ++i)
            scale = std::max(scale, std::fabs(H[i]));
        if (scale == 0.0f)
            scale = 1.0f;
        for (int i = 0; i < 9; ++i)
            W[i] = H[i] / scale;
        const float eyeM[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        std::memcpy(U, eyeM, sizeof eyeM);
        std::memcpy(V, eyeM, sizeof eyeM);

        const float precision = 2.0f * FLT_EPSILON;
        const float consider_as_zero = FLT_MIN;
        float max_diag = std::max({std::fabs(W[0]), std::fabs(W[4]), std::fabs(W[8])});

        bool finished = false;
        int guard = 0;
        while (!finished && guard++ < 1000)
        {
            finished = true;
            for (int p = 1; p < 3; ++p)
                for (int q = 0; q < p; ++q)
                {
                    float thr = std::max(consider_as_zero, precision * max_diag);
                    if (std::fabs(W[p * 3 + q]) > thr || std::fabs(W[q * 3 + p]) > thr)
                    {
                        finished = false;
                        // real_2x2_jacobi_svd(W, p, q, &j_left, &j_right)
                        float m00 = W[p * 3 + p], m01 = W[p * 3 + q];
                        float m10 = W[q * 3 + p], m11 = W[q * 3 + q];
                        Rot rot1;
                        float t = m00 + m11;
                        float d = m10 - m01;
                        if (std::fabs(d) < FLT_MIN)
                            rot1 = {1.0f, 0.0f};
                        else
                        {
                            float u = t / d;
                            float tmp = std::sqrt(1.0f + u * u);
                            rot1.s = 1.0f / tmp;
                            rot1.c = u / tmp;
                        }
                        // m.applyOnTheLeft(0,1,rot1): makes the 2x2 block symmetric
                        plane_rot(m00, m10, rot1);
                        plane_rot(m01, m11, rot1);
                        Rot j_right = make_jacobi(m00, m01, m11);
                        Rot j_left = rot_mul(rot1, rot_transpose(j_right));

                        apply_left(W, p, q, j_left);
                        apply_right(U, p, q, rot_transpose(j_left));
                        apply_right(W, p, q, j_right);
                        apply_right(V, p, q, j_right);

                        max_diag = std::max(
                            max_diag, std::max(std::fabs(W[p * 3 + p]), std::fabs(W[q * 3 + q])));
                    }
                }
        }
        for (int i = 0; i < 3; ++i)
        {
            float a = std::fabs(W[i * 3 + i]);
            S[i] = a;
            if (a != 0.0f)
            {
                float sgn = W[i * 3 + i] / a;
                for (int k = 0; k < 3; ++k)
                    U[k * 3 + i] *= sgn;
            }
        }
        for (int i = 0; i < 3; ++i)
            S[i] *= scale;
        // selection sort, descending, swapping columns of U and V alike
        for (int i = 0; i < 3; ++i)
        {
            int pos = i;
            float big = S[i];
            for (int k = i + 1; k < 3; ++k)
                if (S[k] > big)
                {
                    big = S[k];
                    pos = k;
                }
            if (big == 0.0f)
                break;
            if (pos != i)
            {
                std::swap(S[i], S[pos]);
                for (int k = 0; k < 3; ++k)
                {
                    std::swap(U[k * 3 + i], U[k * 3 + pos]);
                    std::swap(V[k * 3 + i], V[k * 3 + pos]);
                }
            }
        }
    }

    // libcpu/libcpu/utils/matrix.hh:213-232 -- val starts at 0, += a(i,k)*b(k,j), k ascending.
    void matmul4(const float a[16], const float b[16], float out[16])
    {
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
            {
                float v = 0;
                for (int k = 0; k < 4; ++k)
                    v += a[i * 4 + k] * b[k * 4 + j];
                out[i * 4 + j] = v;
            }
    }
} // namespace

extern "C"
{
    // libcpu/libcpu/point-3d.cc:177-192 -- each term is divided by len (size_t -> float) first.
    void orc_mean(const float* a, uint64_t n, float out[3])
    {
        size_t len = n;
        float x = 0, y = 0, z = 0;
        for (size_t i = 0; i < len; ++i)
        {
            x += a[3 * i + 0] / len;
            y += a[3 * i + 1] / len;
            z += a[3 * i + 2] / len;
        }
        out[0] = x;
        out[1] = y;
        out[2] = z;
    }

    // libcpu/libcpu/point-3d.cc:194-210
    void orc_subtract(const float* a, uint64_t n, const float mu[3], float* out)
    {
#pragma omp parallel for
        for (size_t i = 0; i < n; ++i)
        {
            out[3 * i + 0] = a[3 * i + 0] - mu[0];
            out[3 * i + 1] = a[3 * i + 1] - mu[1];
            out[3 * i + 2] = a[3 * i + 2] - mu[2];
        }
    }

    // libcpu/libcpu/point-3d.cc:212-236
    void orc_find_covariance(const float* p, const float* y, uint64_t n, float out[9])
    {
        float s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t i = 0; i < n; ++i)
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b)
                    s[3 * a + b] += p[3 * i + a] * y[3 * i + b];
        std::memcpy(out, s, sizeof s);
    }

    // libcpu/libcpu/vp-tree.cc:54 ; returns NULL for an empty cloud.
    void* orc_tree_build(uint32_t capacity, const float* pts, uint64_t n)
    {
        if (n == 0)
            return nullptr;
        std::vector<TaggedPoint> v(n);
        for (size_t i = 0; i < n; ++i)
            v[i] = {{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]}, (uint32_t)i};
        return new VPNode(capacity, std::move(v));
    }
    // Rebuild the pointer tree from a flat SoA tree (include/icp_hip.h) WITHOUT re-running the
    // build: lets the CPU baseline time the recursive reference search on trees whose host build
    // (71 s at 64M points in the reference) would not fit the benchmark's time box.
    // flat: pointers to the icp_flat_tree arrays, passed one by one to keep this file free of the
    // product header.
    void* orc_tree_from_flat(int32_t root, const float* ncx, const float* ncy, const float* ncz,
                             const float* nr, const int32_t* nin, const int32_t* nout,
                             const uint32_t* nrow, const uint32_t* lbeg, const uint32_t* lcnt,
                             const float* sx, const float* sy, const float* sz,
                             const uint32_t* srow)
    {
        struct Rec
        {
            static VPNode* make(int32_t code, const float* ncx, const float* ncy, const float* ncz,
                                const float* nr, const int32_t* nin, const int32_t* nout,
                                const uint32_t* nrow, const uint32_t* lbeg, const uint32_t* lcnt,
                                const float* sx, const float* sy, const float* sz,
                                const uint32_t* srow)
            {
                auto* n = new VPNode();
                if (code < 0)
                {
                    const uint32_t l = (uint32_t)~code;
                    n->bucket.resize(lcnt[l]);
                    for (uint32_t k = 0; k < lcnt[l]; ++k)
                    {
                        const uint64_t s = (uint64_t)lbeg[l] + k;
                        n->bucket[k] = {{sx[s], sy[s], sz[s]}, srow[s]};
                    }
                    return n;
                }
                n->center = {{ncx[code], ncy[code], ncz[code]}, nrow[code]};
                n->radius = nr[code];
                n->inside.reset(make(nin[code], ncx, ncy, ncz, nr, nin, nout, nrow, lbeg, lcnt, sx,
                                     sy, sz, srow));
                n->outside.reset(make(nout[code], ncx, ncy, ncz, nr, nin, nout, nrow, lbeg, lcnt,
                                      sx, sy, sz, srow));
                return n;
            }
        };
        return Rec::make(root, ncx, ncy, ncz, nr, nin, nout, nrow, lbeg, lcnt, sx, sy, sz, srow);
    }

    int orc_max_threads() { return omp_get_max_threads(); }

    void orc_tree_free(void* t)
    {
        delete static_cast<VPNode*>(t);
    }
    int orc_tree_degenerate(void* t)
    {
        return static_cast<VPNode*>(t)->degenerate ? 1 : 0;
    }
    void orc_tree_stats(void* t, uint64_t* internal, uint64_t* leaves, uint32_t* height,
                        uint64_t* min_leaf, uint64_t* max_leaf)
    {
        uint64_t a = 0, b = 0, mn = UINT64_MAX, mx = 0;
        uint32_t h = 0;
        static_cast<VPNode*>(t)->stats(a, b, 0, h, mn, mx);
        *internal = a;
        *leaves = b;
        *height = h;
        *min_leaf = mn;
        *max_leaf = mx;
    }

    // libcpu/libcpu/vp-tree.cc:177-187.  out_* may be NULL.  n_int/n_pts: per-query counts of
    // internal nodes visited / leaf points scanned (for the algorithmic-bytes model).
    void orc_tree_closest(void* t, const float* q, uint64_t nq, float* out_xyz, uint32_t* out_row,
                          float* out_dist, uint32_t* n_int, uint32_t* n_pts)
    {
        const VPNode* tree = static_cast<VPNode*>(t);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < nq; ++i)
        {
            Counters c;
            Found f = tree->search({q[3 * i], q[3 * i + 1], q[3 * i + 2]},
                                   (n_int || n_pts) ? &c : nullptr);
            if (out_xyz)
            {
                out_xyz[3 * i + 0] = f.p.x;
                out_xyz[3 * i + 1] = f.p.y;
                out_xyz[3 * i + 2] = f.p.z;
            }
            if (out_row)
                out_row[i] = f.row;
            if (out_dist)
                out_dist[i] = f.dist;
            if (n_int)
                n_int[i] = c.internal_visited;
            if (n_pts)
                n_pts[i] = c.leaf_points;
        }
    }

    // Pre-order dump sizes then data (two-call protocol).
    void orc_tree_dump_sizes(void* t, uint64_t* n_kind, uint64_t* n_internal, uint64_t* n_leaves,
                             uint64_t* n_leaf_pts)
    {
        uint64_t a = 0, b = 0, mn = UINT64_MAX, mx = 0;
        uint32_t h = 0;
        static_cast<VPNode*>(t)->stats(a, b, 0, h, mn, mx);
        *n_internal = a;
        *n_leaves = b;
        *n_kind = a + b;
        std::vector<uint8_t> kind;
        std::vector<float> nf, lx;
        std::vector<uint32_t> nr, lc, lr;
        static_cast<VPNode*>(t)->dump(kind, nf, nr, lc, lx, lr);
        *n_leaf_pts = lr.size();
    }
    void orc_tree_dump(void* t, uint8_t* kind, float* node_f, uint32_t* node_row,
                       uint32_t* leaf_count, float* leaf_xyz, uint32_t* leaf_row)
    {
        std::vector<uint8_t> k;
        std::vector<float> nf, lx;
        std::vector<uint32_t> nr, lc, lr;
        static_cast<VPNode*>(t)->dump(k, nf, nr, lc, lx, lr);
        std::copy(k.begin(), k.end(), kind);
        std::copy(nf.begin(), nf.end(), node_f);
        std::copy(nr.begin(), nr.end(), node_row);
        std::copy(lc.begin(), lc.end(), leaf_count);
        std::copy(lx.begin(), lx.end(), leaf_xyz);
        std::copy(lr.begin(), lr.end(), leaf_row);
    }

    void orc_svd3(const float H[9], float U[9], float S[3], float V[9])
    {
        jacobi_svd3(H, U, S, V);
    }

    // libcpu/libcpu/icp.cc:27-60 after find_covariance: H -> 4x4.
    // H is row-major with row = P coordinate, col = Y coordinate (:35-44).
    void orc_alignment_from_cov(const float H[9], const float mu_p[3], const float mu_m[3],
                                float T[16])
    {
        float U[9], S[3], V[9];
        jacobi_svd3(H, U, S, V);
        // rotation = U * V^T (:48); Eigen evaluates a 3x3 product coefficient-wise as a
        // left-to-right dot product.
        float rot[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
            {
                // Eigen 3.3 lazy product coefficient = cwiseProduct().sum(), whose unrolled
                // 3-term reduction associates as a0 + (a1 + a2) (recalled; last-bit only).
                rot[i * 3 + j] = U[i * 3 + 0] * V[j * 3 + 0]
                    + (U[i * 3 + 1] * V[j * 3 + 1] + U[i * 3 + 2] * V[j * 3 + 2]);
            }
        float r[9]; // :51-55 r = rotation^T
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                r[i * 3 + j] = rot[j * 3 + i];
        // :57 t = mu_m - dot(r, mu_p) ; point-3d.cc:248-257 (left-to-right adds)
        float t[3];
        for (int i = 0; i < 3; ++i)
        {
            float v = r[i * 3 + 0] * mu_p[0] + r[i * 3 + 1] * mu_p[1] + r[i * 3 + 2] * mu_p[2];
            t[i] = mu_m[i] - v;
        }
        // :11-25
        for (int i = 0; i < 16; ++i)
            T[i] = 0;
        for (int i = 0; i < 3; ++i)
        {
            for (int j = 0; j < 3; ++j)
                T[i * 4 + j] = r[i * 3 + j];
            T[i * 4 + 3] = t[i];
        }
        T[15] = 1;
    }

    // libcpu/libcpu/icp.cc:62-79
    void orc_apply_alignment(float* p, uint64_t n, const float T[16])
    {
        for (size_t i = 0; i < n; ++i)
        {
            float vx = p[3 * i], vy = p[3 * i + 1], vz = p[3 * i + 2];
            float x = vx * T[0] + vy * T[1] + vz * T[2] + T[3];
            float y = vx * T[4] + vy * T[5] + vz * T[6] + T[7];
            float z = vx * T[8] + vy * T[9] + vz * T[10] + T[11];
            p[3 * i] = x;
            p[3 * i + 1] = y;
            p[3 * i + 2] = z;
        }
    }

    // libcpu/libcpu/icp.cc:81-95
    float orc_compute_error(const float* y, const float* p, uint64_t n, const float mu_m[3])
    {
        float error = 0;
        for (size_t i = 0; i < n; ++i)
        {
            float x = y[3 * i] + mu_m[0] - p[3 * i];
            float yy = y[3 * i + 1] + mu_m[1] - p[3 * i + 1];
            float z = y[3 * i + 2] + mu_m[2] - p[3 * i + 2];
            error += x * x + yy * yy + z * z;
        }
        return error;
    }

    void orc_matmul4(const float a[16], const float b[16], float out[16])
    {
        matmul4(a, b, out);
    }

    // libcpu/libcpu/icp.cc:97-128.  errors[] (size >= iterations, may be NULL) receives the
    // error after each iteration.  nn_seconds/total_seconds may be NULL.  Returns 0, or 2 when
    // the tree hit one of the reference's undefined-behaviour cases.
    int orc_icp(const float* m, uint64_t nm, const float* p, uint64_t np, uint64_t iterations,
                float threshold, uint32_t capacity, float T_out[16], float* new_p_out,
                uint64_t* iters_done, float* errors, uint32_t* last_rows)
    {
        float T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; // :101
        std::vector<float> cur(p, p + 3 * np);                             // :103
        float error = std::numeric_limits<float>::infinity();              // :104
        float mu_m[3];
        orc_mean(m, nm, mu_m); // :106
        std::vector<float> mc(3 * nm);
        orc_subtract(m, nm, mu_m, mc.data());
        std::unique_ptr<VPNode> tree(static_cast<VPNode*>(orc_tree_build(capacity, mc.data(), nm)));
        if (!tree || tree->degenerate)
            return 2;
        mc.clear();
        mc.shrink_to_fit();

        std::vector<float> pc(3 * np), y(3 * np);
        uint64_t it = 0;
        for (; it < iterations && error > threshold; ++it) // :110
        {
            float mu_p[3];
            orc_mean(cur.data(), np, mu_p);                 // :114
            orc_subtract(cur.data(), np, mu_p, pc.data()); // :115
            orc_tree_closest(tree.get(), pc.data(), np, y.data(), last_rows, nullptr, nullptr,
                             nullptr); // :117
            float H[9], Tn[16], Tt[16];
            orc_find_covariance(pc.data(), y.data(), np, H);
            orc_alignment_from_cov(H, mu_p, mu_m, Tn); // :119
            matmul4(Tn, T, Tt);                        // :121
            std::memcpy(T, Tt, sizeof T);
            orc_apply_alignment(cur.data(), np, Tn);                   // :122
            error = orc_compute_error(y.data(), cur.data(), np, mu_m); // :123
            if (errors)
                errors[it] = error;
        }
        std::memcpy(T_out, T, sizeof T);
        if (new_p_out)
            std::memcpy(new_p_out, cur.data(), sizeof(float) * 3 * np);
        if (iters_done)
            *iters_done = it;
        return 0;
    }
}

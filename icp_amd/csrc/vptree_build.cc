// Host-side VP-tree builder that emits the flat SoA layout of include/icp_hip.h directly.
//
// Semantics: libcpu::VPTree::VPTree, /root/reference libcpu/libcpu/vp-tree.cc:54-123
//   leaf iff size < capacity (:56); vantage point = last element (:65-66); squared distances
//   centre->point in fp32 (:72-74, point-3d.cc:136-143); radius^2 = element of rank size/2
//   (:48-52,77); stable split d2 < r2 / d2 >= r2 (:90-111); per side the FIRST strict maximum
//   goes to the back and becomes the child's vantage point (:84-88,95-99,104-108,114-115).
//
// Design (not the reference's): no point copies.  The build permutes one uint32 row array in
// place; a subtree owns a contiguous range of it, so subtrees are independent OpenMP tasks.
// Nodes are appended post-order under a lock, then renumbered breadth-first (top levels
// contiguous, L2 friendly) and leaves left-to-right; buckets are written as SoA slots.
//
// Compile WITHOUT fp contraction (-ffp-contract=off, no -ffast-math): squared distances decide
// the tree shape and must be the reference's bits.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "host_common.hh"
#include "icp_hip.h"

namespace icp_host
{
    const char* const kNonFiniteMessage =
        "non-finite coordinate (NaN or infinity): every comparison of the reference's search fails "
        "on it, so each query would walk the whole tree (vp-tree.cc:147-173)";

    bool all_finite(const float* a, uint64_t count)
    {
        // |x| < inf for every x  <=>  no exponent field is all ones
        uint32_t bad = 0;
#pragma omp parallel for reduction(| : bad) if (count > (1u << 20))
        for (uint64_t i = 0; i < count; ++i)
        {
            uint32_t u;
            std::memcpy(&u, a + i, sizeof u);
            bad |= ((u & 0x7f800000u) == 0x7f800000u) ? 1u : 0u;
        }
        return bad == 0;
    }

    namespace
    {
        struct TmpNode
        {
            float cx, cy, cz, r;
            int32_t inside, outside; // temp child codes
            uint32_t row;
        };
        struct TmpLeaf
        {
            uint64_t lo;
            uint32_t count;
        };

        struct Builder
        {
            const float* pts;
            uint32_t capacity;
            std::vector<uint32_t> perm, perm2;
            std::vector<float> dist, dtmp;
            std::vector<TmpNode> nodes;
            std::vector<TmpLeaf> leaves;
            std::mutex lock;
            bool degenerate = false;
            bool oom = false; // an allocation failed inside the task region (no exception may leave it)

            static constexpr uint64_t kTaskMin = 1u << 14;   // spawn tasks above this size
            static constexpr uint64_t kLoopPar = 1u << 20;   // parallelise inner loops above this

            inline float sqdist_c(float cx, float cy, float cz, uint32_t row) const
            {
                const float* p = pts + 3 * (uint64_t)row;
                float dx = cx - p[0];
                float dy = cy - p[1];
                float dz = cz - p[2];
                return dx * dx + dy * dy + dz * dz;
            }

            // Returns the temp child code of the subtree over perm[lo, hi).
            int32_t build(uint64_t lo, uint64_t hi)
            {
                const uint64_t size = hi - lo;
                if (size < capacity)
                {
                    std::lock_guard<std::mutex> g(lock);
                    try
                    {
                        leaves.push_back({lo, (uint32_t)size});
                    }
                    catch (const std::bad_alloc&)
                    {
                        oom = true;
                        return ~0;
                    }
                    return ~(int32_t)(leaves.size() - 1);
                }
                const uint32_t crow = perm[hi - 1];
                const float cx = pts[3 * (uint64_t)crow], cy = pts[3 * (uint64_t)crow + 1],
                            cz = pts[3 * (uint64_t)crow + 2];
                const uint64_t m = size - 1;
                if (m == 0)
                {
                    degenerate = true; // median of an empty array in the reference
                    return ~0;
                }
                float* d = dist.data() + lo;
                float* t = dtmp.data() + lo;
                const uint32_t* pr = perm.data() + lo;

                if (m >= kLoopPar)
                {
#pragma omp taskloop grainsize(1 << 16) shared(d, t, pr)
                    for (uint64_t i = 0; i < m; ++i)
                    {
                        float v = sqdist_c(cx, cy, cz, pr[i]);
                        d[i] = v;
                        t[i] = v;
                    }
                }
                else
                {
                    for (uint64_t i = 0; i < m; ++i)
                    {
                        float v = sqdist_c(cx, cy, cz, pr[i]);
                        d[i] = v;
                        t[i] = v;
                    }
                }
                std::nth_element(t, t + m / 2, t + m);
                const float r2 = t[m / 2];

                // count, then stable scatter into perm2 (inside first, outside after)
                uint64_t ni = 0;
                for (uint64_t i = 0; i < m; ++i)
                    ni += (d[i] < r2);
                const uint64_t no = m - ni;
                if (ni == 0 || no == 0)
                {
                    degenerate = true; // back() of an empty vector in the reference (:114-115)
                    return ~0;
                }
                uint32_t* qi = perm2.data() + lo;
                uint32_t* qo = qi + ni;
                float in_far = 0, out_far = 0;
                uint64_t in_keep = 0, out_keep = 0, ki = 0, ko = 0;
                for (uint64_t i = 0; i < m; ++i)
                {
                    const float v = d[i];
                    if (v < r2)
                    {
                        if (v > in_far)
                        {
                            in_far = v;
                            in_keep = ki;
                        }
                        qi[ki++] = pr[i];
                    }
                    else
                    {
                        if (v > out_far)
                        {
                            out_far = v;
                            out_keep = ko;
                        }
                        qo[ko++] = pr[i];
                    }
                }
                std::swap(qi[ni - 1], qi[in_keep]);
                std::swap(qo[no - 1], qo[out_keep]);
                std::memcpy(perm.data() + lo, perm2.data() + lo, m * sizeof(uint32_t));

                int32_t in_code = 0, out_code = 0;
                if (size >= kTaskMin)
                {
#pragma omp task shared(in_code) firstprivate(lo, ni)
                    in_code = build(lo, lo + ni);
#pragma omp task shared(out_code) firstprivate(lo, ni, m)
                    out_code = build(lo + ni, lo + m);
#pragma omp taskwait
                }
                else
                {
                    in_code = build(lo, lo + ni);
                    out_code = build(lo + ni, lo + m);
                }
                TmpNode nd{cx, cy, cz, std::sqrt(r2), in_code, out_code, crow};
                std::lock_guard<std::mutex> g(lock);
                try
                {
                    nodes.push_back(nd);
                }
                catch (const std::bad_alloc&)
                {
                    oom = true;
                    return ~0;
                }
                return (int32_t)(nodes.size() - 1);
            }
        };
    } // namespace

    FlatTreeOwner::~FlatTreeOwner() = default;

    int build_flat_tree(const float* xyz, uint64_t n, uint32_t capacity, FlatTreeOwner** out,
                        std::string& err)
    {
        *out = nullptr;
        if (!xyz || n == 0)
        {
            err = "icp_hip_tree_build: empty cloud";
            return ICP_HIP_ERR_ARG;
        }
        if (n >= (1ull << 31))
        {
            err = "icp_hip_tree_build: more than 2^31-1 points are not supported";
            return ICP_HIP_ERR_ARG;
        }
        if (capacity < 3)
        {
            err = "icp_hip_tree_build: capacity <= 2 is undefined behaviour in the reference "
                  "(vp-tree.cc:114: empty inside set on 2-point nodes)";
            return ICP_HIP_ERR_DEGENERATE;
        }
        if (!all_finite(xyz, 3 * n))
        {
            err = "icp_hip_tree_build: " + std::string(kNonFiniteMessage);
            return ICP_HIP_ERR_ARG;
        }
        Builder b;
        b.pts = xyz;
        b.capacity = capacity;
        try
        {
            b.perm.resize(n);
            b.perm2.resize(n);
            b.dist.resize(n);
            b.dtmp.resize(n);
        }
        catch (const std::bad_alloc&)
        {
            err = "icp_hip_tree_build: out of host memory";
            return ICP_HIP_ERR_NOMEM;
        }
        for (uint64_t i = 0; i < n; ++i)
            b.perm[i] = (uint32_t)i;

        int32_t root = 0;
#pragma omp parallel if (n > (1u << 14))
#pragma omp single
        root = b.build(0, n);

        if (b.oom)
        {
            err = "icp_hip_tree_build: out of host memory";
            return ICP_HIP_ERR_NOMEM;
        }
        if (b.degenerate)
        {
            err = "icp_hip_tree_build: degenerate cloud: a vantage-point split left one side empty "
                  "(duplicate points / tied distances); the reference's build is undefined "
                  "behaviour here (vp-tree.cc:114-115)";
            return ICP_HIP_ERR_DEGENERATE;
        }
        b.perm2 = std::vector<uint32_t>();
        b.dist = std::vector<float>();
        b.dtmp = std::vector<float>();

        // ---- renumber: internal nodes breadth-first, leaves left to right ----
        const uint32_t n_nodes = (uint32_t)b.nodes.size();
        const uint32_t n_leaves = (uint32_t)b.leaves.size();
        // (held by a unique_ptr until it is handed out: a later resize may throw)
        std::unique_ptr<FlatTreeOwner> owner(new FlatTreeOwner());
        FlatTreeOwner* const T = owner.get();
        T->node_cx.resize(n_nodes);
        T->node_cy.resize(n_nodes);
        T->node_cz.resize(n_nodes);
        T->node_radius.resize(n_nodes);
        T->node_inside.resize(n_nodes);
        T->node_outside.resize(n_nodes);
        T->node_parent.resize(n_nodes);
        T->node_row.resize(n_nodes);
        T->leaf_begin.resize(n_leaves);
        T->leaf_count.resize(n_leaves);

        std::vector<int32_t> new_id(n_nodes, -1);     // tmp internal -> bfs id
        std::vector<int32_t> bfs;                     // bfs order of tmp ids
        std::vector<int32_t> parent_of;               // bfs id -> bfs parent
        std::vector<uint32_t> depth_of;
        bfs.reserve(n_nodes);
        parent_of.reserve(n_nodes);
        depth_of.reserve(n_nodes);
        uint32_t height = 0;
        if (root >= 0)
        {
            bfs.push_back(root);
            parent_of.push_back(-1);
            depth_of.push_back(0);
            new_id[root] = 0;
            for (size_t h = 0; h < bfs.size(); ++h)
            {
                const TmpNode& nd = b.nodes[bfs[h]];
                for (int32_t c : {nd.inside, nd.outside})
                {
                    if (c >= 0)
                    {
                        new_id[c] = (int32_t)bfs.size();
                        bfs.push_back(c);
                        parent_of.push_back((int32_t)h);
                        depth_of.push_back(depth_of[h] + 1);
                    }
                    height = std::max(height, depth_of[h] + 1);
                }
            }
        }
        // leaves left to right: iterative DFS, inside before outside
        std::vector<int32_t> new_leaf(n_leaves, -1);
        {
            std::vector<int32_t> st;
            st.push_back(root);
            uint32_t next = 0;
            while (!st.empty())
            {
                int32_t c = st.back();
                st.pop_back();
                if (c < 0)
                    new_leaf[~c] = (int32_t)next++;
                else
                {
                    st.push_back(b.nodes[c].outside);
                    st.push_back(b.nodes[c].inside);
                }
            }
        }
        auto recode = [&](int32_t c) -> int32_t { return c >= 0 ? new_id[c] : ~new_leaf[~c]; };
        for (uint32_t i = 0; i < n_nodes; ++i)
        {
            const TmpNode& nd = b.nodes[bfs[i]];
            T->node_cx[i] = nd.cx;
            T->node_cy[i] = nd.cy;
            T->node_cz[i] = nd.cz;
            T->node_radius[i] = nd.r;
            T->node_inside[i] = recode(nd.inside);
            T->node_outside[i] = recode(nd.outside);
            T->node_parent[i] = parent_of[i];
            T->node_row[i] = nd.row;
        }
        // bucket slots: each bucket starts at a multiple of 16 slots
        std::vector<uint32_t> old_of_new(n_leaves);
        for (uint32_t l = 0; l < n_leaves; ++l)
            old_of_new[new_leaf[l]] = l;
        uint64_t slots = 0;
        uint32_t max_leaf = 0;
        for (uint32_t l = 0; l < n_leaves; ++l)
        {
            const TmpLeaf& lf = b.leaves[old_of_new[l]];
            T->leaf_begin[l] = (uint32_t)slots;
            T->leaf_count[l] = lf.count;
            max_leaf = std::max(max_leaf, lf.count);
            slots += ((uint64_t)lf.count + 15) / 16 * 16;
        }
        if (slots + n_nodes >= (1ull << 32) - 64)
        {
            delete T;
            err = "icp_hip_tree_build: slot index space exceeds 32 bits";
            return ICP_HIP_ERR_ARG;
        }
        const float inf = std::numeric_limits<float>::infinity();
        T->slot_x.assign(slots, inf);
        T->slot_y.assign(slots, inf);
        T->slot_z.assign(slots, inf);
        T->slot_row.assign(slots, 0xFFFFFFFFu);
#pragma omp parallel for schedule(dynamic, 64) if (n_leaves > 1024u)
        for (uint32_t l = 0; l < n_leaves; ++l)
        {
            const TmpLeaf& lf = b.leaves[old_of_new[l]];
            const uint64_t base = T->leaf_begin[l];
            for (uint32_t k = 0; k < lf.count; ++k)
            {
                const uint32_t row = b.perm[lf.lo + k];
                T->slot_x[base + k] = xyz[3 * (uint64_t)row];
                T->slot_y[base + k] = xyz[3 * (uint64_t)row + 1];
                T->slot_z[base + k] = xyz[3 * (uint64_t)row + 2];
                T->slot_row[base + k] = row;
            }
        }
        icp_flat_tree& v = T->view;
        v.n_points = n;
        v.n_slots = slots;
        v.n_nodes = n_nodes;
        v.n_leaves = n_leaves;
        v.height = height;
        v.capacity = capacity;
        v.max_leaf_count = max_leaf;
        v.root = recode(root);
        T->sync_view();
        *out = owner.release();
        return ICP_HIP_OK;
    }

    void FlatTreeOwner::sync_view()
    {
        view.node_cx = node_cx.data();
        view.node_cy = node_cy.data();
        view.node_cz = node_cz.data();
        view.node_radius = node_radius.data();
        view.node_inside = node_inside.data();
        view.node_outside = node_outside.data();
        view.node_parent = node_parent.data();
        view.node_row = node_row.data();
        view.leaf_begin = leaf_begin.data();
        view.leaf_count = leaf_count.data();
        view.slot_x = slot_x.data();
        view.slot_y = slot_y.data();
        view.slot_z = slot_z.data();
        view.slot_row = slot_row.data();
    }

    // Structural validation; the kernels trust what passes here (a malformed tree could otherwise
    // make a traversal loop forever or read out of bounds).
    int validate_flat_tree(const icp_flat_tree* t, std::string& err)
    {
        auto fail = [&](const char* m) {
            err = std::string("icp_hip_tree_validate: ") + m;
            return (int)ICP_HIP_ERR_ARG;
        };
        if (!t)
            return fail("null tree");
        if (t->n_leaves == 0 || t->n_leaves != t->n_nodes + 1)
            return fail("a binary tree needs n_leaves == n_nodes + 1");
        if (t->n_slots + t->n_nodes >= (1ull << 32) - 64)
            return fail("slot index space exceeds 32 bits");
        if (!t->leaf_begin || !t->leaf_count || !t->slot_x || !t->slot_y || !t->slot_z ||
            !t->slot_row)
            return fail("null leaf/slot array");
        if (t->n_nodes &&
            (!t->node_cx || !t->node_cy || !t->node_cz || !t->node_radius || !t->node_inside ||
             !t->node_outside || !t->node_parent || !t->node_row))
            return fail("null node array");
        if (!all_finite(t->node_cx, t->n_nodes) || !all_finite(t->node_cy, t->n_nodes) ||
            !all_finite(t->node_cz, t->n_nodes) || !all_finite(t->node_radius, t->n_nodes))
            return fail(kNonFiniteMessage);
        auto code_ok = [&](int32_t c) {
            return c >= 0 ? (uint32_t)c < t->n_nodes : (uint32_t)(~c) < t->n_leaves;
        };
        if (!code_ok(t->root))
            return fail("root code out of range");
        if (t->n_nodes == 0 ? t->root != ~0 : t->root < 0)
            return fail("root code inconsistent with n_nodes");
        std::vector<uint8_t> node_seen(t->n_nodes, 0), leaf_seen(t->n_leaves, 0);
        std::vector<uint32_t> depth(t->n_nodes, 0);
        uint32_t height = 0;
        uint64_t pts = 0;
        uint32_t max_leaf = 0;
        // walk from the root; every node/leaf must be reached exactly once
        std::vector<int32_t> st;
        if (t->root >= 0)
        {
            if (t->node_parent[t->root] != -1)
                return fail("root parent must be -1");
            st.push_back(t->root);
            node_seen[t->root] = 1;
        }
        else
            leaf_seen[0] = 1;
        while (!st.empty())
        {
            int32_t i = st.back();
            st.pop_back();
            int32_t ch[2] = {t->node_inside[i], t->node_outside[i]};
            if (ch[0] == ch[1])
                return fail("node with identical children");
            if (t->node_row[i] >= t->n_points)
                return fail("node row out of range");
            for (int32_t c : ch)
            {
                if (!code_ok(c))
                    return fail("child code out of range");
                height = std::max(height, depth[i] + 1);
                if (c >= 0)
                {
                    if (node_seen[c])
                        return fail("internal node reached twice");
                    if (t->node_parent[c] != i)
                        return fail("node_parent does not match the child link");
                    node_seen[c] = 1;
                    depth[c] = depth[i] + 1;
                    st.push_back(c);
                }
                else
                {
                    if (leaf_seen[~c])
                        return fail("leaf reached twice");
                    leaf_seen[~c] = 1;
                }
            }
        }
        // every row of the cloud owns exactly one place (a node centre or a bucket slot)
        std::vector<uint8_t> row_seen(t->n_points, 0);
        for (uint32_t i = 0; i < t->n_nodes; ++i)
        {
            if (!node_seen[i])
                return fail("unreachable internal node");
            if (row_seen[t->node_row[i]]++)
                return fail("a row is owned twice");
        }
        // buckets: left to right in slot order, each starting at a multiple of 16 slots (the
        // kernels load them with 16-byte accesses), disjoint; every slot outside a bucket is
        // padding: +inf coordinates, row 0xFFFFFFFF
        const float inf = std::numeric_limits<float>::infinity();
        auto padding_ok = [&](uint64_t from, uint64_t to) {
            for (uint64_t k = from; k < to; ++k)
                if (t->slot_x[k] != inf || t->slot_y[k] != inf || t->slot_z[k] != inf ||
                    t->slot_row[k] != 0xFFFFFFFFu)
                    return false;
            return true;
        };
        uint64_t next_free = 0;
        for (uint32_t l = 0; l < t->n_leaves; ++l)
        {
            if (!leaf_seen[l])
                return fail("unreachable leaf");
            if (t->leaf_count[l] == 0)
                return fail("empty leaf bucket");
            if (t->leaf_begin[l] % 16u != 0)
                return fail("leaf bucket does not start at a multiple of 16 slots");
            if ((uint64_t)t->leaf_begin[l] + t->leaf_count[l] > t->n_slots)
                return fail("leaf bucket exceeds n_slots");
            if (t->leaf_begin[l] < next_free)
                return fail("leaf buckets overlap or are not in left-to-right order");
            if (!padding_ok(next_free, t->leaf_begin[l]))
                return fail("padding slot without +inf coordinates / row 0xFFFFFFFF");
            next_free = (uint64_t)t->leaf_begin[l] + t->leaf_count[l];
            for (uint32_t k = 0; k < t->leaf_count[l]; ++k)
            {
                const uint32_t row = t->slot_row[t->leaf_begin[l] + k];
                if (row >= t->n_points)
                    return fail("slot row out of range");
                if (row_seen[row]++)
                    return fail("a row is owned twice");
            }
            if (!all_finite(t->slot_x + t->leaf_begin[l], t->leaf_count[l]) ||
                !all_finite(t->slot_y + t->leaf_begin[l], t->leaf_count[l]) ||
                !all_finite(t->slot_z + t->leaf_begin[l], t->leaf_count[l]))
                return fail(kNonFiniteMessage);
            pts += t->leaf_count[l];
            max_leaf = std::max(max_leaf, t->leaf_count[l]);
        }
        if (!padding_ok(next_free, t->n_slots))
            return fail("padding slot without +inf coordinates / row 0xFFFFFFFF");
        if (pts + t->n_nodes != t->n_points)
            return fail("n_points != n_nodes + sum(leaf_count)");
        if (height != t->height)
            return fail("height field does not match the structure");
        if (max_leaf != t->max_leaf_count)
            return fail("max_leaf_count field does not match the structure");
        return ICP_HIP_OK;
    }

    // libcpu::mean, point-3d.cc:177-192
    void mean_host(const float* a, uint64_t n, float out[3])
    {
        const size_t len = n;
        float x = 0, y = 0, z = 0;
        for (size_t i = 0; i < len; ++i)
        {
            x += a[3 * i] / len;
            y += a[3 * i + 1] / len;
            z += a[3 * i + 2] / len;
        }
        out[0] = x;
        out[1] = y;
        out[2] = z;
    }

    // libcpu::subtract(list, point), point-3d.cc:194-210
    void center_host(const float* a, uint64_t n, const float mu[3], float* out)
    {
        // small clouds stay on the calling thread (waking the team costs more than the loop)
#pragma omp parallel for if (n > (1u << 16))
        for (uint64_t i = 0; i < n; ++i)
        {
            out[3 * i] = a[3 * i] - mu[0];
            out[3 * i + 1] = a[3 * i + 1] - mu[1];
            out[3 * i + 2] = a[3 * i + 2] - mu[2];
        }
    }
} // namespace icp_host

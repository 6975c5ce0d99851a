// integration/flatten_from_libcpu.hh -- libcpu::VPTree -> icp_flat_tree.
//
// The reference lets exactly one class look inside its private tree: `friend class
// libgpu::GPUVPTree` (libcpu/libcpu/vp-tree.hh:8-11,17), which its CUDA backend uses to copy the
// tree to the device node by node (libgpu/libgpu/vp-tree.cu:39-112).  This is that class for the
// MI355X backend: it walks a libcpu::VPTree once and fills the flat structure-of-arrays layout of
// include/icp_hip.h, which icp_hip_set_tree validates and uploads in one bulk copy per array.
//
// A maintainer who wants to keep building the tree with libcpu::VPTree (as libgpu::icp does,
// libgpu/libgpu/icp.cu:141-142) uses this; icp_hip_build_tree / icp_hip_tree_build make the same
// tree without libcpu (tests/test_host_tree.py: bit for bit).
//
// Layout rules reproduced here (include/icp_hip.h):
//   internal nodes numbered breadth-first, leaves left to right (inside before outside), every
//   bucket starts at a multiple of 16 slots and keeps the reference's point order, padding slots
//   hold +inf and row 0xFFFFFFFF.
// Rows: libcpu::VPTree keeps coordinates, not indices.  When the cloud the tree was built from is
// passed, each point gets its index in that cloud back (equal coordinates: lowest free index
// first); otherwise rows number the tree's places 0 .. n-1 in the order they are written.
#pragma once

#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <stdexcept>
#include <tuple>
#include <vector>

#include "icp_hip.h"
#include "libcpu/vp-tree.hh"

namespace libgpu
{
    class GPUVPTree
    {
    public:
        // owns the arrays `view` points into
        struct Flat
        {
            icp_flat_tree view{};
            std::vector<float> node_cx, node_cy, node_cz, node_radius;
            std::vector<int32_t> node_inside, node_outside, node_parent;
            std::vector<uint32_t> node_row, leaf_begin, leaf_count, slot_row;
            std::vector<float> slot_x, slot_y, slot_z;
            Flat() = default;
            Flat(const Flat&) = delete;
            Flat& operator=(const Flat&) = delete;
        };

        static void from_cpu(const libcpu::VPTree& root, uint32_t capacity,
                             const libcpu::point_list* cloud, Flat& out)
        {
            // pass 1, breadth-first: number the internal nodes, record parents and depth
            struct Item
            {
                const libcpu::VPTree* t;
                int32_t parent;
                uint32_t depth;
            };
            std::vector<Item> nodes;
            uint32_t height = 0;
            if (root.inside)
                nodes.push_back({&root, -1, 0});
            for (size_t h = 0; h < nodes.size(); ++h)
            {
                const Item it = nodes[h];
                for (const libcpu::VPTree* c : {it.t->inside.get(), it.t->outside.get()})
                {
                    if (!c)
                        throw std::runtime_error("GPUVPTree::from_cpu: internal node with one child");
                    height = height > it.depth + 1 ? height : it.depth + 1;
                    if (c->inside)
                        nodes.push_back({c, (int32_t)h, it.depth + 1});
                }
            }
            const size_t nn = nodes.size();
            std::map<const libcpu::VPTree*, int32_t> id;
            for (size_t i = 0; i < nn; ++i)
                id[nodes[i].t] = (int32_t)i;

            RowFinder rows(cloud);
            out.node_cx.resize(nn);
            out.node_cy.resize(nn);
            out.node_cz.resize(nn);
            out.node_radius.resize(nn);
            out.node_inside.resize(nn);
            out.node_outside.resize(nn);
            out.node_parent.resize(nn);
            out.node_row.resize(nn);

            // pass 2, depth-first, inside before outside: leaves left to right, buckets
            const float inf = std::numeric_limits<float>::infinity();
            uint64_t n_points = 0;
            uint32_t max_leaf = 0;
            std::vector<std::tuple<const libcpu::VPTree*, int32_t, bool>> st; // (tree, parent id, is inside child)
            st.emplace_back(&root, -1, false);
            int32_t root_code = 0;
            while (!st.empty())
            {
                auto [t, parent, is_inside] = st.back();
                st.pop_back();
                int32_t code;
                if (t->inside)
                {
                    const int32_t i = id.at(t);
                    code = i;
                    out.node_cx[i] = t->center.x;
                    out.node_cy[i] = t->center.y;
                    out.node_cz[i] = t->center.z;
                    out.node_radius[i] = t->radius;
                    out.node_parent[i] = nodes[i].parent;
                    out.node_row[i] = rows.take(t->center);
                    n_points += 1;
                    st.emplace_back(t->outside.get(), i, false);
                    st.emplace_back(t->inside.get(), i, true);
                }
                else
                {
                    if (t->points.empty())
                        throw std::runtime_error("GPUVPTree::from_cpu: empty leaf");
                    const uint32_t leaf = (uint32_t)out.leaf_begin.size();
                    code = ~(int32_t)leaf;
                    out.leaf_begin.push_back((uint32_t)out.slot_x.size());
                    out.leaf_count.push_back((uint32_t)t->points.size());
                    for (const libcpu::Point3D& p : t->points)
                    {
                        out.slot_x.push_back(p.x);
                        out.slot_y.push_back(p.y);
                        out.slot_z.push_back(p.z);
                        out.slot_row.push_back(rows.take(p));
                    }
                    while (out.slot_x.size() % 16 != 0)
                    {
                        out.slot_x.push_back(inf);
                        out.slot_y.push_back(inf);
                        out.slot_z.push_back(inf);
                        out.slot_row.push_back(0xFFFFFFFFu);
                    }
                    n_points += t->points.size();
                    max_leaf = max_leaf > t->points.size() ? max_leaf : (uint32_t)t->points.size();
                }
                if (parent < 0)
                    root_code = code;
                else if (is_inside)
                    out.node_inside[parent] = code;
                else
                    out.node_outside[parent] = code;
            }

            icp_flat_tree& v = out.view;
            v.n_points = n_points;
            v.n_slots = out.slot_x.size();
            v.n_nodes = (uint32_t)nn;
            v.n_leaves = (uint32_t)out.leaf_begin.size();
            v.height = height;
            v.capacity = capacity;
            v.max_leaf_count = max_leaf;
            v.root = root_code;
            v.node_cx = out.node_cx.data();
            v.node_cy = out.node_cy.data();
            v.node_cz = out.node_cz.data();
            v.node_radius = out.node_radius.data();
            v.node_inside = out.node_inside.data();
            v.node_outside = out.node_outside.data();
            v.node_parent = out.node_parent.data();
            v.node_row = out.node_row.data();
            v.leaf_begin = out.leaf_begin.data();
            v.leaf_count = out.leaf_count.data();
            v.slot_x = out.slot_x.data();
            v.slot_y = out.slot_y.data();
            v.slot_z = out.slot_z.data();
            v.slot_row = out.slot_row.data();
        }

    private:
        // coordinates -> index in the cloud the tree was built from
        struct RowFinder
        {
            struct Key
            {
                uint32_t x, y, z;
                bool operator<(const Key& o) const
                {
                    return std::tie(x, y, z) < std::tie(o.x, o.y, o.z);
                }
            };
            std::map<Key, std::vector<uint32_t>> where; // indices in DEscending order: pop_back = lowest
            uint32_t next = 0;
            bool have = false;
            static Key key(const libcpu::Point3D& p)
            {
                Key k;
                std::memcpy(&k.x, &p.x, 4);
                std::memcpy(&k.y, &p.y, 4);
                std::memcpy(&k.z, &p.z, 4);
                return k;
            }
            explicit RowFinder(const libcpu::point_list* cloud)
            {
                if (!cloud)
                    return;
                have = true;
                for (size_t i = cloud->size(); i-- > 0;)
                    where[key((*cloud)[i])].push_back((uint32_t)i);
            }
            uint32_t take(const libcpu::Point3D& p)
            {
                if (!have)
                    return next++;
                auto it = where.find(key(p));
                if (it == where.end() || it->second.empty())
                    throw std::runtime_error("GPUVPTree::from_cpu: a tree point is not in the cloud");
                const uint32_t r = it->second.back();
                it->second.pop_back();
                return r;
            }
        };
    };
} // namespace libgpu

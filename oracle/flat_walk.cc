// oracle/flat_walk.cc -- TEST INFRASTRUCTURE ONLY.
//
// A host emulation of the traversal state machine that kernels.hip runs per lane, over the flat
// SoA tree of include/icp_hip.h (parent links instead of a node stack, the near side's answer
// parked per level, the "first slot whose squared distance is below the sqrt ceiling" form of the
// leaf scan).  It lets the CPU test-suite check the flat layout, the builder and the state
// machine against the recursive oracle (oracle/icp_oracle.cc) without a GPU.  It is never linked
// into, loaded by or called from the product.
//
// Reference semantics: libcpu/libcpu/vp-tree.cc:125-175.

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/icp_hip.h"

namespace
{
    inline float sqd(float ax, float ay, float az, float bx, float by, float bz)
    {
        float dx = ax - bx, dy = ay - by, dz = az - bz;
        return dx * dx + dy * dy + dz * dz;
    }

    inline float sq_ceiling_for_root(float r)
    {
        uint32_t rb;
        std::memcpy(&rb, &r, 4);
        rb += 1;
        float r_up;
        std::memcpy(&r_up, &rb, 4);
        const double m = 0.5 * ((double)r + (double)r_up);
        const double m2 = m * m;
        float f = (float)m2;
        if ((double)f > m2)
            f = std::nextafterf(f, 0.0f);
        return f;
    }
} // namespace

extern "C"
{
    // For each query: id (slot, or n_slots + node), row, distance.
    void flat_walk_closest(const icp_flat_tree* t, const float* q, uint64_t nq, uint32_t* out_id,
                           uint32_t* out_row, float* out_dist, float* out_xyz)
    {
        const uint32_t n_slots = (uint32_t)t->n_slots;
#pragma omp parallel for schedule(static)
        for (uint64_t qi = 0; qi < nq; ++qi)
        {
            const float qx = q[3 * qi], qy = q[3 * qi + 1], qz = q[3 * qi + 2];
            std::vector<float> near_d(t->height + 1);
            std::vector<uint32_t> near_i(t->height + 1);
            enum
            {
                DESC,
                RET,
                WAIT,
                DONE
            } mode = DESC;
            int32_t node = t->root, prev = 0;
            int level = 0;
            float rd = 0;
            uint32_t rid = 0, lbeg = 0, lcnt = 0;
            if (t->root < 0)
            {
                lbeg = t->leaf_begin[~t->root];
                lcnt = t->leaf_count[~t->root];
                prev = t->root;
                node = -1;
                mode = WAIT;
            }
            while (mode != DONE)
            {
                if (mode == WAIT)
                {
                    float smin = INFINITY;
                    for (uint32_t j = 0; j < lcnt; ++j)
                        smin = std::fmin(smin, sqd(qx, qy, qz, t->slot_x[lbeg + j],
                                                   t->slot_y[lbeg + j], t->slot_z[lbeg + j]));
                    const float r = std::sqrt(smin);
                    const float hi = sq_ceiling_for_root(r);
                    uint32_t win = lbeg;
                    for (uint32_t j = 0; j < lcnt; ++j)
                        if (sqd(qx, qy, qz, t->slot_x[lbeg + j], t->slot_y[lbeg + j],
                                t->slot_z[lbeg + j]) <= hi)
                        {
                            win = lbeg + j;
                            break;
                        }
                    rd = r;
                    rid = win;
                    mode = node < 0 ? DONE : RET;
                    continue;
                }
                const float cx = t->node_cx[node], cy = t->node_cy[node], cz = t->node_cz[node];
                const float rad = t->node_radius[node];
                const float d = std::sqrt(sqd(qx, qy, qz, cx, cy, cz));
                const bool in_near = d < rad;
                const int32_t nearc = in_near ? t->node_inside[node] : t->node_outside[node];
                const int32_t farc = in_near ? t->node_outside[node] : t->node_inside[node];
                int32_t go = nearc;
                bool down = true;
                if (mode == RET)
                {
                    bool up;
                    if (prev == nearc)
                    {
                        if (rd < std::fabs(rad - d))
                            up = true;
                        else
                        {
                            near_d[level] = rd;
                            near_i[level] = rid;
                            go = farc;
                            up = false;
                        }
                    }
                    else
                    {
                        if (!(rd < near_d[level]))
                        {
                            rd = near_d[level];
                            rid = near_i[level];
                        }
                        up = true;
                    }
                    down = !up;
                    if (up)
                    {
                        if (d < rd)
                        {
                            rd = d;
                            rid = n_slots + (uint32_t)node;
                        }
                        prev = node;
                        node = t->node_parent[node];
                        level -= 1;
                        mode = node < 0 ? DONE : RET;
                    }
                }
                if (down)
                {
                    if (go >= 0)
                    {
                        node = go;
                        level += 1;
                        mode = DESC;
                    }
                    else
                    {
                        lbeg = t->leaf_begin[~go];
                        lcnt = t->leaf_count[~go];
                        prev = go;
                        mode = WAIT;
                    }
                }
            }
            uint32_t row;
            float ox, oy, oz;
            if (rid < n_slots)
            {
                row = t->slot_row[rid];
                ox = t->slot_x[rid];
                oy = t->slot_y[rid];
                oz = t->slot_z[rid];
            }
            else
            {
                const uint32_t nd = rid - n_slots;
                row = t->node_row[nd];
                ox = t->node_cx[nd];
                oy = t->node_cy[nd];
                oz = t->node_cz[nd];
            }
            if (out_id)
                out_id[qi] = rid;
            if (out_row)
                out_row[qi] = row;
            if (out_dist)
                out_dist[qi] = rd;
            if (out_xyz)
            {
                out_xyz[3 * qi] = ox;
                out_xyz[3 * qi + 1] = oy;
                out_xyz[3 * qi + 2] = oz;
            }
        }
    }
}

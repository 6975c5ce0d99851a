// Host emulation of the NN kernel's way back up the tree (development tool, not product code):
//   plain   every level is stepped (the reference's recursion, vp-tree.cc:125-175);
//   chain   the kernel before: if the bucket's answer is below the descent chain's bound the lane
//           jumps back to where the descent began, else it climbs level by level;
//   heap    a scheme that was built, measured and dropped ("tried and
//           dropped"): internal nodes are in heap order (children of i are 2i+1 / 2i+2), the
//           descent marks in a bit mask the levels whose tests could fail for an answer <= G (a
//           per-batch guess), and the way up visits only those by ancestor arithmetic.
//   single  the kernel now: chain + "exactly one level of the descent can matter" (the answer is
//           below the second smallest bound of the chain): the lane goes to that level, takes its
//           real step, and jumps from there back to where the descent began.
// Checks that all of them give the same answer for every query and prints, per 64-query batch in
// leaf order, the number of lock-step "general steps" (max over the batch's lanes) of each scheme.
// build: g++ -O2 -fopenmp -Iinclude scripts/emulate_level_skip.cc -Licp_amd/lib -licp_hip \
//            -Wl,-rpath,$PWD/icp_amd/lib -o /tmp/emu_skip
// usage: /tmp/emu_skip [n] [capacity] [angle_deg] [g_factor] [far_weight] [jitter]
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>

#include "icp_hip.h"

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
    inline int clz32(uint32_t v) { return v ? __builtin_clz(v) : 32; }

    struct Result
    {
        uint32_t rid;
        float rd;
        uint32_t steps;   // general steps (everything after the first root-to-bucket descent)
        uint32_t n_int;   // descents counted like the kernel
        float gmax;       // largest bucket answer seen (heap scheme's feedback)
    };

    enum Scheme
    {
        PLAIN,
        CHAIN,
        HEAP,
        SINGLE // chain + "exactly one level of the descent can matter": go there, then back to the start
    };
    constexpr int32_t kNearMark = INT32_MIN; // `prev` of a lane that jumped to the one level

    void leaf_scan(const icp_flat_tree* t, float qx, float qy, float qz, uint32_t lbeg, uint32_t lcnt,
                   float& rd, uint32_t& rid)
    {
        float smin = INFINITY;
        for (uint32_t j = 0; j < lcnt; ++j)
            smin = std::fmin(smin, sqd(qx, qy, qz, t->slot_x[lbeg + j], t->slot_y[lbeg + j],
                                       t->slot_z[lbeg + j]));
        const float r = std::sqrt(smin);
        const float hi = sq_ceiling_for_root(r);
        uint32_t win = lbeg;
        for (uint32_t j = 0; j < lcnt; ++j)
            if (sqd(qx, qy, qz, t->slot_x[lbeg + j], t->slot_y[lbeg + j], t->slot_z[lbeg + j]) <= hi)
            {
                win = lbeg + j;
                break;
            }
        rd = r;
        rid = win;
    }

    Result walk(const icp_flat_tree* t, float qx, float qy, float qz, Scheme scheme, float G,
                float far_w)
    {
        const uint32_t n_slots = (uint32_t)t->n_slots;
        const float G2 = G * G * 1.000004f;
        std::vector<float> st_d;
        std::vector<uint32_t> st_i;
        enum
        {
            DESC,
            RET,
            WAIT,
            DONE
        } mode = DESC;
        int32_t node = t->root, prev = 0;
        float rd = 0;
        uint32_t rid = 0, lbeg = 0, lcnt = 0;
        // chain scheme
        int32_t ret_node = -1, top = t->root;
        float chain_lb = INFINITY, chain_d2 = INFINITY;
        // single scheme: smallest and second smallest of min(|radius - d|, d) - slack, and the node
        // of the smallest
        float chain_m = INFINITY, chain_m2 = INFINITY;
        int32_t arg = -1;
        std::vector<uint8_t> st_flag; // pushed with the near answer: "done after this far search"
        // heap scheme
        uint32_t mask = 0, origins = 0;
        Result R{0, 0.f, 0, 0, 0.f};
        bool first_descent = true;
        if (t->root < 0)
        {
            lbeg = t->leaf_begin[~t->root];
            lcnt = t->leaf_count[~t->root];
            prev = t->root;
            node = -1;
            mode = WAIT;
        }
        auto heap_jump = [&]() {
            const uint32_t M = mask | origins;
            if (M == 0)
            {
                prev = node;
                node = -1;
                mode = DONE;
                return;
            }
            const uint32_t np1 = (uint32_t)node + 1u;
            const int sh = clz32(M) - clz32(np1);
            if (sh < 0)
            {
                std::fprintf(stderr, "invariant broken: a marked level below the lane\n");
                std::abort();
            }
            if (sh > 0)
            {
                prev = (int32_t)(np1 >> (sh - 1)) - 1;
                node = (int32_t)(np1 >> sh) - 1;
            }
            mode = RET;
        };
        while (mode != DONE)
        {
            if (mode == WAIT)
            {
                first_descent = false;
                leaf_scan(t, qx, qy, qz, lbeg, lcnt, rd, rid);
                if (node < 0)
                {
                    mode = DONE;
                    continue;
                }
                const float m = rd * rd; // stand-in for the squared distance the kernel holds
                if (scheme == CHAIN)
                {
                    if (rd < chain_lb && chain_d2 > m * 1.000002f)
                    {
                        node = ret_node;
                        prev = top;
                    }
                    mode = node < 0 ? DONE : RET;
                }
                else if (scheme == SINGLE)
                {
                    if (rd < chain_m)
                    {
                        node = ret_node;
                        prev = top;
                    }
                    else if (rd < chain_m2)
                    {
                        node = arg;
                        prev = kNearMark;
                    }
                    mode = node < 0 ? DONE : RET;
                }
                else if (scheme == HEAP)
                {
                    R.gmax = std::fmax(R.gmax, origins ? rd * far_w : rd);
                    if (!(rd <= G))
                    {
                        const uint32_t below = 0xFFFFFFFFu >> clz32((uint32_t)node + 1u);
                        const uint32_t upto_origin = origins ? (0xFFFFFFFFu >> clz32(origins)) : 0u;
                        mask |= below & ~upto_origin;
                    }
                    heap_jump();
                }
                else
                    mode = RET;
                continue;
            }
            if (!first_descent)
                R.steps += 1;
            const float cx = t->node_cx[node], cy = t->node_cy[node], cz = t->node_cz[node];
            const float rad = t->node_radius[node];
            const float d2 = sqd(qx, qy, qz, cx, cy, cz);
            const float d = std::sqrt(d2);
            const bool in_near = d < rad;
            const int32_t nearc = in_near ? t->node_inside[node] : t->node_outside[node];
            const int32_t farc = in_near ? t->node_outside[node] : t->node_inside[node];
            if (mode == DESC)
            {
                R.n_int += 1;
                const float lb = std::fabs(rad - d) - ((d + rad) * 0x1p-20f + 1e-18f);
                if (scheme == CHAIN)
                {
                    chain_lb = std::fmin(chain_lb, lb);
                    chain_d2 = std::fmin(chain_d2, d2);
                }
                else if (scheme == HEAP)
                {
                    const uint32_t bit = 0x80000000u >> clz32((uint32_t)node + 1u);
                    const bool skip = (G < lb) && (d2 > G2);
                    mask |= skip ? 0u : bit;
                }
                else if (scheme == SINGLE)
                {
                    const float slack = (d + rad) * 0x1p-20f + 1e-18f;
                    const float m = std::fmin(std::fabs(rad - d), d) - slack;
                    if (m < chain_m)
                    {
                        chain_m2 = chain_m;
                        chain_m = m;
                        arg = node;
                    }
                    else
                        chain_m2 = std::fmin(chain_m2, m);
                }
            }
            int32_t go = nearc;
            bool down = true;
            if (mode == RET)
            {
                bool up;
                const bool marked = prev == kNearMark;
                bool done_after = false;
                if (prev == nearc || marked)
                {
                    if (rd < std::fabs(rad - d))
                        up = true;
                    else
                    {
                        st_d.push_back(rd);
                        st_i.push_back(rid);
                        // the one level that matters, on the chain from the root: nothing above it
                        // can change the answer, so the lane is done once this far side is searched
                        st_flag.push_back(marked && ret_node < 0);
                        go = farc;
                        up = false;
                        if (scheme == SINGLE)
                        {
                            ret_node = node;
                            top = farc;
                            chain_m = chain_m2 = INFINITY;
                        }
                        if (scheme == CHAIN)
                        {
                            ret_node = node;
                            top = farc;
                            chain_lb = chain_d2 = INFINITY;
                        }
                        else if (scheme == HEAP)
                        {
                            const uint32_t bit = 0x80000000u >> clz32((uint32_t)node + 1u);
                            origins |= bit;
                            mask &= ~bit;
                        }
                    }
                }
                else
                {
                    const float nd = st_d.back();
                    const uint32_t ni = st_i.back();
                    done_after = st_flag.back() != 0;
                    st_d.pop_back();
                    st_i.pop_back();
                    st_flag.pop_back();
                    if (!(rd < nd))
                    {
                        rd = nd;
                        rid = ni;
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
                    if (scheme == HEAP)
                    {
                        const uint32_t bit = 0x80000000u >> clz32((uint32_t)node + 1u);
                        mask &= ~bit;
                        origins &= ~bit;
                        heap_jump();
                    }
                    else if (scheme == SINGLE && (marked || done_after))
                    {
                        // back to where this chain began (marked), or done (after the far side of
                        // the one level of the root chain)
                        if (done_after)
                        {
                            prev = node;
                            node = -1;
                        }
                        else
                        {
                            node = ret_node;
                            prev = top;
                        }
                        mode = node < 0 ? DONE : RET;
                    }
                    else
                    {
                        prev = node;
                        node = t->node_parent[node];
                        mode = node < 0 ? DONE : RET;
                    }
                }
            }
            if (down)
            {
                if (go >= 0)
                {
                    node = go;
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
        R.rid = rid;
        R.rd = rd;
        return R;
    }

    // ---- the ring walk (nn_tile.hip, ring_walk) with the kernel's arithmetic ----
    // ring words of one bucket: (rho with the low 8 mantissa bits cleared) | slot, ascending
    std::vector<uint32_t> ring_words(const icp_flat_tree* t, uint32_t leaf, uint32_t parent)
    {
        const uint32_t beg = t->leaf_begin[leaf], cnt = t->leaf_count[leaf];
        std::vector<uint32_t> w(cnt);
        for (uint32_t j = 0; j < cnt; ++j)
        {
            const float rho = std::sqrt(sqd(t->slot_x[beg + j], t->slot_y[beg + j], t->slot_z[beg + j],
                                            t->node_cx[parent], t->node_cy[parent], t->node_cz[parent]));
            uint32_t bits;
            std::memcpy(&bits, &rho, 4);
            w[j] = (bits & 0xFFFFFF00u) | j;
        }
        std::sort(w.begin(), w.end());
        return w;
    }
    inline float word_rho(uint32_t w)
    {
        w &= 0xFFFFFF00u;
        float f;
        std::memcpy(&f, &w, 4);
        return f;
    }
    // returns true if the walk finished; m1 = smallest d2 among the points examined, s1 its slot,
    // ok = the runner-up is more than 4 ulps above
    bool ring_walk(const icp_flat_tree* t, const std::vector<uint32_t>& rw, uint32_t beg, float qx,
                   float qy, float qz, float dq, int walk_max, float& m1, uint32_t& s1, bool& ok,
                   uint32_t& examined)
    {
        const uint32_t cnt = (uint32_t)rw.size();
        uint32_t pos = 0;
        while (pos < cnt && word_rho(rw[pos]) < dq)
            ++pos;
        int up = (int)pos, dn = (int)pos - 1;
        float m2 = INFINITY;
        m1 = INFINITY;
        s1 = 0;
        examined = 0;
        bool done = false;
        for (int it = 0; it <= walk_max && !done; ++it)
        {
            const float ru = (uint32_t)up < cnt ? word_rho(rw[up]) : INFINITY;
            const float rl = dn >= 0 ? word_rho(rw[dn]) : 0.f;
            const float gu = (uint32_t)up < cnt ? ru - dq : INFINITY;
            const float gd = dn >= 0 ? dq - (rl + rl * 0x1p-15f) : INFINITY;
            const bool take_up = gu <= gd;
            const float gap = take_up ? gu : gd, ref = take_up ? ru : rl;
            const float g = gap - ((ref + dq) * 0x1p-20f + 1e-18f);
            if (g > 0.f && g * g > m1 * 1.00001f)
                done = true;
            else if (it < walk_max)
            {
                const uint32_t slot = (take_up ? rw[up] : rw[dn]) & 0xFFu;
                up += take_up ? 1 : 0;
                dn -= take_up ? 0 : 1;
                const float d2 = sqd(qx, qy, qz, t->slot_x[beg + slot], t->slot_y[beg + slot],
                                     t->slot_z[beg + slot]);
                m2 = d2 < m1 ? m1 : std::fmin(m2, d2);
                s1 = d2 < m1 ? slot : s1;
                m1 = std::fmin(m1, d2);
                examined += 1;
            }
        }
        uint32_t b1, b2;
        std::memcpy(&b1, &m1, 4);
        std::memcpy(&b2, &m2, 4);
        ok = (int32_t)(b2 - b1) > 4;
        return done;
    }

    bool heap_ordered(const icp_flat_tree* t)
    {
        for (uint32_t i = 0; i < t->n_nodes; ++i)
        {
            if (i > 0 && t->node_parent[i] != (int32_t)((i - 1) >> 1))
                return false;
            if (t->node_inside[i] >= 0 && (uint32_t)t->node_inside[i] != 2 * i + 1)
                return false;
            if (t->node_outside[i] >= 0 && (uint32_t)t->node_outside[i] != 2 * i + 2)
                return false;
        }
        return true;
    }
} // namespace

int main(int argc, char** argv)
{
    const uint64_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1000000;
    const uint32_t cap = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 256;
    const double ang = (argc > 3 ? std::atof(argv[3]) : 1.0) * M_PI / 180.0;
    const float g_factor = argc > 4 ? (float)std::atof(argv[4]) : 1.25f;
    const float far_w = argc > 5 ? (float)std::atof(argv[5]) : 0.f;
    const float jitter = argc > 6 ? (float)std::atof(argv[6]) : 0.f;
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    std::vector<float> m(3 * n), q(3 * n);
    for (auto& v : m)
        v = U(rng);
    const float ca = (float)std::cos(ang), sa = (float)std::sin(ang);
    for (uint64_t i = 0; i < n; ++i)
    {
        const float x = m[3 * i], y = m[3 * i + 1], z = m[3 * i + 2];
        q[3 * i] = ca * x - sa * y + jitter * U(rng);
        q[3 * i + 1] = sa * x + ca * y + jitter * U(rng);
        q[3 * i + 2] = z + jitter * U(rng);
    }
    icp_flat_tree* t = nullptr;
    if (icp_hip_tree_build(m.data(), n, cap, &t) != 0)
    {
        std::fprintf(stderr, "tree build failed: %s\n", icp_hip_last_error(nullptr));
        return 1;
    }
    std::printf("n %llu capacity %u height %u nodes %u heap-ordered %d spacing %.5f max displacement %.5f\n",
                (unsigned long long)n, cap, t->height, t->n_nodes, (int)heap_ordered(t),
                2.0 / std::cbrt((double)n), 1.414 * ang);
    if (!heap_ordered(t))
        return 0;
    // leaf order: near-only descent
    std::vector<uint32_t> leaf(n);
#pragma omp parallel for
    for (uint64_t i = 0; i < n; ++i)
    {
        int32_t c = t->root;
        while (c >= 0)
        {
            const float d = std::sqrt(sqd(q[3 * i], q[3 * i + 1], q[3 * i + 2], t->node_cx[c],
                                          t->node_cy[c], t->node_cz[c]));
            c = d < t->node_radius[c] ? t->node_inside[c] : t->node_outside[c];
        }
        leaf[i] = (uint32_t)~c;
    }
    // ---- the ring walk on every query's first bucket against the full scan of that bucket ----
    if (t->max_leaf_count <= 256 && t->n_nodes > 0)
    {
        std::vector<int32_t> parent(t->n_leaves, -1);
        for (uint32_t i = 0; i < t->n_nodes; ++i)
        {
            if (t->node_inside[i] < 0)
                parent[~t->node_inside[i]] = (int32_t)i;
            if (t->node_outside[i] < 0)
                parent[~t->node_outside[i]] = (int32_t)i;
        }
        std::vector<std::vector<uint32_t>> rings(t->n_leaves);
#pragma omp parallel for
        for (uint32_t l = 0; l < t->n_leaves; ++l)
            rings[l] = ring_words(t, l, (uint32_t)parent[l]);
        uint64_t finished = 0, answered = 0, wrong = 0, seen = 0;
#pragma omp parallel for reduction(+ : finished, answered, wrong, seen)
        for (uint64_t i = 0; i < n; ++i)
        {
            const uint32_t l = leaf[i], beg = t->leaf_begin[l], cnt = t->leaf_count[l];
            const int32_t pn = parent[l];
            const float qx = q[3 * i], qy = q[3 * i + 1], qz = q[3 * i + 2];
            const float dq = std::sqrt(sqd(qx, qy, qz, t->node_cx[pn], t->node_cy[pn], t->node_cz[pn]));
            float m1, rd;
            uint32_t s1, rid, ex;
            bool ok;
            if (!ring_walk(t, rings[l], beg, qx, qy, qz, dq, 6, m1, s1, ok, ex))
                continue;
            finished += 1;
            seen += ex;
            if (!ok)
                continue; // the kernel sends these to the exact scan
            answered += 1;
            leaf_scan(t, qx, qy, qz, beg, cnt, rd, rid);
            const float r1 = std::sqrt(m1);
            if (rid != beg + s1 || std::memcmp(&rd, &r1, 4))
                wrong += 1;
        }
        std::printf("ring walk (6 points at most): finished %.4f of the scans after %.2f points, "
                    "answered %.4f, different from the full scan: %llu\n",
                    (double)finished / n, finished ? (double)seen / finished : 0.0, (double)answered / n,
                    (unsigned long long)wrong);
        if (wrong)
            return 1;
    }
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return leaf[a] < leaf[b]; });

    const uint64_t n_batches = (n + 63) / 64;
    // a "wave" runs a contiguous range of batches and carries G from batch to batch
    const uint64_t per_wave = 64;
    uint64_t steps_plain = 0, steps_chain = 0, steps_heap = 0, lanes_invalid = 0, mismatches = 0;
    uint64_t lane_steps_plain = 0, lane_steps_chain = 0, lane_steps_heap = 0;
    uint64_t steps_single = 0, lane_steps_single = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : steps_plain, steps_chain, steps_heap, lanes_invalid, mismatches, lane_steps_plain, lane_steps_chain, lane_steps_heap, steps_single, lane_steps_single)
    for (uint64_t w0 = 0; w0 < n_batches; w0 += per_wave)
    {
        float G = INFINITY;
        for (uint64_t b = w0; b < std::min(n_batches, w0 + per_wave); ++b)
        {
            uint32_t mp = 0, mc = 0, mh = 0, ms_ = 0;
            float gmax = 0.f;
            for (uint64_t k = b * 64; k < std::min<uint64_t>(n, b * 64 + 64); ++k)
            {
                const uint32_t i = order[k];
                const float qx = q[3 * i], qy = q[3 * i + 1], qz = q[3 * i + 2];
                const Result p = walk(t, qx, qy, qz, PLAIN, 0.f, 0.f);
                const Result c = walk(t, qx, qy, qz, CHAIN, 0.f, 0.f);
                const Result h = walk(t, qx, qy, qz, HEAP, G, far_w);
                const Result sg = walk(t, qx, qy, qz, SINGLE, 0.f, 0.f);
                if (p.rid != sg.rid || std::memcmp(&p.rd, &sg.rd, 4) || p.n_int != sg.n_int)
                    mismatches += 1;
                ms_ = std::max(ms_, sg.steps);
                lane_steps_single += sg.steps;
                if (p.rid != c.rid || p.rid != h.rid || std::memcmp(&p.rd, &c.rd, 4) ||
                    std::memcmp(&p.rd, &h.rd, 4) || p.n_int != c.n_int || p.n_int != h.n_int)
                    mismatches += 1;
                mp = std::max(mp, p.steps);
                mc = std::max(mc, c.steps);
                mh = std::max(mh, h.steps);
                lane_steps_plain += p.steps;
                lane_steps_chain += c.steps;
                lane_steps_heap += h.steps;
                gmax = std::fmax(gmax, h.gmax);
                lanes_invalid += h.rd > G;
            }
            steps_plain += mp;
            steps_chain += mc;
            steps_heap += mh;
            steps_single += ms_;
            G = gmax * g_factor;
        }
    }
    std::printf("mismatches %llu\n", (unsigned long long)mismatches);
    std::printf("general steps per batch (max over lanes): plain %.2f chain %.2f heap %.2f single %.2f\n",
                (double)steps_plain / n_batches, (double)steps_chain / n_batches,
                (double)steps_heap / n_batches, (double)steps_single / n_batches);
    std::printf("general steps per lane: single %.2f\n", (double)lane_steps_single / n);
    std::printf("general steps per lane: plain %.2f chain %.2f heap %.2f   final answer above G: %.4f of lanes\n",
                (double)lane_steps_plain / n, (double)lane_steps_chain / n, (double)lane_steps_heap / n,
                (double)lanes_invalid / n);
    icp_hip_tree_free(t);
    return mismatches != 0;
}

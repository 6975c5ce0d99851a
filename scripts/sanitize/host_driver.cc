// Host code of the library (tree builder, loaders, solve) under AddressSanitizer + UBSan:
// scripts/sanitize/run.sh builds and runs this (CPU only; GPU sanitizers are not available).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "host_common.hh"

static uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static int fails = 0;
#define CHECK(cond)                                                        \
    do                                                                     \
    {                                                                      \
        if (!(cond))                                                       \
        {                                                                  \
            std::fprintf(stderr, "FAILED %s:%d %s\n", __FILE__, __LINE__, #cond); \
            ++fails;                                                       \
        }                                                                  \
    } while (0)

int main()
{
    using namespace icp_host;
    std::string err;
    // ---- tree build + validation over sizes around every boundary ----
    for (uint64_t n : {1ull, 2ull, 3ull, 7ull, 8ull, 9ull, 63ull, 64ull, 65ull, 1000ull, 20000ull, 150000ull})
        for (uint32_t cap : {3u, 8u, 32u, 256u, 100000u})
        {
            std::vector<float> xyz(3 * n);
            for (uint64_t i = 0; i < 3 * n; ++i)
                xyz[i] = (float)((double)(splitmix64(77 ^ i) >> 40) * (1.0 / 16777216.0) * 2.0 - 1.0);
            float mu[3];
            mean_host(xyz.data(), n, mu);
            std::vector<float> c(3 * n);
            center_host(xyz.data(), n, mu, c.data());
            FlatTreeOwner* t = nullptr;
            const int rc = build_flat_tree(c.data(), n, cap, &t, err);
            if (rc == ICP_HIP_OK)
            {
                CHECK(validate_flat_tree(&t->view, err) == ICP_HIP_OK);
                CHECK(t->view.n_points == n);
                delete t;
            }
            else
                CHECK(rc == ICP_HIP_ERR_DEGENERATE || rc == ICP_HIP_ERR_ARG);
        }
    // degenerate clouds: all points equal, many duplicates
    {
        std::vector<float> xyz(3 * 500, 0.25f);
        FlatTreeOwner* t = nullptr;
        CHECK(build_flat_tree(xyz.data(), 500, 8, &t, err) != ICP_HIP_OK && t == nullptr);
        CHECK(build_flat_tree(xyz.data(), 500, 2, &t, err) != ICP_HIP_OK);
        CHECK(build_flat_tree(nullptr, 0, 8, &t, err) != ICP_HIP_OK);
    }
    // ---- loaders: good file, extra columns, bad header, bad token, ragged line, missing file ----
    {
        const std::string dir = "/tmp/icp_sanitize_";
        auto write = [&](const char* name, const char* text) {
            std::ofstream f(dir + name);
            f << text;
            return dir + name;
        };
        std::vector<float> v;
        CHECK(read_csv(write("a.txt", "Points_0,Points_1,Points_2\n1,2,3\n4,5,6\n").c_str(), "Points_0",
                       "Points_1", "Points_2", v, err) == ICP_HIP_OK && v.size() == 6);
        v.clear(); // read_csv appends
        CHECK(read_csv(write("b.txt", "id,Points_2,Points_0,w,Points_1\n0,3,1,9,2\n").c_str(), "Points_0",
                       "Points_1", "Points_2", v, err) == ICP_HIP_OK && v.size() == 3 && v[0] == 1 && v[2] == 3);
        CHECK(read_csv(write("c.txt", "x,y,z\n1,2,3\n").c_str(), "Points_0", "Points_1", "Points_2", v, err) != ICP_HIP_OK);
        CHECK(read_csv(write("d.txt", "Points_0,Points_1,Points_2\n1,abc,3\n").c_str(), "Points_0", "Points_1",
                       "Points_2", v, err) != ICP_HIP_OK);
        (void)read_csv(write("e.txt", "Points_0,Points_1,Points_2\n1,2\n\n,,\n").c_str(), "Points_0", "Points_1",
                       "Points_2", v, err);
        (void)read_csv(write("f.txt", "").c_str(), "Points_0", "Points_1", "Points_2", v, err);
        CHECK(read_csv("/nonexistent/file.txt", "Points_0", "Points_1", "Points_2", v, err) != ICP_HIP_OK);
        {
            std::ofstream f(dir + "g.f32", std::ios::binary);
            const float pts[7] = {1, 2, 3, 4, 5, 6, 7};
            f.write((const char*)pts, sizeof pts); // 7 floats: not a multiple of 3
        }
        (void)read_f32((dir + "g.f32").c_str(), v, err);
        CHECK(read_f32("/nonexistent/file.f32", v, err) != ICP_HIP_OK);
    }
    // ---- the solve: regular, rank-deficient, zero, huge, tiny ----
    {
        const float mu[3] = {0.1f, -0.2f, 0.3f};
        const float Hs[][9] = {{132, 3, -2, 1, 38, 4, -3, 2, 8.5f},
                               {1, 1, 1, 1, 1, 1, 1, 1, 1},
                               {0, 0, 0, 0, 0, 0, 0, 0, 0},
                               {1e30f, 0, 0, 0, 1e30f, 0, 0, 0, 1e30f},
                               {1e-38f, 0, 0, 0, 1e-39f, 0, 0, 0, 0},
                               {-1, 0, 0, 0, -1, 0, 0, 0, -1}};
        for (const auto& H : Hs)
        {
            float T[16];
            alignment_from_covariance(H, mu, mu, T);
            for (int i = 0; i < 16; ++i)
                CHECK(std::isfinite(T[i]));
            float P[16];
            matmul4(T, T, P);
        }
    }
    std::printf(fails ? "host sanitize driver: %d check(s) failed\n" : "host sanitize driver: clean\n", fails);
    return fails != 0;
}

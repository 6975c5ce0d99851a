// cli/icp_bench -- the reference's benchmark harness (bench/bench/benchmark.cc:12-92) over the C ABI.
//
// The reference registers BM_CPU and BM_GPU with Google Benchmark (fetched by URL, absent offline)
// and sweeps vp_threshold = 8, 16, .. 1024 over five file pairs (CustomArguments, :21-26); each
// timed iteration is ONE whole icp(q, p, 200, 1e-5, vp_threshold) call, in real time, reported in
// milliseconds (:78, :87-90).  This program times the same thing for the MI355X backend and prints
// the same CSV columns as the reference's logs (analysis/logs/v18.csv:1), so its analysis scripts
// read it unchanged:
//
//   name,iterations,real_time,cpu_time,time_unit,bytes_per_second,items_per_second,label,
//   error_occurred,error_message            + the user counters icp_iterations, nn_kernel_ms
//
//   "BM_GPU/<capacity>/<file index>/real_time", label "<transformed> -> <reference>"
//
// What one timed call contains is what libgpu::icp contains (icp.cu:128-161): mean, centring, the
// tree, the upload of both clouds, the loop, the download of the moved cloud.  By default the
// device context is created and destroyed inside every timed call, as the reference allocates and
// frees everything inside the call (matrix.cu:6-17); --reuse-context keeps one context (stream,
// device buffers) across calls, which is what a long-lived caller of icp_hip_icp gets.
//
//   icp_bench [--data DIR] [--npy] [--capacities 8,16,..] [--files 0,1,..|none] [--min-time SECONDS]
//             [--reuse-context] [--phases] [--timers] [--synthetic N[:ANGLE_DEG]]
//             [--device cpu|gpu]
//
//   --data DIR     the reference's data directory (default ../data, as the reference's paths are
//                  relative to build/); with --npy: a directory of <name>.npy float32 (n,3) files
//                  (tests/golden/data)
//   --phases       additionally time the steps of one call separately (stderr, one line per case)
//   --timers       fill the nn_kernel_ms column (HIP events around every NN pass; they cost about
//                  20 us per iteration, so they are off when the call is what is timed)
//   --synthetic    add the synthetic cloud of bench.py (N points, rigid motion of ANGLE degrees,
//                  default 1) as one more file pair
//   --device cpu   BM_CPU is the reference's own libcpu, which this repository does not contain:
//                  refused, like cli/icp does

#include <sys/time.h>
#include <time.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "icp_hip.h"

namespace
{
    struct FilePair
    {
        const char* ref;
        const char* transformed;
    };
    // bench/bench/benchmark.cc:12-18 (paths below the data directory)
    const FilePair kFiles[] = {
        {"line/line_ref.txt", "line/line_translated_2_3_4.txt"},
        {"cow/cow_ref.txt", "cow/cow_tr1.txt"},
        {"cow/cow_ref.txt", "cow/cow_tr2.txt"},
        {"horse/horse_ref.txt", "horse/horse_tr1.txt"},
        {"horse/horse_ref.txt", "horse/horse_tr2.txt"},
    };
    constexpr unsigned kNbFiles = sizeof(kFiles) / sizeof(kFiles[0]);

    std::string base_name(const std::string& file) // get_name, benchmark.cc:28-33
    {
        const size_t slash = file.rfind('/') + 1;
        const size_t dot = file.rfind('.');
        return file.substr(slash, dot - slash);
    }

    double now_real()
    {
        struct timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
    }
    double now_cpu()
    {
        struct timespec ts;
        clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts);
        return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
    }

    void check(int rc, const icp_hip_ctx* c, const char* what)
    {
        if (rc != ICP_HIP_OK)
            throw std::runtime_error(std::string(what) + ": " +
                                     (c ? icp_hip_last_error(c) : icp_hip_last_global_error()));
    }

    struct Cloud
    {
        std::vector<float> xyz;
        uint64_t n() const { return xyz.size() / 3; }
    };

    // NumPy .npy, version 1/2/3, little-endian float32, C order, shape (n, 3)
    Cloud read_npy(const std::string& path)
    {
        std::ifstream f(path, std::ios::binary);
        if (!f)
            throw std::runtime_error("could not open " + path);
        unsigned char head[12];
        f.read((char*)head, 10);
        if (!f || std::memcmp(head, "\x93NUMPY", 6) != 0)
            throw std::runtime_error(path + ": not a .npy file");
        size_t hlen = head[8] | (head[9] << 8);
        if (head[6] >= 2)
        {
            f.read((char*)head + 10, 2);
            hlen |= ((size_t)head[10] << 16) | ((size_t)head[11] << 24);
        }
        std::string hdr(hlen, '\0');
        f.read(&hdr[0], (std::streamsize)hlen);
        if (hdr.find("'<f4'") == std::string::npos || hdr.find("'fortran_order': False") == std::string::npos)
            throw std::runtime_error(path + ": expected little-endian float32 in C order");
        const size_t sh = hdr.find("'shape': (");
        if (sh == std::string::npos)
            throw std::runtime_error(path + ": no shape");
        unsigned long long n = 0, k = 0;
        if (std::sscanf(hdr.c_str() + sh, "'shape': (%llu, %llu", &n, &k) != 2 || k != 3)
            throw std::runtime_error(path + ": expected shape (n, 3)");
        Cloud c;
        c.xyz.resize(3 * (size_t)n);
        f.read((char*)c.xyz.data(), (std::streamsize)(c.xyz.size() * sizeof(float)));
        if (!f)
            throw std::runtime_error(path + ": truncated");
        return c;
    }

    Cloud read_txt(const std::string& path) // read_csv as the harness calls it (benchmark.cc:71-72)
    {
        float* xyz = nullptr;
        uint64_t n = 0;
        check(icp_hip_read_csv(path.c_str(), "Points_0", "Points_1", "Points_2", &xyz, &n), nullptr,
              "read_csv");
        Cloud c;
        c.xyz.assign(xyz, xyz + 3 * n);
        icp_hip_free_points(xyz);
        return c;
    }

    // the synthetic workload of bench.py / icp_amd/synth.py
    uint64_t splitmix64(uint64_t x)
    {
        x += 0x9E3779B97F4A7C15ull;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        return x ^ (x >> 31);
    }
    Cloud synthetic_cloud(uint64_t n, uint64_t seed = 12345)
    {
        Cloud c;
        c.xyz.resize(3 * n);
        for (uint64_t i = 0; i < 3 * n; ++i)
        {
            const uint64_t u = splitmix64(seed ^ i);
            c.xyz[i] = (float)((double)(u >> 40) * (1.0 / 16777216.0) * 2.0 - 1.0);
        }
        return c;
    }
    Cloud synthetic_moved(const Cloud& m, double angle_deg)
    {
        // rotation about (1,2,3)/sqrt(14) by angle, translation (0.05,-0.03,0.02); applied in
        // fp32 in the order of apply_alignment (icp.cc:66-74)
        const double a = angle_deg * M_PI / 180.0, s14 = std::sqrt(14.0);
        const double ux = 1 / s14, uy = 2 / s14, uz = 3 / s14, co = std::cos(a), si = std::sin(a);
        const double R[9] = {co + ux * ux * (1 - co),      ux * uy * (1 - co) - uz * si, ux * uz * (1 - co) + uy * si,
                             uy * ux * (1 - co) + uz * si, co + uy * uy * (1 - co),      uy * uz * (1 - co) - ux * si,
                             uz * ux * (1 - co) - uy * si, uz * uy * (1 - co) + ux * si, co + uz * uz * (1 - co)};
        float T[12];
        const float t[3] = {0.05f, -0.03f, 0.02f};
        for (int i = 0; i < 3; ++i)
        {
            for (int j = 0; j < 3; ++j)
                T[4 * i + j] = (float)R[3 * i + j];
            T[4 * i + 3] = t[i];
        }
        Cloud p;
        p.xyz.resize(m.xyz.size());
        for (uint64_t i = 0; i < m.n(); ++i)
        {
            const float x = m.xyz[3 * i], y = m.xyz[3 * i + 1], z = m.xyz[3 * i + 2];
            for (int r = 0; r < 3; ++r)
            {
                // built with -ffp-contract=off: separate multiplies and adds
                p.xyz[3 * i + r] = ((x * T[4 * r] + y * T[4 * r + 1]) + z * T[4 * r + 2]) + T[4 * r + 3];
            }
        }
        return p;
    }

    std::vector<unsigned> parse_list(const char* s)
    {
        std::vector<unsigned> v;
        while (*s)
        {
            char* end = nullptr;
            const unsigned long x = std::strtoul(s, &end, 10);
            if (end == s)
                throw std::runtime_error(std::string("bad list: ") + s);
            v.push_back((unsigned)x);
            s = (*end == ',') ? end + 1 : end;
        }
        return v;
    }

    struct Case
    {
        std::string label;
        Cloud q, p; // reference, transformed
    };

    bool g_timers = false; // --timers: collect nn_kernel_ms (costs ~20 us per iteration)

    struct OneCall
    {
        uint64_t iterations = 0;
        float error = 0.f;
        double nn_ms = 0.0;
    };

    // one libgpu::icp call (benchmark.cc:78)
    OneCall call_icp(icp_hip_ctx* persistent, const Case& c, unsigned capacity,
                     std::vector<float>& new_p)
    {
        icp_hip_ctx* ctx = persistent;
        if (!ctx)
            check(icp_hip_create(&ctx, -1), nullptr, "icp_hip_create");
        icp_hip_set_phase_timers(ctx, g_timers ? 1 : 0);
        float T[16];
        OneCall r;
        const int rc = icp_hip_icp(ctx, c.q.xyz.data(), c.q.n(), c.p.xyz.data(), c.p.n(), 200, 1e-5f,
                                   capacity, T, new_p.data(), &r.iterations, &r.error, nullptr,
                                   nullptr);
        if (rc != ICP_HIP_OK)
        {
            const std::string msg = icp_hip_last_error(ctx);
            if (!persistent)
                icp_hip_destroy(ctx);
            throw std::runtime_error(msg);
        }
        icp_hip_timing tm;
        if (icp_hip_get_timing(ctx, &tm) == ICP_HIP_OK)
            r.nn_ms = tm.nn_ms;
        if (!persistent)
            icp_hip_destroy(ctx);
        return r;
    }

    // the steps of one call, each closed by its own synchronisation (they all are in the library)
    void print_phases(const Case& c, unsigned capacity)
    {
        double t[8];
        std::vector<float> centred(c.q.xyz.size()), new_p(c.p.xyz.size());
        float mu_m[3], T[16];
        uint64_t iters = 0;
        float err = 0.f;
        icp_hip_ctx* ctx = nullptr;
        for (int rep = 0; rep < 2; ++rep) // the second repetition is reported
        {
            t[0] = now_real();
            check(icp_hip_create(&ctx, -1), nullptr, "create");
            t[1] = now_real();
            icp_hip_mean_host(c.q.xyz.data(), c.q.n(), mu_m);
            icp_hip_center_host(c.q.xyz.data(), c.q.n(), mu_m, centred.data());
            t[2] = now_real();
            check(icp_hip_build_tree(ctx, centred.data(), c.q.n(), capacity), ctx, "build_tree");
            t[3] = now_real();
            check(icp_hip_set_queries(ctx, c.p.xyz.data(), c.p.n(), c.p.n()), ctx, "set_queries");
            t[4] = now_real();
            check(icp_hip_run(ctx, mu_m, 200, 1e-5f, T, &iters, &err, nullptr, nullptr), ctx, "run");
            t[5] = now_real();
            check(icp_hip_get_points(ctx, new_p.data()), ctx, "get_points");
            t[6] = now_real();
            icp_hip_destroy(ctx);
            ctx = nullptr;
            t[7] = now_real();
        }
        std::fprintf(stderr,
                     "[phases] %-38s cap %4u: create %.3f  mean+centre %.3f  tree %.3f  queries %.3f  "
                     "loop %.3f (%llu it, %.3f/it)  download %.3f  destroy %.3f  total %.3f ms\n",
                     c.label.c_str(), capacity, 1e3 * (t[1] - t[0]), 1e3 * (t[2] - t[1]),
                     1e3 * (t[3] - t[2]), 1e3 * (t[4] - t[3]), 1e3 * (t[5] - t[4]),
                     (unsigned long long)iters, iters ? 1e3 * (t[5] - t[4]) / (double)iters : 0.0,
                     1e3 * (t[6] - t[5]), 1e3 * (t[7] - t[6]), 1e3 * (t[7] - t[0]));
    }
} // namespace

int main(int argc, char* argv[])
{
    std::string data = "../data";
    bool npy = false, reuse = false, phases = false, no_files = false;
    double min_time = 0.5; // Google Benchmark's default --benchmark_min_time
    std::vector<unsigned> capacities, files;
    std::vector<std::pair<uint64_t, double>> synthetic;
    try
    {
        for (int i = 1; i < argc; ++i)
        {
            const std::string a = argv[i];
            auto value = [&]() -> const char* {
                if (i + 1 >= argc)
                    throw std::runtime_error("Missing value of " + a);
                return argv[++i];
            };
            if (a == "--data")
                data = value();
            else if (a == "--npy")
                npy = true;
            else if (a == "--capacities")
                capacities = parse_list(value());
            else if (a == "--files")
            {
                const char* v = value();
                no_files = std::strcmp(v, "none") == 0;
                if (!no_files)
                    files = parse_list(v);
            }
            else if (a == "--min-time")
                min_time = std::stod(value());
            else if (a == "--reuse-context")
                reuse = true;
            else if (a == "--phases")
                phases = true;
            else if (a == "--timers")
                g_timers = true;
            else if (a == "--synthetic")
            {
                const std::string v = value();
                const size_t colon = v.find(':');
                synthetic.emplace_back(std::stoull(v.substr(0, colon)),
                                       colon == std::string::npos ? 1.0 : std::stod(v.substr(colon + 1)));
            }
            else if (a == "--device")
            {
                const std::string v = value();
                if (v == "cpu")
                {
                    std::cerr << "BM_CPU times the reference's libcpu::icp, which this build does "
                                 "not contain; link the reference's libcpu or use --device gpu."
                              << std::endl;
                    return 1;
                }
                if (v != "gpu")
                    throw std::runtime_error("Invalid device: " + v);
            }
            else
                throw std::runtime_error("Invalid argument: " + a);
        }
        if (capacities.empty())
            for (unsigned c = 8; c <= 1024; c *= 2) // CustomArguments, benchmark.cc:23
                capacities.push_back(c);
        if (files.empty() && !no_files)
            for (unsigned i = 0; i < kNbFiles; ++i)
                files.push_back(i);

        std::vector<Case> cases;
        std::vector<unsigned> case_index;
        for (unsigned f : files)
        {
            if (f >= kNbFiles)
                throw std::runtime_error("file index out of range");
            Case c;
            c.label = base_name(kFiles[f].transformed) + " -> " + base_name(kFiles[f].ref); // :35-38
            if (npy)
            {
                c.q = read_npy(data + "/" + base_name(kFiles[f].ref) + ".npy");
                c.p = read_npy(data + "/" + base_name(kFiles[f].transformed) + ".npy");
            }
            else
            {
                c.q = read_txt(data + "/" + kFiles[f].ref);
                c.p = read_txt(data + "/" + kFiles[f].transformed);
            }
            cases.push_back(std::move(c));
            case_index.push_back(f);
        }
        unsigned next_index = kNbFiles;
        for (const auto& s : synthetic)
        {
            Case c;
            char buf[96];
            std::snprintf(buf, sizeof buf, "synthetic_%llu_rot%g -> synthetic_%llu",
                          (unsigned long long)s.first, s.second, (unsigned long long)s.first);
            c.label = buf;
            c.q = synthetic_cloud(s.first);
            c.p = synthetic_moved(c.q, s.second);
            cases.push_back(std::move(c));
            case_index.push_back(next_index++);
        }

        icp_hip_ctx* persistent = nullptr;
        if (reuse)
            check(icp_hip_create(&persistent, -1), nullptr, "icp_hip_create");

        std::cout << "name,iterations,real_time,cpu_time,time_unit,bytes_per_second,items_per_second,"
                     "label,error_occurred,error_message,icp_iterations,nn_kernel_ms"
                  << std::endl;
        for (unsigned capacity : capacities)
            for (size_t k = 0; k < cases.size(); ++k)
            {
                const Case& c = cases[k];
                char name[96];
                std::snprintf(name, sizeof name, "BM_GPU/%u/%u/real_time", capacity, case_index[k]);
                std::vector<float> new_p(c.p.xyz.size());
                try
                {
                    call_icp(persistent, c, capacity, new_p); // code load, first-touch: untimed
                    uint64_t reps = 0;
                    double real = 0.0, cpu = 0.0;
                    OneCall last;
                    while (real < min_time || reps < 1)
                    {
                        const double r0 = now_real(), c0 = now_cpu();
                        last = call_icp(persistent, c, capacity, new_p);
                        real += now_real() - r0;
                        cpu += now_cpu() - c0;
                        ++reps;
                    }
                    std::printf("\"%s\",%llu,%g,%g,ms,,,\"%s\",,,%llu,%g\n", name,
                                (unsigned long long)reps, 1e3 * real / (double)reps,
                                1e3 * cpu / (double)reps, c.label.c_str(),
                                (unsigned long long)last.iterations, last.nn_ms);
                    std::fflush(stdout);
                    if (phases)
                        print_phases(c, capacity);
                }
                catch (const std::exception& e)
                {
                    // Google Benchmark's error columns (State::SkipWithError)
                    std::printf("\"%s\",0,0,0,ms,,,\"%s\",true,\"%s\",0,0\n", name, c.label.c_str(),
                                e.what());
                    std::fflush(stdout);
                }
            }
        icp_hip_destroy(persistent);
    }
    catch (const std::exception& e)
    {
        std::cerr << e.what() << std::endl;
        return 1;
    }
    return 0;
}

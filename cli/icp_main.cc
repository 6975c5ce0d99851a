// cli/icp -- the reference's command line (main/main/main.cc:9-57, options.cc) over the C ABI.
//
//   icp <path/to/reference.txt> <path/to/transformed.txt> [--device cpu|gpu] [--capacity <n>]
//       [--iterations <n>] [--error <e>]
//
// Same positionals, flags, defaults (capacity 0 = per-device default: 256 on gpu, 32 on cpu,
// options.cc:146-149,175-178), messages, and output: parameters, "Execution time", and the first
// ten "reference point - closest moved point" pairs on stderr; "Transformation: " + the 4x4 +
// "Points" on stdout (main.cc:18-22,37-56).  The registration runs on the MI355X backend
// (include/icp_hip.h).  `--device cpu` selects the reference's own libcpu, which is not part of
// this repository: the flag is parsed as the reference parses it and then refused.
// Self-contained on purpose (no reference headers): the GPU box has no copy of the reference.

#include <sys/time.h>

#include <cstdint>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "icp_hip.h"

namespace
{
    struct Options
    {
        const char* reference = nullptr;
        const char* transformed = nullptr;
        bool gpu = true;
        unsigned capacity = 0;
        unsigned long iterations = 200;
        float error = 1e-5f;
    };

    // one handler per flag: consumes 2 argv slots, 0 on error (options.cc:16-111)
    int want_value(int argc, const char* what)
    {
        if (argc <= 1)
        {
            std::cerr << "Missing " << what << "." << std::endl;
            return 0;
        }
        return 2;
    }
    int on_device(Options& o, int argc, char* argv[])
    {
        if (!want_value(argc, "device"))
            return 0;
        if (std::strcmp(argv[1], "gpu") == 0)
            o.gpu = true;
        else if (std::strcmp(argv[1], "cpu") == 0)
            o.gpu = false;
        else
        {
            std::cerr << "Invalid device: " << argv[1] << std::endl;
            return 0;
        }
        return 2;
    }
    template <class T, class Parse>
    int on_number(T& field, const char* what, int argc, char* argv[], Parse parse)
    {
        if (!want_value(argc, what))
            return 0;
        try
        {
            auto v = parse(argv[1]);
            if (v >= 0)
                field = (T)v;
            else
            {
                std::cerr << "Invalid " << what << ": " << v << std::endl;
                return 0;
            }
        }
        catch (const std::exception&)
        {
            std::cerr << "Invalid " << what << ": " << argv[1] << std::endl;
            return 0;
        }
        return 2;
    }
    int on_capacity(Options& o, int argc, char* argv[])
    {
        return on_number(o.capacity, "capacity", argc, argv, [](const char* s) { return std::stoi(s); });
    }
    int on_iterations(Options& o, int argc, char* argv[])
    {
        return on_number(o.iterations, "iterations", argc, argv,
                         [](const char* s) { return std::stoi(s); });
    }
    int on_error(Options& o, int argc, char* argv[])
    {
        return on_number(o.error, "error", argc, argv, [](const char* s) { return std::stof(s); });
    }

    struct Flag
    {
        const char* flag;
        const char* help;
        int (*handler)(Options&, int, char*[]);
    };
    const Flag kFlags[] = {
        {"--device", "cpu|gpu: the device to run on [default: gpu]", on_device},
        {"--capacity", "<capacity>: the vp tree capacity [default: 32 on cpu, 256 on gpu]",
         on_capacity},
        {"--iterations", "<iterations>: the maximum number of iterations [default: 200]",
         on_iterations},
        {"--error", "<error>: the error threshold [default: 1e-5]", on_error},
    };

    bool parse(Options& o, int argc, char* argv[]) // argv past the program name (options.cc:142-180)
    {
        if (argc < 2)
            return false;
        o.reference = argv[0];
        o.transformed = argv[1];
        for (int i = 2; i < argc;)
        {
            bool found = false;
            for (const Flag& f : kFlags)
                if (std::strcmp(argv[i], f.flag) == 0)
                {
                    found = true;
                    const int used = f.handler(o, argc - i, argv + i);
                    if (used == 0)
                        return false;
                    i += used;
                    if (i >= argc)
                        break;
                }
            if (!found)
            {
                std::cerr << "Invalid argument: " << argv[i] << std::endl;
                return false;
            }
        }
        if (o.capacity == 0)
            o.capacity = o.gpu ? 256 : 32;
        return true;
    }

    void help(const char* program)
    {
        std::cerr << "Usage: " << program
                  << " <path/to/reference.txt> <path/to/transformed.txt> [options]" << std::endl;
        std::cerr << "Options:" << std::endl;
        for (const Flag& f : kFlags)
            std::cerr << "    " << f.flag << ' ' << f.help << std::endl;
    }

    struct Points
    {
        float* xyz = nullptr;
        uint64_t n = 0;
        ~Points() { icp_hip_free_points(xyz); }
    };
    struct Ctx
    {
        icp_hip_ctx* c = nullptr;
        ~Ctx() { icp_hip_destroy(c); }
    };
    void check(int rc, const icp_hip_ctx* c)
    {
        if (rc != ICP_HIP_OK)
            throw std::runtime_error(c ? icp_hip_last_error(c) : icp_hip_last_global_error());
    }
    void progress(uint64_t it, uint64_t max_iters, float error, void*)
    {
        std::cerr << "Starting iter " << it << "/" << max_iters << std::endl; // icp.cu:146-147
        std::cerr << "Error: " << error << std::endl;                          // icp.cu:157
    }
    void print_point(const float* p)
    {
        std::cerr << "(" << p[0] << ", " << p[1] << ", " << p[2] << ")"; // point-3d.cc:92-96
    }
} // namespace

int main(int argc, char* argv[])
{
    Options o;
    if (!parse(o, argc - 1, argv + 1))
    {
        help(argv[0]);
        return 1;
    }
    std::cerr << "Parameters:" << std::endl;
    std::cerr << "    Device: " << (o.gpu ? "GPU" : "CPU") << std::endl;
    std::cerr << "    Capacity: " << o.capacity << std::endl;
    std::cerr << "    Max iterations: " << o.iterations << std::endl;
    std::cerr << "    Error threshold: " << o.error << std::endl;
    if (!o.gpu)
    {
        std::cerr << "--device cpu runs the reference's libcpu::icp, which this build does not "
                     "contain; link the reference's libcpu or use --device gpu."
                  << std::endl;
        return 1;
    }

    // like the reference, errors are not caught: the message and a non-zero exit (main.cc has no try)
    Points q, p;
    check(icp_hip_read_csv(o.reference, "Points_0", "Points_1", "Points_2", &q.xyz, &q.n), nullptr);
    check(icp_hip_read_csv(o.transformed, "Points_0", "Points_1", "Points_2", &p.xyz, &p.n), nullptr);

    struct timeval start, end;
    gettimeofday(&start, nullptr);
    float T[16];
    std::vector<float> new_p(3 * p.n);
    {
        Ctx ctx;
        check(icp_hip_create(&ctx.c, -1), nullptr);
        uint64_t iters = 0;
        float err = 0.f;
        // libgpu::icp(q, p, iterations, error, capacity) (main.cc:31-34)
        check(icp_hip_icp(ctx.c, q.xyz, q.n, p.xyz, p.n, o.iterations, o.error, o.capacity, T,
                          new_p.data(), &iters, &err, progress, nullptr),
              ctx.c);
    }
    gettimeofday(&end, nullptr);
    const long start_ms = start.tv_sec * 1000 + start.tv_usec / 1000;
    const long end_ms = end.tv_sec * 1000 + end.tv_usec / 1000;
    std::cerr << "Execution time: " << (end_ms - start_ms) << "ms" << std::endl;

    std::cout << "Transformation: " << std::endl;
    for (int i = 0; i < 4; ++i)
    {
        std::cout << T[4 * i];
        for (int j = 1; j < 4; ++j)
            std::cout << ' ' << T[4 * i + j];
        std::cout << std::endl;
    }
    std::cout << "Points" << std::endl;
    // first <= 10 reference points and their brute-force closest moved point
    // (libcpu::closest, point-3d.cc:145-175: first strict minimum of the squared distance)
    const uint64_t shown = q.n > 10 ? 10 : q.n;
    for (uint64_t i = 0; i < shown && p.n > 0; ++i)
    {
        const float* a = q.xyz + 3 * i;
        uint64_t best = 0;
        float best_d = 0.f;
        for (uint64_t j = 0; j < p.n; ++j)
        {
            const float dx = a[0] - new_p[3 * j], dy = a[1] - new_p[3 * j + 1],
                        dz = a[2] - new_p[3 * j + 2];
            const float d = dx * dx + dy * dy + dz * dz;
            if (j == 0 || d < best_d)
            {
                best_d = d;
                best = j;
            }
        }
        print_point(a);
        std::cerr << " - ";
        print_point(new_p.data() + 3 * best);
        std::cerr << std::endl;
    }
    return 0;
}

// Point-cloud loaders behind the C ABI (host code).
//
// icp_hip_read_csv restates libcpu::read_csv (libcpu/libcpu/point-3d.cc:8-134): the first line is
// a comma-separated header, the three coordinate columns are found BY NAME (exact match, a later
// duplicate wins, :62-79), every other column is ignored, every following line yields one point
// whose tokens go through std::stof (:25-38: leading blanks skipped, trailing garbage ignored,
// inf/nan/hex accepted, a token without a number is an error), a line with fewer columns than the
// largest wanted index is an error (:44-47).  Same messages as the reference's runtime_errors.
//
// icp_hip_read_f32 is the loader for clouds where parsing text would dwarf the registration
// (64M points): raw little-endian float32 x,y,z triples, the Point3D layout itself.

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "host_common.hh"
#include "icp_hip.h"

namespace icp_host
{
    namespace
    {
        constexpr size_t kNone = (size_t)-1;

        // index of each wanted field in the header (point-3d.cc:50-87)
        void find_fields(const std::string& header, const std::string want[3], size_t id[3])
        {
            id[0] = id[1] = id[2] = kNone;
            size_t col = 0, start = 0, end;
            do
            {
                end = header.find(',', start);
                const std::string token(header, start, end - start);
                for (int k = 0; k < 3; ++k)
                    if (token == want[k])
                    {
                        id[k] = col;
                        break; // x is tested first, then y, then z (:69-74)
                    }
                start = end + 1;
                ++col;
            } while (end != std::string::npos);
            if (id[0] == kNone || id[1] == kNone || id[2] == kNone)
                throw std::runtime_error("could not find one of the coordinates");
        }

        // one data line (point-3d.cc:10-48)
        void parse_point(const std::string& line, const size_t id[3], float out[3])
        {
            const size_t last = std::max(std::max(id[0], id[1]), id[2]);
            out[0] = out[1] = out[2] = 0.f;
            size_t col = 0, start = 0, end;
            do
            {
                end = line.find(',', start);
                // x before y before z: with two names on one column only the first is filled
                const int k = col == id[0] ? 0 : col == id[1] ? 1 : col == id[2] ? 2 : -1;
                if (k >= 0)
                    out[k] = std::stof(std::string(line, start, end - start));
                start = end + 1;
                ++col;
            } while (end != std::string::npos && col <= last);
            if (col <= last)
                throw std::runtime_error("could not find one of the coordinates");
        }
    } // namespace

    int read_csv(const char* path, const char* x_field, const char* y_field, const char* z_field,
                 std::vector<float>& xyz, std::string& err)
    {
        try
        {
            std::ifstream stream(path);
            if (!stream)
                throw std::runtime_error("could not open file");
            std::string header;
            if (!std::getline(stream, header))
                throw std::runtime_error("header line not found");
            const std::string want[3] = {x_field, y_field, z_field};
            size_t id[3];
            find_fields(header, want, id);
            std::string line;
            float p[3];
            while (std::getline(stream, line))
            {
                parse_point(line, id, p);
                xyz.insert(xyz.end(), p, p + 3);
            }
        }
        catch (const std::bad_alloc&)
        {
            err = "icp_hip_read_csv: out of host memory";
            return ICP_HIP_ERR_NOMEM;
        }
        catch (const std::exception& e)
        {
            // std::runtime_error texts above; std::stof's invalid_argument / out_of_range ("stof")
            err = std::string("icp_hip_read_csv: ") + e.what();
            return ICP_HIP_ERR_ARG;
        }
        return ICP_HIP_OK;
    }

    int read_f32(const char* path, std::vector<float>& xyz, std::string& err)
    {
        std::FILE* f = std::fopen(path, "rb");
        if (!f)
        {
            err = "icp_hip_read_f32: could not open file";
            return ICP_HIP_ERR_ARG;
        }
        int rc = ICP_HIP_OK;
        try
        {
            std::fseek(f, 0, SEEK_END);
            const long long bytes = std::ftell(f);
            std::fseek(f, 0, SEEK_SET);
            if (bytes < 0 || bytes % 12 != 0)
            {
                err = "icp_hip_read_f32: file size is not a multiple of 12 bytes (x,y,z float32)";
                rc = ICP_HIP_ERR_ARG;
            }
            else
            {
                xyz.resize((size_t)bytes / 4);
                if (bytes && std::fread(xyz.data(), 1, (size_t)bytes, f) != (size_t)bytes)
                {
                    err = "icp_hip_read_f32: short read";
                    rc = ICP_HIP_ERR_ARG;
                }
            }
        }
        catch (const std::bad_alloc&)
        {
            err = "icp_hip_read_f32: out of host memory";
            rc = ICP_HIP_ERR_NOMEM;
        }
        std::fclose(f);
        return rc;
    }
} // namespace icp_host

// Host instance of the rigid-transform solve (align3.hh): icp_hip_alignment_host and the
// host-driven registration loop call it; nine floats in, sixteen out, a microsecond.
#include "align3.hh"
#include "host_common.hh"

namespace icp_host
{
    void alignment_from_covariance(const float H[9], const float mu_p[3], const float mu_m[3],
                                   float T[16])
    {
        icp_align::alignment_from_covariance(H, mu_p, mu_m, T);
    }

    void matmul4(const float a[16], const float b[16], float out[16])
    {
        icp_align::matmul4(a, b, out);
    }
} // namespace icp_host

// Rigid-transform solve on the host: 3x3 cross-covariance -> rotation by SVD -> 4x4.
//
// Replaces libgpu's find_rotation (libgpu/libgpu/svd.cc:7-28: Eigen::JacobiSVD<Matrix3f>,
// ComputeFullU|ComputeFullV, returns (U V^T)^T) and the transpose/dot/subtract/
// to_transformation kernel chain of libgpu/libgpu/icp.cu:35-62; semantics are libcpu's
// find_alignment (libcpu/libcpu/icp.cc:27-60).  Nine floats in, sixteen out: a microsecond on
// the host, not worth a kernel (and no MFMA anywhere: nothing here is a dense contraction).
//
// The SVD is a two-sided Jacobi in fp32 that follows the published Eigen 3.3 algorithm
// (scale by max |h|; sweep pairs (1,0),(2,0),(2,1); per pair: symmetrising rotation, then a
// symmetric 2x2 Jacobi; stop below 2*eps*maxDiag; fold signs into U; sort descending).  There is
// NO det < 0 (reflection) correction, like the reference.

#include <cfloat>
#include <cmath>
#include <cstring>
#include <utility>

#include "host_common.hh"

namespace icp_host
{
    namespace
    {
        struct Givens
        {
            float c = 1.0f, s = 0.0f;
            Givens transposed() const { return {c, -s}; }
            // composition in Eigen's convention for real rotations
            Givens then(const Givens& o) const { return {c * o.c - s * o.s, c * o.s + s * o.c}; }
            // (x, y) <- (c x + s y, -s x + c y)
            void rotate(float& x, float& y) const
            {
                const float a = x, b = y;
                x = c * a + s * b;
                y = -s * a + c * b;
            }
        };

        struct M3
        {
            float v[3][3];
            void rows(int p, int q, const Givens& g)
            {
                for (int k = 0; k < 3; ++k)
                    g.rotate(v[p][k], v[q][k]);
            }
            void cols(int p, int q, const Givens& g)
            {
                const Givens t = g.transposed();
                for (int k = 0; k < 3; ++k)
                    t.rotate(v[k][p], v[k][q]);
            }
            static M3 identity()
            {
                M3 m;
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j)
                        m.v[i][j] = (i == j) ? 1.0f : 0.0f;
                return m;
            }
        };

        // Jacobi rotation diagonalising the symmetric [[x, y], [y, z]]
        Givens symmetric_jacobi(float x, float y, float z)
        {
            const float two_abs_y = 2.0f * std::fabs(y);
            if (two_abs_y < FLT_MIN)
                return {};
            const float tau = (x - z) / two_abs_y;
            const float w = std::sqrt(tau * tau + 1.0f);
            const float t = tau > 0.0f ? 1.0f / (tau + w) : 1.0f / (tau - w);
            const float sgn = t > 0.0f ? 1.0f : -1.0f;
            const float n = 1.0f / std::sqrt(t * t + 1.0f);
            Givens g;
            g.s = -sgn * (y / std::fabs(y)) * std::fabs(t) * n;
            g.c = n;
            return g;
        }

        void svd3(const float H[9], M3& U, float S[3], M3& V)
        {
            float scale = 0.0f;
            for (int i = 0; i < 9; ++i)
                scale = std::fmax(scale, std::fabs(H[i]));
            if (scale == 0.0f)
                scale = 1.0f;
            M3 W;
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j)
                    W.v[i][j] = H[3 * i + j] / scale;
            U = M3::identity();
            V = M3::identity();
            const float precision = 2.0f * FLT_EPSILON;
            float max_diag = std::fmax(std::fabs(W.v[0][0]),
                                       std::fmax(std::fabs(W.v[1][1]), std::fabs(W.v[2][2])));
            for (int sweep = 0; sweep < 1000; ++sweep)
            {
                bool rotated = false;
                for (int p = 1; p < 3; ++p)
                    for (int q = 0; q < p; ++q)
                    {
                        const float thr = std::fmax(FLT_MIN, precision * max_diag);
                        if (!(std::fabs(W.v[p][q]) > thr || std::fabs(W.v[q][p]) > thr))
                            continue;
                        rotated = true;
                        float a = W.v[p][p], b = W.v[p][q], c = W.v[q][p], d = W.v[q][q];
                        Givens sym;
                        const float tr = a + d, df = c - b;
                        if (!(std::fabs(df) < FLT_MIN))
                        {
                            const float u = tr / df;
                            const float h = std::sqrt(1.0f + u * u);
                            sym.s = 1.0f / h;
                            sym.c = u / h;
                        }
                        sym.rotate(a, c);
                        sym.rotate(b, d);
                        const Givens right = symmetric_jacobi(a, b, d);
                        const Givens left = sym.then(right.transposed());
                        W.rows(p, q, left);
                        U.cols(p, q, left.transposed());
                        W.cols(p, q, right);
                        V.cols(p, q, right);
                        max_diag = std::fmax(
                            max_diag, std::fmax(std::fabs(W.v[p][p]), std::fabs(W.v[q][q])));
                    }
                if (!rotated)
                    break;
            }
            for (int i = 0; i < 3; ++i)
            {
                const float a = W.v[i][i];
                S[i] = std::fabs(a) * scale;
                if (a < 0.0f)
                    for (int k = 0; k < 3; ++k)
                        U.v[k][i] = -U.v[k][i];
            }
            for (int i = 0; i < 3; ++i)
            {
                int pos = i;
                for (int k = i + 1; k < 3; ++k)
                    if (S[k] > S[pos])
                        pos = k;
                if (S[pos] == 0.0f)
                    break;
                if (pos != i)
                {
                    std::swap(S[i], S[pos]);
                    for (int k = 0; k < 3; ++k)
                    {
                        std::swap(U.v[k][i], U.v[k][pos]);
                        std::swap(V.v[k][i], V.v[k][pos]);
                    }
                }
            }
        }
    } // namespace

    void alignment_from_covariance(const float H[9], const float mu_p[3], const float mu_m[3],
                                   float T[16])
    {
        M3 U, V;
        float S[3];
        svd3(H, U, S, V);
        // r = (U V^T)^T = V U^T, icp.cc:48-55
        float r[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                r[j][i] = U.v[i][0] * V.v[j][0] + (U.v[i][1] * V.v[j][1] + U.v[i][2] * V.v[j][2]);
        std::memset(T, 0, 16 * sizeof(float));
        for (int i = 0; i < 3; ++i)
        {
            // t = mu_m - r * mu_p, icp.cc:57 with point-3d.cc:248-266
            const float rp = r[i][0] * mu_p[0] + r[i][1] * mu_p[1] + r[i][2] * mu_p[2];
            for (int j = 0; j < 3; ++j)
                T[4 * i + j] = r[i][j];
            T[4 * i + 3] = mu_m[i] - rp;
        }
        T[15] = 1.0f;
    }

    void matmul4(const float a[16], const float b[16], float out[16])
    {
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
            {
                float acc = 0.0f;
                for (int k = 0; k < 4; ++k)
                    acc += a[4 * i + k] * b[4 * k + j];
                out[4 * i + j] = acc;
            }
    }
} // namespace icp_host

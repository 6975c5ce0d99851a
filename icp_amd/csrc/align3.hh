// Rigid-transform solve: 3x3 cross-covariance -> rotation by SVD -> 4x4.  One source for the host
// (align3.cc, g++) and the device (loop.hip: the iteration loop that stays on the GPU); both are
// built with -ffp-contract=off and use only IEEE add/mul/div/sqrt in fp32, so the two give the
// same bits (tests/test_gpu_icp_parity.py compares the host-driven and the device-resident loop).
//
// Replaces libgpu's find_rotation (libgpu/libgpu/svd.cc:7-28: Eigen::JacobiSVD<Matrix3f>,
// ComputeFullU|ComputeFullV, returns (U V^T)^T) and the transpose/dot/subtract/
// to_transformation kernel chain of libgpu/libgpu/icp.cu:35-62; semantics are libcpu's
// find_alignment (libcpu/libcpu/icp.cc:27-60).
//
// The SVD is a two-sided Jacobi in fp32 that follows the published Eigen 3.3 algorithm
// (scale by max |h|; sweep pairs (1,0),(2,0),(2,1); per pair: symmetrising rotation, then a
// symmetric 2x2 Jacobi; stop below 2*eps*maxDiag; fold signs into U; sort descending).  There is
// NO det < 0 (reflection) correction, like the reference.
#pragma once

#include <float.h>
#include <math.h>

#if defined(__HIPCC__)
#define ICP_HD __host__ __device__
#else
#define ICP_HD
#endif

namespace icp_align
{
    ICP_HD inline void swapf(float& a, float& b)
    {
        const float t = a;
        a = b;
        b = t;
    }

    struct Givens
    {
        float c = 1.0f, s = 0.0f;
        ICP_HD Givens transposed() const { { Givens g; g.c = c; g.s = -s; return g; } }
        // composition in Eigen's convention for real rotations
        ICP_HD Givens then(const Givens& o) const { { Givens g; g.c = c * o.c - s * o.s; g.s = c * o.s + s * o.c; return g; } }
        // (x, y) <- (c x + s y, -s x + c y)
        ICP_HD void rotate(float& x, float& y) const
        {
            const float a = x, b = y;
            x = c * a + s * b;
            y = -s * a + c * b;
        }
    };

    struct M3
    {
        float v[3][3];
        ICP_HD void rows(int p, int q, const Givens& g)
        {
            for (int k = 0; k < 3; ++k)
                g.rotate(v[p][k], v[q][k]);
        }
        ICP_HD void cols(int p, int q, const Givens& g)
        {
            const Givens t = g.transposed();
            for (int k = 0; k < 3; ++k)
                t.rotate(v[k][p], v[k][q]);
        }
        ICP_HD static M3 identity()
        {
            M3 m;
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j)
                    m.v[i][j] = (i == j) ? 1.0f : 0.0f;
            return m;
        }
    };

    // Jacobi rotation diagonalising the symmetric [[x, y], [y, z]]
    ICP_HD inline Givens symmetric_jacobi(float x, float y, float z)
    {
        const float two_abs_y = 2.0f * fabsf(y);
        if (two_abs_y < FLT_MIN)
            return Givens();
        const float tau = (x - z) / two_abs_y;
        const float w = sqrtf(tau * tau + 1.0f);
        const float t = tau > 0.0f ? 1.0f / (tau + w) : 1.0f / (tau - w);
        const float sgn = t > 0.0f ? 1.0f : -1.0f;
        const float n = 1.0f / sqrtf(t * t + 1.0f);
        Givens g;
        g.s = -sgn * (y / fabsf(y)) * fabsf(t) * n;
        g.c = n;
        return g;
    }

    ICP_HD inline void svd3(const float H[9], M3& U, float S[3], M3& V)
    {
        float scale = 0.0f;
        for (int i = 0; i < 9; ++i)
            scale = fmaxf(scale, fabsf(H[i]));
        if (scale == 0.0f)
            scale = 1.0f;
        M3 W;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                W.v[i][j] = H[3 * i + j] / scale;
        U = M3::identity();
        V = M3::identity();
        const float precision = 2.0f * FLT_EPSILON;
        float max_diag = fmaxf(fabsf(W.v[0][0]),
                                   fmaxf(fabsf(W.v[1][1]), fabsf(W.v[2][2])));
        for (int sweep = 0; sweep < 1000; ++sweep)
        {
            bool rotated = false;
            for (int p = 1; p < 3; ++p)
                for (int q = 0; q < p; ++q)
                {
                    const float thr = fmaxf(FLT_MIN, precision * max_diag);
                    if (!(fabsf(W.v[p][q]) > thr || fabsf(W.v[q][p]) > thr))
                        continue;
                    rotated = true;
                    float a = W.v[p][p], b = W.v[p][q], c = W.v[q][p], d = W.v[q][q];
                    Givens sym;
                    const float tr = a + d, df = c - b;
                    if (!(fabsf(df) < FLT_MIN))
                    {
                        const float u = tr / df;
                        const float h = sqrtf(1.0f + u * u);
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
                    max_diag = fmaxf(
                        max_diag, fmaxf(fabsf(W.v[p][p]), fabsf(W.v[q][q])));
                }
            if (!rotated)
                break;
        }
        for (int i = 0; i < 3; ++i)
        {
            const float a = W.v[i][i];
            S[i] = fabsf(a) * scale;
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
                swapf(S[i], S[pos]);
                for (int k = 0; k < 3; ++k)
                {
                    swapf(U.v[k][i], U.v[k][pos]);
                    swapf(V.v[k][i], V.v[k][pos]);
                }
            }
        }
    }

    // H (row = P coordinate, col = Y coordinate) -> row-major 4x4 [r t; 0 0 0 1]
    ICP_HD inline void alignment_from_covariance(const float H[9], const float mu_p[3],
                                                 const float mu_m[3], float T[16])
    {
        M3 U, V;
        float S[3];
        svd3(H, U, S, V);
        // r = (U V^T)^T = V U^T, icp.cc:48-55
        float r[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                r[j][i] = U.v[i][0] * V.v[j][0] + (U.v[i][1] * V.v[j][1] + U.v[i][2] * V.v[j][2]);
        for (int i = 0; i < 16; ++i)
            T[i] = 0.0f;
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

    // utils::dot for 4x4, libcpu/libcpu/utils/matrix.hh:213-232
    ICP_HD inline void matmul4(const float a[16], const float b[16], float out[16])
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
} // namespace icp_align

// Probe: operand and result layout of v_mfma_f32_16x16x4_f32 on gfx950 (run on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* D)
{
    const int l = threadIdx.x;
    const float a = A[(l % 16) * 4 + l / 16];  // A[i][k]
    const float b = B[(l / 16) * 16 + l % 16]; // B[k][j]
    v4f c = {0.f, 0.f, 0.f, 0.f};
    const v4f d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r)
        D[(4 * (l / 16) + r) * 16 + l % 16] = d[r]; // D[i][j]
}
int main()
{
    float hA[64], hB[64], hD[256], *dA, *dB, *dD;
    for (int i = 0; i < 64; ++i) { hA[i] = (float)(i % 7 + 1); hB[i] = (float)(i % 5 - 2); }
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j)
        {
            float e = 0.f;
            for (int kk = 0; kk < 4; ++kk) e += hA[i * 4 + kk] * hB[kk * 16 + j];
            if (e != hD[i * 16 + j]) ++bad;
        }
    std::printf("mfma 16x16x4 layout: %d mismatches\n", bad);
    return bad != 0;
}

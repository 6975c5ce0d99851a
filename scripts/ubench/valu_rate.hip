// Micro-benchmark: issue rate of the VALU instructions the bucket scan is made of (gfx950).
// Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = 1.0001f, b1 = 0.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, q = {b0, b1};
    for (int i = 0; i < iters; ++i)
    {
        if (MODE == 0) // v_add_f32 x16 (4 independent chains)
        {
            asm volatile(REP16("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));
        }
        else if (MODE == 1) // v_pk_add_f32
        {
            asm volatile(REP16("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));
        }
        else if (MODE == 2) // v_pk_mul_f32
        {
            asm volatile(REP16("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));
        }
        else if (MODE == 3) // v_min_f32
        {
            asm volatile(REP16("v_min_f32 %0, %0, %4\n v_min_f32 %1, %1, %4\n v_min_f32 %2, %2, %4\n v_min_f32 %3, %3, %4\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));
        }
        else if (MODE == 4) // v_med3_f32
        {
            asm volatile(REP16("v_med3_f32 %0, %0, %4, %5\n v_med3_f32 %1, %1, %4, %5\n v_med3_f32 %2, %2, %4, %5\n v_med3_f32 %3, %3, %4, %5\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
        }
        else if (MODE == 5) // v_cmp + v_cndmask pairs
        {
            asm volatile(REP16("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc");
        }
        else if (MODE == 6) // v_fma_f32
        {
            asm volatile(REP16("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
        }
        else if (MODE == 7) // v_mov_b32
        {
            asm volatile(REP16("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));
        }
        else if (MODE == 8) // v_sqrt_f32
        {
            asm volatile(REP16("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        }
        else if (MODE == 9) // v_min3_f32
        {
            asm volatile(REP16("v_min3_f32 %0, %0, %4, %5\n v_min3_f32 %1, %1, %4, %5\n v_min3_f32 %2, %2, %4, %5\n v_min3_f32 %3, %3, %4, %5\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
        }
        else if (MODE == 10) // v_min_u32
        {
            asm volatile(REP16("v_min_u32 %0, %0, %4\n v_min_u32 %1, %1, %4\n v_min_u32 %2, %2, %4\n v_min_u32 %3, %3, %4\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));
        }
        else if (MODE == 11) // v_med3_u32
        {
            asm volatile(REP16("v_med3_u32 %0, %0, %4, %5\n v_med3_u32 %1, %1, %4, %5\n v_med3_u32 %2, %2, %4, %5\n v_med3_u32 %3, %3, %4, %5\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
        }
        else if (MODE == 12) // v_and_or_b32
        {
            asm volatile(REP16("v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %4, %5\n v_and_or_b32 %2, %2, %4, %5\n v_and_or_b32 %3, %3, %4, %5\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
        }
        else if (MODE == 13) // v_min3_u32
        {
            asm volatile(REP16("v_min3_u32 %0, %0, %4, %5\n v_min3_u32 %1, %1, %4, %5\n v_min3_u32 %2, %2, %4, %5\n v_min3_u32 %3, %3, %4, %5\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
        }
        else if (MODE == 14) // v_mul_f32
        {
            asm volatile(REP16("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));
        }
        else if (MODE == 15) // v_max_f32
        {
            asm volatile(REP16("v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_max_f32 %3, %3, %4\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));
        }
        else if (MODE == 16) // v_add_u32
        {
            asm volatile(REP16("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));
        }
        else if (MODE == 17) // v_min_i32
        {
            asm volatile(REP16("v_min_i32 %0, %0, %4\n v_min_i32 %1, %1, %4\n v_min_i32 %2, %2, %4\n v_min_i32 %3, %3, %4\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));
        }
        else if (MODE == 18) // v_sub_f32 with an SGPR operand
        {
            asm volatile(REP16("v_sub_f32 %0, %4, %0\n v_sub_f32 %1, %4, %1\n v_sub_f32 %2, %4, %2\n v_sub_f32 %3, %4, %3\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(iters));
        }
        else if (MODE == 19) // v_pk_fma_f32
        {
            asm volatile(REP16("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
void run(const char* name, float* d, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd; // 256 CUs x (4 waves per block -> 1 per SIMD) x w
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_wave = 64.0 * iters;                  // wave-instructions per wave
    const double cyc = ms * 1e-3 * 2.4e9;                        // at the nominal 2.4 GHz
    const double cyc_per_inst_per_simd = cyc / (insts_per_wave * waves_per_simd);
    printf("%-14s waves/SIMD %d  %.3f ms  -> %.2f cycles (2.4 GHz) per wave-instruction per SIMD\n", name,
           waves_per_simd, ms, cyc_per_inst_per_simd);
}

int main()
{
    float* d;
    hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    for (int w : {4, 8})
    {
        run<0>("v_add_f32", d, w);
        run<6>("v_fma_f32", d, w);
        run<1>("v_pk_add_f32", d, w);
        run<2>("v_pk_mul_f32", d, w);
        run<3>("v_min_f32", d, w);
        run<9>("v_min3_f32", d, w);
        run<4>("v_med3_f32", d, w);
        run<5>("v_cmp+cndmask", d, w);
        run<7>("v_mov_b32", d, w);
        run<8>("v_sqrt_f32", d, w);
        run<14>("v_mul_f32", d, w);
        run<15>("v_max_f32", d, w);
        run<10>("v_min_u32", d, w);
        run<17>("v_min_i32", d, w);
        run<13>("v_min3_u32", d, w);
        run<11>("v_med3_u32", d, w);
        run<12>("v_and_or_b32", d, w);
        run<16>("v_add_u32", d, w);
        run<18>("v_sub_f32 sgpr", d, w);
        run<19>("v_pk_fma_f32", d, w);
    }
    return 0;
}

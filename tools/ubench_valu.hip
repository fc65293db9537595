// ubench_valu.hip -- issue cost of the VALU instructions the DP kernels are made of, on gfx950.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
// Every wave runs ITER iterations of 32 independent instructions of one kind (8 chains x 4); the grid
// puts W waves on every SIMD.  Printed: SIMD cycles per wave64 instruction = time x clock x 1024 SIMDs /
// (total wave-instructions), clock taken as 2.4 GHz (the measured figure is an upper bound when the
// chip clocks lower).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ void __launch_bounds__(64) k(int iters, unsigned* out) {
    unsigned r[8];
    for (int q = 0; q < 8; ++q) r[q] = threadIdx.x * 2654435761u + q;
    const unsigned c = threadIdx.x | 0x10001u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#define STEP(q)                                                                                                    \
    if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[q]) : "v"(c));                                     \
    if (KIND == 1) asm volatile("v_max_i32 %0, %0, %1" : "+v"(r[q]) : "v"(c));                                     \
    if (KIND == 2) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(r[q]) : "v"(c));                                  \
    if (KIND == 3) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(r[q]) : "v"(c));                                  \
    if (KIND == 4) asm volatile("v_pk_mad_u16 %0, %0, 2, %1 op_sel_hi:[1,0,1]" : "+v"(r[q]) : "v"(c));             \
    if (KIND == 5) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[q]) : "v"(c));                                     \
    if (KIND == 6) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[q]) : "v"(c));                                 \
    if (KIND == 7) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(r[q]) : "v"(c));                            \
    if (KIND == 8) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[q]));       \
    if (KIND == 9) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(r[q]) : "v"(c));                                \
    if (KIND == 10) asm volatile("v_max_f64 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&r[q & 6])) : "v"(1.5)); \
    if (KIND == 11) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[q]) : "v"(c));                                    \
    if (KIND == 12) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&r[q & 6])) : "v"(*reinterpret_cast<const double*>(&r[(q & 6) ^ 2])));
            REP8(STEP)
        }
    }
    unsigned s = 0;
    for (int q = 0; q < 8; ++q) s ^= r[q];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int KIND>
static void run(const char* name, int waves_per_simd) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int grid = p.multiProcessorCount * 4 * waves_per_simd;
    unsigned* d;
    hipMalloc(&d, sizeof(unsigned) * grid * 64);
    const int iters = 20000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<KIND><<<grid, 64>>>(100, d);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<KIND><<<grid, 64>>>(iters, d);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double instr = double(grid) * iters * 32;
    const double cyc = ms * 1e-3 * 2.4e9 * p.multiProcessorCount * 4 / instr;
    printf("%-28s %d waves/SIMD: %.2f SIMD-cycles per wave64 instruction (%.2f ms)\n", name, waves_per_simd, cyc, ms);
    hipFree(d);
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_add_u32", w);
        run<1>("v_max_i32", w);
        run<2>("v_pk_add_i16", w);
        run<3>("v_pk_max_i16", w);
        run<4>("v_pk_mad_u16", w);
        run<5>("v_xor_b32", w);
        run<6>("v_fma_f32", w);
        run<11>("v_add_f32", w);
        run<12>("v_pk_add_f32", w);
        run<7>("v_alignbit_b32", w);
        run<8>("v_mov_b32_dpp wave_shr:1", w);
        run<9>("v_perm_b32", w);
        run<10>("v_max_f64", w);
    }
    return 0;
}

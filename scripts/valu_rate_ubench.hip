// valu_rate_ubench.hip -- issue cost of the 64-bit integer instructions the counting kernels lean on, against their
// 32-bit counterparts (gfx950).  Every lane runs a long chain of independent instructions of one kind in 8 registers;
// 4 waves per SIMD, every CU busy; reported: shader cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 512
template <int OP>
__global__ __launch_bounds__(256) void k(uint64_t *out, uint32_t sh, unsigned long long *cyc) {
    uint64_t a[8];
    uint32_t b[8];
    for (int i = 0; i < 8; ++i) { a[i] = (uint64_t)threadIdx.x * 0x9E3779B97F4A7C15ULL + i; b[i] = threadIdx.x * 2654435761u + i; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(a[i]) : "v"(sh));
            if (OP == 1) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(b[i]) : "v"(sh));
            if (OP == 2) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(a[i]) : "v"(sh));
            if (OP == 3) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(b[i]) : "v"(sh));
            if (OP == 4) asm volatile("v_cmp_ne_u64 vcc, %0, %1" :: "v"(a[i]), "v"(a[(i + 1) & 7]) : "vcc");
            if (OP == 5) asm volatile("v_cmp_ne_u32 vcc, %0, %1" :: "v"(b[i]), "v"(b[(i + 1) & 7]) : "vcc");
            if (OP == 6) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            if (OP == 7) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(b[i]) : "v"(b[(i + 1) & 7]), "v"(sh));
            if (OP == 8) asm volatile("v_bfe_u32 %0, %0, %1, 8" : "+v"(b[i]) : "v"(sh));
            if (OP == 9) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(b[i]) : "v"(sh));
            if (OP == 10) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(sh) : "vcc");
            if (OP == 11) asm volatile("v_mov_b64 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint64_t s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + b[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) atomicAdd(cyc, t1 - t0);
}
template <int OP> void run(const char *name) {
    uint64_t *out; unsigned long long *cyc, h = 0;
    const int grid = 256 * 4;   // 4 workgroups of 4 waves per CU = 4 waves per SIMD
    hipMalloc(&out, (size_t)grid * 256 * 8); hipMalloc(&cyc, 8); hipMemset(cyc, 0, 8);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, out, 3u, cyc);
    hipMemset(cyc, 0, 8);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, out, 3u, cyc);
    hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    // per workgroup: REP * 8 instructions per wave; a SIMD runs 4 waves (one of each of 4 workgroups)
    printf("%-16s %.2f cycles per wave-instruction and SIMD\n", name, (double)h / grid / (REP * 8.0) / 4.0);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<1>("v_lshrrev_b32"); run<0>("v_lshrrev_b64"); run<2>("v_lshlrev_b64"); run<3>("v_xor_b32"); run<5>("v_cmp_ne_u32");
    run<4>("v_cmp_ne_u64"); run<6>("v_lshl_add_u64"); run<7>("v_alignbit_b32"); run<8>("v_bfe_u32"); run<9>("v_mul_lo_u32");
    run<10>("v_mad_u64_u32"); run<11>("v_mov_b64");
    return 0;
}

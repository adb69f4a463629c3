// Micro-benchmark for the segment build: 1024 threads, 128 KiB of LDS, every lane inserts NK random keys with a
// DEPENDENT chain of 64-bit LDS CAS (quadratic probing, like build_segments_*): cycles per wave-level CAS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t xs(uint32_t &s){ s^=s<<13; s^=s>>17; s^=s<<5; return s; }
template <int MODE>   // 0: b64 CAS chain, 1: b32 CAS chain, 2: b64 CAS + ds_read_b64 per round, 3: first probe only (no chain)
__global__ __launch_bounds__(1024) void k(int nk, unsigned long long *out) {
    extern __shared__ uint64_t lds[];
    const uint32_t tid = threadIdx.x;
    uint32_t *l32 = (uint32_t *)lds;
    unsigned long long tot_cas = 0, t_all = 0;
    for (int rep = 0; rep < 16; ++rep) {
        for (uint32_t i = tid; i < 16384; i += 1024) lds[i] = 0;
        __syncthreads();
        uint32_t s = (tid + 1) * 2654435761u + blockIdx.x * 40503u + rep * 977u;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        int left = nk;
        uint32_t q = xs(s) & 16383u, i = 1;
        uint64_t key = ((uint64_t)xs(s) << 8) | 1u;
        unsigned long long ncas = 0;
        uint64_t acc = 0;
        while (__ballot(left > 0)) {
            if (left > 0) {
                bool placed;
                if (MODE == 1) {
                    const uint32_t old = atomicCAS(&l32[q * 2], 0u, (uint32_t)key | i);
                    placed = old == 0;
                } else {
                    const unsigned long long old = atomicCAS((unsigned long long *)&lds[q], 0ULL, (unsigned long long)(key | i));
                    placed = old == 0;
                    if (MODE == 2) acc += lds[(q * 7 + 5) & 16383u];
                }
                if (MODE == 3) placed = true;
                ++i; q = (q + i) & 16383u;
                if (placed) { --left; q = xs(s) & 16383u; i = 1; key = ((uint64_t)xs(s) << 8) | 1u; }
            }
            ++ncas;
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if ((tid & 63) == 0) { tot_cas += ncas; t_all += t1 - t0; }
        if (acc == 0x12345) out[3] = acc;
        __syncthreads();
    }
    if ((tid & 63) == 0) { atomicAdd(&out[0], tot_cas); atomicAdd(&out[1], t_all); }
}
int main() {
    unsigned long long *out, h[4];
    (void)hipMalloc(&out, 32);
    const char *names[4] = {"cas_b64 chain", "cas_b32 chain", "cas_b64 + read_b64", "cas_b64 first probe only"};
    for (int mode = 0; mode < 4; ++mode) {
        (void)hipMemset(out, 0, 32);
        hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 << 10);
        hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 << 10);
        hipFuncSetAttribute((const void *)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 << 10);
        hipFuncSetAttribute((const void *)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 << 10);
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        (void)hipEventRecord(a);
        switch (mode) {
            case 0: k<0><<<256, 1024, 128 << 10>>>(12, out); break;
            case 1: k<1><<<256, 1024, 128 << 10>>>(12, out); break;
            case 2: k<2><<<256, 1024, 128 << 10>>>(12, out); break;
            default: k<3><<<256, 1024, 128 << 10>>>(12, out); break;
        }
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        (void)hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
        // per CU: 16 waves; wave-rounds per rep = h[0] / (256*16*16)
        printf("%-26s %7.3f ms  rounds/wave/rep %6.1f  cycles/round(wave) %7.1f  => LDS pipe per wave-CAS if saturated: %5.1f cyc\n",
               names[mode], ms, (double)h[0] / (256.0 * 16 * 16), (double)h[1] / (double)h[0], (double)h[1] / (double)h[0] / 16.0);
    }
    return 0;
}

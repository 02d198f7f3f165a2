// Experiment: issue cost of a 64-bit shift by a wave-uniform amount (v_lshrrev_b64) against 32-bit forms on gfx950.
// Eight independent chains per lane, every SIMD of every CU busy with 8 waves; time per (wave-instruction x 64 lanes).
//   hipcc --offload-arch=gfx950 -O3 -o build/shift64_rate tools/exp/shift64_rate.hip && build/shift64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32;
typedef unsigned long long u64;
template <int MODE>
__global__ __launch_bounds__(256) void k(u64 *out, int n, int s) {
    u64 x[8];
    for (int i = 0; i < 8; i++) x[i] = (u64)(blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull + i;
    for (int r = 0; r < n; r++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) x[i] = (x[i] >> s) ^ 0x5555555555555555ull ^ (x[i] << 7);           // 64-bit shift right by SGPR + a 64-bit shift left by a constant
            else if (MODE == 1) x[i] = (x[i] >> s) + 0x1234567ull;                                 // 64-bit shift right by SGPR + 64-bit add
            else if (MODE == 2) { u32 lo = (u32)x[i], hi = (u32)(x[i] >> 32); lo = __funnelshift_r(lo, hi, s) + 0x1234567u; x[i] = ((u64)hi << 32) | lo; }   // alignbit + 32-bit add
            else { u32 lo = (u32)x[i], hi = (u32)(x[i] >> 32); lo = (lo >> s) + 0x1234567u; x[i] = ((u64)hi << 32) | lo; }   // 32-bit shift + add
        }
    }
    u64 t = 0;
    for (int i = 0; i < 8; i++) t ^= x[i];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}
template <int MODE>
static void run(const char *name, u64 *d, int n) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 8;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, n, 13);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, n, 13);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    // wave-iterations per SIMD: blocks*4 waves / (256 CUs * 4 SIMDs) * n * 8 chains
    const double per = ms * 1e-3 * 2.4e9 / ((double)blocks * 4 / 1024 * n * 8);
    printf("%-60s %8.3f ms  %6.2f clocks per chain step and wave (at 2.4 GHz)\n", name, ms, per);
}
int main() {
    u64 *d;
    hipMalloc(&d, 256 * 8 * 256 * 8);
    run<0>("x = (x >> s) ^ c ^ (x << 7)   [64-bit]", d, 4096);
    run<1>("x = (x >> s) + c              [64-bit]", d, 4096);
    run<2>("lo = alignbit(hi, lo, s) + c  [32-bit]", d, 4096);
    run<3>("lo = (lo >> s) + c            [32-bit]", d, 4096);
    return 0;
}

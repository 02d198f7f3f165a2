// Experiment: what bounds an LDS digit histogram on gfx950?  Rate of no-return / returning LDS atomic adds per CU and
// clock for several address patterns, and of the alternatives that avoid them.  Every block runs ROUNDS wave-instructions
// per wave on digits it derives from a hash (no memory traffic), so the figure is the LDS / VALU cost alone.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_hist_rate tools/exp/lds_hist_rate.hip && /tmp/lds_hist_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32;
typedef unsigned long long u64;

__device__ __forceinline__ u32 mix(u32 x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// mode 0: bin = lane (conflict-free)   1: random in 256   2: all one bin   3: random in 256, a histogram per wave
// mode 4: random in 256, returning atomics   5: random in 64 bins   6: random in 4096 bins
// mode 7: random in 256, lanes that share a bin folded by a wave-wide match first (one add per distinct bin)
// mode 8: random in 256, two digits per atomic: 16-bit packed counters, bins d and d^1 share a word
template <int MODE>
__global__ __launch_bounds__(256) void k(u32 *out, int rounds) {
    __shared__ u32 h[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) h[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 s = mix(blockIdx.x * 256 + threadIdx.x + 1);
    u32 acc = 0;
    for (int r = 0; r < rounds; r++) {
        s = s * 1664525u + 1013904223u;
        const u32 rnd = s >> 8;
        if (MODE == 0) atomicAdd(&h[lane], 1u);
        else if (MODE == 1) atomicAdd(&h[rnd & 255], 1u);
        else if (MODE == 2) atomicAdd(&h[7], 1u);
        else if (MODE == 3) atomicAdd(&h[wave * 256 + (rnd & 255)], 1u);
        else if (MODE == 4) acc += atomicAdd(&h[rnd & 255], 1u);
        else if (MODE == 5) atomicAdd(&h[rnd & 63], 1u);
        else if (MODE == 6) atomicAdd(&h[rnd & 4095], 1u);
        else if (MODE == 7) {
            const u32 d = rnd & 255;
            u64 peers = ~0ull;
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const u64 m = __ballot((d >> b) & 1);
                peers &= ((d >> b) & 1) ? m : ~m;
            }
            if ((peers & ((1ull << lane) - 1)) == 0) atomicAdd(&h[d], (u32)__popcll(peers));
        } else if (MODE == 8) {
            const u32 d = rnd & 255;
            atomicAdd(&h[d >> 1], (d & 1) ? 0x10000u : 1u);
        }
    }
    __syncthreads();
    u32 t = acc;
    for (int i = threadIdx.x; i < 4096; i += 256) t += h[i];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int MODE>
static void run(const char *name, u32 *d_out, int blocks, int rounds) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, rounds);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, rounds);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    const double adds = (double)blocks * 256 * rounds;
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    printf("%-46s %8.3f ms  %7.1f G digits/s  %6.2f digits per CU and clock (%d CUs, %d MHz nominal)\n", name, ms, adds / ms * 1e-6,
           adds / (ms * 1e-3) / cus / (clk_khz * 1e3), cus, clk_khz / 1000);
    hipEventDestroy(a); hipEventDestroy(b);
}

int main() {
    const int blocks = 256 * 8 * 4, rounds = 2048;          // 16 KB of LDS per block: 8 blocks per CU resident
    u32 *d_out;
    if (hipMalloc(&d_out, (size_t)blocks * 256 * 4) != hipSuccess) return 1;
    run<0>("no-return, bin = lane", d_out, blocks, rounds);
    run<1>("no-return, random of 256", d_out, blocks, rounds);
    run<2>("no-return, one bin", d_out, blocks, rounds);
    run<3>("no-return, random of 256, histogram per wave", d_out, blocks, rounds);
    run<4>("returning, random of 256", d_out, blocks, rounds);
    run<5>("no-return, random of 64", d_out, blocks, rounds);
    run<6>("no-return, random of 4096", d_out, blocks, rounds);
    run<7>("wave match first, one add per distinct bin", d_out, blocks, rounds);
    run<8>("no-return, 16-bit packed pairs", d_out, blocks, rounds);
    hipFree(d_out);
    return 0;
}

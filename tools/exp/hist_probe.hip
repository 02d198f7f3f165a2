// Experiment: what bounds rs_hist_kernel (radix.h)?  120 M random 64-bit keys, the digit histogram per 4096-key tile
// into a digit-major table, 16 tiles per block as the library does it -- and the same with parts taken out or changed.
//   hipcc --offload-arch=gfx950 -O3 -o build/hist_probe tools/exp/hist_probe.hip && build/hist_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32;
typedef unsigned long long u64;
#define BLOCK 256
#define ITEMS 16
#define TILE (BLOCK * ITEMS)
#define TPB 16

// V 0: as the library.  1: loads only.  2: atomics on a conflict-free digit (lane).  3: next tile's loads issued before this
// tile's adds.  4: a histogram per wave, summed at the end.  5: as 0 with 512-thread blocks' worth of loads in flight (two
// tiles at once).  6: digit from the key's low bits instead of bits 40-47 (same distribution; sanity)
template <int V>
__global__ __launch_bounds__(BLOCK) void hist(const u64 *__restrict__ keys, u32 n, int shift, u32 *__restrict__ table, u32 ntiles) {
    __shared__ u32 h[(V == 4 ? 4 : 1) * 256][TPB + 1];
    for (int r = 0; r < (V == 4 ? 4 : 1); r++)
        for (int t = 0; t < TPB; t++) h[r * 256 + threadIdx.x][t] = 0;
    __syncthreads();
    const u32 tile0 = blockIdx.x * TPB;
    const int wave = threadIdx.x >> 6;
    u64 sink = 0;
    if (V == 3) {
        u64 cur[ITEMS], nxt[ITEMS];
        {
            const u32 base = tile0 * TILE;
#pragma unroll
            for (int i = 0; i < ITEMS; i++) cur[i] = base + i * BLOCK + threadIdx.x < n ? keys[base + i * BLOCK + threadIdx.x] : 0;
        }
        for (int t = 0; t < TPB; t++) {
            const u32 nb = (tile0 + t + 1) * TILE;
            if (t + 1 < TPB) {
#pragma unroll
                for (int i = 0; i < ITEMS; i++) nxt[i] = nb + i * BLOCK + threadIdx.x < n ? keys[nb + i * BLOCK + threadIdx.x] : 0;
            }
            if ((tile0 + t) * TILE < n) {
#pragma unroll
                for (int i = 0; i < ITEMS; i++) atomicAdd(&h[(u32)(cur[i] >> shift) & 255][t], 1u);
            }
#pragma unroll
            for (int i = 0; i < ITEMS; i++) cur[i] = nxt[i];
        }
    } else {
        for (int t = 0; t < TPB; t++) {
            const u32 base = (tile0 + t) * TILE;
            if (base >= n) break;
            u64 kk[ITEMS];
#pragma unroll
            for (int i = 0; i < ITEMS; i++) kk[i] = keys[base + i * BLOCK + threadIdx.x];      // (n is a multiple of TILE here)
#pragma unroll
            for (int i = 0; i < ITEMS; i++) {
                if (V == 1) sink ^= kk[i];
                else if (V == 2) atomicAdd(&h[threadIdx.x & 255][t], 1u + (u32)(kk[i] & 1));
                else if (V == 4) atomicAdd(&h[wave * 256 + ((u32)(kk[i] >> shift) & 255)][t], 1u);
                else if (V == 6) atomicAdd(&h[(u32)kk[i] & 255][t], 1u);
                else atomicAdd(&h[(u32)(kk[i] >> shift) & 255][t], 1u);
            }
        }
    }
    __syncthreads();
    u32 *row = table + (u64)threadIdx.x * ntiles + tile0;
    for (int t = 0; t < TPB; t++)
        if (tile0 + t < ntiles) {
            u32 v = h[threadIdx.x][t];
            if (V == 4) v += h[256 + threadIdx.x][t] + h[512 + threadIdx.x][t] + h[768 + threadIdx.x][t];
            row[t] = v + (u32)sink;
        }
}

template <int V>
static void run(const char *name, const u64 *d_keys, u32 n, u32 *d_table, u32 ntiles) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const u32 blocks = (ntiles + TPB - 1) / TPB;
    hipLaunchKernelGGL(hist<V>, dim3(blocks), dim3(BLOCK), 0, 0, d_keys, n, 40, d_table, ntiles);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL(hist<V>, dim3(blocks), dim3(BLOCK), 0, 0, d_keys, n, 40, d_table, ntiles);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 10;
    printf("%-60s %7.4f ms  %7.1f GB/s of keys\n", name, ms, (double)n * 8 / ms * 1e-6);
    hipEventDestroy(a); hipEventDestroy(b);
}

__global__ void fill(u64 *keys, u32 n) {
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        u64 x = i * 0x9E3779B97F4A7C15ull + 0x1234567;
        x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29; x *= 0x94D049BB133111EBull; x ^= x >> 32;
        keys[i] = x;
    }
}

int main() {
    const u32 ntiles = 29296, n = ntiles * TILE;          // ~120 M keys, as config 2's read build
    u64 *d_keys; u32 *d_table;
    if (hipMalloc(&d_keys, (size_t)n * 8) != hipSuccess || hipMalloc(&d_table, (size_t)256 * ntiles * 4) != hipSuccess) return 1;
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, d_keys, n);
    hipDeviceSynchronize();
    run<0>("as the library (16 tiles per block, loads then adds)", d_keys, n, d_table, ntiles);
    run<1>("loads only", d_keys, n, d_table, ntiles);
    run<2>("adds on a conflict-free digit", d_keys, n, d_table, ntiles);
    run<3>("next tile's loads before this tile's adds", d_keys, n, d_table, ntiles);
    run<4>("a histogram per wave", d_keys, n, d_table, ntiles);
    run<6>("digit from the low bits", d_keys, n, d_table, ntiles);
    run<0>("as the library, again", d_keys, n, d_table, ntiles);
    {
        // the same launch right behind a kernel that has just written all the keys (as the extractor has in the library)
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        float tot = 0;
        for (int i = 0; i < 10; i++) {
            hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, d_keys, n);
            hipEventRecord(a);
            hipLaunchKernelGGL(hist<0>, dim3((ntiles + TPB - 1) / TPB), dim3(BLOCK), 0, 0, d_keys, n, 40, d_table, ntiles);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            tot += ms;
        }
        printf("%-60s %7.4f ms\n", "as the library, behind a kernel that wrote the keys", tot / 10);
        tot = 0;
        for (int i = 0; i < 10; i++) {
            hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, d_keys, n);
            hipDeviceSynchronize();
            hipEventRecord(a);
            hipLaunchKernelGGL(hist<0>, dim3((ntiles + TPB - 1) / TPB), dim3(BLOCK), 0, 0, d_keys, n, 40, d_table, ntiles);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            tot += ms;
        }
        printf("%-60s %7.4f ms\n", "the same with the device idle in between", tot / 10);
    }
    hipFree(d_keys); hipFree(d_table);
    return 0;
}

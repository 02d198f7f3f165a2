// Experiment: how fast can a kernel of the histogram's shape READ keys that are cold (not in the 256 MB Infinity Cache)?
// 4 buffers of 0.96 GB are read in turn, so that a launch finds nothing of its buffer cached; several ways of dealing the
// addresses out to blocks, waves and lanes.  (tools/exp/hist_probe.hip re-read ONE buffer and saw 4.7 TB/s where the
// library, reading keys the extractor has just written, sees 3.6.)
//   hipcc --offload-arch=gfx950 -O3 -o build/read_shape_probe tools/exp/read_shape_probe.hip && build/read_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32;
typedef unsigned long long u64;

// Every variant reads n keys (n a multiple of 65536) and writes one word per thread.
// SHAPE 0: block = 16 consecutive tiles of 4096 keys, one after the other; per tile a thread's 16 loads are 2 KB apart
//          (rs_hist_kernel).  1: the same with 16-byte loads.  2: tiles dealt out round robin (block b: tiles b, b + B, ...).
//       3: as 0, but a wave's loads are consecutive (a wave sweeps 8 KB).  4: as 2 with 16-byte loads
//       5: as 0 with the next tile's loads in flight while this one's are consumed (two tiles deep)
template <int SHAPE>
__global__ __launch_bounds__(256) void rd(const u64 *__restrict__ keys, u32 n, u32 *__restrict__ out) {
    const u32 ntiles = n / 4096;
    u64 sink = 0;
    const u32 per = 16;
    if (SHAPE == 0 || SHAPE == 3) {
        for (u32 t = 0; t < per; t++) {
            const u32 tile = blockIdx.x * per + t;
            if (tile >= ntiles) break;
            const u64 *src = keys + (u64)tile * 4096;
            u64 kk[16];
#pragma unroll
            for (int i = 0; i < 16; i++)
                kk[i] = SHAPE == 0 ? src[i * 256 + threadIdx.x] : src[(threadIdx.x >> 6) * 1024 + i * 64 + (threadIdx.x & 63)];
#pragma unroll
            for (int i = 0; i < 16; i++) sink ^= kk[i] >> (i + 1);
        }
    } else if (SHAPE == 1) {
        for (u32 t = 0; t < per; t++) {
            const u32 tile = blockIdx.x * per + t;
            if (tile >= ntiles) break;
            const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(keys + (u64)tile * 4096);
            ulonglong2 kk[8];
#pragma unroll
            for (int i = 0; i < 8; i++) kk[i] = src[i * 256 + threadIdx.x];
#pragma unroll
            for (int i = 0; i < 8; i++) sink ^= (kk[i].x >> (i + 1)) ^ (kk[i].y >> (i + 9));
        }
    } else if (SHAPE == 2 || SHAPE == 4) {
        for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            if (SHAPE == 2) {
                const u64 *src = keys + (u64)tile * 4096;
                u64 kk[16];
#pragma unroll
                for (int i = 0; i < 16; i++) kk[i] = src[i * 256 + threadIdx.x];
#pragma unroll
                for (int i = 0; i < 16; i++) sink ^= kk[i] >> (i + 1);
            } else {
                const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(keys + (u64)tile * 4096);
                ulonglong2 kk[8];
#pragma unroll
                for (int i = 0; i < 8; i++) kk[i] = src[i * 256 + threadIdx.x];
#pragma unroll
                for (int i = 0; i < 8; i++) sink ^= (kk[i].x >> (i + 1)) ^ (kk[i].y >> (i + 9));
            }
        }
    } else if (SHAPE == 5) {
        u64 cur[16], nxt[16];
        const u32 tile0 = blockIdx.x * per;
#pragma unroll
        for (int i = 0; i < 16; i++) cur[i] = tile0 < ntiles ? keys[(u64)tile0 * 4096 + i * 256 + threadIdx.x] : 0;
        for (u32 t = 0; t < per; t++) {
            const u32 tile = tile0 + t;
            if (tile >= ntiles) break;
            const bool more = t + 1 < per && tile + 1 < ntiles;
#pragma unroll
            for (int i = 0; i < 16; i++) nxt[i] = more ? keys[(u64)(tile + 1) * 4096 + i * 256 + threadIdx.x] : 0;
#pragma unroll
            for (int i = 0; i < 16; i++) sink ^= cur[i] >> (i + 1);
#pragma unroll
            for (int i = 0; i < 16; i++) cur[i] = nxt[i];
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (u32)sink ^ (u32)(sink >> 32);
}

__global__ void fill(u64 *keys, u64 n) {
    for (u64 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        u64 x = i * 0x9E3779B97F4A7C15ull + 0x1234567;
        x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29;
        keys[i] = x;
    }
}

template <int SHAPE>
static void run(const char *name, const u64 *d_keys, u32 n, u32 *d_out, u32 blocks) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float tot = 0;
    const int reps = 12;
    for (int i = 0; i < reps + 4; i++) {
        const u64 *src = d_keys + (u64)(i % 4) * n;
        hipEventRecord(a);
        hipLaunchKernelGGL(rd<SHAPE>, dim3(blocks), dim3(256), 0, 0, src, n, d_out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (i >= 4) tot += ms;
    }
    printf("%-72s %4u blocks %7.4f ms  %7.1f GB/s\n", name, blocks, tot / reps, (double)n * 8 / (tot / reps) * 1e-6);
    hipEventDestroy(a); hipEventDestroy(b);
}

int main() {
    const u32 ntiles = 29296, n = ntiles * 4096;
    u64 *d_keys; u32 *d_out;
    if (hipMalloc(&d_keys, (size_t)n * 8 * 4) != hipSuccess || hipMalloc(&d_out, (size_t)4096 * 256 * 4) != hipSuccess) return 1;
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, d_keys, (u64)n * 4);
    hipDeviceSynchronize();
    const u32 b16 = (ntiles + 15) / 16;
    run<0>("16 consecutive tiles per block, a thread's loads 2 KB apart (library)", d_keys, n, d_out, b16);
    run<1>("the same, 16-byte loads", d_keys, n, d_out, b16);
    run<3>("the same, a wave's loads consecutive", d_keys, n, d_out, b16);
    run<5>("the same, two tiles in flight", d_keys, n, d_out, b16);
    run<2>("tiles round robin over the blocks", d_keys, n, d_out, 2048);
    run<4>("tiles round robin, 16-byte loads", d_keys, n, d_out, 2048);
    run<2>("tiles round robin over the blocks", d_keys, n, d_out, 1024);
    run<2>("tiles round robin over the blocks", d_keys, n, d_out, 4096);
    run<4>("tiles round robin, 16-byte loads", d_keys, n, d_out, 4096);
    run<0>("16 consecutive tiles per block (library), again", d_keys, n, d_out, b16);
    hipFree(d_keys); hipFree(d_out);
    return 0;
}

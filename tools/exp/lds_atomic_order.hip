// Experiment: are same-address LDS atomics of one wave instruction applied in ascending lane order?
// (If so, atomicAdd's return value is a stable rank.)  Prints the number of violations over many patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(const unsigned *dig, unsigned *rank_out, int rounds, int packed) {
    __shared__ unsigned cnt[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    if (wave == 0) {
        for (int r = 0; r < rounds; r++) {
            const unsigned d = dig[(blockIdx.x * rounds + r) * 64 + threadIdx.x];
            unsigned old;
            if (packed) { old = atomicAdd(&cnt[d >> 1], (d & 1) ? 0x10000u : 1u); old = (d & 1) ? (old >> 16) : (old & 0xFFFFu); }
            else old = atomicAdd(&cnt[d], 1u);
            rank_out[(blockIdx.x * rounds + r) * 64 + threadIdx.x] = old;
        }
    }
}
int main() {
    const int blocks = 4096, rounds = 8;
    std::vector<unsigned> h(blocks * rounds * 64), out(h.size());
    srand(1);
    for (int b = 0; b < blocks; b++) {
        const int mode = b % 6;   // digit distributions: uniform 256, 16, 4, 2 values, all equal, stride patterns
        for (int i = 0; i < rounds * 64; i++) {
            unsigned d;
            switch (mode) {
                case 0: d = rand() & 255; break;
                case 1: d = rand() & 15; break;
                case 2: d = (rand() & 3) * 64; break;
                case 3: d = (rand() & 1) + 32 * (rand() & 1); break;
                case 4: d = 77; break;
                default: d = (i * 33 + (rand() & 1)) & 255; break;
            }
            h[(b * rounds) * 64 + i] = d;
        }
    }
    unsigned *dd, *dr;
    hipMalloc(&dd, h.size() * 4); hipMalloc(&dr, h.size() * 4);
    hipMemcpy(dd, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int packed = 0; packed < 2; packed++) {
        long bad = 0;
        for (int rep = 0; rep < 20; rep++) {
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, dd, dr, rounds, packed);
            hipMemcpy(out.data(), dr, h.size() * 4, hipMemcpyDeviceToHost);
            for (int b = 0; b < blocks; b++) {
                unsigned c[256] = {0};
                for (int i = 0; i < rounds * 64; i++) {
                    const unsigned d = h[b * rounds * 64 + i];
                    if (out[b * rounds * 64 + i] != c[d]) bad++;
                    c[d]++;
                }
            }
        }
        printf("packed=%d violations=%ld of %ld\n", packed, bad, 20L * (long)h.size());
    }
    return 0;
}

// Microbenchmark (diagnostics): latency of returning atomics / loads to random distinct addresses as a function of how many
// wavefronts issue them at once.  hipcc --offload-arch=gfx950 -O3 -o atomic_probe atomic_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// mode 0: plain load, 1: atomicAdd (returning) u32, 2: atomicCAS u64, 3: non-returning atomicAdd then a dependent load elsewhere
__global__ void probe(uint32_t *a, unsigned long long *b, uint32_t mask, int mode, int lanes, int reps, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t acc = 0, idx = mix(wave * 64u + lane + 1u);
    const uint32_t t0 = (uint32_t)wall_clock64();
    for (int r = 0; r < reps; ++r) {
        idx = mix(idx + acc);                       // dependent chain
        if ((int)lane < lanes) {
            if (mode == 0) acc += a[idx & mask];
            else if (mode == 1) acc += atomicAdd(&a[idx & mask], 1u);
            else if (mode == 2) acc += (uint32_t)atomicCAS(&b[idx & mask], 0xFFFFFFFFFFFFFFFFull, (unsigned long long)idx);
            else { atomicAdd(&a[idx & mask], 1u); acc += a[(idx >> 3) & mask]; }
        }
    }
    const uint32_t t1 = (uint32_t)wall_clock64();
    if (lane == 0) { out[2 * wave] = t1 - t0; out[2 * wave + 1] = acc; }
}
int main(int argc, char **argv)
{
    const uint32_t n = 1u << (argc > 1 ? atoi(argv[1]) : 24);   // 2^24: 64 MB of u32 / 128 MB of u64 (the working set decides the TLB reach)
    uint32_t *a, *out; unsigned long long *b;
    hipMalloc(&a, n * 4); hipMalloc(&b, (size_t)n * 8); hipMalloc(&out, 2 * 16384 * 4);
    hipMemset(a, 0, n * 4); hipMemset(b, 0xFF, (size_t)n * 8);
    std::vector<uint32_t> h(2 * 16384);
    const char *names[] = { "load", "atomicAdd ret", "CAS64 ret", "atomicAdd noret + load" };
    const bool quick = argc > 2;
    printf("working set: %.0f MB (u32), %.0f MB (u64)\n", n * 4.0 / 1048576, n * 8.0 / 1048576);
    for (int mode = 0; mode < 4; ++mode)
        for (int lanes : { 1, 4, 64 })
            for (int blocks : { 1, 64, 1024, 2048 }) {
                if (quick && !(lanes == 4 && blocks >= 1024)) continue;
                const int reps = 8;
                hipMemset(b, 0xFF, (size_t)n * 8);
                hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, a, b, n - 1, mode, lanes, reps, out);
                hipDeviceSynchronize();
                hipMemcpy(h.data(), out, 2 * 4 * blocks * 4, hipMemcpyDeviceToHost);
                std::vector<double> t;
                for (int w = 0; w < blocks * 4; ++w) t.push_back(h[2 * w] / 100.0 / reps);
                std::sort(t.begin(), t.end());
                printf("%-24s lanes %2d waves %5d: per op median %.2f us  p90 %.2f  max %.2f\n", names[mode], lanes, blocks * 4, t[t.size() / 2], t[t.size() * 9 / 10], t.back());
            }
    return 0;
}

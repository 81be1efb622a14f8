// Calibration of the L2's memory-side request counters (TCC_EA0_RDREQ / _32B / _64B / _128B, TCC_EA0_WRREQ / _64B, FETCH_SIZE,
// WRITE_SIZE) on access patterns with KNOWN byte counts, in the shapes the chunk pass uses (MI355X_MICROARCH.md, HBM: "calibrate
// on a known byte count in your own access pattern before trusting an absolute").  One kernel per pattern, each launched once
// per run so that a rocprofv3 --pmc pass attributes its counters to it by name:
//   cal_stream_read16   1 GiB read, 16 B per lane, coalesced          (the guide's reference pattern)
//   cal_stream_read4    1 GiB read, 4 B per lane, coalesced            (member words of a home-sorted list)
//   cal_gather4         2^26 random 4-byte loads from a 4 GiB table    (citizen words by index, hash keys: one line per load)
//   cal_gather32        2^24 random 32-byte records from a 4 GiB table (item records)
//   cal_stream_write16  1 GiB written, 16 B per lane
//   cal_scatter4        2^26 random 4-byte stores into a 4 GiB table
//   cal_atomic4         2^26 random 4-byte no-return atomic adds into a 4 GiB table
// hipcc --offload-arch=gfx950 -O3 -o tcc_calib tcc_calib.hip ; rocprofv3 --kernel-trace --pmc <counters> -- ./tcc_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void cal_stream_read16(const uint4 *a, size_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = a[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) *sink = acc;
}
__global__ void cal_stream_read4(const uint32_t *a, size_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += a[i];
    if (acc == 0x12345u) *sink = acc;
}
__global__ void cal_gather4(const uint32_t *a, uint32_t mask, uint32_t per_thread, uint32_t *sink)
{
    uint32_t acc = 0, idx = mix(blockIdx.x * blockDim.x + threadIdx.x + 1u);
    for (uint32_t r = 0; r < per_thread; ++r) { idx = mix(idx + r); acc += a[idx & mask]; }
    if (acc == 0x12345u) *sink = acc;
}
__global__ void cal_gather32(const uint4 *a, uint32_t mask, uint32_t per_thread, uint32_t *sink)
{
    uint32_t acc = 0, idx = mix(blockIdx.x * blockDim.x + threadIdx.x + 7u);
    for (uint32_t r = 0; r < per_thread; ++r) { idx = mix(idx + r); const size_t rec = (size_t)(idx & mask) * 2u; const uint4 v = a[rec], w = a[rec + 1]; acc += v.x ^ w.w; }
    if (acc == 0x12345u) *sink = acc;
}
__global__ void cal_stream_write16(uint4 *a, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
__global__ void cal_scatter4(uint32_t *a, uint32_t mask, uint32_t per_thread)
{
    uint32_t idx = mix(blockIdx.x * blockDim.x + threadIdx.x + 3u);
    for (uint32_t r = 0; r < per_thread; ++r) { idx = mix(idx + r); a[idx & mask] = idx; }
}
__global__ void cal_atomic4(uint32_t *a, uint32_t mask, uint32_t per_thread)
{
    uint32_t idx = mix(blockIdx.x * blockDim.x + threadIdx.x + 5u);
    for (uint32_t r = 0; r < per_thread; ++r) { idx = mix(idx + r); atomicAdd(&a[idx & mask], 1u); }
}
int main()
{
    const size_t table_bytes = 4ull << 30, stream_bytes = 1ull << 30;
    uint32_t *table = nullptr, *sink = nullptr;
    CHECK(hipMalloc(&table, table_bytes));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(table, 1, table_bytes));
    CHECK(hipDeviceSynchronize());
    const uint32_t mask4 = (uint32_t)(table_bytes / 4 - 1), mask32 = (uint32_t)(table_bytes / 32 - 1);
    const dim3 grid(4096), block(256);                     // 2^20 threads
    hipLaunchKernelGGL(cal_stream_read16, grid, block, 0, 0, (const uint4 *)table, stream_bytes / 16, sink);
    hipLaunchKernelGGL(cal_stream_read4, grid, block, 0, 0, table + (stream_bytes / 4), stream_bytes / 4, sink);
    hipLaunchKernelGGL(cal_gather4, grid, block, 0, 0, table, mask4, 64u, sink);                       // 2^26 loads
    hipLaunchKernelGGL(cal_gather32, grid, block, 0, 0, (const uint4 *)table, mask32, 16u, sink);      // 2^24 records
    hipLaunchKernelGGL(cal_stream_write16, grid, block, 0, 0, (uint4 *)table, stream_bytes / 16);
    hipLaunchKernelGGL(cal_scatter4, grid, block, 0, 0, table, mask4, 64u);
    hipLaunchKernelGGL(cal_atomic4, grid, block, 0, 0, table, mask4, 64u);
    CHECK(hipDeviceSynchronize());
    std::printf("known bytes: stream_read16 %zu, stream_read4 %zu, gather4 %zu useful (%u loads), gather32 %zu useful (%u records), stream_write16 %zu, scatter4 %zu useful, atomic4 %zu useful\n",
                stream_bytes, stream_bytes, (size_t)4 << 26, 1u << 26, (size_t)32 << 24, 1u << 24, stream_bytes, (size_t)4 << 26, (size_t)4 << 26);
    return 0;
}

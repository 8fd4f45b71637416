// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes of k_render
// (the counters are calibrated in the hardware guide for 16-byte-per-lane streams only):
//   k_dword_store   one dword per lane, a wavefront's 64 lanes contiguous (the shape of scratch spills)
//   k_dword_load    the same, loads (scratch reloads)
//   k_record_store  16 bytes per lane, lanes 1600 bytes apart (path records of an 8x8 tile at 100 spp)
//   k_vec4_store    16 bytes per lane, contiguous (the guide's calibrated case, as a check)
// Each kernel moves exactly BYTES bytes once; run under
//   rocprofv3 --pmc WRITE_SIZE -- ./microbench_traffic      and     rocprofv3 --pmc FETCH_SIZE -- ./microbench_traffic
// and compare the counters (KiB) per dispatch with the byte counts printed here.
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench_traffic.hip -o gpurun_out/microbench_traffic
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_dword_store(unsigned* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned)i;
}
__global__ void k_dword_load(const unsigned* p, size_t n, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345u) *sink = acc;
}
__global__ void k_record_store(float4* p, size_t records, size_t stride_records) {
    // record r of "pixel" q lives at q * stride + r: consecutive lanes are consecutive pixels, 1600 bytes apart
    const size_t pixels = records / stride_records;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < records; i += (size_t)gridDim.x * blockDim.x) {
        const size_t q = i % pixels, r = i / pixels;
        p[q * stride_records + r] = make_float4((float)i, 0.f, 0.f, 1.f);
    }
}
__global__ void k_vec4_store(float4* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4((float)i, 0.f, 0.f, 1.f);
}

int main() {
    const size_t BYTES = (size_t)1 << 30;                 // 1 GiB per kernel: well past L2 and the Infinity Cache
    void* buf; unsigned* sink;
    if (hipMalloc(&buf, BYTES) != hipSuccess || hipMalloc((void**)&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, BYTES);
    hipDeviceSynchronize();
    const int blocks = 256 * 8, threads = 256;
    k_dword_store<<<blocks, threads>>>((unsigned*)buf, BYTES / 4);
    hipDeviceSynchronize();
    k_dword_load<<<blocks, threads>>>((const unsigned*)buf, BYTES / 4, sink);
    hipDeviceSynchronize();
    k_record_store<<<blocks, threads>>>((float4*)buf, BYTES / 16, 100);
    hipDeviceSynchronize();
    k_vec4_store<<<blocks, threads>>>((float4*)buf, BYTES / 16);
    hipDeviceSynchronize();
    printf("each kernel moved %zu bytes = %zu KiB: k_dword_store (write), k_dword_load (read), k_record_store (write), k_vec4_store (write)\n",
           BYTES, BYTES / 1024);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

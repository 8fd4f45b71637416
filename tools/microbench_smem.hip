// Scalar-load (s_load) throughput microbenchmark for gfx950: every wave streams
// the same small read-only table through the scalar data cache, one record ahead
// of its use, the way the group loops of the render kernel do.
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench_smem.hip -o tools/microbench_smem
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));

template <int DWORDS>
__global__ void k(const int* table, int records, int stride_bytes, int iters, long long* cyc, int* out, int valu_per_load) {
    int acc = 0;
    float f = threadIdx.x;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const char* p = reinterpret_cast<const char*>(table);
        for (int j = 0; j < records; ++j) {
            if (DWORDS == 16) { v16i r; asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p)); acc += r[0]; }
            if (DWORDS == 8) { v8i r; asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p)); acc += r[0]; }
            if (DWORDS == 18) {   // the render loop's pattern: 16 + 2 dwords of one 80-byte record, one wait
                v16i r; long long q;
                asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx2 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=s"(r), "=s"(q) : "s"(p));
                acc += r[0] + (int)q;
            }
            if (DWORDS == 64) {   // four independent 64-byte loads in flight, one wait
                v16i r0, r1, r2, r3;
                asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %4, 0x40\n\ts_load_dwordx16 %2, %4, 0x80\n\ts_load_dwordx16 %3, %4, 0xc0\n\ts_waitcnt lgkmcnt(0)"
                             : "=s"(r0), "=s"(r1), "=s"(r2), "=s"(r3) : "s"(p));
                acc += r0[0] + r1[0] + r2[0] + r3[0];
            }
            if (DWORDS == 4) { v4i r; asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p)); acc += r[0]; }
            for (int v = 0; v < valu_per_load; ++v) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f));
            p += stride_bytes;
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + (int)f;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int DWORDS> void run(int blocks, int waves_per_simd, int records, int stride, int valu) {
    int *table, *out; long long* cyc;
    hipMalloc(&table, 1 << 20); hipMemset(table, 0, 1 << 20); hipMalloc(&out, 8 << 20); hipMalloc(&cyc, 1 << 20);
    const int iters = 200;
    const int tpb = 64 * 4 * waves_per_simd > 1024 ? 1024 : 64 * 4 * waves_per_simd;
    const int bpc = 64 * 4 * waves_per_simd / tpb;
    k<DWORDS><<<blocks * bpc, tpb>>>(table, records, stride, 2, cyc, out, valu);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<DWORDS><<<blocks * bpc, tpb>>>(table, records, stride, iters, cyc, out, valu); hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double loads_per_wave = (double)iters * records;
    const double waves_per_cu = 4.0 * waves_per_simd;
    const double cyc_per_load_per_cu = ms * 1e-3 * 2.4e9 / (loads_per_wave * waves_per_cu);
    printf("x%-2d blocks %3d waves/simd %d records %4d stride %3d valu/load %2d: %.3f ms, %.2f cyc@2.4GHz per load per CU -> %.1f B/clk/CU\n",
           DWORDS, blocks, waves_per_simd, records, stride, valu, ms, cyc_per_load_per_cu, DWORDS * 4 / cyc_per_load_per_cu);
    hipFree(table); hipFree(out); hipFree(cyc);
}
int main() {
    for (int blocks : {1, 2, 32, 256}) {
        run<16>(blocks, 8, 322, 80, 0);
        run<8>(blocks, 8, 322, 80, 0);
        run<4>(blocks, 8, 322, 80, 0);
    }
    run<18>(256, 8, 322, 80, 0);
    run<18>(256, 8, 322, 80, 40);
    run<64>(256, 8, 80, 256, 0);
    run<64>(256, 4, 80, 256, 0);
    run<64>(1, 8, 80, 256, 0);
    run<16>(256, 8, 322, 64, 0);
    run<16>(256, 8, 322, 80, 30);
    run<16>(256, 8, 322, 80, 60);
    run<16>(256, 8, 322, 80, 100);
    run<16>(256, 4, 322, 80, 0);
    run<16>(256, 2, 322, 80, 0);
    run<16>(256, 8, 3000, 80, 0);
    return 0;
}

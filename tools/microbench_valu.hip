// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction on one
// SIMD for plain and packed FP32 ops at 1..8 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench_valu.hip -o gpurun_out/microbench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP 64
template <int MODE>
__global__ void k(float* out, long long* cyc, int iters, float seed) {
    float a[8]; v2f p[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = (v2f){a[i], a[i] + 1.f}; }
    const float m = 1.0001f, c = 0.5f; const v2f pm = {m, m}, pc = {c, c};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (MODE == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
                if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
                if (MODE == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if (MODE == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "s"(m));
                if (MODE == 6) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (MODE == 7) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a[i]), "v"(m) : "vcc");
                if (MODE == 8) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(m));
                if (MODE == 9) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(m), "v"(c));
                if (MODE == 10) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (MODE == 11) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "s"(m));
                if (MODE == 12) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
                if (MODE == 13) asm volatile("v_mul_f32_dpp %0, %1, %0 row_ror:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(m));
                if (MODE == 14) asm volatile("v_mov_b32_dpp %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(m));
                if (MODE == 15) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "s"(m));
                if (MODE == 16) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (MODE == 17) asm volatile("v_mul_f32 %0, 0x3f800054, %0" : "+v"(a[i]));
                if (MODE == 18) asm volatile("v_mul_f32 %0, 2.0, %0" : "+v"(a[i]));
                if (MODE == 19) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "s"(pm));
                if (MODE == 20) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "s"(pm), "v"(pc));
                if (MODE == 21) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p[i]) : "v"(pm));
                if (MODE == 22) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "s"(pm));
                if (MODE == 23) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(p[i]) : "v"(pc), "s"(pm));
                if (MODE == 24) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "+v"(p[i]) : "v"(pc), "s"(pm));
                if (MODE == 26) asm volatile("v_pk_mul_f32 %0, %0, %1\n\ts_mov_b32 s20, s21" : "+v"(p[i]) : "s"(pm) : "s20");
                if (MODE == 27) asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_mov_b32 s20, s21" : "+v"(a[i]) : "v"(m), "v"(c) : "s20");
                if (MODE == 28) asm volatile("v_pk_mul_f32 %0, %0, %1\n\ts_mov_b32 s20, s21\n\ts_mov_b32 s22, s23" : "+v"(p[i]) : "s"(pm) : "s20", "s22");
                if (MODE == 29) asm volatile("s_mov_b32 s20, s21\n\ts_mov_b32 s22, s23" : : : "s20", "s22");
                if (MODE == 30) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p[i]) : "v"(p[(i + 3) & 7]), "s"(pm));
                if (MODE == 31) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p[i]) : "v"(p[(i + 7) & 7]), "s"(pm));
                if (MODE == 32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (MODE == 33) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (MODE == 34) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "+v"(p[i]) : "v"(m), "v"(c) : "vcc");
                if (MODE == 35) asm volatile("v_add_f64 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
                if (MODE == 36) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
                if (MODE == 37) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (MODE == 38) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
                if (MODE == 39) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[i]) : "v"(m));
                if (MODE == 40) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
                if (MODE == 41) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(m));
                if (MODE == 42) asm volatile("v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(a[i]) : : "vcc");
                if (MODE == 43) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
                if (MODE == 44) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(m) : "s20", "s21");
                if (MODE == 45) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (MODE == 46) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (MODE == 47) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (MODE == 48) asm volatile("v_cmp_gt_i32 vcc, 0, %0" :: "v"(a[i]) : "vcc");
                if (MODE == 49) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (MODE == 50) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (MODE == 51) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(m));
                if (MODE == 52) asm volatile("v_bfe_u32 %0, %0, 3, 7" : "+v"(a[i]));
                if (MODE == 53) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]));
                if (MODE == 54) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
                if (MODE == 55) asm volatile("v_lshrrev_b32 %0, 11, %0" : "+v"(a[i]));
                if (MODE == 56) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (MODE == 57) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (MODE == 58) asm volatile("v_fma_f32 %0, %0, %1, -%2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (MODE == 59) asm volatile("v_fma_f32 %0, -%0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (MODE == 60) asm volatile("v_mul_f32_e64 %0, -%0, %1" : "+v"(a[i]) : "v"(m));
                if (MODE == 61) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(p[i]) : "v"(a[i]));
                if (MODE == 62) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(p[i]));
                if (MODE == 63) asm volatile("v_rcp_f64 %0, %0" : "+v"(p[i]));
                if (MODE == 64) asm volatile("v_sqrt_f64 %0, %0" : "+v"(p[i]));
                if (MODE == 65) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (MODE == 66) asm volatile("v_readlane_b32 s20, %0, 5" :: "v"(a[i]) : "s20");
                if (MODE == 25) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(p[i]) : "v"(pm));
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
template <int MODE> void run(const char* name) {
    float* out; long long* cyc; hipMalloc(&out, 4 << 20); hipMalloc(&cyc, 1 << 20);
    const int iters = 100000;
    printf("%-14s", name);
    for (int waves_per_simd : {1, 2, 4, 8}) {
        int threads = 64 * 4 * waves_per_simd;   // one block per CU, 4 SIMDs
        int tpb = threads > 1024 ? 1024 : threads; int bpc = threads / tpb;
        k<MODE><<<256 * bpc, tpb>>>(out, cyc, 10, 1.0f);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); k<MODE><<<256 * bpc, tpb>>>(out, cyc, iters, 1.0f); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(256 * bpc * tpb / 64); hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= h.size();
        double inst = (double)iters * REP;
        // cycles of wave lifetime per instruction / waves per simd = SIMD cycles per instruction
        printf("  w%d: %.2f ns-cyc@2.4GHz/inst/SIMD (memtime %.2f)", waves_per_simd, ms * 1e-3 * 2.4e9 / (inst * waves_per_simd), avg / inst / waves_per_simd);
    }
    printf("\n");
}
int main() {
    run<0>("v_fma_f32"); run<1>("v_mul_f32"); run<2>("v_pk_fma_f32"); run<3>("v_pk_mul_f32");
    run<4>("v_rcp_f32"); run<5>("v_add_f32 sgpr"); run<6>("v_min3_f32"); run<7>("v_cmp_lt_f32");
    run<8>("v_mul_f32 sgpr"); run<9>("v_fma_f32 sgpr"); run<10>("v_add_f32 vgpr"); run<11>("v_sub_f32 sgpr"); run<12>("v_fma 2src");
    run<13>("v_mul dpp ror3"); run<14>("v_mov dpp ror1"); run<15>("v_mov sgpr"); run<16>("v_max_f32"); run<17>("v_mul literal"); run<18>("v_mul inline2.0");
    run<19>("pk_mul sgpr"); run<20>("pk_fma sgpr"); run<21>("pk_mul bcast"); run<22>("pk_add sgpr");
    run<23>("pk_mul s+bcast"); run<24>("pk_fma s+b+neg"); run<25>("pk_add neg");
    run<26>("pk_mul+1smov"); run<27>("fma+1smov"); run<28>("pk_mul+2smov"); run<29>("2 smov only");
    run<30>("pk_mul dist3"); run<31>("pk_mul dep chain");
    run<32>("v_mul_lo_u32"); run<33>("v_mul_hi_u32"); run<34>("v_mad_u64_u32"); run<35>("v_add_f64"); run<36>("v_fma_f64");
    run<37>("v_xor_b32"); run<38>("v_mul_f64"); run<39>("v_alignbit"); run<40>("v_cndmask");
    run<41>("cndmask sgpr"); run<42>("v_addc vcc"); run<43>("cmp+cndmask vcc"); run<44>("cmp+cndmask sgpr");
    run<45>("v_or3_b32"); run<46>("v_and_or_b32"); run<47>("v_or_b32"); run<48>("v_cmp_gt_i32"); run<49>("v_max3_f32");
    run<50>("v_add3_u32"); run<51>("v_lshl_add_u32"); run<52>("v_bfe_u32"); run<53>("v_cvt_f32_u32"); run<54>("v_sqrt_f32");
    run<55>("v_lshrrev_b32"); run<56>("v_add_u32"); run<57>("v_sub_f32 vgpr"); run<58>("v_fma neg c"); run<59>("v_fma neg a");
    run<60>("v_mul e64 neg"); run<61>("v_cvt_f64_f32"); run<62>("v_cvt_f32_f64"); run<63>("v_rcp_f64"); run<64>("v_sqrt_f64");
    run<65>("v_min_f32"); run<66>("v_readlane");
    run<0>("v_fma_f32 again");
    return 0;
}

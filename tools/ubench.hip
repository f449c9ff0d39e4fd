// ubench.hip -- VALU issue-rate microbenchmark for gfx950.
// Measures cycles per wave64 instruction per SIMD for the instructions the
// 256-bit modular multiplier can be built from, so that DESIGN.md can state a
// measured integer-multiply roofline next to the HBM one.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e = (x);                                                           \
        if (e != hipSuccess) {                                                        \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

constexpr int ITERS = 16384;
constexpr int UNROLL = 8;  // independent chains per lane

#define BODY8(stmt) stmt(0) stmt(1) stmt(2) stmt(3) stmt(4) stmt(5) stmt(6) stmt(7)

template <int OP>
__global__ __launch_bounds__(256) void bench(unsigned* out, unsigned seed, unsigned long long* cyc) {
    unsigned long long t0 = __builtin_readcyclecounter();
    unsigned a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
    unsigned long long acc[UNROLL];
    double facc[UNROLL];
    for (int k = 0; k < UNROLL; k++) {
        acc[k] = (unsigned long long)a * (k + 3) + b;
        facc[k] = (double)(a & 0xfffff) + k;
    }
    unsigned long long mask = __ballot((a & 1) != 0);
    double fa = (double)(a & 0xffff) + 1.0, fb = (double)(b & 0xffff) + 3.0;
    for (int it = 0; it < ITERS; it++) {
        if (OP == 0) {
#define S(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b) : "vcc");
            BODY8(S)
#undef S
        } else if (OP == 1) {
#define S(k) { unsigned lo = (unsigned)acc[k]; asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(b)); acc[k] = lo; }
            BODY8(S)
#undef S
        } else if (OP == 2) {
#define S(k) { unsigned lo = (unsigned)acc[k]; asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo) : "v"(b)); acc[k] = lo; }
            BODY8(S)
#undef S
        } else if (OP == 3) {
#define S(k) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(facc[k]) : "v"(fa), "v"(fb));
            BODY8(S)
#undef S
        } else if (OP == 4) {
#define S(k) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[k]) : "v"(acc[(k + 1) & 7]));
            BODY8(S)
#undef S
        } else if (OP == 5) {
#define S(k) { unsigned lo = (unsigned)acc[k]; asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(lo) : "v"(b) : "vcc"); acc[k] = lo; }
            BODY8(S)
#undef S
        } else if (OP == 6) {
#define S(k) { unsigned lo = (unsigned)acc[k]; asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(lo) : "v"(b), "v"(a)); acc[k] = lo; }
            BODY8(S)
#undef S
        } else if (OP == 7) {
#define S(k) { unsigned lo = (unsigned)acc[k]; asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(b)); acc[k] = lo; }
            BODY8(S)
#undef S
        } else if (OP == 8) {
#define S(k) { unsigned lo = (unsigned)acc[k]; asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(lo) : "v"(b) : "vcc"); acc[k] = lo; }
            BODY8(S)
#undef S
        } else if (OP == 9) {
#define S(k) { unsigned lo = (unsigned)acc[k]; asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(lo) : "v"(b)); acc[k] = lo; }
            BODY8(S)
#undef S
        } else if (OP == 10) {
#define S(k) { unsigned lo = (unsigned)acc[k]; asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(lo) : "v"(a), "v"(b)); acc[k] = lo; }
            BODY8(S)
#undef S
        } else if (OP == 11) {
#define S(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(facc[k]) : "v"(fa));
            BODY8(S)
#undef S
        } else if (OP == 12) {
#define S(k) { unsigned lo = (unsigned)acc[k]; asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo) : "v"(b) : "vcc"); acc[k] = lo; }
            BODY8(S)
#undef S
        } else if (OP == 13) {
#define S(k) { unsigned lo = (unsigned)acc[k]; asm volatile("v_mov_b32 %0, %1" : "+v"(lo) : "v"(b)); acc[k] = lo; }
            BODY8(S)
#undef S
        } else if (OP == 14) {
#define S(k) { unsigned lo = (unsigned)acc[k]; asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(lo) : "v"(a), "v"(b)); acc[k] = lo; }
            BODY8(S)
#undef S
        } else if (OP == 15) {
#define S(k) { unsigned lo = (unsigned)acc[k]; asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(lo) : "v"(a), "v"(b)); acc[k] = lo; }
            BODY8(S)
#undef S
        } else if (OP == 16) {  // the cross-lane moves a wavefront-cooperative multiplier would be made of
#define S(k) { unsigned lo = (unsigned)acc[k]; asm volatile("v_mov_b32_dpp %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(lo)); acc[k] = lo; }
            BODY8(S)
#undef S
        } else if (OP == 17) {  // broadcast of one lane through an SGPR into a multiply-add (v_readfirstlane-style m_i broadcast)
#define S(k) { unsigned s; asm volatile("v_readlane_b32 %0, %2, 0\n\tv_mad_u64_u32 %1, vcc, %0, %3, %1" : "=&s"(s), "+v"(acc[k]) : "v"((unsigned)acc[(k + 1) & 7]), "v"(b) : "vcc"); }
            BODY8(S)
#undef S
        } else if (OP == 20) {  // the carry extraction of the 29-bit-limb multiplier as one 64-bit shift ...
#define S(k) asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(acc[k]));
            BODY8(S)
#undef S
        } else if (OP == 21) {  // ... or as its 32-bit halves
#define S(k) { unsigned lo = (unsigned)acc[k], hi = (unsigned)(acc[k] >> 32); asm volatile("v_alignbit_b32 %0, %1, %0, 29" : "+v"(lo) : "v"(hi)); acc[k] = lo | ((unsigned long long)hi << 32); }
            BODY8(S)
#undef S
        } else if (OP == 22) {
#define S(k) { unsigned lo = (unsigned)acc[k]; asm volatile("v_lshrrev_b32 %0, 29, %0" : "+v"(lo)); acc[k] = lo; }
            BODY8(S)
#undef S
        } else if (OP == 23) {
#define S(k) { unsigned lo = (unsigned)acc[k]; asm volatile("v_and_b32 %0, 0x1fffffff, %0" : "+v"(lo)); acc[k] = lo; }
            BODY8(S)
#undef S
        } else if (OP == 24) {
#define S(k) { unsigned lo = (unsigned)acc[k]; asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(lo) : "v"(a), "v"(b)); acc[k] = lo; }
            BODY8(S)
#undef S
        } else if (OP == 25) {
#define S(k) { unsigned lo = (unsigned)acc[k]; asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(lo) : "v"(a), "v"(b)); acc[k] = lo; }
            BODY8(S)
#undef S
        } else if (OP == 26) {  // compare + select through VCC, as a conditional subtraction compiles
#define S(k) { unsigned lo = (unsigned)acc[k]; asm volatile("v_cmp_gt_u32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(lo) : "v"(a), "v"(b) : "vcc"); acc[k] = lo; }
            BODY8(S)
#undef S
        } else if (OP == 27) {  // select through an SGPR pair (VOP3)
#define S(k) { unsigned lo = (unsigned)acc[k]; asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(lo) : "v"(b), "s"(mask)); acc[k] = lo; }
            BODY8(S)
#undef S
        } else if (OP == 28) {  // compare alone
#define S(k) { unsigned lo = (unsigned)acc[k]; asm volatile("v_cmp_gt_u32 vcc, %1, %0" : : "v"(lo), "v"(a) : "vcc"); acc[k] = lo; }
            BODY8(S)
#undef S
        } else if (OP == 18) {  // ONE dependent chain (no ILP): what a latency-bound lane actually sees per v_mad_u64_u32
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                         "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0"
                         : "+v"(acc[0]) : "v"(a), "v"(b) : "vcc");
        } else if (OP == 19) {  // one dependent chain of plain VALU
            unsigned lo = (unsigned)acc[0];
            asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1"
                         : "+v"(lo) : "v"(b));
            acc[0] = lo;
        }
    }
    unsigned long long r = 0;
    for (int k = 0; k < UNROLL; k++) r ^= acc[k] ^ (unsigned long long)__double_as_longlong(facc[k]);
    if (r == 0x1234567887654321ull) out[0] = (unsigned)r;  // keep results live
    unsigned long long t1 = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int OP>
void run(const char* name, unsigned* d_out, int cus, double ghz, int waves_per_simd = 4) {
    static unsigned long long* d_cyc = nullptr;
    if (!d_cyc) CHECK(hipMalloc(&d_cyc, 8));
    int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    bench<OP><<<blocks, 256>>>(d_out, 1, d_cyc);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(e0));
        bench<OP><<<blocks, 256>>>(d_out, rep, d_cyc);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    double wave_instr_per_simd = (double)ITERS * UNROLL * waves_per_simd;
    double cycles = best * 1e-3 * ghz * 1e9 / wave_instr_per_simd;
    double lane_ops_per_s = (double)blocks * 256 * ITERS * UNROLL / (best * 1e-3);
    unsigned long long hc = 0;
    CHECK(hipMemcpy(&hc, d_cyc, 8, hipMemcpyDeviceToHost));
    double cyc_ctr = (double)hc / wave_instr_per_simd;
    printf("%-18s w/SIMD=%d %8.3f ms  %6.2f cyc/instr (wall@%.2fGHz)  %6.2f cyc/instr (s_memtime)  %8.2f Tlane-op/s\n", name,
           waves_per_simd, best, cycles, ghz, cyc_ctr, lane_ops_per_s * 1e-12);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    double ghz = prop.clockRate * 1e-6;
    printf("device %s  CUs %d  clock %.2f GHz\n", prop.name, prop.multiProcessorCount, ghz);
    unsigned* d_out;
    CHECK(hipMalloc(&d_out, 64));
    int cus = prop.multiProcessorCount;
    run<7>("v_add_u32", d_out, cus, ghz, 1);
    run<0>("v_mad_u64_u32", d_out, cus, ghz, 1);
    run<0>("v_mad_u64_u32", d_out, cus, ghz, 2);
    run<7>("v_add_u32", d_out, cus, ghz);
    run<13>("v_mov_b32", d_out, cus, ghz);
    run<5>("v_add_co_u32", d_out, cus, ghz);
    run<8>("v_addc_co_u32", d_out, cus, ghz);
    run<12>("v_cndmask_b32", d_out, cus, ghz);
    run<4>("v_lshl_add_u64", d_out, cus, ghz);
    run<0>("v_mad_u64_u32", d_out, cus, ghz);
    run<1>("v_mul_lo_u32", d_out, cus, ghz);
    run<2>("v_mul_hi_u32", d_out, cus, ghz);
    run<6>("v_mad_u32_u24", d_out, cus, ghz);
    run<9>("v_mul_hi_u32_u24", d_out, cus, ghz);
    run<15>("v_mad_u32_u16", d_out, cus, ghz);
    run<10>("v_dot4_u32_u8", d_out, cus, ghz);
    run<14>("v_dot2_u32_u16", d_out, cus, ghz);
    run<20>("v_lshrrev_b64", d_out, cus, ghz);
    run<21>("v_alignbit_b32", d_out, cus, ghz);
    run<22>("v_lshrrev_b32", d_out, cus, ghz);
    run<23>("v_and_b32", d_out, cus, ghz);
    run<24>("v_add3_u32", d_out, cus, ghz);
    run<25>("v_bfi_b32", d_out, cus, ghz);
    run<26>("v_cmp+v_cndmask", d_out, cus, ghz);
    run<27>("v_cndmask sgpr", d_out, cus, ghz);
    run<28>("v_cmp_gt_u32", d_out, cus, ghz);
    run<3>("v_fma_f64", d_out, cus, ghz);
    run<11>("v_add_f64", d_out, cus, ghz);
    // one wavefront per SIMD: the issue costs a latency-bound (tree-top, small-tree) hash sees
    run<7>("v_add_u32", d_out, cus, ghz, 1);
    run<19>("v_add_u32 1 chain", d_out, cus, ghz, 1);
    run<0>("v_mad_u64_u32", d_out, cus, ghz, 1);
    run<18>("v_mad_u64 1 chain", d_out, cus, ghz, 1);
    run<16>("v_mov_dpp row_shl", d_out, cus, ghz, 1);
    run<17>("readlane+mad pair", d_out, cus, ghz, 1);
    return 0;
}

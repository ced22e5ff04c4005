// Issue-rate microbenchmark for the instructions of the box-filter cascade (gfx950).
// hipcc --offload-arch=gfx950 -O3 -o op_rates op_rates.hip && ./op_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_IT 4096
template <int OP>
__global__ void k(double* out, float seed, int waves_note) {
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    float f0 = seed, f1 = seed + 1, f2 = seed + 2, f3 = seed + 3, f4 = seed + 4, f5 = seed + 5, f6 = seed + 6, f7 = seed + 7;
    unsigned s0 = (unsigned)waves_note, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3, s4 = s0 + 4, s5 = s0 + 5, s6 = s0 + 6, s7 = s0 + 7;
    unsigned long long m0 = (unsigned long long)waves_note * 0x9E3779B97F4A7C15ull, m1 = ~m0, m2 = m0 >> 7, m3 = m0 << 9, m4 = m0 ^ 0x5555555555555555ull, m5 = ~m4;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N_IT; i++) {
        if (OP == 0) {  // v_add_f64, 8 independent chains
            asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                         "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(1.5));
        } else if (OP == 1) {  // v_cvt_f64_f32
            asm volatile("v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %9\n v_cvt_f64_f32 %2, %10\n v_cvt_f64_f32 %3, %11\n"
                         "v_cvt_f64_f32 %4, %12\n v_cvt_f64_f32 %5, %13\n v_cvt_f64_f32 %6, %14\n v_cvt_f64_f32 %7, %15\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7));
        } else if (OP == 2) {  // v_cvt_f32_f64
            asm volatile("v_cvt_f32_f64 %0, %8\n v_cvt_f32_f64 %1, %9\n v_cvt_f32_f64 %2, %10\n v_cvt_f32_f64 %3, %11\n"
                         "v_cvt_f32_f64 %4, %12\n v_cvt_f32_f64 %5, %13\n v_cvt_f32_f64 %6, %14\n v_cvt_f32_f64 %7, %15\n"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)
                         : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
        } else if (OP == 3) {  // v_add_f32
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(1.5f));
        } else if (OP == 4) {  // dependent v_add_f64 chain (latency)
            asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                         "v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                         : "+v"(a0) : "v"(1.5));
        } else if (OP == 5) {  // the cascade's dependent chain: add, cvt to f32, cvt back, add (one stage per step)
            asm volatile("v_add_f64 %0, %0, %2\n v_cvt_f32_f64 %1, %0\n v_add_f64 %0, %0, %3\n v_cvt_f64_f32 %2, %1\n"
                         "v_add_f64 %0, %0, %2\n v_cvt_f32_f64 %1, %0\n v_add_f64 %0, %0, %3\n v_cvt_f64_f32 %2, %1\n"
                         : "+v"(a0), "+v"(f0), "+v"(a1) : "v"(a2));
        } else if (OP == 6) {  // v_fma_f32 (division helper class)
            asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                         "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(0.5f));
        } else if (OP == 7) {  // v_rcp_f32 + v_div_scale + v_div_fmas + v_div_fixup mix as in one IEEE division (approx: 2 scale, rcp, 6 fma, fmas, fixup)
            asm volatile("v_div_scale_f32 %0, vcc, %1, %2, %1\n v_rcp_f32 %3, %0\n v_fma_f32 %4, %0, %3, %3\n v_fma_f32 %4, %4, %3, %3\n"
                         "v_fma_f32 %4, %4, %3, %3\n v_fma_f32 %4, %4, %3, %3\n v_div_fmas_f32 %4, %4, %3, %0\n v_div_fixup_f32 %5, %4, %2, %1\n"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5) : : "vcc");
        } else if (OP == 8) {  // v_mov_b32 from / to AGPR
            asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_read_b32 %1, a0\n v_accvgpr_write_b32 a1, %2\n v_accvgpr_read_b32 %3, a1\n"
                         "v_accvgpr_write_b32 a2, %4\n v_accvgpr_read_b32 %5, a2\n v_accvgpr_write_b32 a3, %6\n v_accvgpr_read_b32 %7, a3\n"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : : "a0", "a1", "a2", "a3");
        } else if (OP == 9) {  // scalar ALU, 8 independent 32-bit adds (round 4: is the scalar unit per SIMD or per CU?)
            asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                         "s_add_u32 %4, %4, 1\n s_add_u32 %5, %5, 1\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1\n"
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : : "scc");
        } else if (OP == 10) {  // 64-bit lane-mask algebra (the SumThreshold kernel's flag rings): s_or_b64 / s_and_b64
            asm volatile("s_or_b64 %0, %0, %4\n s_and_b64 %1, %1, %4\n s_or_b64 %2, %2, %4\n s_and_b64 %3, %3, %4\n"
                         "s_or_b64 %0, %0, %5\n s_and_b64 %1, %1, %5\n s_or_b64 %2, %2, %5\n s_and_b64 %3, %3, %5\n"
                         : "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3) : "s"(m4), "s"(m5) : "scc");
        } else if (OP == 11) {  // half vector, half scalar: v_add_f32 x4 + s_or_b64 x4 (do the two units overlap?)
            asm volatile("v_add_f32 %0, %0, %8\n s_or_b64 %4, %4, %9\n v_add_f32 %1, %1, %8\n s_and_b64 %5, %5, %9\n"
                         "v_add_f32 %2, %2, %8\n s_or_b64 %6, %6, %9\n v_add_f32 %3, %3, %8\n s_and_b64 %7, %7, %9\n"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3) : "v"(1.5f), "s"(m4) : "scc");
        } else if (OP == 12) {  // compare into a lane mask + mask algebra + mask back into a select: v_cmp_gt_f32, s_and_b64, v_cndmask x 2 + 2 more mask ops
            asm volatile("v_cmp_gt_f32 %4, %0, %8\n s_and_b64 %4, %4, %5\n v_cndmask_b32 %1, %1, %0, %4\n s_or_b64 %6, %6, %4\n"
                         "v_cmp_lt_f32 %7, %2, %8\n s_and_b64 %7, %7, %5\n v_cndmask_b32 %3, %3, %2, %7\n s_or_b64 %6, %6, %7\n"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3) : "v"(1.5f) : "scc");
        } else if (OP == 13) {  // eight independent float64 compares into lane masks (throughput of VALU -> SGPR results)
            asm volatile("v_cmp_gt_f64 %0, %4, %5\n v_cmp_lt_f64 %1, %4, %5\n v_cmp_gt_f64 %2, %5, %4\n v_cmp_lt_f64 %3, %5, %4\n"
                         "v_cmp_gt_f64 %0, %4, %6\n v_cmp_lt_f64 %1, %4, %6\n v_cmp_gt_f64 %2, %6, %4\n v_cmp_lt_f64 %3, %6, %4\n"
                         : "=s"(m0), "=s"(m1), "=s"(m2), "=s"(m3) : "v"(a0), "v"(a1), "v"(a2));
        } else if (OP == 14) {  // eight selects by lane masks held in scalar registers (throughput of SGPR-mask operands)
            asm volatile("v_cndmask_b32 %0, %0, %4, %8\n v_cndmask_b32 %1, %1, %4, %9\n v_cndmask_b32 %2, %2, %4, %10\n v_cndmask_b32 %3, %3, %4, %11\n"
                         "v_cndmask_b32 %0, %0, %5, %9\n v_cndmask_b32 %1, %1, %5, %10\n v_cndmask_b32 %2, %2, %5, %11\n v_cndmask_b32 %3, %3, %5, %8\n"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(f4), "v"(f5), "v"(f6), "v"(f7), "s"(m0), "s"(m1), "s"(m2), "s"(m3));
        } else if (OP == 15) {  // float32 compares into lane masks
            asm volatile("v_cmp_gt_f32 %0, %4, %5\n v_cmp_lt_f32 %1, %4, %5\n v_cmp_gt_f32 %2, %5, %4\n v_cmp_lt_f32 %3, %5, %4\n"
                         "v_cmp_gt_f32 %0, %4, %6\n v_cmp_lt_f32 %1, %4, %6\n v_cmp_gt_f32 %2, %6, %4\n v_cmp_lt_f32 %3, %6, %4\n"
                         : "=s"(m0), "=s"(m1), "=s"(m2), "=s"(m3) : "v"(f0), "v"(f1), "v"(f2));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if ((s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7) == 0x7fffffffu && (m0 ^ m1 ^ m2 ^ m3) == 0x123456789ull) out[3] = 1.0;
    double s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    if (threadIdx.x == 0) out[blockIdx.x * 2] = (double)(t1 - t0) / (N_IT * 8.0);
    if (s == 12345.678) out[1] = s;
}
template <int OP>
void run(const char* name) {
    double* d;
    hipMalloc(&d, 4096 * sizeof(double));
    // one block per CU of 1 ... 16 waves, then (round 4) two 16-wave blocks per CU: 8 waves per SIMD -- what one WAVE gets
    // shows the issue cost, what a full SIMD gets shows the pipe's throughput (cycles per instruction per SIMD = per wave / waves)
    for (int cfg = 0; cfg < 6; cfg++) {
        const int wpb = cfg == 0 ? 64 : cfg == 1 ? 128 : cfg == 2 ? 256 : cfg == 3 ? 512 : 1024;
        const int grid = cfg == 5 ? 512 : 256;
        k<OP><<<grid, wpb>>>(d, 1.0f, wpb);
        hipDeviceSynchronize();
        k<OP><<<grid, wpb>>>(d, 1.0f, wpb);
        hipDeviceSynchronize();
        std::vector<double> h(1024);
        hipMemcpy(h.data(), d, 1024 * sizeof(double), hipMemcpyDeviceToHost);
        double m = 0;
        for (int b = 0; b < grid; b++) m += h[2 * b];
        const double wps = (double)wpb / 256.0 * (grid / 256);
        printf("%-28s block %4d threads x %d per CU (%.2f wave/SIMD): %.2f cycles per instruction per wave, %.2f per SIMD\n", name, wpb, grid / 256,
               wps, m / grid, wps >= 1 ? m / grid / wps : m / grid);
    }
    hipFree(d);
}
int main() {
    run<0>("v_add_f64 x8 indep");
    run<1>("v_cvt_f64_f32 x8");
    run<2>("v_cvt_f32_f64 x8");
    run<3>("v_add_f32 x8");
    run<4>("v_add_f64 dependent");
    run<5>("cascade chain add,cvt,add,cvt");
    run<6>("v_fma_f32 x8");
    run<7>("ieee div mix (8 ops)");
    run<8>("accvgpr write/read");
    run<9>("s_add_u32 x8 indep");
    run<10>("s_or_b64 / s_and_b64 x8");
    run<11>("v_add_f32 x4 + s_or/and_b64 x4");
    run<12>("v_cmp -> s_and -> v_cndmask -> s_or x2");
    run<13>("v_cmp_f64 -> lane mask x8 indep");
    run<14>("v_cndmask by scalar masks x8");
    run<15>("v_cmp_f32 -> lane mask x8 indep");
    return 0;
}

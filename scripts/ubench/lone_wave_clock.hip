// Does a kernel that occupies one or two CUs run at the full shader clock?  One wave spins on dependent adds;
// shader cycles (s_memtime) against wall time (HIP events) and the 100 MHz reference counter (s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out, long long* cyc, int iters) {
    double a = 1.0;
    long long t0 = __builtin_amdgcn_s_memtime();
    long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                     "v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n" : "+v"(a) : "v"(1.5));
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = r1 - r0; out[blockIdx.x] = a; }
}
int main() {
    double* d; long long* c;
    hipMalloc(&d, 8 * 4096); hipMalloc(&c, 16 * 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {1, 2, 64, 1024, 4096}) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            k<<<blocks, 64>>>(d, c, 2000000);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[2]; hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
            printf("%4d blocks of one wave: %.2f ms wall, %lld shader cycles, %lld x 10 ns -> %.0f MHz shader clock, %.2f cycles per instruction\n",
                   blocks, ms, h[0], h[1], (double)h[0] / ((double)h[1] * 0.01), (double)h[0] / (2000000.0 * 8));
        }
    }
    return 0;
}

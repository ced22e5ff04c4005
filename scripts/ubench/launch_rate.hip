// How much work must one thread of a streaming byte kernel carry before the
// kernel stops being bound by the rate at which waves can be launched?
// Copies / ORs a 1 GiB byte image with 16-byte accesses, ITERS accesses per
// thread (stride = one block's worth, so every access stays coalesced).
//   hipcc --offload-arch=gfx950 -O3 -o launch_rate.bin launch_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int ITERS, int OP>
__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n16) {
    size_t base = (size_t)blockIdx.x * (256 * ITERS) + threadIdx.x;
    uint4 v[ITERS];
#pragma unroll
    for (int j = 0; j < ITERS; j++) {
        size_t i = base + (size_t)j * 256;
        v[j] = i < n16 ? a[i] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < ITERS; j++) {
        size_t i = base + (size_t)j * 256;
        if (i < n16) {
            if (OP == 1) { uint4 o = b[i]; v[j].x |= o.x; v[j].y |= o.y; v[j].z |= o.z; v[j].w |= o.w; }
            b[i] = v[j];
        }
    }
}

template <int ITERS, int OP>
void run(const uint4* a, uint4* b, size_t n16) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned blocks = (unsigned)((n16 + 256 * ITERS - 1) / (256 * ITERS));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k_copy<ITERS, OP>), dim3(blocks), dim3(256), 0, 0, a, b, n16);
    hipEventRecord(e0);
    const int reps = 10;
    for (int w = 0; w < reps; w++) hipLaunchKernelGGL((k_copy<ITERS, OP>), dim3(blocks), dim3(256), 0, 0, a, b, n16);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    double bytes = (double)n16 * 16 * (OP == 1 ? 3 : 2);
    printf("%s  16 B x %d per thread: %8.3f ms  %6.2f TB/s  %7.2f waves/ns\n", OP ? "b |= a" : "b  = a", ITERS, ms,
           bytes / ms / 1e9, (double)blocks * 4 / ms / 1e6);
}

int main() {
    size_t n = (size_t)1 << 30, n16 = n / 16;
    uint4 *a, *b;
    hipMalloc(&a, n); hipMalloc(&b, n);
    hipMemset(a, 1, n); hipMemset(b, 0, n);
    run<1, 0>(a, b, n16); run<2, 0>(a, b, n16); run<4, 0>(a, b, n16); run<8, 0>(a, b, n16); run<16, 0>(a, b, n16);
    run<1, 1>(a, b, n16); run<2, 1>(a, b, n16); run<4, 1>(a, b, n16); run<8, 1>(a, b, n16);
    return 0;
}

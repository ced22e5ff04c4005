// Development harness of K4w (kernels_boxweight.hpp): checks k_boxw<2r> against a plain one-thread-per-column cascade on
// random TF4 flag words and times it in the bench slab's launch geometry.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
//         -mllvm -amdgpu-sched-strategy=max-ilp scripts/ubench/boxw_dev.hip -o scripts/ubench/boxw_dev.bin
//   scripts/ubench/boxw_dev.bin [windows]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
struct BoxDenom { float b, y; };
#include "../../tricolour_amd/csrc/kernels_boxweight.hpp"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static BoxDenom recip(int r) {
    volatile float a = (float)(2 * r + 1); volatile float a2 = a * a; volatile float a4 = a2 * a2;
    float b = a4, y = (float)(1.0 / (double)b);
    const float cand[3] = {std::nextafterf(y, 0.0f), y, std::nextafterf(y, INFINITY)};
    double best = INFINITY;
    for (float c : cand) { double e = std::fabs(1.0 - (double)b * (double)c); if (e < best) { best = e; y = c; } }
    return BoxDenom{b, y};
}

// reference: the cascade with its masks, one thread per column, delay lines in global scratch [4][2r][ncol]
__global__ void k_ref(const uint8_t* flags, float* out, unsigned* scratch, int n, int C, int r, float denom, size_t sws, size_t dws, size_t scr_ws) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const size_t win = blockIdx.y;
    const uint8_t* f = flags + win * sws;
    unsigned* d = scratch + win * scr_ws;
    const int R2 = 2 * r;
    for (int k = 0; k < 4 * R2; k++) d[(size_t)k * C + c] = 0;
    unsigned s1 = 0, s2 = 0, s3 = 0, s4 = 0;
    for (int t = 0; t < n + 2 * R2; t++) {
        const int slot = t % R2;
        unsigned in1 = 0;
        if (t < n) in1 = f[((size_t)(t >> 2) * C + c) * 4 + (t & 3)] == 0 ? 1u : 0u;
        unsigned* d1 = d + ((size_t)(0 * R2 + slot)) * C + c;
        unsigned* d2 = d + ((size_t)(1 * R2 + slot)) * C + c;
        unsigned* d3 = d + ((size_t)(2 * R2 + slot)) * C + c;
        unsigned* d4 = d + ((size_t)(3 * R2 + slot)) * C + c;
        s1 += in1; const unsigned o1 = s1; s1 -= *d1; *d1 = in1;
        const unsigned in2 = t < n + R2 ? o1 : 0u;
        s2 += in2; const unsigned o2 = s2; s2 -= *d2; *d2 = in2;
        s3 += o2; const unsigned o3 = s3; s3 -= *d3; *d3 = o2;
        const unsigned in4 = t >= R2 ? o3 : 0u;
        s4 += in4; const unsigned o4 = s4; s4 -= *d4; *d4 = in4;
        const int i = t - 2 * R2;
        if (i >= 0 && i < n) out[win * dws + (size_t)i * C + c] = (float)o4 / denom;
    }
}

template <int R2>
static void run(int W, int n, int C, bool timing) {
    const int r = R2 / 2;
    const size_t N = (size_t)n * C;
    uint8_t* flags; float *a, *b; unsigned* scr;
    CK(hipMalloc(&flags, (size_t)W * N));
    CK(hipMalloc(&a, (size_t)W * N * 4));
    std::vector<uint8_t> h((size_t)W * N);
    unsigned x = 12345u + R2;
    for (size_t k = 0; k < h.size(); k++) {
        x = x * 1664525u + 1013904223u;
        const unsigned v = x >> 8;
        h[k] = (v % 100 < 7) ? (uint8_t)(1 + (v >> 8) % 255) : 0;
    }
    // a fully flagged stretch wider than the filter and a clean one
    for (int w = 0; w < W; w++)
        for (int t = n / 3; t < n / 3 + 5 * r && t < n; t++)
            for (int c = 0; c < C; c += 3) h[(size_t)w * N + ((size_t)(t >> 2) * C + c) * 4 + (t & 3)] = 1;
    CK(hipMemcpy(flags, h.data(), h.size(), hipMemcpyHostToDevice));
    CK(hipMemset(a, 0xff, (size_t)W * N * 4));
    const BoxDenom dn = recip(r);
    dim3 grid((C + 63) / 64, W);
    hipLaunchKernelGGL(k_boxw<R2>, grid, dim3(64), 0, 0, flags, a, n, C, dn, N, N);
    CK(hipDeviceSynchronize());
    if (!timing) {
        CK(hipMalloc(&b, (size_t)W * N * 4));
        CK(hipMalloc(&scr, (size_t)W * 4 * R2 * C * 4));
        CK(hipMemset(b, 0xee, (size_t)W * N * 4));
        hipLaunchKernelGGL(k_ref, dim3((C + 63) / 64, W), dim3(64), 0, 0, flags, b, scr, n, C, r, dn.b, N, N, (size_t)4 * R2 * C);
        CK(hipDeviceSynchronize());
        std::vector<uint32_t> ha((size_t)W * N), hb((size_t)W * N);
        CK(hipMemcpy(ha.data(), a, ha.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hb.data(), b, hb.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0, first = 0;
        for (size_t k = 0; k < ha.size(); k++) if (ha[k] != hb[k]) { if (!bad) first = k; bad++; }
        printf("2r=%3d  n=%d C=%d W=%d  mismatches %zu of %zu", R2, n, C, W, bad, ha.size());
        if (bad) printf("  first at %zu: %08x vs %08x (t=%zu c=%zu)", first, ha[first], hb[first], (first % N) / C, first % C);
        printf("\n");
        CK(hipFree(b)); CK(hipFree(scr));
    } else {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int reps = 5;
        CK(hipEventRecord(e0, 0));
        for (int k = 0; k < reps; k++) hipLaunchKernelGGL(k_boxw<R2>, grid, dim3(64), 0, 0, flags, a, n, C, dn, N, N);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        hipFuncAttributes at;
        CK(hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_boxw<R2>)));
        printf("2r=%3d  %d windows of %d x %d: %.3f ms per launch, %.2f TB/s at 5 B/sample, %d registers\n", R2, W, n, C, ms,
               (double)W * N * 5 / (ms * 1e-3) / 1e12, at.numRegs);
    }
    CK(hipFree(flags)); CK(hipFree(a));
}

int main(int argc, char** argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 1008;
    // correctness: ragged column counts, short lines, lines shorter than the filter
#define CHECK(R2) run<R2>(2, 256, 130, false); run<R2>(1, 64, 70, false); run<R2>(1, 1024, 64, false);
    CHECK(16) CHECK(20) CHECK(22) CHECK(32) CHECK(42) CHECK(44) CHECK(56) CHECK(64) CHECK(86) CHECK(108)
#define TIME(R2) run<R2>(W, 1024, 4096, true);
    TIME(16) TIME(20) TIME(22) TIME(32) TIME(42) TIME(44) TIME(56) TIME(64) TIME(86) TIME(108)
    return 0;
}

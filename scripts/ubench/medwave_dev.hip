// Development harness of the wave medians (kernels_median.hpp k_median_wave): SPW segments per wave with the next segment's loads in flight
// against one segment per wave on the two shapes of the stage-1 step -- rows of F channels in chunks of ~F/10 (KS = 8) and whole lines of T samples (KS = 16).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
//         -mllvm -amdgpu-sched-strategy=max-ilp scripts/ubench/medwave_dev.hip -o scripts/ubench/medwave_dev.bin
//   scripts/ubench/medwave_dev.bin [windows] [rows] [row_len] [segments]
#include "../../tricolour_amd/csrc/tri_common.hpp"
#include "../../tricolour_amd/csrc/kernels_elementwise.hpp"
#include "../../tricolour_amd/csrc/kernels_median.hpp"
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_fill(float* r, uint8_t* f, uint8_t* f2, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)(i * 2654435761u) ^ seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    unsigned y = x * 747796405u + 2891336453u; y ^= y >> 13; y *= 0x5bd1e995u; y ^= y >> 15;
    const float u1 = ((x >> 8) + 1) * (1.0f / 16777217.0f), u2 = (y >> 8) * (1.0f / 16777216.0f);
    float v = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
    if ((y & 1023u) == 7u) v *= 20.0f;
    r[i] = v;
    f[i] = ((x & 31u) == 3u) ? 1 : 0;
    f2[i] = ((x & 127u) == 9u) ? 1 : 0;
}

template <int KS, int SPW>
static float run(const float* d, const uint8_t* f, const uint8_t* f2, double* med, int W, int R, int L, int G, const int64_t* st, const int64_t* ln, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < reps; rep++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_median_wave<KS, true, SPW>), dim3((unsigned)(((int64_t)((R + SPW - 1) / SPW) * G + 3) / 4), (unsigned)W), dim3(256), 0, 0,
                           d, f, med, (size_t)R * L, (size_t)R * L, (size_t)L, (size_t)1, st, ln, R, G, f2, (const uint8_t*)nullptr, (size_t)0, 0);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
    }
    return best;
}

int main(int argc, char** argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 252, R = argc > 2 ? atoi(argv[2]) : 1024, L = argc > 3 ? atoi(argv[3]) : 4096, G = argc > 4 ? atoi(argv[4]) : 10;
    const size_t N = (size_t)W * R * L;
    float* d; uint8_t *f, *f2; double *ma, *mb; int64_t *st, *ln;
    CK(hipMalloc(&d, N * 4)); CK(hipMalloc(&f, N)); CK(hipMalloc(&f2, N));
    CK(hipMalloc(&ma, (size_t)W * R * G * 8)); CK(hipMalloc(&mb, (size_t)W * R * G * 8));
    std::vector<int64_t> hs(G), hl(G);
    for (int g = 0; g < G; g++) { int64_t a = (int64_t)((double)g * L / G), b = g + 1 == G ? L : (int64_t)((double)(g + 1) * L / G); hs[g] = a; hl[g] = b - a; }
    CK(hipMalloc(&st, G * 8)); CK(hipMalloc(&ln, G * 8));
    CK(hipMemcpy(st, hs.data(), G * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(ln, hl.data(), G * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, 0, d, f, f2, N, 777u);
    CK(hipDeviceSynchronize());
    int64_t maxlen = 0; for (int g = 0; g < G; g++) maxlen = hl[g] > maxlen ? hl[g] : maxlen;
    float t[5];
    size_t bad = 0;
    std::vector<double> ha((size_t)W * R * G), hb((size_t)W * R * G);
    auto cmp = [&] {
        CK(hipMemcpy(hb.data(), mb, hb.size() * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < ha.size(); i++) bad += memcmp(&ha[i], &hb[i], 8) != 0;
        CK(hipMemset(mb, 0xEE, hb.size() * 8));
    };
    if (maxlen + 3 <= 512) {
        t[0] = run<8, 1>(d, f, f2, ma, W, R, L, G, st, ln, 4); CK(hipMemcpy(ha.data(), ma, ha.size() * 8, hipMemcpyDeviceToHost));
        t[1] = run<8, 2>(d, f, f2, mb, W, R, L, G, st, ln, 4); cmp();
        t[2] = run<8, 4>(d, f, f2, mb, W, R, L, G, st, ln, 4); cmp();
        t[3] = run<8, 8>(d, f, f2, mb, W, R, L, G, st, ln, 4); cmp();
        t[4] = run<8, 16>(d, f, f2, mb, W, R, L, G, st, ln, 4); cmp();
    } else {
        t[0] = run<16, 1>(d, f, f2, ma, W, R, L, G, st, ln, 4); CK(hipMemcpy(ha.data(), ma, ha.size() * 8, hipMemcpyDeviceToHost));
        t[1] = run<16, 2>(d, f, f2, mb, W, R, L, G, st, ln, 4); cmp();
        t[2] = run<16, 4>(d, f, f2, mb, W, R, L, G, st, ln, 4); cmp();
        t[3] = run<16, 8>(d, f, f2, mb, W, R, L, G, st, ln, 4); cmp();
        t[4] = run<16, 16>(d, f, f2, mb, W, R, L, G, st, ln, 4); cmp();
    }
    printf("W=%d R=%d L=%d G=%d (maxlen %lld): rows per wave 1 / 2 / 4 / 8 / 16: %.3f %.3f %.3f %.3f %.3f ms  (%.0f -> %.0f Gsample/s at 4); differing medians %zu\n", W, R, L, G,
           (long long)maxlen, t[0], t[1], t[2], t[3], t[4], (double)N / t[0] / 1e6, (double)N / t[2] / 1e6, bad);
    return bad != 0;
}

// Development harness of K3r (kernels_reject.hpp): the fused block median + rejection against the two kernels it replaces
// (k_median2 + k_reject4_t) on synthetic |data - background| images: flags (FT bytes and TF4 words) and medians must be identical.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
//         -mllvm -amdgpu-sched-strategy=max-ilp scripts/ubench/medrej_dev.hip -o scripts/ubench/medrej_dev.bin
//   scripts/ubench/medrej_dev.bin [windows] [T] [F] [chunks]
#include "../../tricolour_amd/csrc/tri_common.hpp"
#include "../../tricolour_amd/csrc/kernels_elementwise.hpp"
#include "../../tricolour_amd/csrc/kernels_median.hpp"
#include "../../tricolour_amd/csrc/kernels_reject.hpp"
#include "../../tricolour_amd/csrc/kernels_reject_tile.hpp"
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_fill(float* r, uint8_t* f, size_t n, unsigned seed, int T) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // cheap hash -> two uniforms -> |gaussian| (Box-Muller), occasional outliers, NaN stripes under flags
    unsigned x = (unsigned)(i * 2654435761u) ^ seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    unsigned y = x * 747796405u + 2891336453u; y ^= y >> 13; y *= 0x5bd1e995u; y ^= y >> 15;
    const float u1 = ((x >> 8) + 1) * (1.0f / 16777217.0f), u2 = (y >> 8) * (1.0f / 16777216.0f);
    float v = fabsf(sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2));
    if ((y & 1023u) == 7u) v *= 20.0f;
    const size_t row = i / T;
    uint8_t fl = ((x & 31u) == 3u) ? 1 : 0;
    if (row % 97 == 5) { fl = 1; v = NAN; }
    r[i] = v;
    f[i] = fl;
}

int main(int argc, char** argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 64, T = argc > 2 ? atoi(argv[2]) : 1024, F = argc > 3 ? atoi(argv[3]) : 4096, G = argc > 4 ? atoi(argv[4]) : 10;
    const size_t N = (size_t)T * F;
    float* resid; uint8_t *fin, *fa, *fb, *ta, *tb; double *ma, *mb; unsigned* scratch; int64_t *ends, *sst, *sln; int* chunk_of;
    CK(hipMalloc(&resid, W * N * 4)); CK(hipMalloc(&fin, W * N)); CK(hipMalloc(&fa, W * N)); CK(hipMalloc(&fb, W * N));
    CK(hipMalloc(&ta, W * N)); CK(hipMalloc(&tb, W * N)); CK(hipMalloc(&ma, W * G * 8)); CK(hipMalloc(&mb, W * G * 8));
    const size_t wsS = 2 * N;                                    // scratch words per window
    CK(hipMalloc(&scratch, (size_t)W * wsS * 4));
    std::vector<int64_t> he(G + 1), hs(G), hl(G);
    std::vector<int> hc(F);
    for (int g = 0; g <= G; g++) he[g] = (int64_t)((double)g * ((double)F / G));
    he[G] = F;
    for (int g = 0; g < G; g++) { hs[g] = he[g] * T; hl[g] = (he[g + 1] - he[g]) * T; for (int64_t c = he[g]; c < he[g + 1]; c++) hc[c] = g; }
    CK(hipMalloc(&ends, (G + 1) * 8)); CK(hipMalloc(&sst, G * 8)); CK(hipMalloc(&sln, G * 8)); CK(hipMalloc(&chunk_of, F * 4));
    CK(hipMemcpy(ends, he.data(), (G + 1) * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(sst, hs.data(), G * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(sln, hl.data(), G * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(chunk_of, hc.data(), F * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((W * N + 255) / 256)), dim3(256), 0, 0, resid, fin, W * N, 12345u, T);
    CK(hipDeviceSynchronize());
    const double scale = 1.4826 * 2.0;
    const size_t per_block = (wsS / G) & ~(size_t)7;
    const unsigned capq = (unsigned)(per_block / 2);
    hipEvent_t e0, e1, e2, e3, e4, e5, e6, e7;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3)); CK(hipEventCreate(&e4));
    CK(hipEventCreate(&e5)); CK(hipEventCreate(&e6)); CK(hipEventCreate(&e7));
    uint8_t *fc, *tc; double* mc;
    CK(hipMalloc(&fc, W * N)); CK(hipMalloc(&tc, W * N)); CK(hipMalloc(&mc, W * G * 8));
    for (int rep = 0; rep < 2; rep++) {
        // reference: median, then rejection in place on a copy of the flags
        CK(hipMemcpy(fa, fin, W * N, hipMemcpyDeviceToDevice));
        CK(hipMemset(ta, 0xEE, W * N)); CK(hipMemset(tb, 0xDD, W * N)); CK(hipMemset(fb, 0xCC, W * N));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_median2<true>, dim3(G, W), dim3(256), 0, 0, resid, fa, ma, N, N, (size_t)0, (size_t)1, sst, sln, 1, G, scratch, wsS, (unsigned)(per_block & ~(size_t)3));
        hipLaunchKernelGGL(k_reject4_t, dim3((T / 4 + 63) / 64, (F + 63) / 64, W), dim3(64, 4), 0, 0, resid, fa, ta, ma, chunk_of, scale, F, T / 4, G, N, N);
        CK(hipEventRecord(e1, 0));
        hipLaunchKernelGGL(k_median_reject, dim3(G, W), dim3(256), 0, 0, resid, fin, fb, tb, mb, ends, scale, F, T / 4, G, N, N, scratch, wsS, capq, capq, 0);
        CK(hipEventRecord(e2, 0));
        // tile-parallel form: predict, pass, finish, redo of failed blocks
        CK(hipMemset(tc, 0xBB, W * N)); CK(hipMemset(fc, 0xAA, W * N));
        int ytiles = 0;
        for (int g = 0; g < G; g++) ytiles += (int)((he[g + 1] - he[g] + 63) / 64);
        const size_t bs = (size_t)((he[1] - he[0]) * T);
        const size_t ccap = (bs / 4) & ~(size_t)3, ucap = (bs / 8) & ~(size_t)3;
        if (mrt_scratch_words(G, ccap, ucap) > wsS) { printf("scratch too small\n"); return 1; }
        CK(hipEventRecord(e3, 0));
        hipLaunchKernelGGL(k_mr_predict, dim3(G, W), dim3(256), 0, 0, resid, fin, ends, scale, T / 4, G, N, N, scratch, wsS, ccap, ucap, 0);
        CK(hipEventRecord(e4, 0));
        hipLaunchKernelGGL(k_mr_pass, dim3((T / 4 + 63) / 64, ytiles, W), dim3(256), 0, 0, resid, fin, fc, tc, ends, F, T / 4, G, N, N, scratch, wsS, ccap, ucap, 1u);
        CK(hipEventRecord(e5, 0));
        hipLaunchKernelGGL(k_mr_finish, dim3(G, W), dim3(256), 0, 0, resid, fin, fc, tc, mc, ends, scale, F, T / 4, G, N, N, scratch, wsS, ccap, ucap, 1u);
        hipLaunchKernelGGL(k_mr_pass, dim3((T / 4 + 63) / 64, ytiles, W), dim3(256), 0, 0, resid, fin, fc, tc, ends, F, T / 4, G, N, N, scratch, wsS, ccap, ucap, 2u);
        hipLaunchKernelGGL(k_mr_finish, dim3(G, W), dim3(256), 0, 0, resid, fin, fc, tc, mc, ends, scale, F, T / 4, G, N, N, scratch, wsS, ccap, ucap, 2u);
        CK(hipEventRecord(e6, 0));
        hipLaunchKernelGGL(k_median_reject, dim3(G, W), dim3(256), 0, 0, resid, fin, fc, tc, mc, ends, scale, F, T / 4, G, N, N, scratch, wsS, 0u, 0u, 1,
                           (const unsigned*)scratch, wsS, MRT_PARW, 11);
        CK(hipEventRecord(e7, 0));
        CK(hipEventSynchronize(e7));
        float t1, t2, t3, t4, t5, t6;
        CK(hipEventElapsedTime(&t1, e0, e1)); CK(hipEventElapsedTime(&t2, e1, e2));
        CK(hipEventElapsedTime(&t3, e3, e4)); CK(hipEventElapsedTime(&t4, e4, e5)); CK(hipEventElapsedTime(&t5, e5, e6)); CK(hipEventElapsedTime(&t6, e6, e7));
        if (rep == 1) printf("%d windows of %d x %d, %d chunks: k_median2 + k_reject4_t %.3f ms, k_median_reject %.3f ms, tile form %.3f ms = predict %.3f + pass %.3f (%.2f TB/s at 7 B/sample) + finish and round 2 %.3f + redo %.3f\n",
                             W, T, F, G, t1, t2, t3 + t4 + t5 + t6, t3, t4, (double)W * N * 7 / (t4 * 1e-3) / 1e12, t5, t6);
    }
    std::vector<uint8_t> a(W * N), b(W * N);
    size_t badf = 0, badt = 0, badm = 0, nset = 0;
    CK(hipMemcpy(a.data(), fa, W * N, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), fb, W * N, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < a.size(); i++) { badf += a[i] != b[i]; nset += a[i]; }
    CK(hipMemcpy(a.data(), ta, W * N, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), tb, W * N, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < a.size(); i++) badt += a[i] != b[i];
    std::vector<double> m1(W * G), m2(W * G);
    CK(hipMemcpy(m1.data(), ma, W * G * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(m2.data(), mb, W * G * 8, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < m1.size(); i++) badm += memcmp(&m1[i], &m2[i], 8) != 0;
    {
        size_t bf = 0, bt = 0, bm = 0;
        CK(hipMemcpy(a.data(), fa, W * N, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), fc, W * N, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < a.size(); i++) bf += a[i] != b[i];
        CK(hipMemcpy(a.data(), ta, W * N, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), tc, W * N, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < a.size(); i++) bt += a[i] != b[i];
        CK(hipMemcpy(m2.data(), mc, W * G * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < m1.size(); i++) bm += memcmp(&m1[i], &m2[i], 8) != 0;
        std::vector<unsigned> par((size_t)G * MRT_PARW);
        size_t redo = 0, nc = 0, nu = 0, brk = 0; int why[16] = {0};
        for (int w = 0; w < W; w++) {
            CK(hipMemcpy(par.data(), scratch + (size_t)w * wsS, par.size() * 4, hipMemcpyDeviceToHost));
            for (int g = 0; g < G; g++) { if (par[g * MRT_PARW + 11] == 0) { why[par[g * MRT_PARW + 13] & 15]++; const unsigned* q = &par[g * MRT_PARW];
                printf("  redo w %d g %d: lo %08x S %u wlo %u whi %u ncand %u nund %u below1 %08x rounds %u why %u\n", w, g, q[0], q[1], q[2], q[3], q[8], q[9], q[10], q[12], q[13]); } redo += par[g * MRT_PARW + 11] == 0; brk += par[g * MRT_PARW + 12] & 0xFFFFu; nc += par[g * MRT_PARW + 8]; nu += par[g * MRT_PARW + 9]; }
        }
        printf("tile form: FT flags differing %zu, TF4 %zu, medians %zu; blocks redone by one workgroup %zu, second rounds %zu of %d, window keys %.2f %%, undecided %.2f %% of the samples\n",
               bf, bt, bm, redo, brk, W * G, 100.0 * nc / (W * (double)N), 100.0 * nu / (W * (double)N));
        printf("redo reasons (1 nothing to predict from, 2 window-key list overflow, 10 undecided list overflow, 3 empty, 4 end bin, 5 global list, 6 window miss in round 2, 7 bracket miss in round 2):");
        for (int k = 0; k < 16; k++) if (why[k]) printf(" %d:%d", k, why[k]);
        printf("\n");
        badf += bf; badt += bt; badm += bm;
    }
    unsigned long long st[4];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_medrej_stats), sizeof(st)));
    printf("FT flags differing %zu, TF4 flags differing %zu, medians differing %zu (flags set %zu of %zu); blocks %llu fallbacks %llu / %llu / %llu\n",
           badf, badt, badm, nset, a.size(), st[0], st[1], st[2], st[3]);
    return (badf || badt || badm) ? 1 : 0;
}

// What HBM delivers for the access shapes a column kernel can use on a [W][L][C] float image + [W][L][C] byte image
// (L = 1024 sequential rows, C = 4096 contiguous columns): per-lane dword row loads + byte stores (the SumThreshold
// kernel's shape) against wider shapes.  hipcc --offload-arch=gfx950 -O3 stream_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define L 1024
#define C 4096
template <int MODE, int PFD>
__global__ void __launch_bounds__(256) k(const float* __restrict__ in, uint8_t* __restrict__ out) {
    const size_t win = blockIdx.y;
    const float* p = in + win * (size_t)L * C;
    uint8_t* o = out + win * (size_t)L * C;
    if (MODE <= 2 || MODE >= 4) {
        // lane = column; PFD rows in flight
        const int c = blockIdx.x * 256 + threadIdx.x;
        float acc = 0;
        for (int t0 = 0; t0 < L; t0 += PFD) {
            float v[PFD];
            if (MODE != 1) {
#pragma unroll
                for (int u = 0; u < PFD; u++) v[u] = (MODE == 5 || MODE == 4) ? __builtin_nontemporal_load(&p[(size_t)(t0 + u) * C + c]) : p[(size_t)(t0 + u) * C + c];
            } else {
#pragma unroll
                for (int u = 0; u < PFD; u++) v[u] = (float)(t0 + u);
            }
#pragma unroll
            for (int u = 0; u < PFD; u++) {
                acc += v[u];
                if (MODE == 4) __builtin_nontemporal_store((uint8_t)(acc > 3.0f), &o[(size_t)(t0 + u) * C + c]);
                else if (MODE != 0) o[(size_t)(t0 + u) * C + c] = (uint8_t)(acc > 3.0f);
            }
        }
        if (MODE == 0 && acc == 12345.f) o[c] = 1;
    } else {
        // a wave covers 64 columns x 4 rows per float4 load (lanes 0-15 row t, 16-31 row t+1, ...): 1 KB per instruction;
        // flags leave as one dword (4 columns) per lane: 4 row segments of 64 bytes per instruction
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int c4 = blockIdx.x * 256 + wave * 64 + (lane & 15) * 4;
        const int rsub = lane >> 4;
        float acc = 0;
        for (int t0 = 0; t0 < L; t0 += 4 * PFD) {
            float4 v[PFD];
#pragma unroll
            for (int u = 0; u < PFD; u++) v[u] = *reinterpret_cast<const float4*>(p + (size_t)(t0 + 4 * u + rsub) * C + c4);
#pragma unroll
            for (int u = 0; u < PFD; u++) {
                acc += v[u].x + v[u].y + v[u].z + v[u].w;
                uchar4 f = make_uchar4(v[u].x > 3.f, v[u].y > 3.f, v[u].z > 3.f, acc > 3.f);
                *reinterpret_cast<uchar4*>(o + (size_t)(t0 + 4 * u + rsub) * C + c4) = f;
            }
        }
    }
}
template <int MODE, int PFD>
void run(const char* name, const float* in, uint8_t* out, int W, double bytes_per_sample) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(C / 256, W);
    k<MODE, PFD><<<grid, 256>>>(in, out);
    hipEventRecord(e0);
    for (int i = 0; i < 5; i++) k<MODE, PFD><<<grid, 256>>>(in, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-58s %.3f ms  %.2f TB/s\n", name, ms, (double)W * L * C * bytes_per_sample / ms / 1e9);
}
int main() {
    const int W = 1008;
    float* in; uint8_t* out;
    hipMalloc(&in, (size_t)W * L * C * 4);
    hipMalloc(&out, (size_t)W * L * C);
    hipMemset(in, 0, (size_t)W * L * C * 4);
    run<0, 16>("dword row loads only, 16 rows in flight (4 B)", in, out, W, 4);
    run<0, 32>("dword row loads only, 32 rows in flight (4 B)", in, out, W, 4);
    run<1, 16>("byte row stores only (1 B)", in, out, W, 1);
    run<2, 16>("dword loads + byte stores, 16 in flight (5 B)", in, out, W, 5);
    run<2, 32>("dword loads + byte stores, 32 in flight (5 B)", in, out, W, 5);
    run<4, 16>("nontemporal dword loads + nontemporal byte stores (5 B)", in, out, W, 5);
    run<5, 16>("nontemporal dword loads + plain byte stores (5 B)", in, out, W, 5);
    run<3, 4>("float4 loads (4 rows x 64 cols) + dword stores, 4 in flight (5 B)", in, out, W, 5);
    run<3, 8>("float4 loads (4 rows x 64 cols) + dword stores, 8 in flight (5 B)", in, out, W, 5);
    return 0;
}

// Is the 4.3 TB/s "ceiling" of a 4 B-read + 1 B-write stream (profiles/r02_stream_shapes.txt) a property of the byte
// mix or of the access order?    Same images as stream_shapes.hip ([W][L][C] float in, byte out).
//   copy4      float4 -> float4 linear copy (the guide's 6.3 TB/s shape), 8 B/sample-of-4-bytes moved
//   lin1       linear: dword read, byte write per lane
//   lin4       linear: float4 read, uchar4 write per lane
//   lin16      linear: 4 x float4 read, uint4 (16 flag bytes) write per lane
//   row16      the column kernel's row walk (lane = column, 16 rows in flight), flags leaving as ONE 16-byte store per
//              lane and 16 rows (64 columns x 16 rows of bytes = 1 KB per wave instruction), as after an in-wave transpose
//   rowbits    row walk, flags leaving bit-packed: one 8-byte lane mask per wave and row (0.125 B/sample)
//   rd_only / wr16_only   the two halves alone
//   row1       the row walk with one byte store per lane and row (= stream_shapes' "dword loads + byte stores")
//   panel64 / panel256   the SAME walk (lane = column, one dword in and one byte out per lane and row, 16 rows in flight)
//              over a column-panel-major image [C / PW][L][PW]: the rows a wave (workgroup) visits one after the other are
//              contiguous in memory, so every wave streams one contiguous region instead of hopping 16 KB / 4 KB per row
// hipcc --offload-arch=gfx950 -O3 stream_mix.hip -o stream_mix.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define L 1024
#define C 4096
enum { COPY4, LIN1, LIN4, LIN16, ROW16, ROWBITS, RD_ONLY, WR16_ONLY, LIN16_RD, LIN16_WR, PANEL64, PANEL256, PANEL256_WR, PANEL256_RD, ROW1, WRF_ROW256, WRF_ROW1K, WRF_PANEL64, WRF_LIN, SEG64_RD };
template <int MODE>
__global__ void __launch_bounds__(256) k(const float* __restrict__ in, uint8_t* __restrict__ out, float* __restrict__ out4, size_t n) {
    if (MODE == COPY4) {
        const size_t n4 = n / 4, stride = (size_t)gridDim.x * 256;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride)
            reinterpret_cast<float4*>(out4)[i] = reinterpret_cast<const float4*>(in)[i];
    } else if (MODE == LIN1) {
        // each workgroup owns a contiguous span: 256 lanes x 16 consecutive dword loads in flight
        const size_t span = 256 * 16;
        for (size_t b = (size_t)blockIdx.x * span; b < n; b += (size_t)gridDim.x * span) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = in[b + u * 256 + threadIdx.x];
#pragma unroll
            for (int u = 0; u < 16; u++) out[b + u * 256 + threadIdx.x] = (uint8_t)(v[u] > 3.0f);
        }
    } else if (MODE == LIN4) {
        const size_t n4 = n / 4, span = 256 * 8;
        for (size_t b = (size_t)blockIdx.x * span; b < n4; b += (size_t)gridDim.x * span) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = reinterpret_cast<const float4*>(in)[b + u * 256 + threadIdx.x];
#pragma unroll
            for (int u = 0; u < 8; u++)
                reinterpret_cast<uchar4*>(out)[b + u * 256 + threadIdx.x] = make_uchar4(v[u].x > 3.f, v[u].y > 3.f, v[u].z > 3.f, v[u].w > 3.f);
        }
    } else if (MODE == LIN16 || MODE == LIN16_RD || MODE == LIN16_WR) {
        const size_t n16 = n / 16, span = 256 * 2;
        unsigned keep = 0;
        for (size_t b = (size_t)blockIdx.x * span; b < n16; b += (size_t)gridDim.x * span) {
            float4 v[2][4];
            if (MODE != LIN16_WR) {
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int q = 0; q < 4; q++) v[u][q] = reinterpret_cast<const float4*>(in)[(b + u * 256 + threadIdx.x) * 4 + q];
            } else {
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int q = 0; q < 4; q++) v[u][q] = make_float4((float)b, 1.f, 2.f, 5.f);
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                uint4 f;
                unsigned* fp = &f.x;
#pragma unroll
                for (int q = 0; q < 4; q++)
                    fp[q] = (v[u][q].x > 3.f) | ((v[u][q].y > 3.f) << 8) | ((v[u][q].z > 3.f) << 16) | ((v[u][q].w > 3.f) << 24);
                if (MODE != LIN16_RD) reinterpret_cast<uint4*>(out)[b + u * 256 + threadIdx.x] = f;
                else keep += f.x + f.y + f.z + f.w;
            }
        }
        if (MODE == LIN16_RD && keep == 0x12345678u) out[0] = 1;
    } else if (MODE == PANEL64 || MODE == PANEL256 || MODE == PANEL256_WR || MODE == PANEL256_RD) {
        constexpr int PW = MODE == PANEL64 ? 64 : 256;
        const size_t win = blockIdx.y;
        const int c = blockIdx.x * 256 + threadIdx.x;
        const size_t base = win * (size_t)L * C + (size_t)(c / PW) * L * PW + (c % PW);
        const float* p = in + base;
        uint8_t* o = out + base;
        float acc = 0;
        for (int t0 = 0; t0 < L; t0 += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = MODE == PANEL256_WR ? (float)(t0 + u) : p[(size_t)(t0 + u) * PW];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                acc += v[u];
                if (MODE != PANEL256_RD) o[(size_t)(t0 + u) * PW] = (uint8_t)(acc > 3.0f);
            }
        }
        if (MODE == PANEL256_RD && acc == 12345.f) o[0] = 1;
    } else if (MODE == WRF_ROW256 || MODE == WRF_ROW1K || MODE == WRF_PANEL64) {
        // float image writes only (what a time-axis filter stage does to its output images): per wave and row
        // 256 B (one dword per lane), 1 KB (one float4 per lane), or 256 B into a [C/64][L][64] panel image
        const size_t win = blockIdx.y;
        float* o = out4 + win * (size_t)L * C;
        if (MODE == WRF_ROW1K) {
            const int c4 = blockIdx.x * 1024 + threadIdx.x * 4;          // (grid.x = C / 1024)
            for (int t = 0; t < L; t++) *reinterpret_cast<float4*>(o + (size_t)t * C + c4) = make_float4((float)t, 1.f, 2.f, 3.f);
        } else {
            const int c = blockIdx.x * 256 + threadIdx.x;
            const size_t base = MODE == WRF_PANEL64 ? (size_t)(c / 64) * L * 64 + (c % 64) : (size_t)c;
            const size_t rs = MODE == WRF_PANEL64 ? 64 : C;
            for (int t = 0; t < L; t++) o[base + (size_t)t * rs] = (float)t;
        }
    } else if (MODE == SEG64_RD) {
        // the fused frequency stage's input shape: one float4 per lane, a wave instruction covers 16 rows x 64 bytes
        // (lane = (row, quarter)); rows are image rows 16 KB apart, the walk goes along the row.  Reads only -- used
        // under rocprofv3 --pmc FETCH_SIZE to calibrate the counter for this shape (known bytes: 4 per sample).
        const size_t win = blockIdx.y;
        const float* p = in + win * (size_t)L * C;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int row0 = (blockIdx.x * 4 + wave) * 16 + (lane >> 2);      // (grid.x = L / 64)
        const float* src = p + (size_t)row0 * C + (lane & 3) * 4;
        float acc = 0;
        for (int f0 = 0; f0 < C; f0 += 16 * 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const float4*>(src + f0 + 16 * u);
#pragma unroll
            for (int u = 0; u < 4; u++) acc += v[u].x + v[u].y + v[u].z + v[u].w;
        }
        if (acc == 12345.f) out[threadIdx.x] = 1;
    } else if (MODE == WRF_LIN) {
        const size_t n4 = n / 4, stride = (size_t)gridDim.x * 256;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride)
            reinterpret_cast<float4*>(out4)[i] = make_float4((float)i, 1.f, 2.f, 3.f);
    } else if (MODE == ROW1) {
        const size_t win = blockIdx.y;
        const float* p = in + win * (size_t)L * C;
        uint8_t* o = out + win * (size_t)L * C;
        const int c = blockIdx.x * 256 + threadIdx.x;
        float acc = 0;
        for (int t0 = 0; t0 < L; t0 += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = p[(size_t)(t0 + u) * C + c];
#pragma unroll
            for (int u = 0; u < 16; u++) { acc += v[u]; o[(size_t)(t0 + u) * C + c] = (uint8_t)(acc > 3.0f); }
        }
    } else {
        // row walks: lane = column, blockIdx.y = window
        const size_t win = blockIdx.y;
        const float* p = in + win * (size_t)L * C;
        uint8_t* o = out + win * (size_t)L * C;
        const int c = blockIdx.x * 256 + threadIdx.x;
        const int lane = threadIdx.x & 63;
        const int wc0 = blockIdx.x * 256 + (threadIdx.x & ~63);      // first column of this wave
        float acc = 0;
        for (int t0 = 0; t0 < L; t0 += 16) {
            float v[16];
            if (MODE != WR16_ONLY) {
#pragma unroll
                for (int u = 0; u < 16; u++) v[u] = p[(size_t)(t0 + u) * C + c];
            } else {
#pragma unroll
                for (int u = 0; u < 16; u++) v[u] = (float)(t0 + u);
            }
#pragma unroll
            for (int u = 0; u < 16; u++) acc += v[u];
            if (MODE == ROW16 || MODE == WR16_ONLY) {
                // lane j: row t0 + j / 4, columns wc0 + (j % 4) * 16 .. + 15
                uint4 f = make_uint4(acc > 3.f, acc > 4.f, acc > 5.f, acc > 6.f);
                *reinterpret_cast<uint4*>(o + (size_t)(t0 + (lane >> 2)) * C + wc0 + (lane & 3) * 16) = f;
            } else if (MODE == ROWBITS) {
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(v[u] + acc > 3.f);
                    if (lane == 0) reinterpret_cast<unsigned long long*>(o)[((size_t)(t0 + u) * C + wc0) / 64] = m;
                }
            }
        }
        if (MODE == RD_ONLY && acc == 12345.f) o[c] = 1;
    }
}
template <int MODE>
void run(const char* name, const float* in, uint8_t* out, float* out4, int W, double bytes_per_sample) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t n = (size_t)W * L * C;
    const bool rows = (MODE == ROW16 || MODE == ROWBITS || MODE == RD_ONLY || MODE == WR16_ONLY || MODE >= PANEL64) && MODE != WRF_LIN;
    dim3 grid = MODE == SEG64_RD ? dim3(L / 64, W) : (rows ? dim3(MODE == WRF_ROW1K ? C / 1024 : C / 256, W) : dim3(256 * 32, 1));
    k<MODE><<<grid, 256>>>(in, out, out4, n);
    hipEventRecord(e0);
    for (int i = 0; i < 5; i++) k<MODE><<<grid, 256>>>(in, out, out4, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-78s %.3f ms  %.2f TB/s\n", name, ms, (double)n * bytes_per_sample / ms / 1e9);
}
int main() {
    const int W = 1008;
    const size_t n = (size_t)W * L * C;
    float *in, *out4; uint8_t* out;
    hipMalloc(&in, n * 4);
    hipMalloc(&out4, n * 4);
    hipMalloc(&out, n);
    hipMemset(in, 0, n * 4);
    run<COPY4>("copy4: float4 -> float4 linear copy (4 B read + 4 B written)", in, out, out4, W, 8);
    run<LIN1>("lin1: linear dword read + byte write per lane (5 B)", in, out, out4, W, 5);
    run<LIN4>("lin4: linear float4 read + uchar4 write per lane (5 B)", in, out, out4, W, 5);
    run<LIN16>("lin16: linear 4 x float4 read + 16-byte flag write per lane (5 B)", in, out, out4, W, 5);
    run<LIN16_RD>("lin16 reads only (4 B)", in, out, out4, W, 4);
    run<LIN16_WR>("lin16 writes only (1 B)", in, out, out4, W, 1);
    run<RD_ONLY>("row walk, dword loads only, 16 rows in flight (4 B)", in, out, out4, W, 4);
    run<WR16_ONLY>("row walk, 16-byte flag stores only (1 B)", in, out, out4, W, 1);
    run<ROW16>("row16: row walk + one 16-byte flag store per lane and 16 rows (5 B)", in, out, out4, W, 5);
    run<ROWBITS>("rowbits: row walk + bit-packed flags, 8 B per wave and row (4.125 B)", in, out, out4, W, 4.125);
    run<ROW1>("row1: row walk, dword load + byte store per lane and row (5 B)", in, out, out4, W, 5);
    run<PANEL64>("panel64: same walk over [C/64][L][64] panels (5 B)", in, out, out4, W, 5);
    run<PANEL256>("panel256: same walk over [C/256][L][256] panels (5 B)", in, out, out4, W, 5);
    run<PANEL256_RD>("panel256 reads only (4 B)", in, out, out4, W, 4);
    run<PANEL256_WR>("panel256 byte writes only (1 B)", in, out, out4, W, 1);
    run<WRF_ROW256>("float writes only, row walk, 256 B per wave and row (4 B)", in, out, out4, W, 4);
    run<WRF_ROW1K>("float writes only, row walk, 1 KB per wave and row (4 B)", in, out, out4, W, 4);
    run<WRF_PANEL64>("float writes only, [C/64][L][64] panels, 256 B per wave and row (4 B)", in, out, out4, W, 4);
    run<WRF_LIN>("float writes only, linear float4 (4 B)", in, out, out4, W, 4);
    run<SEG64_RD>("reads only, 64-byte row segments: float4 per lane, 16 rows per wave instruction (4 B)", in, out, out4, W, 4);
    return 0;
}

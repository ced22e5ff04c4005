// tricolour_amd.hip -- host orchestration and C ABI of the MI355X SumThreshold flagger.
// The kernels live in the headers included below (one translation unit):
//   tri_common.hpp            overview, error plumbing, device helpers
//   kernels_elementwise.hpp   K1, K2, K5, K6, K8, pack / unpack, strategy steps
//   kernels_median.hpp        K3  exact medians
//   kernels_reject.hpp        K3r block median + rejection of the background loop in one pass (one workgroup per block; the redo path)
//   kernels_reject_tile.hpp   K3t the same, tile-parallel: predict / pass / finish
//   kernels_boxfilter.hpp     K4  box-Gaussian filter (four forms)
//   kernels_boxline.hpp / kernels_boxpipe.hpp   K4r register delay lines, K4p / K4q / K4qf stage pipelines
//   kernels_boxweight.hpp     K4w integer weight image of the time-axis stage (bit / byte / halfword-packed delay lines)
//   kernels_boxexact.hpp      K4x exact row filter for any radius (LDS-resident line, checked exactness)
//   kernels_sumthreshold.hpp  K7  fused SumThreshold
#include <unordered_set>
#include "tri_common.hpp"
#include "kernels_elementwise.hpp"
#include "kernels_median.hpp"
#include "kernels_reject.hpp"
#include "kernels_reject_tile.hpp"
#include "kernels_boxfilter.hpp"
#include "kernels_boxline.hpp"
#include "kernels_boxpipe.hpp"
#include "kernels_boxweight.hpp"
#include "kernels_boxexact.hpp"
#include "kernels_sumthreshold.hpp"

// ===========================================================================
// host side
// ===========================================================================
namespace {

// Opt-in to more than 64 KB of dynamic LDS for one kernel.  The attribute belongs to the (kernel, device) pair and a
// process may drive several devices (one calling thread per device): it is set once per calling thread, kernel
// and device -- not once per process -- and a transient failure is not cached.
static hipError_t lds_optin(const void* fn, hipFuncAttribute attr, int value) {
    thread_local std::unordered_set<uint64_t> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t key = (uint64_t)(uintptr_t)fn * 64u + (uint64_t)(dev & 63);
    if (done.count(key)) return hipSuccess;
    e = hipFuncSetAttribute(fn, attr, value);
    if (e == hipSuccess) done.insert(key);
    return e;
}

struct Bump {
    char* base;
    size_t cap, off;
    bool dry;
    Bump(void* b, size_t c, bool d) : base((char*)b), cap(c), off(0), dry(d) {}
    template <typename T>
    T* get(size_t count) {
        size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
        size_t o = off;
        off += bytes;
        if (dry) return nullptr;
        return reinterpret_cast<T*>(base + o);
    }
};

// int(0.5 * sqrt(12 sigma^2 / passes + 1)), flagging.py:451
int64_t box_radius(double sigma) {
    return (int64_t)(0.5 * std::sqrt(12.0 * (sigma * sigma) / 4.0 + 1.0));
}

// float32(d) ** 4 as numba evaluates it: square-and-multiply in float32
// (numba cpython/numbers.py:207-245)
float box_denominator(int64_t r) {
    volatile float a = (float)(2 * r + 1);
    volatile float a2 = a * a;
    volatile float a4 = a2 * a2;
    volatile float res = 1.0f * a4;
    return res;
}

// RN(1 / b) for the reciprocal-based exact division of the register-ring filter kernels
// (box_divide): b and the candidates are 24-bit, so b * y is exact in float64 and the
// candidate closest to 1 / b is found without rounding.
BoxDenom box_reciprocal(float b) {
    float y = (float)(1.0 / (double)b);
    const float cand[3] = {std::nextafterf(y, 0.0f), y, std::nextafterf(y, INFINITY)};
    double best = INFINITY;
    for (float c : cand) {
        const double e = std::fabs(1.0 - (double)b * (double)c);
        if (e < best) { best = e; y = c; }
    }
    return BoxDenom{b, y};
}

struct Plan {
    int64_t T, F, Fa, avg, G;
    int64_t r0max, r1max;   // largest box radii along time / frequency
    int64_t PT, PF;         // padded line lengths
    int nit;                // background_iterations
    bool vec;               // 16-byte vectorised elementwise kernels usable
    int64_t maxchunk;       // longest frequency chunk (averaged channels)
    StWin swT, swF;
};

int make_stwin(const int64_t* w, int64_t nw, double rho, StWin* s) {
    memset(s, 0, sizeof(*s));
    if (nw <= 0) return set_err(TRI_EINVAL, "no SumThreshold window fits the data (np.max of an empty window list)");
    if (nw > TRI_MAX_WINDOWS) return set_err(TRI_EINVAL, "too many windows");
    s->nw = (int)nw;
    int off = 0, d = 0, maxw = 0;
    for (int j = 0; j < nw; j++) {
        if (w[j] <= 0) return set_err(TRI_EINVAL, "SumThreshold window of size %lld (the reference fails here with a broadcasting ValueError)", (long long)w[j]);
        s->w[j] = (int)w[j];
        s->tf[j] = std::pow(rho, std::log2((double)w[j]));
        s->scale[j] = (double)(float)(1.0 / (double)w[j]);
        s->ringoff[j] = off;
        off += (int)w[j];
        s->delay[j] = d;
        d += (int)w[j] - 1;
        maxw = std::max(maxw, (int)w[j]);
    }
    s->delay[nw] = d;
    s->ringtot = off;
    s->acccap = d + 1;
    s->maxw = maxw;
    return TRI_OK;
}

int make_plan(int64_t T, int64_t F, const tri_params* p, Plan* pl) {
    if (T <= 0 || F <= 0) return set_err(TRI_EINVAL, "empty window (ntime=%lld, nchan=%lld)", (long long)T, (long long)F);
    if (p->average_freq < 1 || p->average_freq > 255) return set_err(TRI_EUNSUPPORTED, "average_freq must be in [1, 255]");
    if (p->n_chunk_ends < 2 || !p->chunk_ends) return set_err(TRI_EINVAL, "freq_chunks must be >= 1");
    if (p->background_iterations < 0 || p->num_major_iterations < 0) return set_err(TRI_EINVAL, "negative iteration count");
    pl->T = T; pl->F = F; pl->avg = p->average_freq;
    pl->Fa = (F + pl->avg - 1) / pl->avg;
    pl->G = p->n_chunk_ends - 1;
    for (int64_t g = 0; g < p->n_chunk_ends; g++) {
        if (p->chunk_ends[g] < 0 || p->chunk_ends[g] > pl->Fa || (g > 0 && p->chunk_ends[g] < p->chunk_ends[g - 1]))
            return set_err(TRI_EINVAL, "freq chunk ends must be non-decreasing within [0, averaged channels]");
    }
    if (p->chunk_ends[0] != 0 || p->chunk_ends[pl->G] != pl->Fa) return set_err(TRI_EINVAL, "freq chunk ends must start at 0 and end at the averaged channel count");
    pl->nit = (int)p->background_iterations;
    pl->maxchunk = 0;
    for (int64_t g = 0; g + 1 < p->n_chunk_ends; g++) pl->maxchunk = std::max(pl->maxchunk, p->chunk_ends[g + 1] - p->chunk_ends[g]);
    pl->vec = (pl->avg == 1 && F % 16 == 0 && T % 4 == 0);
    int64_t emax = std::max<int64_t>(pl->nit, 1);
    pl->r0max = box_radius((double)emax * p->spike_width_time);
    pl->r1max = box_radius((double)emax * p->spike_width_freq);
    if (pl->r0max > (1 << 20) || pl->r1max > (1 << 20)) return set_err(TRI_EUNSUPPORTED, "spike width too large");
    pl->PT = T + 4 * pl->r0max;
    pl->PF = pl->Fa + 4 * pl->r1max;
    // with no major iteration the reference never looks at its window lists
    int rc = TRI_OK;
    memset(&pl->swT, 0, sizeof(pl->swT));
    memset(&pl->swF, 0, sizeof(pl->swF));
    if (p->num_major_iterations > 0) {
        rc = make_stwin(p->windows_time, p->n_windows_time, p->rho, &pl->swT);
        if (rc) return rc;
        rc = make_stwin(p->windows_freq, p->n_windows_freq, p->rho, &pl->swF);
        if (rc) return rc;
    }
    if (T * pl->Fa >= ((int64_t)1 << 31) || T * F >= ((int64_t)1 << 31))
        return set_err(TRI_EUNSUPPORTED, "a single window must hold fewer than 2^31 samples");
    return TRI_OK;
}

// Workspace carve-up for a batch of Wb windows.  `dry` only measures.
#define TRI_MAX_CHUNKS 255
struct ChunkTab {
    int n;                       // number of chunk ends
    int64_t ends[TRI_MAX_CHUNKS + 1];
};

struct Ws {
    // per-window images
    float *dataTF, *dataFT, *Aw, *Ao, *Bw, *Bo;
    uint8_t *iter, *flagsTF, *flagsFT, *bgfTF, *bgfFT, *tflTF, *fflFT, *fflTF, *comb, *dil;
    int *rowcnt, *colcnt;
    double* med;      // medians: max(Fa, T*G) per window
    // spectrum layout [Fa][Wb]
    float *sdata, *sw, *so, *sres;
    uint8_t *sflags, *sbgf, *sout, *srows;   // srows: sout as rows [Wb][Fa]
    uint16_t* nanmask;                       // cached-amplitude path: one NaN bit per sample, 16 samples per word
    int* itab;                               // segment table of the interpolation (K6p)
    float* srowsf;                           // spectrum residuals as rows [Wb][Fa] (median input)
    uint8_t* srowsu;                         // their flags as rows
    double* smed;     // [Wb][G]
    // SumThreshold scratch
    double* ring;
    uint8_t* acc;
    // small tables (device)
    int64_t* d_chunk_ends;   // G+1, averaged-channel units
    int64_t* d_tends;        // {0, T}
    int64_t *segC_start, *segC_len;   // chunk segments in line units
    int64_t *segB_start, *segB_len;   // chunk blocks of the FT layout (x T)
    int64_t *segT_start, *segT_len;   // the single segment [0, T)
    int* d_chunk_of;         // Fa
    size_t total;
};

static void carve(const Plan& pl, int64_t Wb, void* base, size_t cap, bool dry, Ws* ws) {
    Bump b(base, cap, dry);
    size_t W = (size_t)Wb, T = (size_t)pl.T, F = (size_t)pl.F, Fa = (size_t)pl.Fa, G = (size_t)pl.G;
    size_t N = T * Fa;
    // small tables first so that their addresses do not depend on the batch
    ws->d_chunk_ends = b.get<int64_t>(G + 1);
    ws->d_tends = b.get<int64_t>(2);
    ws->segC_start = b.get<int64_t>(G);
    ws->segC_len = b.get<int64_t>(G);
    ws->segB_start = b.get<int64_t>(G);
    ws->segB_len = b.get<int64_t>(G);
    ws->segT_start = b.get<int64_t>(1);
    ws->segT_len = b.get<int64_t>(1);
    ws->d_chunk_of = b.get<int>(Fa);
    ws->iter = b.get<uint8_t>(W * T * F);
    ws->dataTF = b.get<float>(W * N);
    ws->dataFT = b.get<float>(W * N);
    // time-axis scratch: the weight and the weight * data image of a window sit next to each other
    // (window stride 2 * PT * Fa), so one buffer descriptor of the fused frequency stage reaches both
    ws->Aw = b.get<float>(W * 2 * (size_t)pl.PT * Fa);
    ws->Ao = dry ? nullptr : ws->Aw + (size_t)pl.PT * Fa;
    ws->Bw = b.get<float>(W * (size_t)pl.PF * T);
    ws->Bo = b.get<float>(W * (size_t)pl.PF * T);
    ws->flagsTF = b.get<uint8_t>(W * N);
    ws->flagsFT = b.get<uint8_t>(W * N);
    ws->bgfTF = b.get<uint8_t>(W * N);
    ws->bgfFT = b.get<uint8_t>(W * N);
    ws->tflTF = b.get<uint8_t>(W * N);
    ws->fflFT = b.get<uint8_t>(W * N);
    ws->fflTF = b.get<uint8_t>(W * N);
    ws->comb = b.get<uint8_t>(W * N);
    ws->dil = b.get<uint8_t>(W * T * F);
    ws->rowcnt = b.get<int>(W * T);
    ws->colcnt = b.get<int>(W * F);
    ws->nanmask = b.get<uint16_t>(W * N / 16 + 8);
    ws->itab = b.get<int>(3 * W * (size_t)cdiv(Fa, INTERP_SEG) * std::max<size_t>(T, 1));
    ws->med = b.get<double>(W * std::max(Fa, T * G));
    size_t PS = (size_t)pl.PF;  // spectrum padded length
    ws->sdata = b.get<float>(Fa * W);
    ws->sw = b.get<float>(PS * W);
    ws->so = b.get<float>(PS * W);
    ws->sres = b.get<float>(Fa * W);
    ws->sflags = b.get<uint8_t>(Fa * W);
    ws->sbgf = b.get<uint8_t>(Fa * W);
    ws->sout = b.get<uint8_t>(Fa * W);
    ws->srows = b.get<uint8_t>(Fa * W);
    ws->srowsf = b.get<float>(Fa * W);
    ws->srowsu = b.get<uint8_t>(Fa * W);
    ws->smed = b.get<double>(W * G);
    // SumThreshold scratch: one slot set per (window, chunk, column) thread
    size_t thrT = W * Fa, thrF = W * T * G;
    size_t ringn = std::max(thrT * (size_t)pl.swT.ringtot, thrF * (size_t)pl.swF.ringtot);
    size_t accn = std::max(thrT * (size_t)pl.swT.acccap, thrF * (size_t)pl.swF.acccap);
    ws->ring = b.get<double>(ringn);
    ws->acc = b.get<uint8_t>(accn);
    ws->total = b.off;
}

static inline dim3 grid1(size_t n, size_t W) { return dim3((unsigned)cdiv((int64_t)n, 256), (unsigned)W, 1); }

}  // namespace

__global__ void k_tables(ChunkTab ct, int T, int Fa, int64_t* chunk_ends, int64_t* tends,
                         int64_t* segC_start, int64_t* segC_len, int64_t* segB_start,
                         int64_t* segB_len, int64_t* segT_start, int64_t* segT_len,
                         int* chunk_of) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int G = ct.n - 1;
    if (i <= G) chunk_ends[i] = ct.ends[i];
    if (i < G) {
        segC_start[i] = ct.ends[i];
        segC_len[i] = ct.ends[i + 1] - ct.ends[i];
        segB_start[i] = ct.ends[i] * T;
        segB_len[i] = (ct.ends[i + 1] - ct.ends[i]) * T;
    }
    if (i == 0) { tends[0] = 0; tends[1] = T; segT_start[0] = 0; segT_len[0] = T; }
    if (i < Fa) {
        int g = 0;
        // chunk containing channel i: ends[g] <= i < ends[g+1] (skips empty chunks)
        while (g < G - 1 && !(i >= ct.ends[g] && i < ct.ends[g + 1])) g++;
        chunk_of[i] = g;
    }
}

// copies column `col` of a [L][C] array into a contiguous vector (debug taps)
__global__ void k_gather_col_f32(const float* __restrict__ a, float* __restrict__ out, int L, int C, int col) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < L) out[i] = a[(size_t)i * C + col];
}
__global__ void k_gather_col_u8(const uint8_t* __restrict__ a, uint8_t* __restrict__ out, int L, int C, int col) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < L) out[i] = a[(size_t)i * C + col];
}
__global__ void k_normalise_flags(const uint8_t* __restrict__ a, uint8_t* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i] ? 1 : 0;
}

extern "C" size_t tri_workspace_bytes(int64_t batch_windows, int64_t ntime, int64_t nchan,
                                      const tri_params* p) {
    Plan pl;
    if (!p || batch_windows <= 0 || make_plan(ntime, nchan, p, &pl) != TRI_OK) return 0;
    Ws ws;
    carve(pl, batch_windows, nullptr, 0, true, &ws);
    return ws.total;
}

extern "C" int tri_prepare_params(int64_t ntime, int64_t nchan, double outlier_nsigma,
                                  const double* windows_time, int64_t n_windows_time,
                                  const double* windows_freq, int64_t n_windows_freq,
                                  double background_reject, int64_t background_iterations,
                                  double spike_width_time, double spike_width_freq,
                                  int64_t time_extend, int64_t freq_extend, int64_t freq_chunks,
                                  int64_t average_freq, double flag_all_time_frac,
                                  double flag_all_freq_frac, double rho,
                                  int64_t num_major_iterations, int64_t* chunk_ends_buf,
                                  int64_t chunk_cap, tri_params* out) {
    if (!out || !chunk_ends_buf) return set_err(TRI_EINVAL, "NULL output");
    if (ntime <= 0 || nchan <= 0) return set_err(TRI_EINVAL, "empty window");
    if (average_freq < 1) return set_err(TRI_EINVAL, "average_freq must be >= 1");
    if (freq_chunks < 1) return set_err(TRI_EINVAL, "freq_chunks must be >= 1");
    if (chunk_cap < freq_chunks + 1) return set_err(TRI_EINVAL, "chunk_ends_buf too small");
    if (n_windows_time > TRI_MAX_WINDOWS || n_windows_freq > TRI_MAX_WINDOWS || n_windows_time < 0 || n_windows_freq < 0)
        return set_err(TRI_EINVAL, "at most %d windows per axis", TRI_MAX_WINDOWS);
    memset(out, 0, sizeof(*out));
    int64_t fa = (nchan + average_freq - 1) / average_freq;
    // flagging.py:1160-1162: float32 ceil, float32 division, truncation, unique
    std::vector<int64_t> wf;
    for (int64_t i = 0; i < n_windows_freq; i++) {
        float v = std::ceil((float)windows_freq[i]) / (float)average_freq;
        wf.push_back((int64_t)v);
    }
    std::sort(wf.begin(), wf.end());
    wf.erase(std::unique(wf.begin(), wf.end()), wf.end());
    // flagging.py:1172-1173: linspace(0, fa, freq_chunks + 1).astype(int)
    double step = (double)fa / (double)freq_chunks;
    for (int64_t i = 0; i <= freq_chunks; i++) chunk_ends_buf[i] = (int64_t)((double)i * step);
    chunk_ends_buf[freq_chunks] = fa;
    // flagging.py:1176-1179
    out->n_windows_time = 0;
    for (int64_t i = 0; i < n_windows_time; i++)
        if (windows_time[i] <= (double)ntime) out->windows_time[out->n_windows_time++] = (int64_t)windows_time[i];
    out->n_windows_freq = 0;
    for (size_t i = 0; i < wf.size(); i++)
        if (wf[i] <= fa) out->windows_freq[out->n_windows_freq++] = wf[i];
    for (int64_t i = 0; i < out->n_windows_freq; i++)
        if (out->windows_freq[i] <= 0)
            return set_err(TRI_EINVAL, "windows_freq / average_freq produced a window of size %lld; the reference fails here (operands could not be broadcast together, flagging.py:663)", (long long)out->windows_freq[i]);
    for (int64_t i = 0; i < out->n_windows_time; i++)
        if (out->windows_time[i] <= 0) return set_err(TRI_EINVAL, "windows_time must be positive");
    out->outlier_nsigma = outlier_nsigma;
    out->background_reject = background_reject;
    out->background_iterations = background_iterations;
    out->spike_width_time = spike_width_time;
    out->spike_width_freq = spike_width_freq;
    out->time_extend = time_extend;
    out->freq_extend = freq_extend;
    out->n_chunk_ends = freq_chunks + 1;
    out->chunk_ends = chunk_ends_buf;
    out->average_freq = average_freq;
    out->flag_all_time_frac = flag_all_time_frac;
    out->flag_all_freq_frac = flag_all_freq_frac;
    out->rho = rho;
    out->num_major_iterations = num_major_iterations;
    return TRI_OK;
}

namespace {

struct Debug {
    float* f32;     // [spec_resid Fa][background FT Fa*T][residual TF T*Fa]
    uint8_t* u8;    // [spec_flags Fa][time_flags TF][freq_flags TF]
};

struct Run {
    hipStream_t st;
    Plan pl;
    Ws ws;
    const tri_params* p;
    int64_t Wb;      // windows in this batch
    Debug* dbg;      // taps of window 0 (tests only), or NULL
    // true: ws.dataTF holds the batch's UNMASKED amplitudes, computed once (k_amplitude4,
    // NaNs folded into the running flags; the time-axis filter masks by its own flags),
    // ws.dataFT their transpose, masked in place every iteration (the flags only grow)
    bool ampl_cached = false;
    // the iteration's input flags in TF layout while ws.dataTF is unmasked (K4x masks its data row with them:
    // data is zero where the iteration's _average_freq flagged it, flagging.py:858-870), else NULL
    const uint8_t* data_mask = nullptr;
    size_t data_mask_ws = 0;
};

// The register cascade covers windows exactly (1,2,4,8) in that order; the
// environment variable TRI_ST_GENERIC=1 forces the generic kernel (tests).
bool st_use_fused(const StWin& sw) {
    static const bool force_generic = [] {
        const char* e = getenv("TRI_ST_GENERIC");
        return e && e[0] == '1';
    }();
    if (force_generic) return false;
    return sw.nw == 4 && sw.w[0] == 1 && sw.w[1] == 2 && sw.w[2] == 4 && sw.w[3] == 8;
}

// The lane-mask cascade (K7c) addresses a window through signed 32-bit buffer
// offsets (windows below 2 GiB of float32); larger windows and TRI_ST_REGISTER=1 (tests) take the register
// cascade (K7b).
bool st_use_mask(int L, int C) {
    static const bool force_register = [] {
        const char* e = getenv("TRI_ST_REGISTER");
        return e && e[0] == '1';
    }();
    return !force_register && (uint64_t)L * (uint64_t)C * 4u < (1ull << 31);   // signed 32-bit scalar offsets
}

// launch_median() can OR a second flag image and per-column flags into the flags it reads when the segments fit the wave kernels
static bool median_takes_extra_flags(int64_t max_len) { return max_len <= 64 * MW_K; }

int launch_median(const Run& r, const float* data, const uint8_t* flags, double* med, size_t WSd,
                  size_t WSf, size_t RS, size_t ES, const int64_t* seg_start,
                  const int64_t* seg_len, int R, int G, int64_t W, int64_t max_len, bool vec_ok = false,
                  bool rows_aligned = false, bool segs_aligned = false,
                  unsigned* gcand = nullptr, size_t cand_ws = 0, unsigned cand_cap = 0,
                  const uint8_t* flags2 = nullptr, const uint8_t* colflags = nullptr, size_t WScol = 0, int panel_rows = 0) {
    if ((int64_t)R * G <= 0 || W <= 0) return TRI_OK;
    // (data / flags2 as column panels: only the 16-byte wave kernels read them -- see median_takes_panels())
    if (panel_rows > 0 && !(ES == 1 && RS % 64 == 0 && rows_aligned && max_len + (segs_aligned ? 0 : 3) <= 64 * MW_K &&
                            WSd % 4 == 0 && WSf % 4 == 0 && ((uintptr_t)data % 16 == 0) && ((uintptr_t)flags % 4 == 0)))
        return set_err(TRI_EUNSUPPORTED, "panel images need wave-sized segments of 64-column aligned rows");
    // (extra flag sources: the wave kernels only -- see median_takes_extra_flags())
    if ((flags2 || colflags) && (max_len > 64 * MW_K || (colflags && ES != 1) || ((uintptr_t)flags2 % 4 != 0) ||
                                 ((uintptr_t)colflags % 4 != 0) || WScol % 4 != 0))
        return set_err(TRI_EUNSUPPORTED, "extra flag sources need wave-sized segments of contiguous, 4-byte aligned rows");
    if ((int64_t)R * G > 0x7FFFFFFF || W > 65535) return set_err(TRI_EUNSUPPORTED, "median grid too large");
    // segments of contiguous 4-aligned rows can be loaded 16 bytes at a time
    // (misaligned segment ends are masked, costing up to 3 extra slots)
    // (the 16-byte wave kernels address a window's image through 32-bit buffer offsets: below 4 GB)
    const bool row4 = ES == 1 && RS % 4 == 0 && WSd % 4 == 0 && WSf % 4 == 0 &&
                      ((uintptr_t)data % 16 == 0) && ((uintptr_t)flags % 4 == 0) && rows_aligned &&
                      (uint64_t)RS * (uint64_t)std::max(R, panel_rows) * 4u < (1ull << 32);
    if (panel_rows > 0 && !row4) return set_err(TRI_EUNSUPPORTED, "panel images of 4 GB or more per window");
    const int64_t slack = segs_aligned ? 0 : 3;   // misaligned segment starts cost up to 3 masked slots
    // wave medians: MW_SPW rows of a segment per wave, the next row's loads in flight (round 4; TRI_MEDIAN_WAVE_OLD=1: one segment per wave)
    static const bool wave_old = [] { const char* e = getenv("TRI_MEDIAN_WAVE_OLD"); return e && e[0] == '1'; }();
    if (max_len <= 64 * MW_K && !wave_old) {
        if (max_len + slack <= 64 * 8 && row4)
            hipLaunchKernelGGL((k_median_wave<8, true, MW_SPW>), dim3((unsigned)cdiv((int64_t)cdiv((int64_t)R, MW_SPW) * G, 4), (unsigned)W), dim3(256), 0, r.st,
                               data, flags, med, WSd, WSf, RS, ES, seg_start, seg_len, R, G, flags2, colflags, WScol, panel_rows);
        else if (max_len <= 64 * 8 && panel_rows == 0)               // (panel images: the 16-byte kernels only)
            hipLaunchKernelGGL((k_median_wave<8, false, MW_SPW>), dim3((unsigned)cdiv((int64_t)cdiv((int64_t)R, MW_SPW) * G, 4), (unsigned)W), dim3(256), 0, r.st,
                               data, flags, med, WSd, WSf, RS, ES, seg_start, seg_len, R, G, flags2, colflags, WScol);
        else if (max_len + slack <= 64 * MW_K && row4)
            hipLaunchKernelGGL((k_median_wave<MW_K, true, MW_SPW16>), dim3((unsigned)cdiv((int64_t)cdiv((int64_t)R, MW_SPW16) * G, 4), (unsigned)W), dim3(256), 0, r.st,
                               data, flags, med, WSd, WSf, RS, ES, seg_start, seg_len, R, G, flags2, colflags, WScol, panel_rows);
        else if (max_len <= 64 * MW_K)
            hipLaunchKernelGGL((k_median_wave<MW_K, false, MW_SPW16>), dim3((unsigned)cdiv((int64_t)cdiv((int64_t)R, MW_SPW16) * G, 4), (unsigned)W), dim3(256), 0, r.st,
                               data, flags, med, WSd, WSf, RS, ES, seg_start, seg_len, R, G, flags2, colflags, WScol);
        LAUNCHCHK();
        return TRI_OK;
    }
    if (max_len + slack <= 64 * 8 && row4)
        hipLaunchKernelGGL((k_median_wave<8, true, 1>), dim3((unsigned)cdiv((int64_t)cdiv((int64_t)R, 1) * G, 4), (unsigned)W), dim3(256), 0, r.st,
                           data, flags, med, WSd, WSf, RS, ES, seg_start, seg_len, R, G, flags2, colflags, WScol, panel_rows);
    else if (max_len <= 64 * 8 && panel_rows == 0)               // (panel images: the 16-byte kernels only)
        hipLaunchKernelGGL((k_median_wave<8, false, 1>), dim3((unsigned)cdiv((int64_t)cdiv((int64_t)R, 1) * G, 4), (unsigned)W), dim3(256), 0, r.st,
                           data, flags, med, WSd, WSf, RS, ES, seg_start, seg_len, R, G, flags2, colflags, WScol);
    else if (max_len + slack <= 64 * MW_K && row4)
        hipLaunchKernelGGL((k_median_wave<MW_K, true, 1>), dim3((unsigned)cdiv((int64_t)cdiv((int64_t)R, 1) * G, 4), (unsigned)W), dim3(256), 0, r.st,
                           data, flags, med, WSd, WSf, RS, ES, seg_start, seg_len, R, G, flags2, colflags, WScol, panel_rows);
    else if (max_len <= 64 * MW_K)
        hipLaunchKernelGGL((k_median_wave<MW_K, false, 1>), dim3((unsigned)cdiv((int64_t)cdiv((int64_t)R, 1) * G, 4), (unsigned)W), dim3(256), 0, r.st,
                           data, flags, med, WSd, WSf, RS, ES, seg_start, seg_len, R, G, flags2, colflags, WScol);
    else {
        // long segments: two-pass select (K3c); TRI_MEDIAN_3PASS=1 keeps the three-pass kernel (A/B runs, tests)
        static const bool three = [] { const char* e = getenv("TRI_MEDIAN_3PASS"); return e && e[0] == '1'; }();
        const dim3 grid((unsigned)(R * G), (unsigned)W);
        if (three && vec_ok)
            hipLaunchKernelGGL(k_median<true>, grid, dim3(256), 0, r.st, data, flags, med, WSd, WSf, RS, ES, seg_start, seg_len, R, G);
        else if (three)
            hipLaunchKernelGGL(k_median<false>, grid, dim3(256), 0, r.st, data, flags, med, WSd, WSf, RS, ES, seg_start, seg_len, R, G);
        else {
            // scratch for the predicted-window candidates (K3c): one read of the segment instead of two
            static const bool no_predict = [] { const char* e = getenv("TRI_MEDIAN_NO_PREDICT"); return e && e[0] == '1'; }();
            if (no_predict || cand_cap < max_len || cand_ws % 4 != 0 || cand_cap % 4 != 0 || ((uintptr_t)gcand % 16 != 0)) { gcand = nullptr; cand_ws = 0; cand_cap = 0; }
            // few, long segments (fewer workgroups than the machine holds at six per CU): sixteen 16-byte groups in flight per thread
            if ((vec_ok || row4) && (int64_t)R * G * W < 1536 && max_len >= (int64_t)1 << 20)
                hipLaunchKernelGGL((k_median2<true, false, 16>), grid, dim3(256), 0, r.st, data, flags, med, WSd, WSf, RS, ES, seg_start, seg_len, R, G, gcand, cand_ws, cand_cap);
            else if (vec_ok || row4)
                hipLaunchKernelGGL(k_median2<true>, grid, dim3(256), 0, r.st, data, flags, med, WSd, WSf, RS, ES, seg_start, seg_len, R, G, gcand, cand_ws, cand_cap);
            else
                hipLaunchKernelGGL(k_median2<false>, grid, dim3(256), 0, r.st, data, flags, med, WSd, WSf, RS, ES, seg_start, seg_len, R, G, gcand, cand_ws, cand_cap);
        }
    }
    LAUNCHCHK();
    return TRI_OK;
}

// LDS budget of the single-sweep filter: 4 stages x 2r slots x BT threads x 4 B.
// Returns the block size to use, or 0 when the radius is too large and the
// in-place multi-pass kernel must be used instead.
// Radius limits of the single-sweep kernels.  r <= LDS_R_MAX: K4b with 256
// threads (4 rings per thread); r <= LANE4_R_MAX: K4c (one ring per thread);
// beyond: the in-place multi-pass kernel.
// Measured crossovers (scripts/build_variants.sh -DLDS_R_MAX=.. -DLANE4_R_MAX=..): four rings per thread win
// up to r = 15 (256 threads to r = 10, 128 beyond; K4c needs 2r >= 32 anyway), one ring per thread up to
// r = 160 (two waves per CU), the multi-pass kernel beyond.
#ifndef LDS_R_MAX
#define LDS_R_MAX 15
#endif
#ifndef LANE4_R_MAX
#define LANE4_R_MAX 160
#endif
int colfilter_lds_block(int rad, int C) {
    static const bool disabled = [] {
        const char* e = getenv("TRI_FILTER_MULTIPASS");
        return e && e[0] == '1';
    }();
    static const bool no_lane4 = [] {
        const char* e = getenv("TRI_FILTER_NO_LANE4");
        return e && e[0] == '1';
    }();
    if (disabled || rad <= 0) return 0;
    if (no_lane4) {
        int bt = rad <= 10 ? 256 : (rad <= 20 ? 128 : (rad <= 40 ? 64 : 0));   // K4b alone (TRI_FILTER_NO_LANE4)
        while (bt > 64 && bt / 2 >= C) bt /= 2;
        return bt;
    }
    if (rad <= LDS_R_MAX) {
        int bt = rad <= 10 ? 256 : 128;
        while (bt > 64 && bt / 2 >= C) bt /= 2;
        return bt;
    }
    return rad <= LANE4_R_MAX ? 64 : 0;
}
inline bool colfilter_use_lane4(int rad) {
    static const bool no_lane4 = [] {
        const char* e = getenv("TRI_FILTER_NO_LANE4");
        return e && e[0] == '1';
    }();
    return !no_lane4 && rad > LDS_R_MAX && rad <= LANE4_R_MAX;
}

// One axis of masked_gaussian_filter's two box filters (weight image and
// data image) on a column-layout image pair.
//   srcmode 0: images are built on the fly from (srcData, srcFlags) [n][C]
//   srcmode 1: images are float arrays; for the multi-pass kernel they sit in
//              rows [4r, 4r+n) of bufW / bufO, for the LDS kernel in rows [0,n)
// Output: rows [0,n) of dstW / dstO.
// `deferred_denom` (optional): when the single-sweep kernel is used the final
// division by float32(d)**4 is left to the consumer (transpose / masked_div) and
// *deferred_denom receives the denominator; otherwise it is set to 0.
// Register-ring single sweep (K4r, kernels_boxline.hpp): KS register slots per stage, the
// rest of the 2r-deep delay line in LDS.  Returns 0 when the radius is outside its range.
// TRI_FILTER_NO_REGRING=1 keeps the LDS-ring kernels (A/B runs, tests).
#ifndef BOXR_MAX_LDS_SLOTS
#define BOXR_MAX_LDS_SLOTS 60
#endif
thread_local int g_boxr_override = -1;     // measurement hook: 0 = LDS-ring kernels, 1 = register-ring kernels
int boxr_pick_ks(int rad) {
    static const bool env_off = [] { const char* e = getenv("TRI_FILTER_NO_REGRING"); return e && e[0] == '1'; }();
    const bool off = g_boxr_override >= 0 ? g_boxr_override == 0 : env_off;
    if (off || rad < 4 || rad > 107) return 0;
    const int ks = 2 * rad >= 80 ? 80 : (2 * rad >= 64 ? 64 : (2 * rad >= 32 ? 32 : (2 * rad >= 16 ? 16 : 8)));
    return 2 * rad - ks <= BOXR_MAX_LDS_SLOTS ? ks : 0;
}

// the time-axis stage keeps K4b below 16 slots (already at the HBM floor there); the fused
// frequency stage takes the register form from r = 4 on
#ifndef BOXT_MIN_2R
#define BOXT_MIN_2R 32
#endif
#ifndef BOXT_EXACT20
#define BOXT_EXACT20 1                   // r = 10: all 20 slots of a delay line in registers (k_boxt<20, false, *>): no LDS traffic at all,
                                         // 11.7 ms per 1008-window launch pair against 13.9 for the LDS delay lines (K4b)
#endif
int boxr_pick_ks_t(int rad) {
    if (BOXT_EXACT20 && 2 * rad == 20) return boxr_pick_ks(rad) > 0 ? 20 : 0;
    return 2 * rad >= BOXT_MIN_2R ? boxr_pick_ks(rad) : 0;
}
int boxr_pick_ks_f(int rad) {
    static const bool off = [] { const char* e = getenv("TRI_FILTER_NO_REGRING_F"); return e && e[0] == '1'; }();
    return off ? 0 : boxr_pick_ks(rad);
}

// K4w (kernels_boxweight.hpp): the weight image of the time-axis stage in integer arithmetic, one thread per line, for
// 20 <= 2r <= 110 -- every radius the register-delay-line kernels (K4r, K4q) serve on that axis.  Returns false when it does
// not apply; the caller's kernels then filter both images as before.  TRI_FILTER_NO_BOXW=1 switches it off (A/B runs, tests).
thread_local int g_boxw_override = -1;     // tests / benches: 0 = off
static bool boxw_usable(int rad, int n, int C) {
    static const bool off = [] { const char* e = getenv("TRI_FILTER_NO_BOXW"); return e && e[0] == '1'; }();
    if (g_boxw_override == 0 || (off && g_boxw_override < 0)) return false;
    return 2 * rad >= 20 && 2 * rad <= 110 && n % 4 == 0 && (uint64_t)n * (uint64_t)C * 4u < (1ull << 31);
}
template <int R2>
int launch_boxw_r2(const Run& r, const uint8_t* srcFlags, float* dstW, int n, int C, float denom, size_t sws, size_t dws, int64_t W) {
    dim3 grid((unsigned)cdiv(C, 64), (unsigned)W);
    hipLaunchKernelGGL((k_boxw<R2>), grid, dim3(64), 0, r.st, srcFlags, dstW, n, C, box_reciprocal(denom), sws, dws);
    LAUNCHCHK();
    return TRI_OK;
}
int launch_boxw(const Run& r, const uint8_t* srcFlags, float* dstW, int n, int C, int rad, float denom, size_t sws, size_t dws, int64_t W) {
#define BOXW_CASE(R2) case R2: return launch_boxw_r2<R2>(r, srcFlags, dstW, n, C, denom, sws, dws, W);
#define BOXW_CASE5(A) BOXW_CASE(A) BOXW_CASE(A + 2) BOXW_CASE(A + 4) BOXW_CASE(A + 6) BOXW_CASE(A + 8)
    switch (2 * rad) {
        BOXW_CASE5(20) BOXW_CASE5(30) BOXW_CASE5(40) BOXW_CASE5(50) BOXW_CASE5(60) BOXW_CASE5(70) BOXW_CASE5(80) BOXW_CASE5(90) BOXW_CASE5(100)
        BOXW_CASE(110)
    }
#undef BOXW_CASE5
#undef BOXW_CASE
    return set_err(TRI_EINVAL, "no integer weight filter for radius %d", rad);
}

template <int KS>
int launch_boxt_ks(const Run& r, const float* srcData, const uint8_t* srcFlags, float* dstW, float* dstO,
                   int n, int C, int rad, float denom, size_t sws, size_t dws, int64_t W) {
    const int d = 2 * rad - KS;
    const size_t lds = (size_t)4 * d * 64 * sizeof(float);
    dim3 grid((unsigned)cdiv(C, 64), (unsigned)W);
    const hipError_t attr = [] {
        hipError_t e = lds_optin(reinterpret_cast<const void*>(&k_boxt<KS, true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = lds_optin(reinterpret_cast<const void*>(&k_boxt<KS, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return e;
    }();
    HIPCHK(attr);
    // one launch per image: a compute unit then runs (mostly) one of the two long loop bodies at a time
    // (the weight image through the integer kernel K4w where it applies)
    const bool wk = boxw_usable(rad, n, C);
    if (wk) { const int rc = launch_boxw(r, srcFlags, dstW, n, C, rad, denom, sws, dws, W); if (rc) return rc; }
    if (d > 0) {
        if (!wk) hipLaunchKernelGGL((k_boxt<KS, true, 0>), grid, dim3(64), lds, r.st, srcData, srcFlags, dstW, n, C, rad, denom, sws, dws);
        hipLaunchKernelGGL((k_boxt<KS, true, 1>), grid, dim3(64), lds, r.st, srcData, srcFlags, dstO, n, C, rad, denom, sws, dws);
    } else {
        if (!wk) hipLaunchKernelGGL((k_boxt<KS, false, 0>), grid, dim3(64), 0, r.st, srcData, srcFlags, dstW, n, C, rad, denom, sws, dws);
        hipLaunchKernelGGL((k_boxt<KS, false, 1>), grid, dim3(64), 0, r.st, srcData, srcFlags, dstO, n, C, rad, denom, sws, dws);
    }
    LAUNCHCHK();
    return TRI_OK;
}

// Spectrum path (byte flags [n][C], a single "window", both images in one launch)
template <int KS>
int launch_boxt_spec_ks(const Run& r, const float* srcData, const uint8_t* srcFlags, float* dstW, float* dstO,
                        int n, int C, int rad, float denom) {
    const int d = 2 * rad - KS;
    const size_t lds = (size_t)4 * d * 64 * sizeof(float);
    dim3 grid((unsigned)cdiv(C, 64), 2);
    const hipError_t attr = lds_optin(reinterpret_cast<const void*>(&k_boxt_spec<KS, true>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    HIPCHK(attr);
    if (d > 0) hipLaunchKernelGGL((k_boxt_spec<KS, true>), grid, dim3(64), lds, r.st, srcData, srcFlags, dstW, dstO, n, C, rad, denom);
    else hipLaunchKernelGGL((k_boxt_spec<KS, false>), grid, dim3(64), 0, r.st, srcData, srcFlags, dstW, dstO, n, C, rad, denom);
    LAUNCHCHK();
    return TRI_OK;
}

int launch_boxt_spec(const Run& r, int ks, const float* srcData, const uint8_t* srcFlags, float* dstW, float* dstO,
                     int n, int C, int rad, float denom) {
    switch (ks) {
        case 8: return launch_boxt_spec_ks<8>(r, srcData, srcFlags, dstW, dstO, n, C, rad, denom);
        case 16: return launch_boxt_spec_ks<16>(r, srcData, srcFlags, dstW, dstO, n, C, rad, denom);
        case 32: return launch_boxt_spec_ks<32>(r, srcData, srcFlags, dstW, dstO, n, C, rad, denom);
        case 64: return launch_boxt_spec_ks<64>(r, srcData, srcFlags, dstW, dstO, n, C, rad, denom);
        case 80: return launch_boxt_spec_ks<80>(r, srcData, srcFlags, dstW, dstO, n, C, rad, denom);
    }
    return set_err(TRI_EINVAL, "no register-ring kernel for %d slots", ks);
}

// Stage-pipelined form (kernels_boxpipe.hpp): block length B for radius rad, 0 when it does not apply
// (2r >= B, four stage buffers within the CU's 160 KB of LDS).  TRI_SPEC_NO_PIPE=1 keeps the register rings.
static thread_local int g_boxp_override = -1;   // tests / benches: 0 = off, 8 / 16 = only that block length
static int boxp_pick_block(int rad, int C) {
    static const bool off = [] { const char* e = getenv("TRI_SPEC_NO_PIPE"); return e && e[0] == '1'; }();
    if (g_boxp_override == 0 || (off && g_boxp_override < 0)) return 0;
    if (g_boxp_override != 8 && 2 * rad >= 16 && C % 4 == 0 && boxp_lds_bytes(rad, 16) <= 160 * 1024) return 16;
    if (g_boxp_override == 16) return 0;
    if (2 * rad >= 8 && C % 2 == 0 && boxp_lds_bytes(rad, 8) <= 160 * 1024) return 8;
    return 0;
}

template <int B, int P>
int launch_boxp_spec_b(const Run& r, const float* srcData, const uint8_t* srcFlags, float* dstW, float* dstO,
                       int n, int C, int rad, float denom) {
    const hipError_t attr = lds_optin(reinterpret_cast<const void*>(&k_boxp_spec<B, P>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    HIPCHK(attr);
    dim3 grid((unsigned)cdiv(C, 64), 2);
    hipLaunchKernelGGL((k_boxp_spec<B, P>), grid, dim3(256), boxp_lds_bytes(rad, B), r.st, srcData, srcFlags, dstW, dstO, n, C, rad, denom);
    LAUNCHCHK();
    return TRI_OK;
}

// K4q (kernels_boxpipe.hpp): register part of the delay lines for radius rad, 0 when it does not apply.
// TRI_FILTER_NO_PIPE_T=1 keeps the K4r / LDS kernels.
#ifndef BOXQ_MIN_2R
#define BOXQ_MIN_2R 56
#endif
thread_local int g_boxq_override = -1;   // tests / benches: 0 = off, 1 = on wherever it applies
static int boxq_pick_ks(int rad) {
    static const bool off = [] { const char* e = getenv("TRI_FILTER_NO_PIPE_T"); return e && e[0] == '1'; }();
    if (g_boxq_override == 0 || (off && g_boxq_override < 0)) return 0;
    const int ks = 2 * rad / 16 * 16;
    return (ks >= 16 && ks <= 96) ? ks : 0;
}

template <int KS, int B = 16>
int launch_boxq_ks(const Run& r, const float* srcData, const uint8_t* srcFlags, float* dstW, float* dstO,
                   int n, int C, int rad, float denom, size_t sws, size_t dws, int64_t W) {
    const hipError_t attr = [] {
        hipError_t e = lds_optin(reinterpret_cast<const void*>(&k_boxq<KS, 0, B>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = lds_optin(reinterpret_cast<const void*>(&k_boxq<KS, 1, B>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return e;
    }();
    HIPCHK(attr);
    dim3 grid((unsigned)cdiv(C, 64), (unsigned)W);
    const BoxDenom dn = box_reciprocal(denom);
    if (boxw_usable(rad, n, C)) { const int rc = launch_boxw(r, srcFlags, dstW, n, C, rad, denom, sws, dws, W); if (rc) return rc; }
    else hipLaunchKernelGGL((k_boxq<KS, 0, B>), grid, dim3(256), boxq_lds_bytes(B), r.st, srcData, srcFlags, dstW, n, C, rad, dn, sws, dws);
    hipLaunchKernelGGL((k_boxq<KS, 1, B>), grid, dim3(256), boxq_lds_bytes(B), r.st, srcData, srcFlags, dstO, n, C, rad, dn, sws, dws);
    LAUNCHCHK();
    return TRI_OK;
}

// blocks of 8 positions (four workgroups = four waves per SIMD) while the delay line fits 128 registers: KS = 8 * floor(2r / 8)
#ifndef BOXQ_B8_MIN_2R
#define BOXQ_B8_MIN_2R 32
#endif
#ifndef BOXQ_B8_MAX2R
#define BOXQ_B8_MAX2R 88
#endif
thread_local int g_boxq_b8_override = -1;   // tests: 0 = blocks of 16 only (time- and frequency-axis stage pipelines)
static int boxq_pick_ks8(int rad) {
    static const int b8 = [] { const char* e = getenv("TRI_FILTER_PIPE_T_B8"); return e ? atoi(e) : 1; }();
    if (!b8 || g_boxq_b8_override == 0 || 2 * rad < BOXQ_B8_MIN_2R || 2 * rad >= BOXQ_B8_MAX2R) return 0;
    return 2 * rad / 8 * 8;
}

// blocks of 8 with FIFOs one block deeper (k_boxq_deep): KS = 80 for 2r = 88 .. 95, KS = 96 for 2r = 104 .. 111
template <int KS>
int launch_boxq_deep(const Run& r, const float* srcData, const uint8_t* srcFlags, float* dstW, float* dstO,
                     int n, int C, int rad, float denom, size_t sws, size_t dws, int64_t W) {
    const hipError_t attr = [] {
        hipError_t e = lds_optin(reinterpret_cast<const void*>(&k_boxq_deep<KS, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = lds_optin(reinterpret_cast<const void*>(&k_boxq_deep<KS, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return e;
    }();
    HIPCHK(attr);
    dim3 grid((unsigned)cdiv(C, 64), (unsigned)W);
    const BoxDenom dn = box_reciprocal(denom);
    if (boxw_usable(rad, n, C)) { const int rc = launch_boxw(r, srcFlags, dstW, n, C, rad, denom, sws, dws, W); if (rc) return rc; }
    else hipLaunchKernelGGL((k_boxq_deep<KS, 0>), grid, dim3(256), boxq_lds_bytes(8, 4), r.st, srcData, srcFlags, dstW, n, C, rad, dn, sws, dws);
    hipLaunchKernelGGL((k_boxq_deep<KS, 1>), grid, dim3(256), boxq_lds_bytes(8, 4), r.st, srcData, srcFlags, dstO, n, C, rad, dn, sws, dws);
    LAUNCHCHK();
    return TRI_OK;
}
#ifndef BOXQ_DEEP
#define BOXQ_DEEP 1
#endif

int launch_boxq(const Run& r, int ks, const float* srcData, const uint8_t* srcFlags, float* dstW, float* dstO,
                int n, int C, int rad, float denom, size_t sws, size_t dws, int64_t W) {
    if (BOXQ_DEEP && boxq_pick_ks8(40) > 0) {              // (blocks of 8 not switched off)
        if (2 * rad >= 88 && 2 * rad < 96) return launch_boxq_deep<80>(r, srcData, srcFlags, dstW, dstO, n, C, rad, denom, sws, dws, W);
        if (2 * rad >= 104 && 2 * rad < 112) return launch_boxq_deep<96>(r, srcData, srcFlags, dstW, dstO, n, C, rad, denom, sws, dws, W);
    }
    switch (boxq_pick_ks8(rad)) {
        case 32: return launch_boxq_ks<32, 8>(r, srcData, srcFlags, dstW, dstO, n, C, rad, denom, sws, dws, W);
        case 40: return launch_boxq_ks<40, 8>(r, srcData, srcFlags, dstW, dstO, n, C, rad, denom, sws, dws, W);
        case 48: return launch_boxq_ks<48, 8>(r, srcData, srcFlags, dstW, dstO, n, C, rad, denom, sws, dws, W);
        case 56: return launch_boxq_ks<56, 8>(r, srcData, srcFlags, dstW, dstO, n, C, rad, denom, sws, dws, W);
        case 64: return launch_boxq_ks<64, 8>(r, srcData, srcFlags, dstW, dstO, n, C, rad, denom, sws, dws, W);
        // (72: the data image's kernel would spill 7 registers at 128 -- blocks of 16 with KS = 64 take 2r = 72 .. 78)
        case 80: return launch_boxq_ks<80, 8>(r, srcData, srcFlags, dstW, dstO, n, C, rad, denom, sws, dws, W);
    }
    switch (ks) {
        case 16: return launch_boxq_ks<16>(r, srcData, srcFlags, dstW, dstO, n, C, rad, denom, sws, dws, W);
        case 32: return launch_boxq_ks<32>(r, srcData, srcFlags, dstW, dstO, n, C, rad, denom, sws, dws, W);
        case 48: return launch_boxq_ks<48>(r, srcData, srcFlags, dstW, dstO, n, C, rad, denom, sws, dws, W);
        case 64: return launch_boxq_ks<64>(r, srcData, srcFlags, dstW, dstO, n, C, rad, denom, sws, dws, W);
        case 80: return launch_boxq_ks<80>(r, srcData, srcFlags, dstW, dstO, n, C, rad, denom, sws, dws, W);
        case 96: return launch_boxq_ks<96>(r, srcData, srcFlags, dstW, dstO, n, C, rad, denom, sws, dws, W);
    }
    return set_err(TRI_EINVAL, "no stage-pipeline kernel for %d slots", ks);
}

int launch_boxt(const Run& r, int ks, const float* srcData, const uint8_t* srcFlags, float* dstW, float* dstO,
                int n, int C, int rad, float denom, size_t sws, size_t dws, int64_t W) {
    switch (ks) {
        case 16: return launch_boxt_ks<16>(r, srcData, srcFlags, dstW, dstO, n, C, rad, denom, sws, dws, W);
        case 20: return launch_boxt_ks<20>(r, srcData, srcFlags, dstW, dstO, n, C, rad, denom, sws, dws, W);
        case 32: return launch_boxt_ks<32>(r, srcData, srcFlags, dstW, dstO, n, C, rad, denom, sws, dws, W);
        case 64: return launch_boxt_ks<64>(r, srcData, srcFlags, dstW, dstO, n, C, rad, denom, sws, dws, W);
        case 80: return launch_boxt_ks<80>(r, srcData, srcFlags, dstW, dstO, n, C, rad, denom, sws, dws, W);
    }
    return set_err(TRI_EINVAL, "no register-ring kernel for %d slots", ks);
}

int launch_colfilter(const Run& r, int srcmode, float* bufW, float* bufO, const float* srcData,
                     const uint8_t* srcFlags, float* dstW, float* dstO, int n, int C, int rad,
                     size_t bws, size_t sws, size_t dws, int64_t W, float* deferred_denom = nullptr,
                     bool transposed_out = false, bool weights_are_01 = false) {
    float denom = box_denominator(rad);
    // integer arithmetic for the weight image is exact while (2r+1)^4 <= 2^24
    const int intw = (weights_are_01 && rad <= 31) ? 1 : 0;
    int bt = colfilter_lds_block(rad, C);
    if (deferred_denom) *deferred_denom = 0.0f;
    // spectrum path (one "window" of byte flags + data, both images): register-ring kernel from r = 4 on
    if (srcmode == 0 && !deferred_denom && !transposed_out && weights_are_01 && W == 1 && boxp_pick_block(rad, C) > 0 && rad <= 107 &&
        (uint64_t)n * (uint64_t)C * 4u < (1ull << 31) && ((uintptr_t)srcData % 16 == 0) && ((uintptr_t)srcFlags % 4 == 0) &&
        ((uintptr_t)dstW % 16 == 0) && ((uintptr_t)dstO % 16 == 0)) {
        if (boxp_pick_block(rad, C) == 16) return launch_boxp_spec_b<16, 16>(r, srcData, srcFlags, dstW, dstO, n, C, rad, denom);
        return launch_boxp_spec_b<8, 16>(r, srcData, srcFlags, dstW, dstO, n, C, rad, denom);
    }
    if (srcmode == 0 && !deferred_denom && !transposed_out && weights_are_01 && W == 1 && boxr_pick_ks(rad) > 0 &&
        (uint64_t)n * (uint64_t)C * 4u < (1ull << 31))
        return launch_boxt_spec(r, boxr_pick_ks(rad), srcData, srcFlags, dstW, dstO, n, C, rad, denom);
    if (srcmode == 2 && !deferred_denom && !transposed_out && weights_are_01 && boxq_pick_ks(rad) > 0 && rad <= 107 &&
        // measured (1008 windows, both images): K4r 13.4 / 13.5 / 14.6 / 15.9 / 19.2 / 21.9 / 19.9 / 23.0 / 23.7 ms at
        // r = 17 / 21 / 27 / 28 / 32 / 35 / 40 / 43 / 54; K4q with blocks of 8 (four waves per SIMD, r <= 43)
        // 15.3 / 15.8 / 16.0 / 15.9 / 16.1 / 16.3 / 17.2 / 17.3, with blocks of 16 19.3 ... 20.8
        (g_boxq_override == 1 || 2 * rad >= BOXQ_MIN_2R) &&
        n % 4 == 0 && (uint64_t)n * (uint64_t)C * 4u < (1ull << 31))
        return launch_boxq(r, boxq_pick_ks(rad), srcData, srcFlags, dstW, dstO, n, C, rad, denom, sws, dws, W);
    if (srcmode == 2 && !deferred_denom && !transposed_out && weights_are_01 && boxr_pick_ks_t(rad) > 0 &&
        n % 4 == 0 && (uint64_t)n * (uint64_t)C * 4u < (1ull << 31))   // signed 32-bit buffer offsets
        return launch_boxt(r, boxr_pick_ks_t(rad), srcData, srcFlags, dstW, dstO, n, C, rad, denom, sws, dws, W);
    if (bt > 0 && colfilter_use_lane4(rad) && !transposed_out) {
        size_t lds = (size_t)lane4_ring_capacity(rad) * 64 * sizeof(float);
        dim3 grid((unsigned)cdiv(C, 16), (unsigned)W, 2);
        // thread-safe one-time setup (C++11 static initialisation)
        const hipError_t attr4 = [] {
            hipError_t e = lds_optin(reinterpret_cast<const void*>(&k_colfilter_lane4<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = lds_optin(reinterpret_cast<const void*>(&k_colfilter_lane4<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = lds_optin(reinterpret_cast<const void*>(&k_colfilter_lane4<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = lds_optin(reinterpret_cast<const void*>(&k_colfilter_lane4<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = lds_optin(reinterpret_cast<const void*>(&k_colfilter_lane4<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            return e;
        }();
        HIPCHK(attr4);
        if (srcmode == 0) {
            // byte flags + data (spectrum path): build-free variant is not provided;
            // fall through to the 4-ring kernel below when it fits, else multi-pass
        } else if (srcmode == 2) {
            if (deferred_denom) {
                *deferred_denom = denom;
                hipLaunchKernelGGL((k_colfilter_lane4<2, false>), grid, dim3(64), lds, r.st, (const float*)nullptr, (const float*)nullptr,
                                   srcData, srcFlags, dstW, dstO, n, C, rad, denom, (size_t)0, sws, dws);
            } else {
                hipLaunchKernelGGL((k_colfilter_lane4<2, true>), grid, dim3(64), lds, r.st, (const float*)nullptr, (const float*)nullptr,
                                   srcData, srcFlags, dstW, dstO, n, C, rad, denom, (size_t)0, sws, dws);
            }
            LAUNCHCHK();
            return TRI_OK;
        } else if (deferred_denom) {
            *deferred_denom = denom;
            hipLaunchKernelGGL((k_colfilter_lane4<1, false>), grid, dim3(64), lds, r.st, (const float*)bufW, (const float*)bufO,
                               srcData, srcFlags, dstW, dstO, n, C, rad, denom, bws, sws, dws);
            LAUNCHCHK();
            return TRI_OK;
        } else {
            hipLaunchKernelGGL((k_colfilter_lane4<1, true>), grid, dim3(64), lds, r.st, (const float*)bufW, (const float*)bufO,
                               srcData, srcFlags, dstW, dstO, n, C, rad, denom, bws, sws, dws);
            LAUNCHCHK();
            return TRI_OK;
        }
    }
    if (bt > 0 && colfilter_use_lane4(rad) && srcmode == 0) {
        // spectrum path with a medium radius: 4-ring kernel with a small block if it fits
        bt = rad <= 20 ? 128 : (rad <= 40 ? 64 : 0);
        while (bt > 64 && bt / 2 >= C) bt /= 2;
    }
    if (bt > 0) {
        size_t lds = (size_t)4 * 2 * rad * bt * sizeof(float);
        dim3 grid((unsigned)cdiv(C, bt), (unsigned)W, 2);
        const hipError_t attr_set = [] {
            hipError_t e = lds_optin(reinterpret_cast<const void*>(&k_colfilter_lds<0, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = lds_optin(reinterpret_cast<const void*>(&k_colfilter_lds<1, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = lds_optin(reinterpret_cast<const void*>(&k_colfilter_lds<1, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = lds_optin(reinterpret_cast<const void*>(&k_colfilter_lds<1, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = lds_optin(reinterpret_cast<const void*>(&k_colfilter_lds<2, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = lds_optin(reinterpret_cast<const void*>(&k_colfilter_lds<2, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            return e;
        }();
        HIPCHK(attr_set);
        if (srcmode == 2) {
            if (deferred_denom) {
                *deferred_denom = denom;
                hipLaunchKernelGGL((k_colfilter_lds<2, false, false>), grid, dim3(bt), lds, r.st, (const float*)nullptr, (const float*)nullptr,
                                   srcData, srcFlags, dstW, dstO, n, C, rad, denom, (size_t)0, sws, dws, intw);
            } else {
                hipLaunchKernelGGL((k_colfilter_lds<2, true, false>), grid, dim3(bt), lds, r.st, (const float*)nullptr, (const float*)nullptr,
                                   srcData, srcFlags, dstW, dstO, n, C, rad, denom, (size_t)0, sws, dws, intw);
            }
            LAUNCHCHK();
            return TRI_OK;
        }
        if (transposed_out)
            hipLaunchKernelGGL((k_colfilter_lds<1, true, true>), grid, dim3(bt), lds, r.st, (const float*)bufW, (const float*)bufO,
                               srcData, srcFlags, dstW, dstO, n, C, rad, denom, bws, sws, dws, intw);
        else if (srcmode == 0)
            hipLaunchKernelGGL((k_colfilter_lds<0, true, false>), grid, dim3(bt), lds, r.st, (const float*)nullptr, (const float*)nullptr,
                               srcData, srcFlags, dstW, dstO, n, C, rad, denom, (size_t)0, sws, dws, intw);
        else if (deferred_denom) {
            *deferred_denom = denom;
            hipLaunchKernelGGL((k_colfilter_lds<1, false, false>), grid, dim3(bt), lds, r.st, (const float*)bufW, (const float*)bufO,
                               srcData, srcFlags, dstW, dstO, n, C, rad, denom, bws, sws, dws, intw);
        } else
            hipLaunchKernelGGL((k_colfilter_lds<1, true, false>), grid, dim3(bt), lds, r.st, (const float*)bufW, (const float*)bufO,
                               srcData, srcFlags, dstW, dstO, n, C, rad, denom, bws, sws, dws, intw);
        LAUNCHCHK();
        return TRI_OK;
    }
    int blk = C >= 256 ? 256 : (C >= 128 ? 128 : 64);
    dim3 grid((unsigned)cdiv(C, blk), (unsigned)W, 2);
    if (srcmode == 0)
        hipLaunchKernelGGL(k_colfilter<0>, grid, dim3(blk), 0, r.st, bufW, bufO, srcData, srcFlags,
                           dstW, dstO, n, C, rad, denom, bws, sws, dws);
    else
        hipLaunchKernelGGL(k_colfilter<1>, grid, dim3(blk), 0, r.st, bufW, bufO, srcData, srcFlags,
                           dstW, dstO, n, C, rad, denom, bws, sws, dws);
    LAUNCHCHK();
    return TRI_OK;
}

// K7p applies: up to eight windows whose flag-byte ring fits the CU's LDS (there are no prefix rings any more)
// (TRI_ST_NO_PIPE=1 keeps the global-scratch kernel)
bool st_use_pipe(const StWin& sw) {
    static const bool off = [] { const char* e = getenv("TRI_ST_NO_PIPE"); return e && e[0] == '1'; }();
    return !off && sw.nw >= 1 && sw.nw <= 8 && stp_lds_bytes(sw) <= 160 * 1024;
}

int launch_colst(const Run& r, const StWin& sw, const float* data, const double* med,
                 uint8_t* out, const int64_t* d_chunk_ends, int L, int C, int G, size_t ws_data,
                 size_t ws_out, int64_t W, bool panel = false) {
    double thr_scale = r.p->outlier_nsigma * TRI_MAD_NORMAL;  // flagging.py:623
    int blk = C >= 256 ? 256 : (C >= 128 ? 128 : 64);
    if (const char* e = getenv("TRI_ST_BLK")) { int b = atoi(e); if (b >= 64 && b <= ST_MAXBLK && C >= b) blk = b; }
    dim3 grid((unsigned)cdiv(C, blk), (unsigned)G, (unsigned)W);
    if (st_use_fused(sw)) {
        StFusedArgs fa;
        for (int j = 0; j < 4; j++) fa.tf[j] = sw.tf[j];
        if (panel && !(st_use_mask(L, C) && C % 64 == 0)) return set_err(TRI_EUNSUPPORTED, "panel images: lane-mask SumThreshold kernel on 64-column panels only");
        if (panel)
            hipLaunchKernelGGL((k_colst_mask<1, 2, 4, 8, true>), grid, dim3(blk), 0, r.st, data, med, out,
                               d_chunk_ends, fa, thr_scale, L, C, G, ws_data, ws_out);
        else if (st_use_mask(L, C))
            hipLaunchKernelGGL((k_colst_mask<1, 2, 4, 8>), grid, dim3(blk), 0, r.st, data, med, out,
                               d_chunk_ends, fa, thr_scale, L, C, G, ws_data, ws_out);
        else
            hipLaunchKernelGGL((k_colst_fused<1, 2, 4, 8>), grid, dim3(blk), 0, r.st, data, med, out,
                               d_chunk_ends, fa, thr_scale, L, C, G, ws_data, ws_out);
    } else if (panel) {
        return set_err(TRI_EUNSUPPORTED, "panel images: windows (1, 2, 4, 8) only");
    } else if (st_use_pipe(sw)) {
        // any list of up to eight windows: one window per wave, lagging twin accumulators (K7p)
        const hipError_t attr = lds_optin(reinterpret_cast<const void*>(&k_colst_pipe),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        HIPCHK(attr);
        dim3 gridp((unsigned)cdiv(C, 64), (unsigned)G, (unsigned)W);
        hipLaunchKernelGGL(k_colst_pipe, gridp, dim3(64 * sw.nw), stp_lds_bytes(sw), r.st, data, med, out, d_chunk_ends, sw,
                           stp_plan(sw), thr_scale, L, C, G, ws_data, ws_out);
    } else {
        hipLaunchKernelGGL(k_colst_dyn, grid, dim3(blk), 0, r.st, data, med, out, r.ws.ring, r.ws.acc,
                           d_chunk_ends, sw, thr_scale, L, C, G, ws_data, ws_out);
    }
    LAUNCHCHK();
    return TRI_OK;
}

// _linearly_interpolate_nans1d along the line axis of [L][C] images (K6 / K6p)
int launch_interp(const Run& r, float* a, int L, int C, size_t ws, int64_t W, const uint8_t* nanflag,
                  const float* data, size_t ws_data, float* resid) {
    static const bool one_pass = [] { const char* e = getenv("TRI_INTERP_ONE_PASS"); return e && e[0] == '1'; }();
    const int nseg = (int)cdiv(L, INTERP_SEG);
    if (one_pass || nseg < 2 || nseg > 65535) {
        hipLaunchKernelGGL(k_colinterp, dim3((unsigned)cdiv(C, 256), (unsigned)W), dim3(256), 0, r.st, a, L, C, ws, nanflag, data, ws_data, resid);
    } else {
        dim3 grid((unsigned)cdiv(C, 64), (unsigned)nseg, (unsigned)W);
        hipLaunchKernelGGL(k_interp_scan, grid, dim3(64), 0, r.st, (const float*)a, L, C, ws, nanflag, r.ws.itab);
        hipLaunchKernelGGL(k_interp_fix, grid, dim3(64), 0, r.st, a, L, C, ws, (const int*)r.ws.itab, data, ws_data, resid);
    }
    LAUNCHCHK();
    return TRI_OK;
}

template <typename T>
int launch_transpose(const Run& r, const T* src, T* dst, int R, int C, size_t sws, size_t dws,
                     int64_t W, float denom = 0.0f, bool panel = false) {
    dim3 grid((unsigned)cdiv(C, 64), (unsigned)cdiv(R, 64), (unsigned)W);
    if (panel) {
        if (R % 64 != 0) return set_err(TRI_EUNSUPPORTED, "panel output needs a multiple of 64 columns");
        hipLaunchKernelGGL((k_transpose<T, true>), grid, dim3(64, 4), 0, r.st, src, dst, R, C, sws, dws, denom);
        LAUNCHCHK();
        return TRI_OK;
    }
    if (sizeof(T) == 1 && R % 4 == 0 && C % 4 == 0 && sws % 4 == 0 && dws % 4 == 0 &&
        ((uintptr_t)src % 4 == 0) && ((uintptr_t)dst % 4 == 0)) {
        dim3 gridw((unsigned)cdiv(C, 64), (unsigned)cdiv(R, 128), (unsigned)W);
        hipLaunchKernelGGL((k_transpose_u8w<false, false>), gridw, dim3(256), 0, r.st, (const uint8_t*)src, (uint8_t*)dst,
                           (uint8_t*)nullptr, (float*)nullptr, R, C, sws, dws, (size_t)0, (size_t)0);
        LAUNCHCHK();
        return TRI_OK;
    }
    hipLaunchKernelGGL(k_transpose<T>, grid, dim3(64, 4), 0, r.st, src, dst, R, C, sws, dws, denom);
    LAUNCHCHK();
    return TRI_OK;
}


// ---- elementwise launch helpers (vector path when the plan allows) ----------
template <int OP>
int launch_u8(const Run& r, const uint8_t* a, uint8_t* b, size_t nper, size_t ws_a, size_t ws_b, int64_t W) {
    if (r.pl.vec && nper % 16 == 0 && ws_a % 16 == 0 && ws_b % 16 == 0) {
        hipLaunchKernelGGL(k_u8_op16<OP>, grid1(nper / 16, W), dim3(256), 0, r.st, a, b, nper / 16, ws_a, ws_b);
    } else if (OP == 0) {
        for (int64_t w = 0; w < W; w++)
            hipLaunchKernelGGL(k_copy_u8, dim3((unsigned)cdiv(nper, 256)), dim3(256), 0, r.st, a + w * ws_a, b + w * ws_b, nper);
    } else if (OP == 1) {
        hipLaunchKernelGGL(k_or, grid1(nper, W), dim3(256), 0, r.st, b, a, nper, ws_b, ws_a);
    } else {
        for (int64_t w = 0; w < W; w++)
            hipLaunchKernelGGL(k_normalise_flags, dim3((unsigned)cdiv(nper, 256)), dim3(256), 0, r.st, a + w * ws_a, b + w * ws_b, nper);
    }
    LAUNCHCHK();
    return TRI_OK;
}

template <int MODE>
int launch_masked_div(const Run& r, const float* w, float* o, const float* data, size_t nper, size_t ws_wo, size_t ws_data, int64_t W, float denom = 0.0f,
                      uint8_t* nanflag = nullptr, int C = 1) {
    if (nper % 4 == 0 && ws_wo % 4 == 0 && ws_data % 4 == 0 && (nanflag == nullptr || C % 4 == 0) && ((uintptr_t)w % 16 == 0) && ((uintptr_t)o % 16 == 0))
        hipLaunchKernelGGL(k_masked_div4<MODE>, grid1(nper / 4, W), dim3(256), 0, r.st, w, o, data, nper / 4, ws_wo, ws_data, denom, nanflag, C);
    else
        hipLaunchKernelGGL(k_masked_div<MODE>, grid1(nper, W), dim3(256), 0, r.st, w, o, data, nper, ws_wo, ws_data, denom, nanflag, C);
    LAUNCHCHK();
    return TRI_OK;
}

int launch_sub(const Run& r, const float* a, const float* b, float* out, size_t nper, size_t ws_a, size_t ws_b, size_t ws_o, int64_t W) {
    if (nper % 4 == 0 && ws_a % 4 == 0 && ws_b % 4 == 0 && ws_o % 4 == 0 && ((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0) && ((uintptr_t)out % 16 == 0))
        hipLaunchKernelGGL(k_sub4, grid1(nper / 4, W), dim3(256), 0, r.st, a, b, out, nper / 4, ws_a, ws_b, ws_o);
    else
        hipLaunchKernelGGL(k_sub, grid1(nper, W), dim3(256), 0, r.st, a, b, out, nper, ws_a, ws_b, ws_o);
    LAUNCHCHK();
    return TRI_OK;
}

// Frequency-axis stage reading the time-axis stage's TF images directly
// (k_colfilter_lds_t).  Usable for radii whose four rings fit LDS at 128 threads.
bool colfilter_t_usable(int rad) {
    static const bool off = [] { const char* e = getenv("TRI_FILTER_NO_TIN"); return e && e[0] == '1'; }();
    return !off && rad > 0 && rad <= 16;
}

int launch_colfilter_t(const Run& r, const float* srcW, const float* srcO, float* dstW, float* dstO,
                       int n, int C, int ld, int rad, size_t sws_img, size_t dws, int64_t W, float* deferred_denom) {
    float denom = box_denominator(rad);
    size_t lds = ((size_t)4 * 2 * rad * CFT_BT + (size_t)CFT_PF * (CFT_BT + 1)) * sizeof(float);
    const hipError_t attr = [] {
        hipError_t e = lds_optin(reinterpret_cast<const void*>(&k_colfilter_lds_t<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = lds_optin(reinterpret_cast<const void*>(&k_colfilter_lds_t<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return e;
    }();
    HIPCHK(attr);
    dim3 grid((unsigned)cdiv(C, CFT_BT), (unsigned)W, 2);
    if (deferred_denom) {
        *deferred_denom = denom;
        hipLaunchKernelGGL(k_colfilter_lds_t<false>, grid, dim3(CFT_BT), lds, r.st, srcW, srcO, dstW, dstO, n, C, ld, rad, denom, sws_img, dws);
    } else {
        hipLaunchKernelGGL(k_colfilter_lds_t<true>, grid, dim3(CFT_BT), lds, r.st, srcW, srcO, dstW, dstO, n, C, ld, rad, denom, sws_img, dws);
    }
    LAUNCHCHK();
    return TRI_OK;
}

// The same stage fused with the masked division that follows it (k_colfilter_lds_tf):
// MODE 1 leaves |data - background| in dstO, MODE 2 the background in dstO, the signed
// residual in dstW and the per-line NaN markers.  TRI_FILTER_NO_FUSED_DIV=1 keeps the
// separate kernels (A/B runs).
bool colfilter_tf_usable(int rad) {
    static const bool off = [] { const char* e = getenv("TRI_FILTER_NO_FUSED_DIV"); return e && e[0] == '1'; }();
    return !off && colfilter_t_usable(rad);
}

template <int MODE>
int launch_colfilter_tf(const Run& r, const float* srcW, const float* srcO, float* dstW, float* dstO, const float* data,
                        int n, int C, int ld, int rad, size_t sws_img, size_t dws, size_t ws_data, int64_t W, uint8_t* nanflag) {
    float denom = box_denominator(rad);
    size_t lds = (size_t)2 * ((size_t)4 * 2 * rad * CFF_BT + (size_t)CFT_PF * (CFF_BT + 1)) * sizeof(float);
    const hipError_t attr = lds_optin(reinterpret_cast<const void*>(&k_colfilter_lds_tf<MODE>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    HIPCHK(attr);
    dim3 grid((unsigned)cdiv(C, CFF_BT), (unsigned)W);
    hipLaunchKernelGGL(k_colfilter_lds_tf<MODE>, grid, dim3(2 * CFF_BT), lds, r.st, srcW, srcO, dstW, dstO, data, n, C, ld, rad,
                       denom, sws_img, dws, ws_data, nanflag);
    LAUNCHCHK();
    return TRI_OK;
}

// Register-ring form of the fused stage (K4r, k_boxf): any radius boxr_pick_ks() accepts.
template <int KS, int MODE>
int launch_boxf_ks(const Run& r, const float* srcW, const float* srcO, float* dstW, float* dstO, const float* data,
                   int n, int C, int ld, int rad, size_t sws_img, size_t dws, size_t ws_data, int64_t W, uint8_t* nanflag) {
    const BoxDenom denom = box_reciprocal(box_denominator(rad));
    const int d = 2 * rad - KS;
    const size_t lds = ((size_t)4 * d * 64 + (size_t)2 * boxr_pf_f(KS) * boxf_nsub(KS) * boxf_ts(boxr_pf_f(KS) * boxf_nsub(KS))) * sizeof(float);
    // one descriptor per window spans both images: the data image must follow the weight image closely
    if (srcO <= srcW || ((uint64_t)(srcO - srcW) + (uint64_t)C * ld) * 4u >= (1ull << 31))
        return set_err(TRI_EUNSUPPORTED, "fused frequency stage: the data image must follow the weight image within 2^31 bytes");
    const unsigned gap = (unsigned)(srcO - srcW);
    const hipError_t attr = lds_optin(reinterpret_cast<const void*>(&k_boxf<KS, true, MODE>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    HIPCHK(attr);
    dim3 grid((unsigned)cdiv(C, 32), (unsigned)W);
    // 32 register slots: two waves per SIMD only pay while the LDS part leaves room for them (measured: r = 30,
    // 28 LDS slots: 10.0 ms per 252 windows at two waves per SIMD, 7.0 ms with the one-wave register budget)
    bool one_wave = false;
    if constexpr (KS == 32) {
        if (d > 14) {
            const hipError_t attr1 = lds_optin(reinterpret_cast<const void*>(&k_boxf<KS, true, MODE, 1>),
                                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            HIPCHK(attr1);
            hipLaunchKernelGGL((k_boxf<KS, true, MODE, 1>), grid, dim3(64), lds, r.st, srcW, gap, dstW, dstO, data, n, C, ld, rad,
                               denom, sws_img, dws, ws_data, nanflag);
            one_wave = true;
        }
    }
    if (one_wave) {
    } else if (d > 0)
        hipLaunchKernelGGL((k_boxf<KS, true, MODE>), grid, dim3(64), lds, r.st, srcW, gap, dstW, dstO, data, n, C, ld, rad,
                           denom, sws_img, dws, ws_data, nanflag);
    else
        hipLaunchKernelGGL((k_boxf<KS, false, MODE>), grid, dim3(64), lds, r.st, srcW, gap, dstW, dstO, data, n, C, ld, rad,
                           denom, sws_img, dws, ws_data, nanflag);
    LAUNCHCHK();
    return TRI_OK;
}

static int boxq_pick_ks(int rad);
extern thread_local int g_boxq_override;
extern thread_local int g_boxq_b8_override;
// K4qf: the fused frequency stage as an eight-wave stage pipeline (kernels_boxpipe.hpp)
#ifndef BOXQF_MIN_2R
#define BOXQF_MIN_2R 34
#endif
#ifndef BOXQF_FEW_WAVES
#define BOXQF_FEW_WAVES 2048             // waves a K4r launch needs to fill the machine at two per SIMD
#endif
#ifndef BOXQF_B8_DEFAULT
#define BOXQF_B8_DEFAULT 1
#endif
#ifndef BOXQF_B8_MAX2R
#define BOXQF_B8_MAX2R 88                // blocks of 8 positions below this delay (56: up to 48 registers; 72: up to 64; 88: up to 80)
#endif
template <int KS, int MODE, int B = 16>
int launch_boxqf_ks(const Run& r, const float* srcW, unsigned gap, float* dstW, float* dstO, const float* data,
                    int n, int C, int ld, int rad, size_t sws_img, size_t dws, size_t ws_data, int64_t W, uint8_t* nanflag) {
    const hipError_t attr = lds_optin(reinterpret_cast<const void*>(&k_boxqf<KS, MODE, B>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    HIPCHK(attr);
    const BoxDenom denom = box_reciprocal(box_denominator(rad));
    dim3 grid((unsigned)cdiv(C, 64), (unsigned)W);
    hipLaunchKernelGGL((k_boxqf<KS, MODE, B>), grid, dim3(512), boxqf_lds_bytes(B, boxqf_dbl(KS, B)), r.st, srcW, gap, dstW, dstO, data, n, C, ld, rad,
                       denom, sws_img, dws, ws_data, nanflag);
    LAUNCHCHK();
    return TRI_OK;
}

template <int MODE>
int launch_boxf(const Run& r, int ks, const float* srcW, const float* srcO, float* dstW, float* dstO, const float* data,
                int n, int C, int ld, int rad, size_t sws_img, size_t dws, size_t ws_data, int64_t W, uint8_t* nanflag) {
    {
        static const bool off = [] { const char* e = getenv("TRI_FILTER_NO_PIPE_F"); return e && e[0] == '1'; }();
        const int kq = boxq_pick_ks(rad);
        // Few, long lines (an SKA slab: 64 windows of 512 lines x 65536 channels): the one-wave-per-32-lines kernels (K4r) fill
        // half the machine at best; the stage pipeline (a workgroup of eight waves per 64 lines) then takes the small radii too.
        const bool few = (int64_t)W * cdiv(C, 32) < BOXQF_FEW_WAVES;
        const bool want = g_boxq_override == 1 || (g_boxq_override < 0 && !off && (2 * rad >= BOXQF_MIN_2R || (few && 2 * rad >= 16)));
        if (want && kq > 0 && srcO > srcW && n % 4 == 0 && ld % 4 == 0 && sws_img % 4 == 0 && (uint64_t)(srcO - srcW) % 4 == 0 &&
            ((uintptr_t)srcW % 16 == 0) && ((uint64_t)(srcO - srcW) + (uint64_t)C * ld) * 4u < (1ull << 31) &&
            (uint64_t)n * (uint64_t)C * 4u < (1ull << 31)) {
            const unsigned gap = (unsigned)(srcO - srcW);
            // blocks of 8 positions (two workgroups per CU) while the delay line leaves the registers for it
            static const int b8 = [] { const char* e = getenv("TRI_FILTER_PIPE_F_B8"); return e ? atoi(e) : BOXQF_B8_DEFAULT; }();
            if (b8 && g_boxq_b8_override != 0 && 2 * rad >= 16 && 2 * rad < BOXQF_B8_MAX2R) {
                switch (2 * rad / 8 * 8) {
                    case 16: return launch_boxqf_ks<16, MODE, 8>(r, srcW, gap, dstW, dstO, data, n, C, ld, rad, sws_img, dws, ws_data, W, nanflag);
                    case 24: return launch_boxqf_ks<24, MODE, 8>(r, srcW, gap, dstW, dstO, data, n, C, ld, rad, sws_img, dws, ws_data, W, nanflag);
                    case 32: return launch_boxqf_ks<32, MODE, 8>(r, srcW, gap, dstW, dstO, data, n, C, ld, rad, sws_img, dws, ws_data, W, nanflag);
                    case 40: return launch_boxqf_ks<40, MODE, 8>(r, srcW, gap, dstW, dstO, data, n, C, ld, rad, sws_img, dws, ws_data, W, nanflag);
                    case 48: return launch_boxqf_ks<48, MODE, 8>(r, srcW, gap, dstW, dstO, data, n, C, ld, rad, sws_img, dws, ws_data, W, nanflag);
#if BOXQF_B8_MAX2R > 56
                    case 56: return launch_boxqf_ks<56, MODE, 8>(r, srcW, gap, dstW, dstO, data, n, C, ld, rad, sws_img, dws, ws_data, W, nanflag);
                    case 64: return launch_boxqf_ks<64, MODE, 8>(r, srcW, gap, dstW, dstO, data, n, C, ld, rad, sws_img, dws, ws_data, W, nanflag);
#endif
#if BOXQF_B8_MAX2R > 72
                    case 72: return launch_boxqf_ks<72, MODE, 8>(r, srcW, gap, dstW, dstO, data, n, C, ld, rad, sws_img, dws, ws_data, W, nanflag);
                    case 80:                                // (MODE 2 would spill 11 registers at 128: blocks of 16 for it)
                        if (MODE == 1) return launch_boxqf_ks<80, 1, 8>(r, srcW, gap, dstW, dstO, data, n, C, ld, rad, sws_img, dws, ws_data, W, nanflag);
                        break;
#endif
#if BOXQF_B8_MAX2R > 88
                    case 96: return launch_boxqf_ks<96, MODE, 8>(r, srcW, gap, dstW, dstO, data, n, C, ld, rad, sws_img, dws, ws_data, W, nanflag);
#endif
                }
            }
            switch (kq) {
                case 16: return launch_boxqf_ks<16, MODE>(r, srcW, gap, dstW, dstO, data, n, C, ld, rad, sws_img, dws, ws_data, W, nanflag);
                case 32: return launch_boxqf_ks<32, MODE>(r, srcW, gap, dstW, dstO, data, n, C, ld, rad, sws_img, dws, ws_data, W, nanflag);
                case 48: return launch_boxqf_ks<48, MODE>(r, srcW, gap, dstW, dstO, data, n, C, ld, rad, sws_img, dws, ws_data, W, nanflag);
                case 64: return launch_boxqf_ks<64, MODE>(r, srcW, gap, dstW, dstO, data, n, C, ld, rad, sws_img, dws, ws_data, W, nanflag);
                case 80: return launch_boxqf_ks<80, MODE>(r, srcW, gap, dstW, dstO, data, n, C, ld, rad, sws_img, dws, ws_data, W, nanflag);
                case 96: return launch_boxqf_ks<96, MODE>(r, srcW, gap, dstW, dstO, data, n, C, ld, rad, sws_img, dws, ws_data, W, nanflag);
            }
        }
    }
    switch (ks) {
        case 8: return launch_boxf_ks<8, MODE>(r, srcW, srcO, dstW, dstO, data, n, C, ld, rad, sws_img, dws, ws_data, W, nanflag);
        case 16: return launch_boxf_ks<16, MODE>(r, srcW, srcO, dstW, dstO, data, n, C, ld, rad, sws_img, dws, ws_data, W, nanflag);
        case 32: return launch_boxf_ks<32, MODE>(r, srcW, srcO, dstW, dstO, data, n, C, ld, rad, sws_img, dws, ws_data, W, nanflag);
        case 64: return launch_boxf_ks<64, MODE>(r, srcW, srcO, dstW, dstO, data, n, C, ld, rad, sws_img, dws, ws_data, W, nanflag);
        case 80: return launch_boxf_ks<80, MODE>(r, srcW, srcO, dstW, dstO, data, n, C, ld, rad, sws_img, dws, ws_data, W, nanflag);
    }
    return set_err(TRI_EINVAL, "no register-ring kernel for %d slots", ks);
}

// K4x: the frequency-axis stage + masked division for any radius, one line pair resident in LDS, lanes =
// positions, exactness checked per pass with a sequential redo (kernels_boxexact.hpp).  Rows in, rows out.
// The flagger takes it where no stage pipeline exists (r >= 56: final_st_very_broad's 277 / 221 / 166 / 110);
// TRI_FILTER_NO_EXACT=1 restores the lane-per-stage / in-place multi-pass routes (A/B runs, tests).
thread_local int g_boxx_override = -1;   // tests / benches: 0 = off, 1 = on wherever the shape allows
thread_local unsigned long long g_boxx_last_stats[2] = {0, 0};
#ifndef BOXX_MIN_R
#define BOXX_MIN_R 56
#endif
// chunk length (low 8 bits) and threads per image (128 or 256, << 8) for a line of n positions, or 0.
// Long chunks on 128 threads per image where the line fits them (fewer instructions per line: see the kernel);
// TRI_BOXX_NTI=256 forces the short chunks (A/B runs).
int boxx_pick_l(int rad, int n) {
    static const bool env_off = [] { const char* e = getenv("TRI_FILTER_NO_EXACT"); return e && e[0] == '1'; }();
    static const int force_nti = [] { const char* e = getenv("TRI_BOXX_NTI"); return e ? atoi(e) : 0; }();
    if (g_boxx_override == 0 || (g_boxx_override < 0 && (env_off || rad < BOXX_MIN_R))) return 0;
    if (rad < 1 || n % 4 != 0) return 0;
    const int64_t P = (int64_t)n + 4 * (int64_t)rad;
    struct Cand { int nti, l; };
    static const Cand cands[] = {{128, 37}, {128, 41}, {256, 17}, {256, 19}, {256, 21}, {256, 25}};
    for (const Cand& c : cands) {
        if (force_nti && c.nti != force_nti) continue;
        if (c.nti == 128 && (P <= 128 * 17 || rad < 64)) continue;        // (short lines / small radii: the short chunks do)
        if ((int64_t)c.nti * c.l < P || c.l > 2 * rad + 1 || (2 * rad + 1) / c.l > BOXX_AMAX) continue;
        if (boxx_lds_bytes(c.nti, c.l, rad) > 159 * 1024) continue;
        return c.l | (c.nti << 8);
    }
    return 0;
}
// the reciprocal division is verified exhaustively for these radii only (test_division_by_box_denominator)
static bool boxx_recip_ok(int rad) { return rad <= 128 || rad == 166 || rad == 221 || rad == 277 || rad == 397 || rad == 795; }

template <int NTI, int L, int MODE>
int launch_boxx_l(const Run& r, const float* srcW, unsigned gap, const float* data, const uint8_t* mask, float* outA, float* outB,
                  int n, int C, int ld, int rad, size_t sws_img, size_t ws_data, size_t ws_mask, size_t ws_outA, size_t ws_outB,
                  int64_t W, uint8_t* nanflag, unsigned long long* stats) {
    const BoxDenom denom = box_reciprocal(box_denominator(rad));
    const size_t lds = boxx_lds_bytes(NTI, L, rad);
    dim3 grid((unsigned)C, (unsigned)W);
    // (the kernel also has a few static LDS bytes -- __syncthreads_or -- so 160 KB of dynamic LDS is refused: ask for 159)
    if (boxx_recip_ok(rad)) {
        HIPCHK(lds_optin(reinterpret_cast<const void*>(&k_boxx<NTI, L, MODE, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
        hipLaunchKernelGGL((k_boxx<NTI, L, MODE, true>), grid, dim3(2 * NTI), lds, r.st, srcW, gap, data, mask, outA, outB, n, ld, rad, denom, sws_img,
                           ws_data, ws_mask, ws_outA, ws_outB, nanflag, stats);
    } else {
        HIPCHK(lds_optin(reinterpret_cast<const void*>(&k_boxx<NTI, L, MODE, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
        hipLaunchKernelGGL((k_boxx<NTI, L, MODE, false>), grid, dim3(2 * NTI), lds, r.st, srcW, gap, data, mask, outA, outB, n, ld, rad, denom, sws_img,
                           ws_data, ws_mask, ws_outA, ws_outB, nanflag, stats);
    }
    LAUNCHCHK();
    return TRI_OK;
}
// rows of (srcW, srcO) [C][ld] + data rows [C][n] (+ byte mask rows) -> outA (MODE 1: |data - bg|; MODE 2: bg), outB (MODE 2: data - bg),
// rows [C][ld]; outputs may alias the inputs (a workgroup has its whole line in LDS before it writes)
template <int MODE>
int launch_boxx(const Run& r, int L, const float* srcW, const float* srcO, const float* data, const uint8_t* mask, float* outA, float* outB,
                int n, int C, int ld, int rad, size_t sws_img, size_t ws_data, size_t ws_mask, size_t ws_outA, size_t ws_outB,
                int64_t W, uint8_t* nanflag, unsigned long long* stats = nullptr) {
    if (srcO <= srcW || ld % 4 != 0 || sws_img % 4 != 0 || (uint64_t)(srcO - srcW) % 4 != 0 || ((uintptr_t)srcW % 16 != 0) ||
        ((uintptr_t)data % 16 != 0) || ws_data % 4 != 0 || ((uintptr_t)outA % 16 != 0) || ws_outA % 4 != 0 ||
        (mask && (((uintptr_t)mask % 4 != 0) || ws_mask % 4 != 0)) || (uint64_t)(srcO - srcW) > 0xffffffffull ||
        (MODE == 2 && (((uintptr_t)outB % 16 != 0) || ws_outB % 4 != 0)) || C > 65535 * 1024)
        return set_err(TRI_EUNSUPPORTED, "exact row filter: unaligned images");
    const unsigned gap = (unsigned)(srcO - srcW);
#define BOXX_CASE(NTI, LL) case ((NTI) << 8 | (LL)): return launch_boxx_l<NTI, LL, MODE>(r, srcW, gap, data, mask, outA, outB, n, C, ld, rad, sws_img, ws_data, ws_mask, ws_outA, ws_outB, W, nanflag, stats);
    switch (L) {
        BOXX_CASE(128, 37) BOXX_CASE(128, 41)
        BOXX_CASE(256, 17) BOXX_CASE(256, 19) BOXX_CASE(256, 21) BOXX_CASE(256, 25)
    }
#undef BOXX_CASE
    return set_err(TRI_EINVAL, "no exact row filter for code %d", L);
}

// Lane-per-stage variant of the same (radii 17..LANE4_R_MAX): k_colfilter_lane4<3, *>.
bool colfilter_t4_usable(int rad) {
    static const bool off = [] { const char* e = getenv("TRI_FILTER_NO_TIN"); return e && e[0] == '1'; }();
    return !off && colfilter_use_lane4(rad);
}

int launch_colfilter_t4(const Run& r, const float* srcW, const float* srcO, float* dstW, float* dstO,
                        int n, int C, int ld, int rad, size_t sws_img, size_t dws, int64_t W, float* deferred_denom) {
    float denom = box_denominator(rad);
    size_t lds = ((size_t)lane4_ring_capacity(rad) * 64 + 32 * 17) * sizeof(float);
    const hipError_t attr = [] {
        hipError_t e = lds_optin(reinterpret_cast<const void*>(&k_colfilter_lane4<3, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = lds_optin(reinterpret_cast<const void*>(&k_colfilter_lane4<3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return e;
    }();
    HIPCHK(attr);
    dim3 grid((unsigned)cdiv(C, 16), (unsigned)W, 2);
    if (deferred_denom) {
        *deferred_denom = denom;
        hipLaunchKernelGGL((k_colfilter_lane4<3, false>), grid, dim3(64), lds, r.st, srcW, srcO, (const float*)nullptr, (const uint8_t*)nullptr,
                           dstW, dstO, n, C, rad, denom, sws_img, (size_t)ld, dws);
    } else {
        hipLaunchKernelGGL((k_colfilter_lane4<3, true>), grid, dim3(64), lds, r.st, srcW, srcO, (const float*)nullptr, (const uint8_t*)nullptr,
                           dstW, dstO, n, C, rad, denom, sws_img, (size_t)ld, dws);
    }
    LAUNCHCHK();
    return TRI_OK;
}

// _get_background2d (flagging.py:516-579) for the median spectra of the batch,
// held in spectrum layout [Fa][Wb] (line axis = channel, column = window).
// Result: rows [0,Fa) of ws.so hold the background.
// Per (window, chunk) medians of a spectrum-layout image [Fa][Wb]: the segments run along the slow axis there
// (stride Wb floats), so the image and its flags are first turned into rows [Wb][Fa] (a few MB) and the
// segments become contiguous.  Output smed[w][g].
int spectrum_medians(const Run& r, const float* img, const uint8_t* flg) {
    const Plan& pl = r.pl;
    const Ws& ws = r.ws;
    const int Fa = (int)pl.Fa, Wn = (int)r.Wb, G = (int)pl.G;
    int rc = launch_transpose<float>(r, img, ws.srowsf, Fa, Wn, 0, 0, 1);
    if (rc) return rc;
    rc = launch_transpose<uint8_t>(r, flg, ws.srowsu, Fa, Wn, 0, 0, 1);
    if (rc) return rc;
    bool seg4 = Fa % 4 == 0;
    for (int64_t g = 0; g < pl.G + 1 && seg4; g++) seg4 = r.p->chunk_ends[g] % 4 == 0;
    return launch_median(r, ws.srowsf, ws.srowsu, ws.smed, 0, 0, (size_t)Fa, 1, ws.segC_start, ws.segC_len, Wn, G, 1, pl.maxchunk,
                         seg4, Fa % 4 == 0, seg4);
}

int spectrum_background(const Run& r) {
    const Plan& pl = r.pl;
    const Ws& ws = r.ws;
    int Fa = (int)pl.Fa, Wn = (int)r.Wb, G = (int)pl.G;
    size_t nS = (size_t)Fa * Wn;
    hipLaunchKernelGGL(k_copy_u8, dim3((unsigned)cdiv(nS, 256)), dim3(256), 0, r.st, ws.sflags, ws.sbgf, nS);
    LAUNCHCHK();
    double rej = TRI_MAD_NORMAL * r.p->background_reject;  // flagging.py:568
    // the exact row filter (K4x) is picked PER ITERATION (boxx_pick_l depends on the radius: a large first radius may not fit
    // its LDS line while a later, smaller one does), so the row copy of the spectra is made by the first iteration that
    // takes that route, whichever it is (it used to be made by the first iteration only)
    bool rowD_valid = false;
    for (int ext = pl.nit; ext >= 0; ext--) {
        bool final_pass = ext == 0;
        double sigma = (double)(final_pass ? 1 : ext) * r.p->spike_width_freq;  // flagging.py:554, 576
        int rad = (int)box_radius(sigma);
        // Radii beyond the stage pipeline (K4p): the spectra as ROWS through the exact row filter (K4x) -- the in-place
        // multi-pass kernel walks 4 x (Fa + 4 r) positions with one wave per 64 windows: 5.8 ms per launch whatever
        // the batch.  The 2-D time-stage scratch (ws.Aw ...) is idle here and holds the four row images.
        const int xl = (rad > 0 && 2 * (size_t)pl.PT >= 4) ? boxx_pick_l(rad, Fa) : 0;
        if (xl > 0) {
            float* rowW = ws.Aw;                       // [Wn][Fa] weight rows, then the signed residual (unused)
            float* rowO = ws.Aw + nS;                  // weight * data rows
            float* rowD = ws.Aw + 2 * nS;              // the spectra themselves
            float* rowR = ws.Aw + 3 * nS;              // result rows
            hipLaunchKernelGGL(k_build_wo, grid1(nS, 1), dim3(256), 0, r.st, ws.sdata, ws.sbgf, ws.sw, ws.so, nS, (size_t)0, (size_t)0);
            LAUNCHCHK();
            int rc = launch_transpose<float>(r, ws.sw, rowW, Fa, Wn, 0, 0, 1);
            if (!rc) rc = launch_transpose<float>(r, ws.so, rowO, Fa, Wn, 0, 0, 1);
            if (!rc && !rowD_valid) rc = launch_transpose<float>(r, ws.sdata, rowD, Fa, Wn, 0, 0, 1);
            if (rc) return rc;
            rowD_valid = true;
            if (final_pass)
                rc = launch_boxx<2>(r, xl, rowW, rowO, rowD, nullptr, rowR, rowW, Fa, Wn, Fa, rad, 0, 0, 0, 0, 0, 1, reinterpret_cast<uint8_t*>(ws.rowcnt));
            else
                rc = launch_boxx<1>(r, xl, rowW, rowO, rowD, nullptr, rowR, nullptr, Fa, Wn, Fa, rad, 0, 0, 0, 0, 0, 1, nullptr);
            if (!rc) rc = launch_transpose<float>(r, rowR, ws.so, Wn, Fa, 0, 0, 1);
            if (rc) return rc;
            if (!final_pass) {
                rc = spectrum_medians(r, ws.so, ws.sbgf);
                if (rc) return rc;
                hipLaunchKernelGGL(k_reject<false>, grid1(nS, 1), dim3(256), 0, r.st, ws.so, ws.sbgf, ws.smed, ws.d_chunk_of, rej, Fa, Wn, G, (size_t)0, (size_t)0);
                LAUNCHCHK();
            }
            continue;
        }
        if (rad > 0) {
            int rc = launch_colfilter(r, 0, ws.sw, ws.so, ws.sdata, ws.sbgf, ws.sw, ws.so, Fa, Wn, rad, 0, 0, 0, 1, nullptr, false, true);
            if (rc) return rc;
        } else {
            hipLaunchKernelGGL(k_build_wo, grid1(nS, 1), dim3(256), 0, r.st, ws.sdata, ws.sbgf, ws.sw, ws.so, nS, (size_t)0, (size_t)0);
            LAUNCHCHK();
        }
        if (final_pass) {
            { int rc2 = launch_masked_div<0>(r, ws.sw, ws.so, ws.sdata, nS, 0, 0, 1); if (rc2) return rc2; }
        } else {
            { int rc2 = launch_masked_div<1>(r, ws.sw, ws.so, ws.sdata, nS, 0, 0, 1); if (rc2) return rc2; }
            // per (window, chunk) median of the residual: element (f, w) at f*Wn + w
            // -> row = w (RS 1), element stride Wn
            int rc = spectrum_medians(r, ws.so, ws.sbgf);
            if (rc) return rc;
            hipLaunchKernelGGL(k_reject<false>, grid1(nS, 1), dim3(256), 0, r.st, ws.so, ws.sbgf, ws.smed, ws.d_chunk_of, rej, Fa, Wn, G, (size_t)0, (size_t)0);
            LAUNCHCHK();
        }
    }
    return launch_interp(r, ws.so, Fa, Wn, (size_t)0, 1, (const uint8_t*)nullptr, (const float*)nullptr, (size_t)0, (float*)nullptr);
}

// _get_background2d (flagging.py:516-579) for every window of the batch.
// In: dataTF / dataFT, flagsTF (spectral flags already OR-ed in).  Out: the
// background in FT layout in rows [0,Fa) of ws.Bo and the residual
// data - background in rows [0,Fa) of ws.Bw (window stride PF*T).
static bool bg_flags_packed(int T, size_t N) {
    static const bool no_pack = [] { const char* e = getenv("TRI_NO_PACKED_FLAGS"); return e && e[0] == '1'; }();
    return !no_pack && (T % 4 == 0) && (N % 4 == 0);
}

int background2d(const Run& r, bool flagsFT_current) {
    const Plan& pl = r.pl;
    const Ws& ws = r.ws;
    int T = (int)pl.T, Fa = (int)pl.Fa, G = (int)pl.G;
    int64_t W = r.Wb;
    size_t N = (size_t)T * Fa;
    size_t wsA = 2 * (size_t)pl.PT * Fa, wsB = (size_t)pl.PF * T;   // Aw / Ao interleaved per window (carve)
    // Background flags live in FT layout (ws.bgfFT).  For the time-axis stage
    // they are needed per (time, channel) column thread: either as a TF byte
    // image (general case) or, when T % 4 == 0, packed four times per word
    // ("TF4": [T/4][Fa] uint32) -- which is exactly the 32-bit transpose of the
    // FT byte image viewed as [Fa][T/4] words.
    const bool packed = bg_flags_packed(T, N);
    int rc = flagsFT_current ? launch_u8<0>(r, ws.flagsFT, ws.bgfFT, N, N, N, W)
                             : launch_transpose<uint8_t>(r, ws.flagsTF, ws.bgfFT, T, Fa, N, N, W);
    if (rc) return rc;
    if (packed) {
        rc = launch_transpose<float>(r, reinterpret_cast<const float*>(ws.bgfFT), reinterpret_cast<float*>(ws.bgfTF), Fa, T / 4, N / 4, N / 4, W);
        if (rc) return rc;
    } else {
        rc = launch_u8<0>(r, ws.flagsTF, ws.bgfTF, N, N, N, W);
        if (rc) return rc;
    }
    double rej = TRI_MAD_NORMAL * r.p->background_reject;
    bool bgf_ft_stale = false;           // the FT flag bytes lag behind the TF4 words (rows-only rejection iterations)
    // K3r (round 4): block median + rejection in ONE pass over |data - background| (kernels_reject.hpp).  It writes the updated FT
    // flags to a SECOND image (its fallback redoes a block from the input flags), so the FT flag image alternates between
    // ws.bgfFT and ws.fflFT (free until the frequency-axis SumThreshold writes it).  TRI_FUSED_MEDREJ=1 switches it on.
    // (opt-in for now, TRI_FUSED_MEDREJ=1: one workgroup per block does not stream fast enough -- 3.7 against 3.0 ms per 252
    //  windows for the two kernels, scripts/ubench/medrej_dev.hip)
    static const bool no_medrej = [] { const char* e = getenv("TRI_FUSED_MEDREJ"); return !(e && e[0] == '1'); }();
    static const int medrej_fallback = [] { const char* e = getenv("TRI_MEDREJ_FORCE_FALLBACK"); return e ? atoi(e) : 0; }();
    uint8_t* cur_ft = ws.bgfFT;
    uint8_t* alt_ft = ws.fflFT;
    for (int ext = pl.nit; ext >= 0; ext--) {
        bool final_pass = ext == 0;
        double e = (double)(final_pass ? 1 : ext);
        int r0 = (int)box_radius(e * r.p->spike_width_time);
        int r1 = (int)box_radius(e * r.p->spike_width_freq);
        // --- time axis (TF layout: line = time, column = channel) ---
        // Building the weight / data images first (13 B/sample, vectorised) and
        // streaming float images through the sequential kernel is faster than
        // byte loads of the flags inside its per-line loop.
        static const bool prebuild = [] { const char* e = getenv("TRI_TIME_PREBUILD"); return !(e && e[0] == '0'); }();
        float den_t = 0.0f, den_f = 0.0f;   // divisions deferred to the transposes / masked_div
        bool direct_ft = false;
        // frequency stage able to read the time stage's TF images itself (no transposes)
        // register-ring fused frequency stage (signed 32-bit buffer offsets: window below 2^31 bytes)
        const int ksf = ((uint64_t)N * 4u < (1ull << 31) && ((uint64_t)(ws.Ao - ws.Aw) + N) * 4u < (1ull << 31)) ? boxr_pick_ks_f(r1) : 0;
        // exact row filter (K4x) where the stage pipelines end: rows of the time stage's TF images in, rows out
        const int xl = (r1 > 0 && (!ksf || r1 >= BOXX_MIN_R) && (!r.ampl_cached || r.data_mask)) ? boxx_pick_l(r1, Fa) : 0;
        // ... whose rejection iterations then stay in the row layout altogether (TRI_FILTER_NO_TF_REJECT=1: transposes + FT kernels)
        static const bool no_tfr = [] { const char* e = getenv("TRI_FILTER_NO_TF_REJECT"); return e && e[0] == '1'; }();
        const bool tf_native = xl > 0 && !final_pass && packed && !no_tfr && G <= 65535 && (uint64_t)T * Fa * 4u < (1ull << 31);
        // (who reads the FT bytes: a time stage that cannot take the TF4 words, and the median / rejection of an FT-native
        //  iteration; the final pass has neither median nor rejection)
        const bool time_from_tf4 = r0 > 0 && packed && colfilter_lds_block(r0, Fa) > 0;
        if (bgf_ft_stale && (!time_from_tf4 || (!final_pass && !tf_native))) {
            // an iteration that reads the FT flag bytes follows rows-only ones: they are the 32-bit transpose of the TF4 words
            rc = launch_transpose<float>(r, reinterpret_cast<const float*>(ws.bgfTF), reinterpret_cast<float*>(cur_ft), T / 4, Fa, N / 4, N / 4, W);
            if (rc) return rc;
            bgf_ft_stale = false;
        }
        const bool tin4 = !xl && !ksf && !colfilter_t_usable(r1) && colfilter_t4_usable(r1);
        const bool tin = xl > 0 || ksf > 0 || colfilter_t_usable(r1) || tin4;
        float* den_t_ptr = tin ? nullptr : &den_t;
        // (for the in-place multi-pass kernel, used at large radii, building on
        //  the fly measured faster: 11.4 vs 16.8 ms per call at 128 windows)
        if (r0 > 0 && packed && colfilter_lds_block(r0, Fa) > 0) {
            // single sweep straight from (data, packed flags): no image build
            rc = launch_colfilter(r, 2, ws.Aw, ws.Ao, ws.dataTF, ws.bgfTF, ws.Aw, ws.Ao, T, Fa, r0, wsA, N, wsA, W, den_t_ptr, false, true);
            if (rc) return rc;
        } else if (r0 > 0 && !packed && prebuild && colfilter_lds_block(r0, Fa) > 0) {
            const bool lds_path = true;
            const size_t boff = lds_path ? 0 : (size_t)4 * r0 * Fa;
            if (N % 4 == 0 && wsA % 4 == 0)
                hipLaunchKernelGGL(k_build_wo4, grid1(N / 4, W), dim3(256), 0, r.st, ws.dataTF, ws.bgfTF, ws.Aw + boff, ws.Ao + boff, N / 4, N, wsA);
            else
                hipLaunchKernelGGL(k_build_wo, grid1(N, W), dim3(256), 0, r.st, ws.dataTF, ws.bgfTF, ws.Aw + boff, ws.Ao + boff, N, N, wsA);
            LAUNCHCHK();
            // (TRI_FILTER_DIRECT_FT=1: write the filtered images straight into FT
            //  layout; measured no faster than the two transposes it replaces)
            static const bool want_direct = [] { const char* e = getenv("TRI_FILTER_DIRECT_FT"); return e && e[0] == '1'; }();
            direct_ft = lds_path && want_direct && (T % 4 == 0) && (wsB % 4 == 0);
            if (direct_ft) {
                size_t off2 = colfilter_lds_block(r1, T) > 0 ? 0 : (size_t)4 * r1 * T;
                rc = launch_colfilter(r, 1, ws.Aw, ws.Ao, nullptr, nullptr, ws.Bw + off2, ws.Bo + off2, T, Fa, r0, wsA, 0, wsB, W, nullptr, true);
            } else {
                rc = launch_colfilter(r, 1, ws.Aw, ws.Ao, nullptr, nullptr, ws.Aw, ws.Ao, T, Fa, r0, wsA, 0, wsA, W, den_t_ptr, false, true);
            }
            if (rc) return rc;
        } else if (r0 > 0) {
            const uint8_t* fl = ws.bgfTF;
            if (packed) {   // byte image needed: rebuild it next to the packed one
                rc = launch_transpose<uint8_t>(r, cur_ft, ws.comb, Fa, T, N, N, W);
                if (rc) return rc;
                fl = ws.comb;
            }
            rc = launch_colfilter(r, 0, ws.Aw, ws.Ao, ws.dataTF, fl, ws.Aw, ws.Ao, T, Fa, r0, wsA, N, wsA, W);
            if (rc) return rc;
        } else {
            const uint8_t* fl = ws.bgfTF;
            if (packed) {
                rc = launch_transpose<uint8_t>(r, cur_ft, ws.comb, Fa, T, N, N, W);
                if (rc) return rc;
                fl = ws.comb;
            }
            hipLaunchKernelGGL(k_build_wo, grid1(N, W), dim3(256), 0, r.st, ws.dataTF, fl, ws.Aw, ws.Ao, N, N, wsA);
            LAUNCHCHK();
        }
        // --- to FT layout: rows [4 r1, 4 r1 + Fa) of the padded buffers for the
        //     in-place multi-pass filter, rows [0, Fa) for the single-sweep one ---
        size_t off = colfilter_lds_block(r1, T) > 0 ? 0 : (size_t)4 * r1 * T;
        // frequency stage + masked division in one kernel when the four-ring stage applies
        const bool fused_div = tin && !tin4 && !direct_ft && (xl > 0 || ksf > 0 || colfilter_tf_usable(r1));
        if (xl > 0 && !direct_ft) {
            // rows are filtered in place (Ao <- |data - bg| or bg, Aw <- data - bg), then taken to the FT layout
            const uint8_t* mk = r.ampl_cached ? r.data_mask : nullptr;
            if (final_pass) {
                HIPCHK(hipMemsetAsync(ws.rowcnt, 0, (size_t)W * T, r.st));
                rc = launch_boxx<2>(r, xl, ws.Aw, ws.Ao, ws.dataTF, mk, ws.Ao, ws.Aw, Fa, T, Fa, r1, wsA, N, r.data_mask_ws, wsA, wsA, W,
                                    reinterpret_cast<uint8_t*>(ws.rowcnt));
                if (rc) return rc;
                rc = launch_transpose<float>(r, ws.Aw, ws.Bw, T, Fa, wsA, wsB, W);
            } else {
                rc = launch_boxx<1>(r, xl, ws.Aw, ws.Ao, ws.dataTF, mk, ws.Ao, nullptr, Fa, T, Fa, r1, wsA, N, r.data_mask_ws, wsA, 0, W, nullptr);
                if (rc) return rc;
                if (tf_native) {
                    // the rejection where the rows lie: block medians over (all times) x (chunk channels) of the row image
                    // with the TF4 flag words, rejection straight into those words -- no FT image of the residual or of
                    // the flags in this iteration (the FT flags are rebuilt when an FT-native iteration follows)
                    static const bool no_predict = [] { const char* e = getenv("TRI_MEDIAN_NO_PREDICT"); return e && e[0] == '1'; }();
                    unsigned* gc = reinterpret_cast<unsigned*>(ws.Aw);          // the weight rows are dead: candidate scratch
                    unsigned cap = (unsigned)(std::min<size_t>((wsA / 2) / (size_t)G, 0x7fffffffu) & ~(size_t)3);
                    if (no_predict || (int64_t)cap < pl.maxchunk * pl.T || wsA % 4 != 0) { gc = nullptr; cap = 0; }
                    hipLaunchKernelGGL((k_median2<false, true>), dim3((unsigned)G, (unsigned)W), dim3(256), 0, r.st, (const float*)ws.Ao,
                                       (const uint8_t*)ws.bgfTF, ws.med, wsA, N, (size_t)0, (size_t)1, ws.segC_start, ws.segC_len, 1, G,
                                       gc, gc ? wsA : (size_t)0, cap, T / 4, Fa);
                    hipLaunchKernelGGL(k_reject_tf, dim3((unsigned)cdiv(Fa, 256), (unsigned)(T / 4), (unsigned)W), dim3(256), 0, r.st,
                                       (const float*)ws.Ao, reinterpret_cast<unsigned*>(ws.bgfTF), ws.med, ws.d_chunk_of, rej, T / 4, Fa, Fa, G,
                                       wsA, N / 4);
                    LAUNCHCHK();
                    bgf_ft_stale = true;
                    continue;
                }
            }
            if (rc) return rc;
            rc = launch_transpose<float>(r, ws.Ao, ws.Bo, T, Fa, wsA, wsB, W);
            if (rc) return rc;
        } else if (fused_div) {
            if (final_pass) {
                HIPCHK(hipMemsetAsync(ws.rowcnt, 0, (size_t)W * T, r.st));
                if (ksf) rc = launch_boxf<2>(r, ksf, ws.Aw, ws.Ao, ws.Bw, ws.Bo, ws.dataFT, Fa, T, Fa, r1, wsA, wsB, N, W, reinterpret_cast<uint8_t*>(ws.rowcnt));
                else rc = launch_colfilter_tf<2>(r, ws.Aw, ws.Ao, ws.Bw, ws.Bo, ws.dataFT, Fa, T, Fa, r1, wsA, wsB, N, W, reinterpret_cast<uint8_t*>(ws.rowcnt));
            } else {
                if (ksf) rc = launch_boxf<1>(r, ksf, ws.Aw, ws.Ao, ws.Bw, ws.Bo, ws.dataFT, Fa, T, Fa, r1, wsA, wsB, N, W, nullptr);
                else rc = launch_colfilter_tf<1>(r, ws.Aw, ws.Ao, ws.Bw, ws.Bo, ws.dataFT, Fa, T, Fa, r1, wsA, wsB, N, W, nullptr);
            }
            if (rc) return rc;
        } else if (tin && !direct_ft) {
            if (tin4) rc = launch_colfilter_t4(r, ws.Aw, ws.Ao, ws.Bw, ws.Bo, Fa, T, Fa, r1, wsA, wsB, W, &den_f);
            else rc = launch_colfilter_t(r, ws.Aw, ws.Ao, ws.Bw, ws.Bo, Fa, T, Fa, r1, wsA, wsB, W, &den_f);
            if (rc) return rc;
        } else if (!direct_ft) {
            rc = launch_transpose<float>(r, ws.Aw, ws.Bw + off, T, Fa, wsA, wsB, W, den_t);
            if (rc) return rc;
            rc = launch_transpose<float>(r, ws.Ao, ws.Bo + off, T, Fa, wsA, wsB, W, den_t);
            if (rc) return rc;
        }
        // --- frequency axis (FT layout: line = channel, column = time) ---
        if (r1 > 0 && !(tin && !direct_ft)) {
            rc = launch_colfilter(r, 1, ws.Bw, ws.Bo, nullptr, nullptr, ws.Bw, ws.Bo, Fa, T, r1, wsB, 0, wsB, W, &den_f);
            if (rc) return rc;
        }
        if (final_pass) {
            // ws.rowcnt (W * T ints, idle until the end of the iteration) doubles as the
            // per-line "background holds a NaN" marker
            if (!fused_div) {
                HIPCHK(hipMemsetAsync(ws.rowcnt, 0, (size_t)W * T, r.st));
                rc = launch_masked_div<2>(r, ws.Bw, ws.Bo, ws.dataFT, N, wsB, N, W, den_f, reinterpret_cast<uint8_t*>(ws.rowcnt), T);
                if (rc) return rc;
            }
        } else {
            if (!fused_div) {
                rc = launch_masked_div<1>(r, ws.Bw, ws.Bo, ws.dataFT, N, wsB, N, W, den_f);
                if (rc) return rc;
            }
            static const bool no_fuse = [] { const char* e = getenv("TRI_NO_FUSED_REJECT"); return e && e[0] == '1'; }();
            static const bool no_tile = [] { const char* e = getenv("TRI_NO_TILE_MEDREJ"); return e && e[0] == '1'; }();
            if (r.pl.vec && wsB % 4 == 0 && wsA % 4 == 0 && packed && !no_fuse && !no_tile && no_medrej && G <= 65535) {
                // K3t: median + rejection + TF4 re-pack in one pass over |data - background|, tile-parallel (kernels_reject_tile.hpp):
                // predict (per block) -> pass (per 64 x 64-word tile) -> finish (per block) -> redo of the blocks that failed a check.
                // Scratch: the dead time-stage images of the window.
                const size_t nb = (size_t)pl.maxchunk * T;                 // samples of the largest block
                const size_t ccap = (nb / 4) & ~(size_t)3, ucap = (nb / 8) & ~(size_t)3;
                int ytiles = 0;
                for (int g = 0; g < G; g++) ytiles += (int)cdiv(r.p->chunk_ends[g + 1] - r.p->chunk_ends[g], 64);
                // (blocks too small to predict from -- fewer than 65536 samples -- would all take the redo path: the two kernels below)
                if (mrt_scratch_words(G, ccap, ucap) <= wsA && (medrej_fallback || (pl.maxchunk - 1) * (int64_t)T >= 65536) && ytiles > 0 && ytiles <= 65535) {
                    unsigned* sc = reinterpret_cast<unsigned*>(ws.Aw);
                    hipLaunchKernelGGL(k_mr_predict, dim3((unsigned)G, (unsigned)W), dim3(256), 0, r.st, (const float*)ws.Bo, (const uint8_t*)cur_ft,
                                       ws.d_chunk_ends, rej, T / 4, G, wsB, N, sc, wsA, ccap, ucap, medrej_fallback);
                    for (unsigned round = 1; round <= 2; round++) {        // (round 2: only the blocks whose median missed the predicted window)
                        hipLaunchKernelGGL(k_mr_pass, dim3((unsigned)cdiv(T / 4, 64), (unsigned)ytiles, (unsigned)W), dim3(256), 0, r.st, (const float*)ws.Bo,
                                           (const uint8_t*)cur_ft, alt_ft, ws.bgfTF, ws.d_chunk_ends, Fa, T / 4, G, wsB, N, sc, wsA, ccap, ucap, round);
                        hipLaunchKernelGGL(k_mr_finish, dim3((unsigned)G, (unsigned)W), dim3(256), 0, r.st, (const float*)ws.Bo, (const uint8_t*)cur_ft, alt_ft,
                                           ws.bgfTF, ws.med, ws.d_chunk_ends, rej, Fa, T / 4, G, wsB, N, sc, wsA, ccap, ucap, round);
                    }
                    hipLaunchKernelGGL(k_median_reject, dim3((unsigned)G, (unsigned)W), dim3(256), 0, r.st, (const float*)ws.Bo, (const uint8_t*)cur_ft,
                                       alt_ft, ws.bgfTF, ws.med, ws.d_chunk_ends, rej, Fa, T / 4, G, wsB, N, sc, wsA, 0u, 0u, 1,
                                       (const unsigned*)sc, wsA, (int)MRT_PARW, 11);
                    LAUNCHCHK();
                    std::swap(cur_ft, alt_ft);
                    continue;
                }
            }
            {
                // one pass: median + rejection + TF4 re-pack (K3r).  Scratch per (window, chunk) block: the dead time-stage
                // images, half for the window's keys, half for the undecided samples' indices
                const size_t per_block = (wsA / (size_t)G) & ~(size_t)7;
                const unsigned capq = (unsigned)std::min<size_t>(per_block / 2, 0x3ffffffcu);
                if (r.pl.vec && wsB % 4 == 0 && wsA % 4 == 0 && packed && !no_fuse && !no_medrej && capq >= 1024 && G <= 65535 &&
                    (uint64_t)N * 4u < (1ull << 32)) {
                    hipLaunchKernelGGL(k_median_reject, dim3((unsigned)G, (unsigned)W), dim3(256), 0, r.st, (const float*)ws.Bo, (const uint8_t*)cur_ft,
                                       alt_ft, ws.bgfTF, ws.med, ws.d_chunk_ends, rej, Fa, T / 4, G, wsB, N,
                                       reinterpret_cast<unsigned*>(ws.Aw), wsA, capq, capq, medrej_fallback);
                    LAUNCHCHK();
                    std::swap(cur_ft, alt_ft);
                    continue;
                }
            }
            // block medians over (all times) x (chunk channels): contiguous in FT
            // (the time stage's images in ws.Aw / ws.Ao are dead here: candidate scratch, wsA / G keys per block)
            rc = launch_median(r, ws.Bo, cur_ft, ws.med, wsB, N, 0, 1, ws.segB_start, ws.segB_len, 1, G, W, pl.maxchunk * pl.T,
                               T % 4 == 0 && wsB % 4 == 0 && N % 4 == 0, false, false,
                               reinterpret_cast<unsigned*>(ws.Aw), wsA, (unsigned)(std::min<size_t>(wsA / (size_t)G, 0x7fffffffu) & ~(size_t)3));
            if (rc) return rc;
            if (r.pl.vec && wsB % 4 == 0 && packed && !no_fuse) {
                // rejection + TF4 re-pack of the flags in one pass
                hipLaunchKernelGGL(k_reject4_t, dim3((unsigned)cdiv(T / 4, 64), (unsigned)cdiv(Fa, 64), (unsigned)W), dim3(64, 4), 0, r.st,
                                   ws.Bo, cur_ft, ws.bgfTF, ws.med, ws.d_chunk_of, rej, Fa, T / 4, G, wsB, N);
                LAUNCHCHK();
                continue;
            }
            if (r.pl.vec && wsB % 4 == 0)
                hipLaunchKernelGGL(k_reject4, grid1(N / 4, W), dim3(256), 0, r.st, ws.Bo, cur_ft, ws.med, ws.d_chunk_of, rej, T / 4, G, N / 4, wsB, N);
            else
                hipLaunchKernelGGL(k_reject<true>, grid1(N, W), dim3(256), 0, r.st, ws.Bo, cur_ft, ws.med, ws.d_chunk_of, rej, Fa, T, G, wsB, N);
            LAUNCHCHK();
            if (packed)
                rc = launch_transpose<float>(r, reinterpret_cast<const float*>(cur_ft), reinterpret_cast<float*>(ws.bgfTF), Fa, T / 4, N / 4, N / 4, W);
            else
                rc = launch_transpose<uint8_t>(r, cur_ft, ws.bgfTF, Fa, T, N, N, W);
            if (rc) return rc;
        }
    }
    return launch_interp(r, ws.Bo, Fa, T, wsB, W, reinterpret_cast<const uint8_t*>(ws.rowcnt), (const float*)ws.dataFT, N, ws.Bw);
}

// One major iteration (_get_flags_impl, flagging.py:745-781) for a batch.
template <int VD>
int run_iteration(Run& r, const void* vis, uint8_t* iter_flags, uint8_t* out_flags, bool update_iter, bool tap) {
    const Plan& pl = r.pl;
    const Ws& ws = r.ws;
    const tri_params* p = r.p;
    int T = (int)pl.T, F = (int)pl.F, Fa = (int)pl.Fa, G = (int)pl.G;
    int64_t W = r.Wb;
    int Wn = (int)W;
    size_t N = (size_t)T * Fa, NF = (size_t)T * F;
    size_t wsB = (size_t)pl.PF * T;
    int rc;

    // flagging.py:756  _average_freq
    static const bool no_fused_begin = [] { const char* e = getenv("TRI_NO_FUSED_BEGIN"); return e && e[0] == '1'; }();
    const bool fused_begin = r.ampl_cached && !no_fused_begin && T % 4 == 0 && Fa % 4 == 0 && N % 4 == 0 &&
                             ((uintptr_t)iter_flags % 4 == 0);
    if (fused_begin) {
        // flagsTF = running flags (NaNs already folded in), flagsFT = their transpose,
        // cached FT amplitudes masked -- one pass over the flags
        dim3 gridw((unsigned)cdiv(Fa, 64), (unsigned)cdiv(T, 128), (unsigned)W);
        hipLaunchKernelGGL((k_transpose_u8w<true, true>), gridw, dim3(256), 0, r.st, iter_flags, ws.flagsFT, ws.flagsTF, ws.dataFT,
                           T, Fa, N, N, N, N);
        LAUNCHCHK();
    } else if (r.ampl_cached) {
        // amplitudes (both layouts) were made once for the batch; this iteration's
        // flags = running flags (NaNs already folded in)
        rc = launch_u8<0>(r, iter_flags, ws.flagsTF, N, N, N, W);
        if (rc) return rc;
    } else {
        if (pl.vec)
            hipLaunchKernelGGL(k_prepare4<VD>, dim3((unsigned)cdiv(N * (size_t)W / 4, 256)), dim3(256), 0, r.st, vis, iter_flags, ws.dataTF, ws.flagsTF, N * (size_t)W / 4);
        else
            hipLaunchKernelGGL(k_prepare<VD>, grid1(N, W), dim3(256), 0, r.st, vis, iter_flags, ws.dataTF, ws.flagsTF, T, F, Fa, (int)pl.avg);
        LAUNCHCHK();
        rc = launch_transpose<float>(r, ws.dataTF, ws.dataFT, T, Fa, N, N, W);
        if (rc) return rc;
    }
    if (!fused_begin) {
        rc = launch_transpose<uint8_t>(r, ws.flagsTF, ws.flagsFT, T, Fa, N, N, W);
        if (rc) return rc;
    }
    if (r.ampl_cached && !fused_begin) {
        hipLaunchKernelGGL(k_zero_flagged4, grid1(N / 4, W), dim3(256), 0, r.st, ws.flagsFT, ws.dataFT, N / 4, N, N);
        LAUNCHCHK();
    }

    // flagging.py:944  _time_median: rows of the FT layout are contiguous in time
    rc = launch_median(r, ws.dataFT, ws.flagsFT, ws.med, N, N, (size_t)T, 1, ws.segT_start, ws.segT_len, Fa, 1, W, pl.T, false, T % 4 == 0, true);
    if (rc) return rc;
    hipLaunchKernelGGL(k_spec_from_med, dim3((unsigned)cdiv((size_t)Fa * Wn, 256)), dim3(256), 0, r.st, ws.med, ws.sdata, ws.sflags, Fa, Wn);
    LAUNCHCHK();

    // flagging.py:945-952  spectrum background, subtraction, SumThreshold
    rc = spectrum_background(r);
    if (rc) return rc;
    size_t nS = (size_t)Fa * Wn;
    rc = launch_sub(r, ws.sdata, ws.so, ws.sres, nS, 0, 0, 0, 1);
    if (rc) return rc;
    rc = spectrum_medians(r, ws.sres, ws.sflags);
    if (rc) return rc;
    rc = launch_colst(r, pl.swF, ws.sres, ws.smed, ws.sout, ws.d_chunk_ends, Fa, Wn, G, 0, 0, 1);
    if (rc) return rc;

    // flagging.py:954  flags |= spec_flags
    // With whole 16-byte groups along time the FT copy of the flags is updated
    // in place (rows of flagged channels only) instead of being transposed again.
    static const bool no_ft_or = [] { const char* e = getenv("TRI_NO_FT_SPEC_OR"); return e && e[0] == '1'; }();
    const bool ft_current = pl.vec && !no_ft_or && T % 16 == 0;
    // ... and when the background works from the FT image alone (packed flags), the TF image is
    // not read before the time flags are OR-ed in: both updates then share one pass.
    const bool defer_tf = ft_current && bg_flags_packed(T, N);
    if (pl.vec) {
        hipLaunchKernelGGL(k_spec_rows, dim3((unsigned)cdiv(nS, 256)), dim3(256), 0, r.st, ws.sout, ws.srows, Fa, Wn);
        if (!defer_tf)
            hipLaunchKernelGGL(k_or_spec16, grid1(N / 16, W), dim3(256), 0, r.st, ws.flagsTF, ws.srows, T, Fa / 16);
        if (ft_current)
            hipLaunchKernelGGL(k_or_spec_ft16, grid1(N / 16, W), dim3(256), 0, r.st, ws.flagsFT, ws.srows, T / 16, Fa);
    } else {
        hipLaunchKernelGGL(k_or_spec, grid1(N, W), dim3(256), 0, r.st, ws.flagsTF, ws.sout, T, Fa, Wn);
    }
    LAUNCHCHK();

    // flagging.py:957-962  2-D background (FT layout, ws.Bo), then the residual
    // (K4x masks rows of the unmasked TF amplitudes with the iteration's input flags: same layout when nothing is averaged)
    r.data_mask = (r.ampl_cached && Fa == F) ? iter_flags : nullptr;
    r.data_mask_ws = N;
    rc = background2d(r, ft_current);
    if (rc) return rc;
    if (tap && r.dbg) {
        HIPCHK(hipMemcpyAsync(r.dbg->f32 + Fa, ws.Bo, N * sizeof(float), hipMemcpyDeviceToDevice, r.st));
    }
    // the residual data - background was written by the final masked division
    // (and redone by the interpolation pass on repaired lines) into ws.Bw
    float* residFT = ws.Bw;   // window stride wsB
    float* residTF = ws.Aw;   // window stride N (the time-axis scratch is free again)
    // Column panels for the time-axis SumThreshold (round 4): the TF residual and the time flags as [Fa / 64][T][64], so the
    // column kernel's row walk is one linear stream.  Who else touches the two images reads panels too: the frequency-axis MAD
    // (wave medians over row segments: an aligned group of four channels is contiguous either way) and the fused combine /
    // dilate pass.  Only on the route where exactly those kernels run (TRI_ST_NO_PANEL=1: plain rows everywhere).
    static const bool no_panel = [] { const char* e = getenv("TRI_ST_NO_PANEL"); return e && e[0] == '1'; }();
    static const bool no_fused_or_p = [] { const char* e = getenv("TRI_NO_FUSED_OR"); return e && e[0] == '1'; }();
    static const bool no_fdil_p = [] { const char* e = getenv("TRI_NO_FUSED_DILATE"); return e && e[0] == '1'; }();
    const bool panel_t = !no_panel && defer_tf && !no_fused_or_p && !no_fdil_p && Fa % 64 == 0 && Fa == F &&
                         median_takes_extra_flags(pl.maxchunk + 3) && st_use_fused(pl.swT) && st_use_mask(T, Fa) &&
                         [&] { int64_t e = p->freq_extend; int64_t h = e >= 0 ? e / 2 : -((-e + 1) / 2); return -h == -1 && -h + e == 2; }();
    rc = launch_transpose<float>(r, residFT, residTF, Fa, T, wsB, N, W, 0.0f, panel_t);
    if (rc) return rc;

    // flagging.py:964  SumThreshold along time.  MAD per channel over time =
    // contiguous rows of the FT layout; flags = input | spectral flags.
    if (!ft_current) {
        rc = launch_transpose<uint8_t>(r, ws.flagsTF, ws.flagsFT, T, Fa, N, N, W);
        if (rc) return rc;
    }
    rc = launch_median(r, residFT, ws.flagsFT, ws.med, wsB, N, (size_t)T, 1, ws.segT_start, ws.segT_len, Fa, 1, W, pl.T, false, T % 4 == 0, true);
    if (rc) return rc;
    rc = launch_colst(r, pl.swT, residTF, ws.med, ws.tflTF, ws.d_tends, T, Fa, 1, N, N, W, panel_t);
    if (rc) return rc;

    // flagging.py:967-969  flags |= time_flags; SumThreshold along frequency.
    // MAD per (time, chunk) = contiguous row segments of the TF layout.
    // (the union is read by this MAD only -- the next major iteration rebuilds the TF flags -- so when the chunk medians are
    //  wave medians they take the three sources as they are and no pass writes the union; TRI_NO_FUSED_OR=1: the pass)
    static const bool no_fused_or = [] { const char* e = getenv("TRI_NO_FUSED_OR"); return e && e[0] == '1'; }();
    if (defer_tf && !no_fused_or && median_takes_extra_flags(pl.maxchunk) && Fa % 4 == 0) {
        rc = launch_median(r, residTF, ws.flagsTF, ws.med, N, N, (size_t)Fa, 1, ws.segC_start, ws.segC_len, T, G, W, pl.maxchunk, false, Fa % 4 == 0,
                           false, nullptr, 0, 0, ws.tflTF, ws.srows, (size_t)Fa, panel_t ? T : 0);
        if (rc) return rc;
    } else {
        if (panel_t) return set_err(TRI_EUNSUPPORTED, "internal: panel images on a route that reads rows");
        if (defer_tf) {
            hipLaunchKernelGGL(k_or_spec_more16, grid1(N / 16, W), dim3(256), 0, r.st, ws.flagsTF, ws.srows, ws.tflTF, T, Fa / 16);
            LAUNCHCHK();
        } else {
            rc = launch_u8<1>(r, ws.tflTF, ws.flagsTF, N, N, N, W);
            if (rc) return rc;
        }
        rc = launch_median(r, residTF, ws.flagsTF, ws.med, N, N, (size_t)Fa, 1, ws.segC_start, ws.segC_len, T, G, W, pl.maxchunk, false, Fa % 4 == 0);
        if (rc) return rc;
    }
    rc = launch_colst(r, pl.swF, residFT, ws.med, ws.fflFT, ws.d_chunk_ends, Fa, T, G, wsB, N, W);
    if (rc) return rc;
    rc = launch_transpose<uint8_t>(r, ws.fflFT, ws.fflTF, Fa, T, N, N, W);
    if (rc) return rc;

    if (tap && r.dbg) {
        hipLaunchKernelGGL(k_gather_col_f32, dim3((unsigned)cdiv(Fa, 256)), dim3(256), 0, r.st, ws.sres, r.dbg->f32, Fa, Wn, 0);
        hipLaunchKernelGGL(k_gather_col_u8, dim3((unsigned)cdiv(Fa, 256)), dim3(256), 0, r.st, ws.sout, r.dbg->u8, Fa, Wn, 0);
        LAUNCHCHK();
        if (panel_t) {
            hipLaunchKernelGGL(k_unpanel<float>, dim3((unsigned)cdiv(N, 256)), dim3(256), 0, r.st, (const float*)residTF, r.dbg->f32 + Fa + N, T, Fa);
            hipLaunchKernelGGL(k_unpanel<uint8_t>, dim3((unsigned)cdiv(N, 256)), dim3(256), 0, r.st, (const uint8_t*)ws.tflTF, r.dbg->u8 + Fa, T, Fa);
            LAUNCHCHK();
        } else {
            HIPCHK(hipMemcpyAsync(r.dbg->f32 + Fa + N, residTF, N * sizeof(float), hipMemcpyDeviceToDevice, r.st));
            HIPCHK(hipMemcpyAsync(r.dbg->u8 + Fa, ws.tflTF, N, hipMemcpyDeviceToDevice, r.st));
        }
        HIPCHK(hipMemcpyAsync(r.dbg->u8 + Fa + N, ws.fflTF, N, hipMemcpyDeviceToDevice, r.st));
    }

    // flagging.py:973  _combine_flags (time smearing), flagging.py:975  _unaverage_freq
    // (replication, frequency smearing, counts)
    {
        int64_t e = p->time_extend;
        int64_t half = e >= 0 ? e / 2 : -((-e + 1) / 2);   // Python floor division
        int lo = (int)-half, hi = (int)(-half + e);
        int64_t ef = p->freq_extend;
        int64_t halff = ef >= 0 ? ef / 2 : -((-ef + 1) / 2);
        int flo = (int)-halff, fhi = (int)(-halff + ef);
        static const bool no_fuse = [] { const char* e = getenv("TRI_NO_FUSED_DILATE"); return e && e[0] == '1'; }();
        HIPCHK(hipMemsetAsync(ws.rowcnt, 0, (size_t)W * T * sizeof(int), r.st));
        if (pl.vec && flo == -1 && fhi == 2 && !no_fuse) {
            // both smearings in one pass, no intermediate image
            dim3 grid((unsigned)cdiv(F / 16, 64), (unsigned)T, (unsigned)W);
            if (panel_t) hipLaunchKernelGGL(k_combine_dilate16<true>, grid, dim3(64), 0, r.st, ws.srows, ws.tflTF, ws.fflTF, ws.dil, ws.rowcnt, T, F / 16, lo, hi);
            else hipLaunchKernelGGL(k_combine_dilate16<false>, grid, dim3(64), 0, r.st, ws.srows, ws.tflTF, ws.fflTF, ws.dil, ws.rowcnt, T, F / 16, lo, hi);
            hipLaunchKernelGGL(k_colcount, grid1(F / 4, W), dim3(256), 0, r.st, ws.dil, ws.colcnt, T, F / 4);
        } else {
            if (panel_t) return set_err(TRI_EUNSUPPORTED, "internal: panel images on a route that reads rows");
            if (pl.vec)
                hipLaunchKernelGGL(k_combine16, grid1(N / 16, W), dim3(256), 0, r.st, ws.srows, ws.tflTF, ws.fflTF, ws.comb, T, Fa / 16, lo, hi);
            else
                hipLaunchKernelGGL(k_combine, grid1(N, W), dim3(256), 0, r.st, ws.sout, ws.tflTF, ws.fflTF, ws.comb, T, Fa, Wn, lo, hi);
            LAUNCHCHK();
            if (pl.vec && flo == -1 && fhi == 2) {
                dim3 grid((unsigned)cdiv(F / 16, 64), (unsigned)T, (unsigned)W);
                hipLaunchKernelGGL((k_unaverage16<-1, 3>), grid, dim3(64), 0, r.st, ws.comb, ws.dil, ws.rowcnt, T, F / 16);
                hipLaunchKernelGGL(k_colcount, grid1(F / 4, W), dim3(256), 0, r.st, ws.dil, ws.colcnt, T, F / 4);
            } else {
                HIPCHK(hipMemsetAsync(ws.colcnt, 0, (size_t)W * F * sizeof(int), r.st));
                dim3 grid((unsigned)cdiv(F, 256), (unsigned)T, (unsigned)W);
                hipLaunchKernelGGL(k_unaverage, grid, dim3(256), 0, r.st, ws.comb, ws.dil, ws.rowcnt, ws.colcnt, T, Fa, F, (int)pl.avg, flo, fhi);
            }
        }
        LAUNCHCHK();
    }
    // flagging.py:910-918 whole-row / whole-column rules; :777-781 NaN OR; :1193
    double row_limit = p->flag_all_freq_frac * (double)F;
    double col_limit = (double)T * p->flag_all_time_frac;
    if (pl.vec)
        if (r.ampl_cached)   // isnan(|vis|) from the cached amplitudes: half the bytes of the complex visibilities
            hipLaunchKernelGGL(k_final16<TRI_VIS_NANMASK>, grid1(NF / 16, W), dim3(256), 0, r.st, ws.dil, ws.rowcnt, ws.colcnt, (const void*)ws.nanmask, out_flags, iter_flags, T, F / 16, row_limit, col_limit, update_iter ? 1 : 0);
        else
            hipLaunchKernelGGL(k_final16<VD>, grid1(NF / 16, W), dim3(256), 0, r.st, ws.dil, ws.rowcnt, ws.colcnt, vis, out_flags, iter_flags, T, F / 16, row_limit, col_limit, update_iter ? 1 : 0);
    else
        hipLaunchKernelGGL(k_final<VD>, grid1(NF, W), dim3(256), 0, r.st, ws.dil, ws.rowcnt, ws.colcnt, vis, out_flags, iter_flags, T, F, row_limit, col_limit, update_iter ? 1 : 0);
    LAUNCHCHK();
    return TRI_OK;
}

// All batches of the windows [w_begin, w_end) on r.st with the workspace [wsp, wsp + ws_bytes).
int process_windows(Run& r, const void* vis, int vis_dtype, const uint8_t* flags, uint8_t* out_flags,
                    int64_t w_begin, int64_t w_end, int64_t T, int64_t F, void* wsp, size_t ws_bytes, bool tap_first) {
    const tri_params* p = r.p;
    Ws probe;
    carve(r.pl, 1, nullptr, 0, true, &probe);
    if (probe.total > ws_bytes)
        return set_err(TRI_EWORKSPACE, "workspace of %zu bytes is smaller than the %zu needed for one window", ws_bytes, probe.total);
    // largest batch the workspace admits (the carve-up is monotone in Wb)
    int64_t lo = 1, hi = std::min<int64_t>(w_end - w_begin, 16384);
    while (lo < hi) {
        int64_t mid = (lo + hi + 1) / 2;
        carve(r.pl, mid, nullptr, 0, true, &probe);
        if (probe.total <= ws_bytes) lo = mid; else hi = mid - 1;
    }
    const int64_t Wb = lo;
    carve(r.pl, Wb, wsp, ws_bytes, false, &r.ws);

    // device tables
    ChunkTab ct;
    ct.n = (int)p->n_chunk_ends;
    for (int i = 0; i < ct.n; i++) ct.ends[i] = p->chunk_ends[i];
    {
        int nthr = (int)std::max<int64_t>(r.pl.Fa, ct.n);
        hipLaunchKernelGGL(k_tables, dim3((unsigned)cdiv(nthr, 256)), dim3(256), 0, r.st, ct, (int)T, (int)r.pl.Fa,
                           r.ws.d_chunk_ends, r.ws.d_tends, r.ws.segC_start, r.ws.segC_len, r.ws.segB_start,
                           r.ws.segB_len, r.ws.segT_start, r.ws.segT_len, r.ws.d_chunk_of);
        LAUNCHCHK();
    }
    const size_t NF = (size_t)T * F;
    const size_t esz = (vis_dtype == TRI_VIS_C64 || vis_dtype == TRI_VIS_F64) ? 8 : 4;
    int rc = TRI_OK;
    for (int64_t w0 = w_begin; w0 < w_end; w0 += Wb) {
        r.Wb = std::min(Wb, w_end - w0);
        const char* vis_b = (const char*)vis + (size_t)w0 * NF * esz;
        uint8_t* out_b = out_flags + (size_t)w0 * NF;
        // flagging.py:1182  iter_flags = flags.copy()  (non-zero = flagged)
        rc = launch_u8<2>(r, flags + (size_t)w0 * NF, r.ws.iter, NF, NF, NF, r.Wb);
        if (rc) return rc;
        // without channel averaging |vis| is the same in every major iteration:
        // compute it once, in both layouts (TRI_NO_AMPL_CACHE=1 recomputes it per iteration)
        static const bool no_cache = [] { const char* e = getenv("TRI_NO_AMPL_CACHE"); return e && e[0] == '1'; }();
        r.ampl_cached = r.pl.vec && !no_cache && p->num_major_iterations > 0;
        if (r.ampl_cached) {
            const size_t n4 = (size_t)r.Wb * NF / 4;
            if (vis_dtype == TRI_VIS_C64)
                hipLaunchKernelGGL(k_amplitude4<TRI_VIS_C64>, dim3((unsigned)cdiv(n4, 256)), dim3(256), 0, r.st, (const void*)vis_b, r.ws.dataTF, r.ws.iter, r.ws.nanmask, n4);
            else
                hipLaunchKernelGGL(k_amplitude4<TRI_VIS_F32>, dim3((unsigned)cdiv(n4, 256)), dim3(256), 0, r.st, (const void*)vis_b, r.ws.dataTF, r.ws.iter, r.ws.nanmask, n4);
            LAUNCHCHK();
            rc = launch_transpose<float>(r, r.ws.dataTF, r.ws.dataFT, (int)T, (int)r.pl.Fa, NF, NF, r.Wb);
            if (rc) return rc;
        }
        for (int64_t it = 0; it < p->num_major_iterations; it++) {
            bool last = it == p->num_major_iterations - 1;
            bool tap = last && tap_first && w0 == w_begin;
            if (vis_dtype == TRI_VIS_C64) rc = run_iteration<TRI_VIS_C64>(r, vis_b, r.ws.iter, out_b, !last, tap);
            else if (vis_dtype == TRI_VIS_F64) rc = run_iteration<TRI_VIS_F64>(r, vis_b, r.ws.iter, out_b, !last, tap);
            else rc = run_iteration<TRI_VIS_F32>(r, vis_b, r.ws.iter, out_b, !last, tap);
            if (rc) return rc;
        }
    }
    return TRI_OK;
}

// Optional schedule (TRI_SUBSTREAMS=1): two internal streams per calling thread (created once).  A window set
// whose 1-D spectrum path is a long sequential recurrence on a handful of lanes (SKA-sized windows:
// 65536-channel lines, one lane per window) is split in two halves that run concurrently -- the high-priority
// half at full speed, the other one in the gaps its spectrum kernels leave on the device.
struct SubStreams {
    hipStream_t st[2] = {nullptr, nullptr};
    hipEvent_t fork = nullptr, join[2] = {nullptr, nullptr};
    int device = -1;
};
thread_local SubStreams g_sub;

int get_substreams(SubStreams** out) {
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    if (g_sub.device != dev) {
        if (g_sub.device >= 0) return set_err(TRI_EUNSUPPORTED, "a calling thread must stay on one device");
        int lo = 0, hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));    // lo = least, hi = greatest priority (numerically lower)
        HIPCHK(hipStreamCreateWithPriority(&g_sub.st[0], hipStreamNonBlocking, hi));
        HIPCHK(hipStreamCreateWithPriority(&g_sub.st[1], hipStreamNonBlocking, lo));
        HIPCHK(hipEventCreateWithFlags(&g_sub.fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&g_sub.join[0], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&g_sub.join[1], hipEventDisableTiming));
        g_sub.device = dev;
    }
    *out = &g_sub;
    return TRI_OK;
}

int flagger_impl(const void* vis, int vis_dtype, const uint8_t* flags, uint8_t* out_flags,
                 int64_t n_cp, int64_t T, int64_t F, const tri_params* p, void* workspace,
                 size_t workspace_bytes, void* stream, Debug* dbg) {
    if (!vis || !flags || !out_flags || !p) return set_err(TRI_EINVAL, "NULL pointer argument");
    if (n_cp < 0) return set_err(TRI_EINVAL, "negative window count");
    if (vis_dtype != TRI_VIS_C64 && vis_dtype != TRI_VIS_F32 && vis_dtype != TRI_VIS_F64)
        return set_err(TRI_EUNSUPPORTED, "vis dtype must be complex64, float32 or float64 amplitudes");
    Run r;
    r.st = (hipStream_t)stream;
    r.p = p;
    r.dbg = dbg;
    int rc = make_plan(T, F, p, &r.pl);
    if (rc) return rc;
    if (r.pl.G > TRI_MAX_CHUNKS) return set_err(TRI_EUNSUPPORTED, "at most %d frequency chunks", TRI_MAX_CHUNKS);
    // the 16-byte kernels need 16-byte aligned user buffers (torch allocations are)
    if ((((uintptr_t)vis) | ((uintptr_t)flags) | ((uintptr_t)out_flags)) & 15) r.pl.vec = false;
    if (vis_dtype == TRI_VIS_F64) r.pl.vec = false;       // float64 amplitudes: the scalar preparation / final kernels (rare input type)
    if (n_cp == 0) return TRI_OK;
    if (!workspace) return set_err(TRI_EWORKSPACE, "NULL workspace");
    if (((uintptr_t)workspace & 255) != 0) return set_err(TRI_EINVAL, "workspace must be 256-byte aligned");
    if (p->num_major_iterations == 0)
        HIPCHK(hipMemsetAsync(out_flags, 0, (size_t)n_cp * (size_t)T * F, r.st));

    // TRI_SUBSTREAMS=0 / 1 forces the single-stream / two-stream schedule
    const char* fe = getenv("TRI_SUBSTREAMS");      // read per call: tests toggle it
    const int force = fe ? atoi(fe) : -1;
    Ws probe;
    carve(r.pl, 1, nullptr, 0, true, &probe);
    const size_t half = (workspace_bytes / 2) & ~(size_t)255;
    const bool can_split = n_cp >= 2 && half >= probe.total && dbg == nullptr;
    // (off by default: on slabs of 64 SKA-sized windows the halves lose more parallelism in the 2-D kernels --
    //  512 lines per window on the frequency axis -- than the overlap of the spectrum kernels returns:
    //  1.5 against 2.5 Gvis/s, profiles/r02_bench_ska.log)
    const bool want_split = force == 1;
    if (!(can_split && want_split))
        return process_windows(r, vis, vis_dtype, flags, out_flags, 0, n_cp, T, F, workspace, workspace_bytes, true);

    SubStreams* ss = nullptr;
    rc = get_substreams(&ss);
    if (rc) return rc;
    HIPCHK(hipEventRecord(ss->fork, r.st));
    const int64_t mid = (n_cp + 1) / 2;
    for (int k = 0; k < 2 && rc == TRI_OK; k++) {
        Run rk;
        rk.st = ss->st[k];
        rk.p = p;
        rk.dbg = nullptr;
        rk.pl = r.pl;
        HIPCHK(hipStreamWaitEvent(rk.st, ss->fork, 0));
        rc = process_windows(rk, vis, vis_dtype, flags, out_flags, k == 0 ? 0 : mid, k == 0 ? mid : n_cp, T, F,
                             (char*)workspace + (size_t)k * half, half, false);
        // join even after an error: the caller's stream must not run ahead of work already queued
        (void)hipEventRecord(ss->join[k], rk.st);
        (void)hipStreamWaitEvent(r.st, ss->join[k], 0);
    }
    return rc;
}

}  // namespace

extern "C" int tri_sum_threshold_flagger(const void* vis, int vis_dtype, const uint8_t* flags,
                                         uint8_t* out_flags, int64_t n_cp, int64_t ntime,
                                         int64_t nchan, const tri_params* p, void* workspace,
                                         size_t workspace_bytes, void* stream) {
    return flagger_impl(vis, vis_dtype, flags, out_flags, n_cp, ntime, nchan, p, workspace,
                        workspace_bytes, stream, nullptr);
}

// Test hook: as above, additionally taps the last major iteration's
// intermediates of window 0 into caller-provided device buffers:
//   dbg_f32: [spec_resid (Fa)][background, FT layout (Fa*T)][residual, TF layout (T*Fa)]
//   dbg_u8 : [spec_flags (Fa)][time_flags TF (T*Fa)][freq_flags TF (T*Fa)]
extern "C" int tri_sum_threshold_flagger_debug(const void* vis, int vis_dtype, const uint8_t* flags,
                                               uint8_t* out_flags, int64_t n_cp, int64_t ntime,
                                               int64_t nchan, const tri_params* p, void* workspace,
                                               size_t workspace_bytes, void* stream,
                                               float* dbg_f32, uint8_t* dbg_u8) {
    Debug d{dbg_f32, dbg_u8};
    return flagger_impl(vis, vis_dtype, flags, out_flags, n_cp, ntime, nchan, p, workspace,
                        workspace_bytes, stream, (dbg_f32 && dbg_u8) ? &d : nullptr);
}

extern "C" int tri_abs_c64(const void* z, float* out, int64_t n, void* stream) {
    if (!z || !out || n < 0) return set_err(TRI_EINVAL, "bad argument");
    if (n == 0) return TRI_OK;
    hipLaunchKernelGGL(k_abs_c64, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (const float2*)z, out, (size_t)n);
    LAUNCHCHK();
    return TRI_OK;
}

extern "C" int tri_fill_windows(void* vis_windows_c64, uint8_t* flag_windows, int64_t n, void* stream) {
    if (!vis_windows_c64 || !flag_windows || n < 0) return set_err(TRI_EINVAL, "bad argument");
    if (n == 0) return TRI_OK;
    hipLaunchKernelGGL(k_fill_windows, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (float2*)vis_windows_c64, flag_windows, (size_t)n);
    LAUNCHCHK();
    return TRI_OK;
}

extern "C" int tri_pack_data(const void* data_c64, const uint8_t* flag, const int32_t* row_bl,
                             const int32_t* row_time, int64_t rows, int64_t nchan, int64_t ncorr,
                             int64_t nbl, int64_t ntime, void* vis_windows_c64,
                             uint8_t* flag_windows, void* stream) {
    if (!data_c64 || !flag || !row_bl || !row_time || !vis_windows_c64 || !flag_windows)
        return set_err(TRI_EINVAL, "NULL pointer argument");
    if (rows < 0 || nchan <= 0 || ncorr <= 0 || nbl < 0 || ntime < 0) return set_err(TRI_EINVAL, "bad shape");
    if (rows == 0) return TRI_OK;
    if (rows > 0x7FFFFFFF) return set_err(TRI_EUNSUPPORTED, "too many rows in one call");
    dim3 grid((unsigned)cdiv(nchan, 256), 1, 1);
    // gridDim.y is limited to 65535: walk the rows in slabs
    for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
        int64_t nr = std::min<int64_t>(65535, rows - r0);
        grid.y = (unsigned)nr;
        const float2* dsl = (const float2*)data_c64 + (size_t)r0 * nchan * ncorr;
        const uint8_t* fsl = flag + (size_t)r0 * nchan * ncorr;
        // 1 / 2 / 4 correlations with 16-byte aligned rows: vector loads of a thread's (chan, corr) piece
        const bool al = ((uintptr_t)data_c64 % 16 == 0) && ((uintptr_t)flag % 4 == 0);
        if (ncorr == 4 && al)
            hipLaunchKernelGGL(k_pack_v<4>, grid, dim3(256), 0, (hipStream_t)stream, dsl, fsl, row_bl + r0, row_time + r0,
                               (int)nchan, (int)nbl, (int)ntime, (float2*)vis_windows_c64, flag_windows);
        else if (ncorr == 2 && al)
            hipLaunchKernelGGL(k_pack_v<2>, grid, dim3(256), 0, (hipStream_t)stream, dsl, fsl, row_bl + r0, row_time + r0,
                               (int)nchan, (int)nbl, (int)ntime, (float2*)vis_windows_c64, flag_windows);
        else if (ncorr == 1)
            hipLaunchKernelGGL(k_pack_v<1>, grid, dim3(256), 0, (hipStream_t)stream, dsl, fsl, row_bl + r0, row_time + r0,
                               (int)nchan, (int)nbl, (int)ntime, (float2*)vis_windows_c64, flag_windows);
        else
            hipLaunchKernelGGL(k_pack, grid, dim3(256), 0, (hipStream_t)stream, dsl, fsl,
                               row_bl + r0, row_time + r0, (int)nchan, (int)ncorr, (int)nbl, (int)ntime,
                               (float2*)vis_windows_c64, flag_windows);
        LAUNCHCHK();
    }
    return TRI_OK;
}

extern "C" int tri_unpack_data(const uint8_t* flag_windows, const int32_t* row_bl,
                               const int32_t* row_time, int64_t rows, int64_t nchan, int64_t ncorr,
                               int64_t nbl, int64_t ntime, uint8_t* out_flags, int any_corr,
                               void* stream) {
    if (!flag_windows || !row_bl || !row_time || !out_flags) return set_err(TRI_EINVAL, "NULL pointer argument");
    if (rows < 0 || nchan <= 0 || ncorr <= 0 || nbl < 0 || ntime < 0) return set_err(TRI_EINVAL, "bad shape");
    dim3 grid((unsigned)cdiv(nchan, 256), 1, 1);
    for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
        int64_t nr = std::min<int64_t>(65535, rows - r0);
        grid.y = (unsigned)nr;
        uint8_t* osl = out_flags + (size_t)r0 * nchan * ncorr;
        const bool al = (uintptr_t)out_flags % 4 == 0;
        if (ncorr == 4 && al)
            hipLaunchKernelGGL(k_unpack_v<4>, grid, dim3(256), 0, (hipStream_t)stream, flag_windows, row_bl + r0, row_time + r0,
                               (int)nchan, (int)nbl, (int)ntime, osl, any_corr);
        else if (ncorr == 2 && al)
            hipLaunchKernelGGL(k_unpack_v<2>, grid, dim3(256), 0, (hipStream_t)stream, flag_windows, row_bl + r0, row_time + r0,
                               (int)nchan, (int)nbl, (int)ntime, osl, any_corr);
        else if (ncorr == 1)
            hipLaunchKernelGGL(k_unpack_v<1>, grid, dim3(256), 0, (hipStream_t)stream, flag_windows, row_bl + r0, row_time + r0,
                               (int)nchan, (int)nbl, (int)ntime, osl, any_corr);
        else
            hipLaunchKernelGGL(k_unpack, grid, dim3(256), 0, (hipStream_t)stream, flag_windows, row_bl + r0, row_time + r0,
                               (int)nchan, (int)ncorr, (int)nbl, (int)ntime, osl, any_corr);
        LAUNCHCHK();
    }
    return TRI_OK;
}

// Measurement / test hook: ONE rejection step of the background loop (flagging.py:553-574) as the flagger runs it for blocks
// of >= 65536 samples: k_mr_predict, two rounds of k_mr_pass + k_mr_finish, the redo kernel (K3t).  resid: (n_win, n_chan, n_time)
// float32 = |data - background| in the FT layout, flags_in: (n_win, n_chan, n_time) bytes; flags_out (FT bytes), flags_t4 (TF4 words,
// (n_win, n_time / 4, n_chan) x 4 bytes) and med (n_win, n_chunks) float64 receive the results.
extern "C" int tri_bench_reject(const float* resid, const uint8_t* flags_in, uint8_t* flags_out, uint8_t* flags_t4, double* med,
                                int64_t n_win, int64_t n_chan, int64_t n_time, const int64_t* chunk_ends, int64_t n_chunk_ends,
                                double reject, int repeats, float* ms_per_step, void* stream) {
    if (!resid || !flags_in || !flags_out || !flags_t4 || !med || !chunk_ends || !ms_per_step) return set_err(TRI_EINVAL, "NULL pointer argument");
    if (n_win <= 0 || n_win > 65535 || n_chan <= 0 || n_time <= 0 || n_time % 4 != 0 || n_chunk_ends < 2 || n_chunk_ends > 256 || repeats <= 0)
        return set_err(TRI_EINVAL, "bad shape");
    if ((uint64_t)n_chan * (uint64_t)n_time * 4u >= (1ull << 31)) return set_err(TRI_EUNSUPPORTED, "a window of 2^31 bytes or more");
    const int G = (int)n_chunk_ends - 1, T = (int)n_time, Fa = (int)n_chan;
    int64_t maxchunk = 0, minchunk = n_chan;
    int ytiles = 0;
    for (int g = 0; g < G; g++) {
        const int64_t c = chunk_ends[g + 1] - chunk_ends[g];
        if (c <= 0 || chunk_ends[g] < 0 || chunk_ends[g + 1] > n_chan) return set_err(TRI_EINVAL, "bad chunk");
        maxchunk = std::max(maxchunk, c); minchunk = std::min(minchunk, c);
        ytiles += (int)cdiv(c, 64);
    }
    if (chunk_ends[0] != 0 || chunk_ends[G] != n_chan) return set_err(TRI_EINVAL, "chunks must cover the channels");
    if (minchunk * (int64_t)T < 65536 || ytiles > 65535) return set_err(TRI_EUNSUPPORTED, "blocks below 65536 samples take the two-kernel route");
    hipStream_t st = (hipStream_t)stream;
    const size_t N = (size_t)Fa * T;
    const size_t nb = (size_t)maxchunk * T, ccap = (nb / 4) & ~(size_t)3, ucap = (nb / 8) & ~(size_t)3;
    const size_t wsS = mrt_scratch_words(G, ccap, ucap);
    unsigned* sc = nullptr;
    int64_t* d_ends = nullptr;
    HIPCHK(hipMalloc(&sc, (size_t)n_win * wsS * sizeof(unsigned)));
    HIPCHK(hipMalloc(&d_ends, (size_t)n_chunk_ends * sizeof(int64_t)));
    HIPCHK(hipMemcpyAsync(d_ends, chunk_ends, (size_t)n_chunk_ends * sizeof(int64_t), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    const double rej = TRI_MAD_NORMAL * reject;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    const unsigned W = (unsigned)n_win;
    HIPCHK(hipEventRecord(e0, st));
    for (int i = 0; i < repeats; i++) {
        hipLaunchKernelGGL(k_mr_predict, dim3((unsigned)G, W), dim3(256), 0, st, resid, flags_in, d_ends, rej, T / 4, G, N, N, sc, wsS, ccap, ucap, 0);
        for (unsigned round = 1; round <= 2; round++) {
            hipLaunchKernelGGL(k_mr_pass, dim3((unsigned)cdiv(T / 4, 64), (unsigned)ytiles, W), dim3(256), 0, st, resid, flags_in, flags_out, flags_t4,
                               d_ends, Fa, T / 4, G, N, N, sc, wsS, ccap, ucap, round);
            hipLaunchKernelGGL(k_mr_finish, dim3((unsigned)G, W), dim3(256), 0, st, resid, flags_in, flags_out, flags_t4, med, d_ends, rej, Fa, T / 4, G,
                               N, N, sc, wsS, ccap, ucap, round);
        }
        hipLaunchKernelGGL(k_median_reject, dim3((unsigned)G, W), dim3(256), 0, st, resid, flags_in, flags_out, flags_t4, med, d_ends, rej, Fa, T / 4, G,
                           N, N, sc, wsS, 0u, 0u, 1, (const unsigned*)sc, wsS, (int)MRT_PARW, 11);
    }
    HIPCHK(hipEventRecord(e1, st));
    hipError_t le = hipGetLastError();
    hipError_t se = hipEventSynchronize(e1);
    float ms = 0.f;
    if (le == hipSuccess && se == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(sc); (void)hipFree(d_ends);
    if (le != hipSuccess) return set_err(TRI_EHIP, "launch: %s", hipGetErrorString(le));
    if (se != hipSuccess) return set_err(TRI_EHIP, "sync: %s", hipGetErrorString(se));
    *ms_per_step = ms / (float)repeats;
    return TRI_OK;
}

extern "C" int tri_bench_sumthreshold(const float* data, const double* mad, uint8_t* out,
                                      int64_t n_win, int64_t n_line, int64_t n_col,
                                      const int64_t* windows, int64_t n_windows,
                                      double outlier_nsigma, double rho, int variant, int repeats,
                                      float* ms_per_launch, void* stream) {
    if (!data || !mad || !out || !windows || !ms_per_launch) return set_err(TRI_EINVAL, "NULL pointer argument");
    if (n_win <= 0 || n_line <= 0 || n_col <= 0 || repeats <= 0 || n_win > 65535) return set_err(TRI_EINVAL, "bad shape");
    hipStream_t st = (hipStream_t)stream;
    StWin sw;
    int rc = make_stwin(windows, n_windows, rho, &sw);
    if (rc) return rc;
    double thr_scale = outlier_nsigma * TRI_MAD_NORMAL;
    size_t nthreads = (size_t)n_win * n_col;
    double* ring = nullptr;
    uint8_t* acc = nullptr;
    int64_t* d_ends = nullptr;
    HIPCHK(hipMalloc(&ring, nthreads * sw.ringtot * sizeof(double)));
    HIPCHK(hipMalloc(&acc, nthreads * sw.acccap));
    HIPCHK(hipMalloc(&d_ends, 2 * sizeof(int64_t)));
    int64_t ends[2] = {0, n_line};
    HIPCHK(hipMemcpyAsync(d_ends, ends, sizeof(ends), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    int C = (int)n_col, L = (int)n_line;
    int blk = C >= 256 ? 256 : (C >= 128 ? 128 : 64);
    if (const char* e = getenv("TRI_ST_BLK")) { int b = atoi(e); if (b >= 64 && b <= ST_MAXBLK && C >= b) blk = b; }
    dim3 grid((unsigned)cdiv(C, blk), 1, (unsigned)n_win);
    size_t ws = (size_t)n_line * n_col;
    bool can_fuse = sw.nw == 4 && sw.w[0] == 1 && sw.w[1] == 2 && sw.w[2] == 4 && sw.w[3] == 8;
    if ((variant == 2 || variant == 3) && !can_fuse) return set_err(TRI_EUNSUPPORTED, "register cascade needs windows (1,2,4,8)");
    if (variant == 3 && !((uint64_t)L * (uint64_t)C * 4u < (1ull << 31)))
        return set_err(TRI_EUNSUPPORTED, "lane-mask cascade needs a window below 2^31 bytes");
    // variant 5 (round 4): the lane-mask cascade on COLUMN PANELS, as the flagger's time-axis pass runs it -- the row images
    // handed in are re-laid out before the timed region and the flags taken back to rows after it
    static const bool no_panel = [] { const char* e = getenv("TRI_ST_NO_PANEL"); return e && e[0] == '1'; }();
    if (variant == 5 && !(can_fuse && C % 64 == 0 && (uint64_t)L * (uint64_t)C * 4u < (1ull << 31)))
        return set_err(TRI_EUNSUPPORTED, "panel SumThreshold: windows (1,2,4,8), a multiple of 64 columns, a window below 2^31 bytes");
    if (variant == 0 && can_fuse) variant = st_use_mask(L, C) ? ((C % 64 == 0 && !no_panel) ? 5 : 3) : 2;
    float* pdata = nullptr;
    uint8_t* pout = nullptr;
    if (variant == 5) {
        HIPCHK(hipMalloc(&pdata, (size_t)n_win * ws * sizeof(float)));
        HIPCHK(hipMalloc(&pout, (size_t)n_win * ws));
        const bool klog = g_klog_on;
        g_klog_on = false;                                   // (re-layout helpers are not part of what is measured)
        hipLaunchKernelGGL(k_panelize<float>, dim3((unsigned)cdiv((int64_t)ws, 256), (unsigned)n_win), dim3(256), 0, st, data, pdata, L, C, ws);
        g_klog_on = klog;
        LAUNCHCHK();
    }
    if (variant == 4) {
        if (sw.nw > 8 || stp_lds_bytes(sw) > 160 * 1024) return set_err(TRI_EUNSUPPORTED, "stage pipeline: more than eight windows, or the flag ring does not fit LDS");
        HIPCHK(lds_optin(reinterpret_cast<const void*>(&k_colst_pipe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    bool fused = variant == 2;
    StFusedArgs fa;
    for (int j = 0; j < 4; j++) fa.tf[j] = sw.tf[j < sw.nw ? j : 0];
    HIPCHK(hipEventRecord(e0, st));
    for (int i = 0; i < repeats; i++) {
        if (variant == 5)
            hipLaunchKernelGGL((k_colst_mask<1, 2, 4, 8, true>), grid, dim3(blk), 0, st, (const float*)pdata, mad, pout, d_ends, fa, thr_scale, L, C, 1, ws, ws);
        else if (variant == 3)
            hipLaunchKernelGGL((k_colst_mask<1, 2, 4, 8>), grid, dim3(blk), 0, st, data, mad, out, d_ends, fa, thr_scale, L, C, 1, ws, ws);
        else if (fused)
            hipLaunchKernelGGL((k_colst_fused<1, 2, 4, 8>), grid, dim3(blk), 0, st, data, mad, out, d_ends, fa, thr_scale, L, C, 1, ws, ws);
        else if (variant == 4)
            hipLaunchKernelGGL(k_colst_pipe, dim3((unsigned)cdiv(C, 64), 1, (unsigned)n_win), dim3(64 * sw.nw), stp_lds_bytes(sw), st,
                               data, mad, out, d_ends, sw, stp_plan(sw), thr_scale, L, C, 1, ws, ws);
        else
            hipLaunchKernelGGL(k_colst_dyn, grid, dim3(blk), 0, st, data, mad, out, ring, acc, d_ends, sw, thr_scale, L, C, 1, ws, ws);
    }
    HIPCHK(hipEventRecord(e1, st));
    HIPCHK(hipEventSynchronize(e1));
    LAUNCHCHK();
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *ms_per_launch = ms / repeats;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (variant == 5) {
        const bool klog = g_klog_on;
        g_klog_on = false;
        hipLaunchKernelGGL(k_unpanel_w<uint8_t>, dim3((unsigned)cdiv((int64_t)ws, 256), (unsigned)n_win), dim3(256), 0, st, (const uint8_t*)pout, out, L, C, ws);
        g_klog_on = klog;
        LAUNCHCHK();
        HIPCHK(hipStreamSynchronize(st));
        (void)hipFree(pdata);
        (void)hipFree(pout);
    }
    (void)hipFree(ring);
    (void)hipFree(acc);
    (void)hipFree(d_ends);
    return TRI_OK;
}

// Measurement / test hook: one axis stage of the masked box filter in the launch geometry of
// the step (see include/tricolour_amd.h).
extern "C" int tri_bench_boxfilter(const float* data, const uint8_t* flags4, float* out_w, float* out_o,
                                   int64_t n_win, int64_t n_line, int64_t n_col, int64_t radius,
                                   int stage, int variant, int repeats, float* ms_per_launch, void* stream) {
    if (!data || !flags4 || !out_w || !out_o || !ms_per_launch) return set_err(TRI_EINVAL, "NULL pointer argument");
    if (n_win <= 0 || n_line <= 0 || n_col <= 0 || repeats <= 0 || n_win > 65535 || radius <= 0 || (stage != 2 && n_line % 4 != 0))
        return set_err(TRI_EINVAL, "bad shape");
    if (stage < 0 || stage > 2) return set_err(TRI_EUNSUPPORTED, "stage must be 0, 1 or 2");
    if (variant < 0 || variant > 5 || (variant == 4 && stage != 1) || (variant == 5 && stage == 2))
        return set_err(TRI_EINVAL, "variant must be 0 .. 3 (4: exact row filter, stage 1 only; 5: stage pipeline with blocks of 16, stages 0 and 1)");
    if (stage == 2 && n_win != 1) return set_err(TRI_EINVAL, "the spectrum stage takes one window");
    Run r;
    r.st = (hipStream_t)stream;
    r.p = nullptr;
    r.dbg = nullptr;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    const size_t N = (size_t)n_line * n_col;
    g_boxr_override = variant == 0 ? -1 : (variant == 1 ? 0 : 1);
    g_boxw_override = variant == 0 ? -1 : 0;            // the named variants run BOTH images through their own kernels
    // stage 2: 0 = the flagger's route, 1 = register rings, 2 / 3 = stage pipeline with blocks of 16 / 8
    if (stage == 2) { g_boxr_override = -1; g_boxp_override = variant == 0 ? -1 : (variant == 1 ? 0 : (variant == 2 ? 16 : 8)); }
    // stage 0: 0 = the flagger's route, 1 = LDS delay lines, 2 = register delay lines (K4r), 3 = stage pipeline (K4q)
    // (5 = the stage pipeline with blocks of 16 positions wherever 3 takes blocks of 8)
    if (stage == 0 || stage == 1) {
        g_boxq_override = variant == 0 ? -1 : (variant == 3 || variant == 5 ? 1 : 0);
        if (variant == 3 || variant == 5) g_boxr_override = -1;
        g_boxq_b8_override = variant == 5 ? 0 : -1;
    }
    int rc = TRI_OK;
    // stage 1, variant 4: K4x works on rows -- the amplitudes are taken to the TF layout once (untimed), the rows it
    // writes go back to FT inside the timed loop (as in the flagger); out_w doubles as the row buffer.
    // ms_per_launch then covers kernel + transpose; the counters in stats[] are printed with TRI_BOXX_STATS=1
    float* x_data_tf = nullptr;
    unsigned long long* x_stats = nullptr;
    int xl = 0;
    if (stage == 1 && variant == 4) {
        g_boxx_override = 1;
        xl = boxx_pick_l((int)radius, (int)n_col);
        g_boxx_override = -1;
        if (!xl) return set_err(TRI_EUNSUPPORTED, "no exact row filter for radius %d on lines of %d", (int)radius, (int)n_col);
        HIPCHK(hipMalloc(&x_data_tf, (size_t)n_win * N * sizeof(float)));
        HIPCHK(hipMalloc(&x_stats, 2 * sizeof(unsigned long long)));
        HIPCHK(hipMemsetAsync(x_stats, 0, 2 * sizeof(unsigned long long), r.st));
        rc = launch_transpose<float>(r, data, x_data_tf, (int)n_col, (int)n_line, N, N, n_win);
        if (rc) return rc;
    }
    HIPCHK(hipEventRecord(e0, r.st));
    for (int i = 0; i < repeats && rc == TRI_OK; i++) {
        if (xl) {
            const float* srcW = reinterpret_cast<const float*>(flags4);
            rc = launch_boxx<1>(r, xl, srcW, srcW + N, x_data_tf, nullptr, out_w, nullptr, (int)n_col, (int)n_line, (int)n_col, (int)radius,
                                2 * N, N, 0, N, 0, n_win, nullptr, x_stats);
            if (rc == TRI_OK) rc = launch_transpose<float>(r, out_w, out_o, (int)n_line, (int)n_col, N, N, n_win);
        } else if (stage == 2) {
            // spectrum path: byte flags [n_line][n_col] + data -> filtered weight and data images
            if (variant >= 2 && boxp_pick_block((int)radius, (int)n_col) == 0) rc = set_err(TRI_EUNSUPPORTED, "no stage pipeline for this shape");
            else rc = launch_colfilter(r, 0, out_w, out_o, data, flags4, out_w, out_o, (int)n_line, (int)n_col, (int)radius, 0, 0, 0, 1, nullptr, false, true);
        } else if (stage == 0) {
            rc = launch_colfilter(r, 2, out_w, out_o, data, flags4, out_w, out_o, (int)n_line, (int)n_col, (int)radius, N, N, N,
                                  n_win, nullptr, false, true);
        } else {
            // frequency-axis stage + masked division from TF images (line = time row, n_col positions):
            // the flagger's route for this radius (fused register / LDS ring kernel, or lane-per-stage
            // filter followed by the division kernel)
            const float* srcW = reinterpret_cast<const float*>(flags4);   // (n_win, 2, n_line, n_col): weight, weight * data
            const float* srcO = srcW + N;
            const int T = (int)n_line, Fa = (int)n_col, rad = (int)radius;
            const int ksf = boxr_pick_ks_f(rad);
            if (ksf > 0) {
                rc = launch_boxf<1>(r, ksf, srcW, srcO, out_w, out_o, data, Fa, T, Fa, rad, 2 * N, N, N, n_win, nullptr);
            } else if (colfilter_tf_usable(rad)) {
                rc = launch_colfilter_tf<1>(r, srcW, srcO, out_w, out_o, data, Fa, T, Fa, rad, 2 * N, N, N, n_win, nullptr);
            } else if (colfilter_t4_usable(rad)) {
                float den_f = 0.0f;
                rc = launch_colfilter_t4(r, srcW, srcO, out_w, out_o, Fa, T, Fa, rad, 2 * N, N, n_win, &den_f);
                if (rc == TRI_OK) rc = launch_masked_div<1>(r, out_w, out_o, data, N, N, N, n_win, den_f);
            } else {
                rc = set_err(TRI_EUNSUPPORTED, "no single-sweep frequency stage for radius %d", rad);
            }
        }
    }
    g_boxr_override = -1;
    g_boxp_override = -1;
    g_boxq_override = -1;
    g_boxq_b8_override = -1;
    g_boxw_override = -1;
    if (rc) return rc;
    HIPCHK(hipEventRecord(e1, r.st));
    HIPCHK(hipEventSynchronize(e1));
    LAUNCHCHK();
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *ms_per_launch = ms / repeats;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (xl) {
        unsigned long long h[2] = {0, 0};
        HIPCHK(hipMemcpy(h, x_stats, sizeof(h), hipMemcpyDeviceToHost));
        g_boxx_last_stats[0] = h[0];
        g_boxx_last_stats[1] = h[1];
        if (const char* e = getenv("TRI_BOXX_STATS")) if (e[0] == '1')
            fprintf(stderr, "k_boxx r=%d L=%d x %d threads: %llu line passes, %llu redone sequentially\n", (int)radius, xl & 255, xl >> 8, h[0], h[1]);
        (void)hipFree(x_data_tf);
        (void)hipFree(x_stats);
    }
    return TRI_OK;
}

// Kernel log of the calling thread: op 0 = clear and switch on, op 1 = write "name=count;name=count;..." (demangled
// kernel symbols, launches since op 0) into buf and switch off, op 2 = switch off.  Returns the number of distinct kernels.
#include <cxxabi.h>
extern "C" int tri_kernel_log(int op, char* buf, int64_t cap) {
    if (op == 0) {
        if (g_klog) g_klog->clear();
        g_klog_on = true;
        return 0;
    }
    g_klog_on = false;
    if (op != 1) return 0;
    if (!buf || cap <= 0) return set_err(TRI_EINVAL, "kernel log: no buffer");
    buf[0] = 0;
    if (!g_klog) return 0;
    std::vector<std::pair<std::string, long long>> rows;
    for (const auto& kv : *g_klog) {
        const char* m = hipKernelNameRefByPtr(kv.first, nullptr);
        std::string name = m ? m : "?";
        int st = 0;
        char* d = m ? abi::__cxa_demangle(m, nullptr, nullptr, &st) : nullptr;
        if (d && st == 0) { name = d; }
        if (d) free(d);
        const size_t par = name.find('(');                      // drop the argument list and a leading "void "
        if (par != std::string::npos) name.erase(par);
        if (name.compare(0, 5, "void ") == 0) name.erase(0, 5);
        rows.emplace_back(name, kv.second);
    }
    std::sort(rows.begin(), rows.end());
    size_t off = 0;
    for (const auto& r : rows) {
        const int n = snprintf(buf + off, (size_t)cap - off, "%s=%lld;", r.first.c_str(), r.second);
        if (n < 0 || off + (size_t)n >= (size_t)cap) break;
        off += (size_t)n;
    }
    return (int)rows.size();
}

// Test hook: statistics of the fused median + rejection kernel (K3r) since the last reset: blocks run, fallbacks before the
// pass, after it, at the bracket verification.
extern "C" int tri_medrej_stats(uint64_t* out20, int reset) {
    unsigned long long h[20] = {0};
    HIPCHK(hipDeviceSynchronize());
    if (out20) {
        HIPCHK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_medrej_stats), sizeof(h)));
        for (int k = 0; k < 20; k++) out20[k] = h[k];
    }
    if (reset) {
        unsigned long long z[20] = {0};
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_medrej_stats), z, sizeof(z)));
    }
    return TRI_OK;
}

// Test hook: line passes run / redone sequentially by the last tri_bench_boxfilter(stage 1, variant 4) of this thread
extern "C" int tri_boxx_last_stats(uint64_t* passes, uint64_t* sequential) {
    if (!passes || !sequential) return set_err(TRI_EINVAL, "NULL pointer argument");
    *passes = g_boxx_last_stats[0];
    *sequential = g_boxx_last_stats[1];
    return TRI_OK;
}

// Test hook: exhaustive check of box_divide() for one radius (see include/tricolour_amd.h)
__global__ void k_check_box_divide(BoxDenom dn, unsigned long long* out) {
    unsigned long long bad = 0;
    const unsigned stride = gridDim.x * blockDim.x;
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    for (unsigned long long it = 0; it < (1ull << 32) / stride; it++, i += stride) {
        const float a = __uint_as_float(i);
        const float q = box_divide_checked(a, dn);
        const float e = a / dn.b;
        if (!((__float_as_uint(q) == __float_as_uint(e)) || (isnan(q) && isnan(e)))) bad++;
    }
    if (bad) atomicAdd(out, bad);
}
extern "C" int tri_test_box_divide(int64_t radius, uint64_t* mismatches, void* stream) {
    if (!mismatches || radius < 0 || radius > (1 << 20)) return set_err(TRI_EINVAL, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* d = nullptr;
    HIPCHK(hipMalloc(&d, sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(d, 0, sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_check_box_divide, dim3(4096), dim3(256), 0, st, box_reciprocal(box_denominator(radius)), d);
    LAUNCHCHK();
    unsigned long long h = 0;
    HIPCHK(hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    (void)hipFree(d);
    *mismatches = h;
    return TRI_OK;
}

static int launch_median_big(hipStream_t st, const float* data, const uint8_t* flags, size_t N, int64_t B, const double* centre,
                             double* med, MedBigPar* par, unsigned* ghist, unsigned* gcand, bool vec);

// Test hook: exact segmented medians of |x| over unflagged samples of a
// [n_win][rows][row_len] array; segments [ends[g], ends[g+1]) along each row.
// variant: 0 = automatic choice, 1 = wave kernel (segments <= 1024),
// 2 = workgroup kernel with scalar loads, 3 = workgroup kernel with vector
// loads (needs row_len, segment bounds % 4 == 0).  med: [n_win][rows][G].
extern "C" int tri_test_median(const float* data, const uint8_t* flags, double* med,
                               int64_t n_win, int64_t rows, int64_t row_len,
                               const int64_t* seg_ends, int64_t n_seg_ends, int variant,
                               void* stream) {
    if (!data || !flags || !med || !seg_ends || n_seg_ends < 2 || n_win <= 0 || rows <= 0)
        return set_err(TRI_EINVAL, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    int G = (int)n_seg_ends - 1;
    std::vector<int64_t> start(G), len(G);
    int64_t maxlen = 0;
    bool al4 = row_len % 4 == 0;
    for (int g = 0; g < G; g++) {
        start[g] = seg_ends[g];
        len[g] = seg_ends[g + 1] - seg_ends[g];
        if (len[g] < 0 || seg_ends[g] < 0 || seg_ends[g + 1] > row_len) return set_err(TRI_EINVAL, "bad segment");
        maxlen = std::max(maxlen, len[g]);
        al4 = al4 && start[g] % 4 == 0 && len[g] % 4 == 0;
    }
    int64_t *d_start = nullptr, *d_len = nullptr;
    HIPCHK(hipMalloc(&d_start, G * sizeof(int64_t)));
    HIPCHK(hipMalloc(&d_len, G * sizeof(int64_t)));
    HIPCHK(hipMemcpy(d_start, start.data(), G * sizeof(int64_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_len, len.data(), G * sizeof(int64_t), hipMemcpyHostToDevice));
    size_t WS = (size_t)rows * row_len, RS = (size_t)row_len;
    int R = (int)rows;
    if (variant == 8) {
        // multi-workgroup two-pass select (K3d): one segment spanning each whole row
        if (G != 1 || seg_ends[0] != 0 || seg_ends[1] != row_len) return set_err(TRI_EINVAL, "variant 8 takes one segment covering the row");
        const int64_t B = n_win * rows;
        MedBigPar* par = nullptr;
        unsigned *ghist = nullptr, *gcand = nullptr;
        HIPCHK(hipMalloc(&par, B * sizeof(MedBigPar)));
        HIPCHK(hipMalloc(&ghist, B * SEL_BINS * sizeof(unsigned)));
        HIPCHK(hipMalloc(&gcand, B * (size_t)MEDBIG_CAND * sizeof(unsigned)));
        int rcb = launch_median_big(st, data, flags, (size_t)row_len, B, nullptr, med, par, ghist, gcand,
                                    row_len % 4 == 0 && ((uintptr_t)data % 16 == 0) && ((uintptr_t)flags % 4 == 0));
        HIPCHK(hipStreamSynchronize(st));
        (void)hipFree(par); (void)hipFree(ghist); (void)hipFree(gcand);
        (void)hipFree(d_start); (void)hipFree(d_len);
        return rcb;
    }
    if (variant == 0) variant = maxlen <= 64 * MW_K ? 1 : (al4 ? 3 : 2);
    // variants 11 / 14: variants 1 / 4 with one segment per wave (the round-3 launch shape; TRI_MEDIAN_WAVE_OLD=1 in the pipeline)
    const bool wave_old = variant == 11 || variant == 14;
    if (wave_old) variant -= 10;
    if ((variant == 1 || variant == 4) && maxlen > 64 * MW_K) return set_err(TRI_EINVAL, "wave kernel handles segments <= 1024");
    if (variant == 4 && (row_len % 4 != 0 || maxlen + 3 > 64 * MW_K)) return set_err(TRI_EINVAL, "masked vector variant needs row_len % 4 == 0 and segments <= 1021");
    if ((uint64_t)rows * (uint64_t)row_len * 4u >= (1ull << 32) && (variant == 4 || variant == 1)) return set_err(TRI_EUNSUPPORTED, "wave kernels: a window below 4 GB");
    if ((variant == 3 || variant == 5) && !al4) return set_err(TRI_EINVAL, "vector loads need 4-aligned segments");
    if (wave_old && variant == 1 && maxlen <= 64 * 8)
        hipLaunchKernelGGL((k_median_wave<8, false, 1>), dim3((unsigned)cdiv((int64_t)cdiv((int64_t)R, 1) * G, 4), (unsigned)n_win), dim3(256), 0, st,
                           data, flags, med, WS, WS, RS, (size_t)1, d_start, d_len, R, G);
    else if (wave_old && variant == 4 && row_len % 4 == 0 && maxlen + 3 <= 64 * 8)
        hipLaunchKernelGGL((k_median_wave<8, true, 1>), dim3((unsigned)cdiv((int64_t)cdiv((int64_t)R, 1) * G, 4), (unsigned)n_win), dim3(256), 0, st,
                           data, flags, med, WS, WS, RS, (size_t)1, d_start, d_len, R, G);
    else if (wave_old && ((variant == 4 && row_len % 4 == 0 && maxlen + 3 <= 64 * MW_K) ||
             (variant == 1 && G == 1 && al4 && seg_ends[0] == 0 && seg_ends[1] == row_len)))
        hipLaunchKernelGGL((k_median_wave<MW_K, true, 1>), dim3((unsigned)cdiv((int64_t)cdiv((int64_t)R, 1) * G, 4), (unsigned)n_win), dim3(256), 0, st,
                           data, flags, med, WS, WS, RS, (size_t)1, d_start, d_len, R, G);
    else if (wave_old && variant == 1)
        hipLaunchKernelGGL((k_median_wave<MW_K, false, 1>), dim3((unsigned)cdiv((int64_t)cdiv((int64_t)R, 1) * G, 4), (unsigned)n_win), dim3(256), 0, st,
                           data, flags, med, WS, WS, RS, (size_t)1, d_start, d_len, R, G);
    else if (variant == 1 && maxlen <= 64 * 8)
        hipLaunchKernelGGL((k_median_wave<8, false, MW_SPW>), dim3((unsigned)cdiv((int64_t)cdiv((int64_t)R, MW_SPW) * G, 4), (unsigned)n_win), dim3(256), 0, st,
                           data, flags, med, WS, WS, RS, (size_t)1, d_start, d_len, R, G);
    else if (variant == 4 && row_len % 4 == 0 && maxlen + 3 <= 64 * 8)
        hipLaunchKernelGGL((k_median_wave<8, true, MW_SPW>), dim3((unsigned)cdiv((int64_t)cdiv((int64_t)R, MW_SPW) * G, 4), (unsigned)n_win), dim3(256), 0, st,
                           data, flags, med, WS, WS, RS, (size_t)1, d_start, d_len, R, G);
    else if ((variant == 4 && row_len % 4 == 0 && maxlen + 3 <= 64 * MW_K) ||
             (variant == 1 && G == 1 && al4 && seg_ends[0] == 0 && seg_ends[1] == row_len))
        hipLaunchKernelGGL((k_median_wave<MW_K, true, MW_SPW16>), dim3((unsigned)cdiv((int64_t)cdiv((int64_t)R, MW_SPW16) * G, 4), (unsigned)n_win), dim3(256), 0, st,
                           data, flags, med, WS, WS, RS, (size_t)1, d_start, d_len, R, G);
    else if (variant == 1)
        hipLaunchKernelGGL((k_median_wave<MW_K, false, MW_SPW16>), dim3((unsigned)cdiv((int64_t)cdiv((int64_t)R, MW_SPW16) * G, 4), (unsigned)n_win), dim3(256), 0, st,
                           data, flags, med, WS, WS, RS, (size_t)1, d_start, d_len, R, G);
    else if (variant == 3)
        hipLaunchKernelGGL(k_median<true>, dim3((unsigned)(R * G), (unsigned)n_win), dim3(256), 0, st, data, flags,
                           med, WS, WS, RS, (size_t)1, d_start, d_len, R, G);
    else if (variant == 5)
        hipLaunchKernelGGL(k_median2<true>, dim3((unsigned)(R * G), (unsigned)n_win), dim3(256), 0, st, data, flags,
                           med, WS, WS, RS, (size_t)1, d_start, d_len, R, G);
    else if (variant == 9 || variant == 10) {
        // K3c with the predicted-window candidates in global scratch (9: vector loads, 10: scalar)
        if (variant == 9 && row_len % 4 != 0) return set_err(TRI_EINVAL, "variant 9 needs row_len % 4 == 0");
        unsigned* gc = nullptr;
        const size_t cap = ((size_t)maxlen + 3) & ~(size_t)3, per_win = cap * (size_t)R * G;
        HIPCHK(hipMalloc(&gc, per_win * (size_t)n_win * sizeof(unsigned)));
        if (variant == 9)
            hipLaunchKernelGGL(k_median2<true>, dim3((unsigned)(R * G), (unsigned)n_win), dim3(256), 0, st, data, flags,
                               med, WS, WS, RS, (size_t)1, d_start, d_len, R, G, gc, per_win, (unsigned)cap);
        else
            hipLaunchKernelGGL(k_median2<false>, dim3((unsigned)(R * G), (unsigned)n_win), dim3(256), 0, st, data, flags,
                               med, WS, WS, RS, (size_t)1, d_start, d_len, R, G, gc, per_win, (unsigned)cap);
        hipError_t le = hipGetLastError();
        hipError_t se = hipStreamSynchronize(st);
        (void)hipFree(gc); (void)hipFree(d_start); (void)hipFree(d_len);
        if (le != hipSuccess || se != hipSuccess) return set_err(TRI_EHIP, "median variant %d failed", variant);
        return TRI_OK;
    } else if (variant == 7) {
        if (row_len % 4 != 0) return set_err(TRI_EINVAL, "variant 7 needs row_len % 4 == 0");
        hipLaunchKernelGGL(k_median2<true>, dim3((unsigned)(R * G), (unsigned)n_win), dim3(256), 0, st, data, flags,
                           med, WS, WS, RS, (size_t)1, d_start, d_len, R, G);
    } else if (variant == 6)
        hipLaunchKernelGGL(k_median2<false>, dim3((unsigned)(R * G), (unsigned)n_win), dim3(256), 0, st, data, flags,
                           med, WS, WS, RS, (size_t)1, d_start, d_len, R, G);
    else
        hipLaunchKernelGGL(k_median<false>, dim3((unsigned)(R * G), (unsigned)n_win), dim3(256), 0, st, data, flags,
                           med, WS, WS, RS, (size_t)1, d_start, d_len, R, G);
    LAUNCHCHK();
    HIPCHK(hipStreamSynchronize(st));
    (void)hipFree(d_start);
    (void)hipFree(d_len);
    return TRI_OK;
}

extern "C" int tri_flag_nans_and_zeros(const void* vis, int vis_dtype, const uint8_t* flags,
                                       uint8_t* out_flags, int64_t n, void* stream) {
    if (!vis || !flags || !out_flags || n < 0) return set_err(TRI_EINVAL, "bad argument");
    if (vis_dtype != TRI_VIS_C64 && vis_dtype != TRI_VIS_F32) return set_err(TRI_EUNSUPPORTED, "vis dtype must be complex64 or float32");
    if (n == 0) return TRI_OK;
    dim3 grid((unsigned)cdiv(n, 256));
    if (vis_dtype == TRI_VIS_C64)
        hipLaunchKernelGGL(k_flag_nans_zeros<TRI_VIS_C64>, grid, dim3(256), 0, (hipStream_t)stream, vis, flags, out_flags, (size_t)n);
    else
        hipLaunchKernelGGL(k_flag_nans_zeros<TRI_VIS_F32>, grid, dim3(256), 0, (hipStream_t)stream, vis, flags, out_flags, (size_t)n);
    LAUNCHCHK();
    return TRI_OK;
}

extern "C" int tri_stokes_intensity(const void* vis, int vis_dtype, int64_t n, int64_t ncorr,
                                    const int32_t* pol_idx, const double* pol_alpha, int64_t n_pol,
                                    const int32_t* unpol_idx, const double* unpol_alpha, int64_t n_unpol,
                                    int mode, void* out, void* stream) {
    if (n < 0 || ncorr <= 0 || n_pol < 0 || n_unpol < 0) return set_err(TRI_EINVAL, "bad shape");
    if (mode != 0 && mode != 1) return set_err(TRI_EINVAL, "mode must be 0 (polarised) or 1 (unpolarised)");
    if (vis_dtype != TRI_VIS_C64 && vis_dtype != TRI_VIS_C128) return set_err(TRI_EUNSUPPORTED, "visibilities must be complex64 or complex128");
    if (n_pol > TRI_MAX_STOKES_TERMS || n_unpol > TRI_MAX_STOKES_TERMS) return set_err(TRI_EUNSUPPORTED, "at most 4 stokes terms per kind");
    if ((n_pol > 0 && (!pol_idx || !pol_alpha)) || (n_unpol > 0 && (!unpol_idx || !unpol_alpha))) return set_err(TRI_EINVAL, "NULL pointer argument");
    StokesTerms terms;
    terms.n_pol = (int)n_pol;
    terms.n_unpol = (int)n_unpol;
    auto fill = [&](StokesTerm* dst, const int32_t* idx, const double* alpha, int64_t cnt) -> bool {
        for (int64_t k = 0; k < cnt; k++) {
            dst[k].c1 = idx[4 * k];
            dst[k].c2 = idx[4 * k + 1];
            dst[k].s1 = idx[4 * k + 2];
            dst[k].s2 = idx[4 * k + 3];
            dst[k].ar = alpha[2 * k];
            dst[k].ai = alpha[2 * k + 1];
            if (dst[k].c1 < 0 || dst[k].c1 >= ncorr || dst[k].c2 < 0 || dst[k].c2 >= ncorr) return false;
        }
        return true;
    };
    if (!fill(terms.pol, pol_idx, pol_alpha, n_pol) || !fill(terms.unpol, unpol_idx, unpol_alpha, n_unpol))
        return set_err(TRI_EINVAL, "correlation index out of range");
    if (n == 0) return TRI_OK;
    if (!vis || !out) return set_err(TRI_EINVAL, "NULL pointer argument");
    dim3 grid((unsigned)cdiv(n, 256));
    if (vis_dtype == TRI_VIS_C64)
        hipLaunchKernelGGL(k_stokes_intensity<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)vis, (float*)out, (size_t)n, (int)ncorr, terms, mode);
    else
        hipLaunchKernelGGL(k_stokes_intensity<double>, grid, dim3(256), 0, (hipStream_t)stream, (const double*)vis, (double*)out, (size_t)n, (int)ncorr, terms, mode);
    LAUNCHCHK();
    return TRI_OK;
}

extern "C" int tri_window_counts(const uint8_t* flags, int64_t nbl, int64_t ncorr, int64_t ntime,
                                 int64_t nchan, uint64_t* per_bl, uint64_t* per_chan, void* stream) {
    if (!per_bl || !per_chan) return set_err(TRI_EINVAL, "NULL pointer argument");
    if (nbl < 0 || ncorr < 0 || ntime < 0 || nchan < 0) return set_err(TRI_EINVAL, "bad shape");
    hipStream_t st = (hipStream_t)stream;
    if (nbl > 0) HIPCHK(hipMemsetAsync(per_bl, 0, (size_t)nbl * sizeof(uint64_t), st));
    if (nchan > 0) HIPCHK(hipMemsetAsync(per_chan, 0, (size_t)nchan * sizeof(uint64_t), st));
    const int64_t rows = ncorr * ntime;
    if (nbl == 0 || rows == 0 || nchan == 0) return TRI_OK;
    if (!flags) return set_err(TRI_EINVAL, "NULL pointer argument");
    if (nbl > 65535 || rows >= (1ll << 31) || nchan >= (1ll << 31)) return set_err(TRI_EUNSUPPORTED, "window too large for one call");
    // enough row chunks per baseline to fill the device, at most 65535
    int rows_per_block = 256;
    while (cdiv(rows, rows_per_block) > 65535) rows_per_block *= 2;
    const bool vec = nchan % 4 == 0 && ((uintptr_t)flags % 4 == 0);
    dim3 grid((unsigned)cdiv(nchan, vec ? 1024 : 256), (unsigned)cdiv(rows, rows_per_block), (unsigned)nbl);
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "64-bit counters");
    if (vec)
        hipLaunchKernelGGL(k_window_counts<true>, grid, dim3(256), 0, st, flags, (unsigned long long*)per_bl,
                           (unsigned long long*)per_chan, (int)rows, (int)nchan, rows_per_block);
    else
        hipLaunchKernelGGL(k_window_counts<false>, grid, dim3(256), 0, st, flags, (unsigned long long*)per_bl,
                           (unsigned long long*)per_chan, (int)rows, (int)nchan, rows_per_block);
    LAUNCHCHK();
    return TRI_OK;
}

extern "C" int tri_apply_baseline_channel_mask(const uint8_t* flags, uint8_t* out_flags,
                                               const uint8_t* bl_sel, const uint8_t* chan_mask,
                                               int mode, int64_t nbl, int64_t ncorr, int64_t ntime,
                                               int64_t nchan, void* stream) {
    if (!flags || !out_flags || !bl_sel || !chan_mask) return set_err(TRI_EINVAL, "NULL pointer argument");
    if (mode != 0 && mode != 1) return set_err(TRI_EINVAL, "mode must be 0 (or) or 1 (override)");
    if (nbl < 0 || ncorr < 0 || ntime < 0 || nchan < 0) return set_err(TRI_EINVAL, "bad shape");
    if (nbl == 0 || ncorr * ntime == 0 || nchan == 0) return TRI_OK;
    if (ncorr * ntime > 65535 || nbl > 65535) return set_err(TRI_EUNSUPPORTED, "corr*time and bl must each be <= 65535 per call");
    dim3 grid((unsigned)cdiv(nchan, 256), (unsigned)(ncorr * ntime), (unsigned)nbl);
    hipLaunchKernelGGL(k_apply_bl_chan_mask, grid, dim3(256), 0, (hipStream_t)stream, flags, out_flags, bl_sel,
                       chan_mask, mode, (int)nchan, (size_t)(ncorr * ntime));
    LAUNCHCHK();
    return TRI_OK;
}


// ---------------------------------------------------------------------------
// uvcontsub_flagger (flagging.py:989-1073), SURVEY.md 8f-2
// ---------------------------------------------------------------------------
extern "C" size_t tri_uvcontsub_workspace_bytes(int64_t batch, int64_t ntime, int64_t nchan) {
    if (batch <= 0 || ntime <= 0 || nchan <= 0) return 0;
    size_t N = (size_t)ntime * nchan, B = (size_t)batch;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    // |residual| image, median flags, mean / smooth spectra, two medians, flag counts, and the global histogram /
    // candidate list / parameters of the multi-workgroup median (K3d)
    return al(B * N * 4) + al(B * N) + al(B * nchan * 8) * 2 + al(B * 8) * 2 + al(B * 4) +
           al(B * SEL_BINS * 4) + al(B * MEDBIG_CAND * 4) + al(B * sizeof(MedBigPar)) + 4 * 256;
}

// Exact median of | |x| - centre | (centre == nullptr: of |x|) over the unflagged samples of each of B windows of N
// contiguous samples, many workgroups per window (K3d).
static int launch_median_big(hipStream_t st, const float* data, const uint8_t* flags, size_t N, int64_t B, const double* centre,
                             double* med, MedBigPar* par, unsigned* ghist, unsigned* gcand, bool vec) {
    const unsigned slices = (unsigned)cdiv((int64_t)N, MEDBIG_SLICE);
    if (B > 65535) return set_err(TRI_EUNSUPPORTED, "too many windows in one median batch");
    hipLaunchKernelGGL(k_medbig_range, dim3((unsigned)B), dim3(256), 0, st, data, flags, N, centre, par, ghist);
    if (vec) hipLaunchKernelGGL(k_medbig_hist<true>, dim3(slices, (unsigned)B), dim3(256), 0, st, data, flags, N, centre, par, ghist);
    else hipLaunchKernelGGL(k_medbig_hist<false>, dim3(slices, (unsigned)B), dim3(256), 0, st, data, flags, N, centre, par, ghist);
    hipLaunchKernelGGL(k_medbig_pick, dim3((unsigned)B), dim3(256), 0, st, ghist, par);
    if (vec) {
        hipLaunchKernelGGL(k_medbig_compact<true>, dim3(slices, (unsigned)B), dim3(256), 0, st, data, flags, N, centre, par, gcand);
        hipLaunchKernelGGL(k_medbig_select<true>, dim3((unsigned)B), dim3(256), 0, st, data, flags, N, centre, par, gcand, med);
    } else {
        hipLaunchKernelGGL(k_medbig_compact<false>, dim3(slices, (unsigned)B), dim3(256), 0, st, data, flags, N, centre, par, gcand);
        hipLaunchKernelGGL(k_medbig_select<false>, dim3((unsigned)B), dim3(256), 0, st, data, flags, N, centre, par, gcand, med);
    }
    LAUNCHCHK();
    return TRI_OK;
}

extern "C" int tri_uvcontsub_flagger(const void* vis_c64, const uint8_t* flags, uint8_t* out_flags,
                                     int64_t n_cp, int64_t ntime, int64_t nchan,
                                     int64_t major_cycles, int64_t or_original_from_cycle,
                                     int64_t taylor_degrees, double sigma, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    if (!vis_c64 || !flags || !out_flags) return set_err(TRI_EINVAL, "NULL pointer argument");
    if (n_cp < 0 || ntime <= 0 || nchan <= 0) return set_err(TRI_EINVAL, "bad shape");
    if (taylor_degrees < 0 || taylor_degrees > 64) return set_err(TRI_EUNSUPPORTED, "taylor_degrees must be in [0, 64]");
    if ((int64_t)ntime * nchan >= ((int64_t)1 << 31)) return set_err(TRI_EUNSUPPORTED, "window too large");
    if (n_cp == 0) return TRI_OK;
    hipStream_t st = (hipStream_t)stream;
    int T = (int)ntime, F = (int)nchan;
    size_t N = (size_t)T * F;
    size_t one = tri_uvcontsub_workspace_bytes(1, ntime, nchan);
    if (!workspace || workspace_bytes < one) return set_err(TRI_EWORKSPACE, "workspace of %zu bytes is smaller than the %zu needed for one window", workspace_bytes, one);
    int64_t Bmax = std::min<int64_t>(n_cp, 16384);
    while (Bmax > 1 && tri_uvcontsub_workspace_bytes(Bmax, ntime, nchan) > workspace_bytes) Bmax = (Bmax + 1) / 2;
    while (tri_uvcontsub_workspace_bytes(Bmax, ntime, nchan) > workspace_bytes) Bmax--;
    Bump b(workspace, workspace_bytes, false);
    float* absres = b.get<float>((size_t)Bmax * N);
    uint8_t* mflags = b.get<uint8_t>((size_t)Bmax * N);
    float2* avg = b.get<float2>((size_t)Bmax * F);
    float2* smooth = b.get<float2>((size_t)Bmax * F);
    double* med1 = b.get<double>((size_t)Bmax);
    double* mad = b.get<double>((size_t)Bmax);
    unsigned* cnt = b.get<unsigned>((size_t)Bmax);
    unsigned* ghist = b.get<unsigned>((size_t)Bmax * SEL_BINS);
    unsigned* gcand = b.get<unsigned>((size_t)Bmax * MEDBIG_CAND);
    MedBigPar* mpar = b.get<MedBigPar>((size_t)Bmax);
    int K = (int)std::min<int64_t>(taylor_degrees, F);
    bool vec = N % 4 == 0 && (((uintptr_t)workspace) % 16 == 0);
    for (int64_t c0 = 0; c0 < n_cp; c0 += Bmax) {
        int64_t B = std::min(Bmax, n_cp - c0);
        const float2* v = (const float2*)vis_c64 + (size_t)c0 * N;
        uint8_t* rf = out_flags + (size_t)c0 * N;
        // result_flags = flags.copy()  (:1028), normalised to 0/1
        hipLaunchKernelGGL(k_normalise_flags, dim3((unsigned)cdiv((int64_t)(B * N), 256)), dim3(256), 0, st, flags + (size_t)c0 * N, rf, (size_t)B * N);
        LAUNCHCHK();
        for (int64_t mi = 0; mi < major_cycles; mi++) {
            HIPCHK(hipMemsetAsync(cnt, 0, (size_t)B * sizeof(unsigned), st));
            hipLaunchKernelGGL(k_uv_mean, dim3((unsigned)cdiv(F, 256), (unsigned)B), dim3(256), 0, st, v, rf, avg, T, F);
            hipLaunchKernelGGL(k_uv_lowpass, dim3((unsigned)B), dim3(256), 0, st, avg, smooth, F, K);
            // |vis - smooth|, the median flags and (in the same pass) the number of flagged samples per product
            static const bool uv_scalar = [] { const char* e = getenv("TRI_UV_SCALAR"); return e && e[0] == '1'; }();   // (A/B, tests)
            if (!uv_scalar && F % 4 == 0 && ((uintptr_t)v % 16 == 0) && ((uintptr_t)rf % 4 == 0) && ((uintptr_t)smooth % 16 == 0) &&
                ((uintptr_t)absres % 16 == 0) && ((uintptr_t)mflags % 4 == 0))
                hipLaunchKernelGGL(k_uv_resid4, dim3((unsigned)cdiv((int64_t)N, 2048), (unsigned)B), dim3(256), 0, st, v, rf, smooth, absres, mflags, cnt, T, F);
            else
                hipLaunchKernelGGL(k_uv_resid, dim3((unsigned)cdiv((int64_t)N, 2048), (unsigned)B), dim3(256), 0, st, v, rf, smooth, absres, mflags, cnt, T, F);
            LAUNCHCHK();
            // nanmedian over the unflagged, non-NaN residuals of each product (:1061), then the median of
            // | |residual| - median | (:1064-1066) straight from the residual image
            int rcm = launch_median_big(st, absres, mflags, N, B, nullptr, med1, mpar, ghist, gcand, vec);
            if (rcm) return rcm;
            rcm = launch_median_big(st, absres, mflags, N, B, med1, mad, mpar, ghist, gcand, vec);
            if (rcm) return rcm;
            if (!uv_scalar && N % 4 == 0 && ((uintptr_t)absres % 16 == 0) && ((uintptr_t)rf % 4 == 0))
                hipLaunchKernelGGL(k_uv_apply4, dim3((unsigned)cdiv((int64_t)(N / 4), 256), (unsigned)B), dim3(256), 0, st, absres, mad, cnt, rf, (float)sigma, mi >= or_original_from_cycle ? 1 : 0, N / 4);
            else
                hipLaunchKernelGGL(k_uv_apply, dim3((unsigned)cdiv((int64_t)N, 256), (unsigned)B), dim3(256), 0, st, absres, mad, cnt, rf, (float)sigma, mi >= or_original_from_cycle ? 1 : 0, N);
            LAUNCHCHK();
        }
    }
    return TRI_OK;
}

// GPU tile preprocessing: page upload -> ViT tiles, bit-exact with the reference's host pipeline
//   utils/utils.py:354-362 build_transform, :381-417 dynamic_preprocess, :420-452 load_image_2, :463-478 load_image
// whose arithmetic is Pillow's Image.resize (default BICUBIC, antialiased) followed by ToTensor/Normalize/.to(bf16).
//
// Pillow's resize (src/libImaging/Resample.c, third-party; restated here from its published algorithm and pinned by
// tests against the installed Pillow): per axis, per output index, a window [xmin, xmin+n) of fixed-point weights
//   scale = in/out, filterscale = max(scale,1), support = 2*filterscale, center = (xx+0.5)*scale,
//   w_x = bicubic((x + xmin - center + 0.5)/filterscale), normalised by their sum (all in double),
//   kk = (int)(+-0.5 + w * 2^22);   out = clip8((2^21 + sum in[x]*kk[x]) >> 22)
// horizontal pass first into an 8-bit image, then the vertical pass.  Everything after the weights is integer/byte
// work, so the GPU result is bit-identical provided the double arithmetic keeps Pillow's operation order: this file
// is compiled with fp contraction off.  Normalisation uses a 3x256 bf16 table built on the host with the reference's
// own fp32 expression ((p/255 - mean)/std -> bf16), so that step is exact by construction.
#pragma clang fp contract(off)
#include <math.h>

#include <algorithm>
#include <vector>

#include "ctx.hpp"

struct PrepJobDev {
    int sx0, sy0, sw, sh;      // source rectangle in the page
    int ow, oh;                // size after Image.resize
    int mode;                  // 0: one tile, image pasted at (left, top) on white; 1: grid of 448x448 tiles, `cols` per row
    int tile0, cols, left, top;
    int ksx, ksy;              // taps per output index (Pillow's ksize) per axis
    int xtab, ytab;            // offsets (ints) of the weight tables: per output index [xmin, n, kk[0..ks)]
    long long tmp;             // byte offset of the horizontal-pass image [sh][ow][3]
};

namespace {

constexpr int PREC = 32 - 8 - 2;

__device__ inline double bicubic_w(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// one thread per (job, axis, output index)
__global__ __launch_bounds__(256) void prep_coeffs_kernel(const PrepJobDev* __restrict__ jobs, const int* __restrict__ first,
                                                          int n_items, int n_jobs, int* __restrict__ tabs) {
    const int item = blockIdx.x * 256 + threadIdx.x;
    if (item >= n_items) return;
    // find the job: first[j] = first item of job j (x items then y items)
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (first[mid] <= item) lo = mid; else hi = mid - 1; }
    const PrepJobDev jb = jobs[lo];
    int idx = item - first[lo];
    const bool yaxis = idx >= jb.ow;
    if (yaxis) idx -= jb.ow;
    const int in_size = yaxis ? jb.sh : jb.sw, out_size = yaxis ? jb.oh : jb.ow, ks = yaxis ? jb.ksy : jb.ksx;
    int* row = tabs + (yaxis ? jb.ytab : jb.xtab) + (long long)idx * (2 + ks);

    const double scale = (double)in_size / out_size;
    double filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 2.0 * filterscale;
    const double center = (idx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; x++) ww += bicubic_w((x + xmin - center + 0.5) * ss);
    row[0] = xmin;
    row[1] = xmax;
    for (int x = 0; x < ks; x++) {
        int kk = 0;
        if (x < xmax) {
            double w = bicubic_w((x + xmin - center + 0.5) * ss);
            if (ww != 0.0) w /= ww;
            kk = w < 0 ? (int)(-0.5 + w * (double)(1 << PREC)) : (int)(0.5 + w * (double)(1 << PREC));
        }
        row[2 + x] = kk;
    }
}

__device__ __forceinline__ unsigned char clip8(int v) {
    v >>= PREC;
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: tmp[y][xx][c] for y in [0, sh), xx in [0, ow)
__global__ __launch_bounds__(256) void prep_hpass_kernel(const unsigned char* __restrict__ page, int W, const PrepJobDev* __restrict__ jobs,
                                                         const int* __restrict__ tabs, unsigned char* __restrict__ tmp) {
    const PrepJobDev jb = jobs[blockIdx.y];
    const long long n = (long long)jb.sh * jb.ow;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int y = (int)(i / jb.ow), xx = (int)(i - (long long)y * jb.ow);
        const int* row = tabs + jb.xtab + (long long)xx * (2 + jb.ksx);
        const int xmin = row[0], cnt = row[1];
        const unsigned char* src = page + ((long long)(jb.sy0 + y) * W + jb.sx0 + xmin) * 3;
        int s0 = 1 << (PREC - 1), s1 = s0, s2 = s0;
        for (int x = 0; x < cnt; x++) {
            const int k = row[2 + x];
            s0 += src[3 * x] * k; s1 += src[3 * x + 1] * k; s2 += src[3 * x + 2] * k;
        }
        unsigned char* d = tmp + jb.tmp + i * 3;
        d[0] = clip8(s0); d[1] = clip8(s1); d[2] = clip8(s2);
    }
}

// vertical pass + placement + normalisation: out tile planes [T][3][448][448] bf16
__global__ __launch_bounds__(256) void prep_vpass_kernel(const PrepJobDev* __restrict__ jobs, const int* __restrict__ tabs,
                                                         const unsigned char* __restrict__ tmp, const bf16* __restrict__ lut,
                                                         bf16* __restrict__ out) {
    const PrepJobDev jb = jobs[blockIdx.y];
    const long long n = (long long)jb.oh * jb.ow;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int yy = (int)(i / jb.ow), xx = (int)(i - (long long)yy * jb.ow);
        const int* row = tabs + jb.ytab + (long long)yy * (2 + jb.ksy);
        const int ymin = row[0], cnt = row[1];
        const unsigned char* src = tmp + jb.tmp + ((long long)ymin * jb.ow + xx) * 3;
        int s0 = 1 << (PREC - 1), s1 = s0, s2 = s0;
        for (int y = 0; y < cnt; y++) {
            const int k = row[2 + y];
            const unsigned char* p = src + (long long)y * jb.ow * 3;
            s0 += p[0] * k; s1 += p[1] * k; s2 += p[2] * k;
        }
        int tile, px, py;
        if (jb.mode == 0) { tile = jb.tile0; px = jb.left + xx; py = jb.top + yy; }
        else { tile = jb.tile0 + (yy / 448) * jb.cols + xx / 448; px = xx % 448; py = yy % 448; }
        bf16* o = out + (long long)tile * 3 * 448 * 448 + (long long)py * 448 + px;
        o[0] = lut[clip8(s0)];
        o[448 * 448] = lut[256 + clip8(s1)];
        o[2 * 448 * 448] = lut[512 + clip8(s2)];
    }
}

// white canvas for the pasted (mode 0) tiles
__global__ __launch_bounds__(256) void prep_fill_white_kernel(const PrepJobDev* __restrict__ jobs, const bf16* __restrict__ lut,
                                                              bf16* __restrict__ out) {
    const PrepJobDev jb = jobs[blockIdx.y];
    if (jb.mode != 0) return;
    bf16* o = out + (long long)jb.tile0 * 3 * 448 * 448;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 3 * 448 * 448; i += gridDim.x * 256) o[i] = lut[(i / (448 * 448)) * 256 + 255];
}

}  // namespace

extern "C" int cr_preprocess(cr_ctx* c, const void* page_rgb, int H, int W, const cr_prep_job* jobs, int n_jobs, const void* lut,
                             void* out_tiles, int n_tiles, void* stream) {
    if (!c || !page_rgb || !jobs || !lut || !out_tiles || n_jobs <= 0 || H <= 0 || W <= 0) return cr_fail(CR_ERR_ARG, "cr_preprocess: bad argument");
    CR_HIP(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    std::vector<PrepJobDev> dj(n_jobs);
    std::vector<int> first(n_jobs);
    long long tmp_bytes = 0;
    long long tab_ints = 0;
    int n_items = 0;
    long long max_h = 1, max_v = 1;
    for (int j = 0; j < n_jobs; j++) {
        const cr_prep_job& s = jobs[j];
        if (s.sw <= 0 || s.sh <= 0 || s.ow <= 0 || s.oh <= 0 || s.sx0 < 0 || s.sy0 < 0 || s.sx0 + s.sw > W || s.sy0 + s.sh > H)
            return cr_fail(CR_ERR_ARG, "cr_preprocess: job %d rectangle outside the %dx%d page", j, W, H);
        PrepJobDev& d = dj[j];
        d.sx0 = s.sx0; d.sy0 = s.sy0; d.sw = s.sw; d.sh = s.sh; d.ow = s.ow; d.oh = s.oh; d.mode = s.mode;
        d.tile0 = s.tile0; d.cols = s.cols; d.left = s.left; d.top = s.top;
        const int tiles_used = s.mode == 0 ? 1 : (s.ow / 448) * (s.oh / 448);
        if (s.tile0 < 0 || s.tile0 + tiles_used > n_tiles) return cr_fail(CR_ERR_ARG, "cr_preprocess: job %d writes outside the %d output tiles", j, n_tiles);
        if (s.mode == 0 && (s.left < 0 || s.top < 0 || s.left + s.ow > 448 || s.top + s.oh > 448)) return cr_fail(CR_ERR_ARG, "cr_preprocess: job %d does not fit a tile", j);
        if (s.mode != 0 && (s.ow % 448 || s.oh % 448 || s.cols != s.ow / 448)) return cr_fail(CR_ERR_ARG, "cr_preprocess: job %d grid must be whole 448-pixel tiles", j);
        auto ksize = [](int in, int out) { double sc = (double)in / out; if (sc < 1.0) sc = 1.0; return (int)ceil(2.0 * sc) * 2 + 1; };
        d.ksx = ksize(s.sw, s.ow); d.ksy = ksize(s.sh, s.oh);
        d.xtab = (int)tab_ints; tab_ints += (long long)s.ow * (2 + d.ksx);
        d.ytab = (int)tab_ints; tab_ints += (long long)s.oh * (2 + d.ksy);
        d.tmp = tmp_bytes; tmp_bytes += ((long long)s.sh * s.ow * 3 + 15) & ~15LL;
        first[j] = n_items; n_items += s.ow + s.oh;
        max_h = std::max(max_h, (long long)s.sh * s.ow); max_v = std::max(max_v, (long long)s.oh * s.ow);
    }
    if (tab_ints > 0x7fffffffLL) return cr_fail(CR_ERR_ARG, "cr_preprocess: weight tables too large");
    const size_t jb_bytes = (size_t)n_jobs * sizeof(PrepJobDev), fi_bytes = (size_t)n_jobs * 4;
    CR_TRY(ws_ensure(c, jb_bytes + fi_bytes + (size_t)tab_ints * 4 + (size_t)tmp_bytes + 4096));
    Arena ar(c->ws);
    PrepJobDev* d_jobs = ar.take<PrepJobDev>(n_jobs);
    int* d_first = ar.take<int>(n_jobs);
    int* d_tabs = ar.take<int>((size_t)tab_ints);
    unsigned char* d_tmp = ar.take<unsigned char>((size_t)tmp_bytes);
    CR_HIP(hipMemcpyAsync(d_jobs, dj.data(), jb_bytes, hipMemcpyHostToDevice, st));
    CR_HIP(hipMemcpyAsync(d_first, first.data(), fi_bytes, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(prep_coeffs_kernel, dim3((n_items + 255) / 256), dim3(256), 0, st, d_jobs, d_first, n_items, n_jobs, d_tabs);
    const int gx_h = (int)std::min<long long>((max_h + 255) / 256, 4096), gx_v = (int)std::min<long long>((max_v + 255) / 256, 4096);
    hipLaunchKernelGGL(prep_fill_white_kernel, dim3(64, n_jobs), dim3(256), 0, st, d_jobs, (const bf16*)lut, (bf16*)out_tiles);
    hipLaunchKernelGGL(prep_hpass_kernel, dim3(gx_h, n_jobs), dim3(256), 0, st, (const unsigned char*)page_rgb, W, d_jobs, d_tabs, d_tmp);
    hipLaunchKernelGGL(prep_vpass_kernel, dim3(gx_v, n_jobs), dim3(256), 0, st, d_jobs, d_tabs, d_tmp, (const bf16*)lut, (bf16*)out_tiles);
    CR_HIP(hipGetLastError());
    return CR_OK;
}

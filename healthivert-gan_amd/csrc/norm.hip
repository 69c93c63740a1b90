// BatchNorm2d / InstanceNorm2d + LeakyReLU/ReLU(+sigmoid), forward and backward, NHWC fp32.
// HBM-bound: each pass streams the activation once with 16 B per lane; statistics are reduced in
// fp64 (per-thread partials -> per-block -> finalize) so var = E[x^2]-E[x]^2 is safe.
#include <stdlib.h>

#include <string.h>

#include "hv_common.h"

static bool n_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static bool n_vec_ok(int C) { return (C % 4 == 0) && n_pow2(C / 4) && C / 4 <= 256; }
static bool n_shape_ok(int C) { return n_vec_ok(C) || (n_pow2(C) && C <= 256); }

struct NormPlan { int G, R, nchunk, rpb, rstep, CB, slices; };
// Reduction grid: (row chunks) x (groups) x (channel slices).  A block covers CB = min(C/4, 32) channel quads (512 B of a
// row) and 256/CB rows per iteration; chunks are sized for ~2048 blocks so that every CU has enough loads in flight to
// stream from HBM, while the fp64 partials ([G][nchunk][2][C]) stay small against the activation.
static NormPlan norm_plan(int B, int HW, int C, int norm, int groups = 1, bool vec = true) {
    NormPlan p;
    p.G = norm == HV_NORM_INSTANCE ? B : (groups > 0 ? groups : 1);
    p.R = norm == HV_NORM_INSTANCE ? HW : (B / p.G) * HW;
    const int CV = vec ? C / 4 : C;                       // lanes per row
    p.CB = vec ? (CV < 32 ? CV : 32) : CV;
    p.slices = CV / p.CB;
    p.rstep = 256 / p.CB;
    static const int blocks = getenv("HV_NORM_BLOCKS") ? atoi(getenv("HV_NORM_BLOCKS")) : 2048;   // tuning knob
    long long cap = blocks / ((long long)p.G * p.slices);
    if (cap < 16) cap = 16;
    if (cap > 512) cap = 512;
    long long rpb = (long long)p.rstep * 8;              // at least 8 iterations per block
    long long nch = (p.R + rpb - 1) / rpb;
    if (nch > cap) {
        rpb = (p.R + cap - 1) / cap;
        rpb = (rpb + p.rstep - 1) / p.rstep * p.rstep;
        nch = (p.R + rpb - 1) / rpb;
    }
    p.nchunk = (int)nch;
    p.rpb = (int)rpb;
    return p;
}

extern "C" size_t hv_norm_workspace_bytes(int B, int HW, int C) {
    if (B <= 0 || HW <= 0 || C <= 0 || !n_shape_ok(C)) return 0;
    size_t pa = 0, pb = 0;
    for (int vec = 0; vec < 2; ++vec) {
        if (vec ? !n_vec_ok(C) : !(n_pow2(C) && C <= 256)) continue;
        const NormPlan a = norm_plan(B, HW, C, HV_NORM_BATCH, 1, vec), b = norm_plan(B, HW, C, HV_NORM_INSTANCE, 1, vec);
        if ((size_t)a.G * a.nchunk > pa) pa = (size_t)a.G * a.nchunk;
        if ((size_t)b.G * b.nchunk > pb) pb = (size_t)b.G * b.nchunk;
    }
    return (pa > pb ? pa : pb) * 2 * C * sizeof(double) + (size_t)B * 2 * C * sizeof(float) + 64;
}

struct NormK {
    const void* x; const void* y; const void* dy; void* out;      // fp32, or fp16 elements with the H instantiations (all four alike)
    int x_ld, x_coff, y_ld, y_coff, dy_ld, dy_coff, o_ld, o_coff;
    int C, R, rpb, nchunk, act, post_sigmoid, CB, lc;   // CB: lanes per row in the reduction; lc: log2(lanes per row) in the apply pass
    const float* stats; const float* gamma; const float* beta;
};

__device__ __forceinline__ float norm_act_fwd(float v, int act, int post_sigmoid) {
    v = hv_act(v, act);
    if (post_sigmoid) v = 1.f / (1.f + expf(-v));
    return v;
}
// d(final)/d(pre-activation) from the final output y
__device__ __forceinline__ float norm_act_bwd(float y, int act, int post_sigmoid) {
    if (post_sigmoid) {
        const float ds = y * (1.f - y);
        // r = act(bn) > 0  <=>  sigmoid(r) > 0.5
        float da = 1.f;
        if (act == HV_ACT_RELU) da = y > 0.5f ? 1.f : 0.f;
        else if (act == HV_ACT_LRELU) da = y > 0.5f ? 1.f : 0.2f;
        return ds * da;
    }
    return hv_act_grad_from_out(y, act);
}

// MODE 0: (sum x, sum x^2);  MODE 1: (sum g, sum g*xhat) with g = dy*act'(y), xhat = (x-mean)*rstd
// grid (chunk, group, channel slice); V floats per lane; rows of the chunk strided by 256/CB, two rows in flight per lane
template <int MODE, bool VEC, bool H>
__global__ __launch_bounds__(256) void norm_reduce_kernel(const NormK k, double* __restrict__ part) {
    constexpr int V = VEC ? 4 : 1;
    __shared__ double sh[256 * 2 * V];
    const int tid = threadIdx.x, g = blockIdx.y, CB = k.CB;
    const long long r0 = (long long)g * k.R + (long long)blockIdx.x * k.rpb;
    const long long r1 = min((long long)(g + 1) * k.R, r0 + k.rpb);
    const int C = k.C;
    double* dst = part + ((long long)g * k.nchunk + blockIdx.x) * 2 * C;
    const int lane = tid % CB, rp = tid / CB, rstep = 256 / CB;
    const int c0 = (blockIdx.z * CB + lane) * V;           // first channel of this lane
    double a[V], b[V];
    float mean[V], rstd[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        a[e] = 0; b[e] = 0; mean[e] = 0.f; rstd[e] = 1.f;
        if (MODE == 1) { mean[e] = k.stats[(long long)g * 2 * C + c0 + e]; rstd[e] = k.stats[(long long)g * 2 * C + C + c0 + e]; }
    }
    auto ldv = [&](const void* p, long long r, int ld, int coff, float (&o)[V]) __attribute__((always_inline)) {
        if (VEC) { const float4 v = hv_ld4(p, r * ld + coff + c0, H); o[0] = v.x; o[V > 1 ? 1 : 0] = v.y; o[V > 2 ? 2 : 0] = v.z; o[V > 3 ? 3 : 0] = v.w; }
        else o[0] = hv_ld1(p, r * ld + coff + c0, H);
    };
    auto acc1 = [&](const float (&xs)[V], const float (&ds)[V], const float (&ys)[V]) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
            if (MODE == 0) { a[e] += xs[e]; b[e] += (double)xs[e] * xs[e]; }
            else {
                const float gq = (k.act != HV_ACT_NONE || k.post_sigmoid) ? ds[e] * norm_act_bwd(ys[e], k.act, k.post_sigmoid) : ds[e];
                a[e] += gq;
                b[e] += (double)gq * ((xs[e] - mean[e]) * rstd[e]);
            }
        }
    };
    // act == none (the incoming gradient already carries act'(y): the consumer's data-gradient epilogue applied it): y is not read at all
    const bool need_y = MODE == 1 && (k.act != HV_ACT_NONE || k.post_sigmoid);
    long long r = r0 + rp;
    for (; r + rstep < r1; r += 2 * rstep) {     // two independent rows per iteration
        float x0[V], x1[V], d0[V], d1[V], y0[V] = {}, y1[V] = {};
        ldv(k.x, r, k.x_ld, k.x_coff, x0);
        ldv(k.x, r + rstep, k.x_ld, k.x_coff, x1);
        if (MODE == 1) { ldv(k.dy, r, k.dy_ld, k.dy_coff, d0); ldv(k.dy, r + rstep, k.dy_ld, k.dy_coff, d1); }
        if (need_y) { ldv(k.y, r, k.y_ld, k.y_coff, y0); ldv(k.y, r + rstep, k.y_ld, k.y_coff, y1); }
        acc1(x0, d0, y0);
        acc1(x1, d1, y1);
    }
    if (r < r1) {
        float x0[V], d0[V], y0[V] = {};
        ldv(k.x, r, k.x_ld, k.x_coff, x0);
        if (MODE == 1) ldv(k.dy, r, k.dy_ld, k.dy_coff, d0);
        if (need_y) ldv(k.y, r, k.y_ld, k.y_coff, y0);
        acc1(x0, d0, y0);
    }
#pragma unroll
    for (int e = 0; e < V; ++e) { sh[(tid * V + e) * 2] = a[e]; sh[(tid * V + e) * 2 + 1] = b[e]; }
    __syncthreads();
    if (tid < CB) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
            double sa = 0, sb = 0;
            for (int t = tid; t < 256; t += CB) { sa += sh[(t * V + e) * 2]; sb += sh[(t * V + e) * 2 + 1]; }
            dst[c0 + e] = sa;
            dst[C + c0 + e] = sb;
        }
    }
}

// one wave per channel; groups are visited in order so that several batch-norm groups in one launch (fake | real halves of a
// discriminator batch) update the running statistics exactly like consecutive forward calls
// fpart != NULL: the sums come from the producing conv's epilogue (hv_conv_desc.stats: [nchunk][C][2] floats, G == 1) instead of norm_reduce_kernel
__global__ void norm_fwd_finalize_kernel(const double* __restrict__ part, int G, int nchunk, int C, int R, float eps, float momentum,
                                         float* __restrict__ stats, float* running_mean, float* running_var, long long* nbt,
                                         int use_running, int update_running, const float* __restrict__ fpart) {
    const int c = blockIdx.x;
    if (c >= C) return;
    for (int g = 0; g < G; ++g) {
        if (use_running) {
            if (threadIdx.x == 0) {
                stats[(long long)g * 2 * C + c] = running_mean[c];
                stats[(long long)g * 2 * C + C + c] = 1.f / sqrtf(running_var[c] + eps);
            }
            continue;
        }
        double s = 0, q = 0;
        if (fpart) {      // the producing conv's partials are ordered by (image, tile): group g owns the g-th of G equal ranges
            const int per = nchunk / G;
            for (int k = g * per + threadIdx.x; k < (g + 1) * per; k += 64) {
                const float2 v = *reinterpret_cast<const float2*>(fpart + ((long long)k * C + c) * 2);
                s += v.x; q += v.y;
            }
        } else {
            for (int k = threadIdx.x; k < nchunk; k += 64) {
                s += part[((long long)g * nchunk + k) * 2 * C + c];
                q += part[((long long)g * nchunk + k) * 2 * C + C + c];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
        if (threadIdx.x == 0) {
            const double mean = s / R;
            double var = q / R - mean * mean;
            if (var < 0) var = 0;
            stats[(long long)g * 2 * C + c] = (float)mean;
            stats[(long long)g * 2 * C + C + c] = 1.f / sqrtf((float)var + eps);
            if (update_running) {
                const double unb = R > 1 ? var * R / (R - 1) : var;
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
                if (c == 0 && nbt) nbt[0] += 1;
            }
        }
    }
}

// grid (blocks, group): lanes per row CV = C/V is a power of two <= 256, so a thread keeps its channels over the grid-stride
// loop: per-channel constants live in registers and the row index is a shift
template <bool VEC, bool H>
__global__ __launch_bounds__(256) void norm_apply_kernel(const NormK k) {
    constexpr int V = VEC ? 4 : 1;
    const int C = k.C, g = blockIdx.y, lc = k.lc;
    const int cg = threadIdx.x & ((1 << lc) - 1), c0 = cg * V;
    float mean[V], rstd[V], gam[V], bet[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        mean[e] = k.stats[(long long)g * 2 * C + c0 + e];
        rstd[e] = k.stats[(long long)g * 2 * C + C + c0 + e];
        gam[e] = k.gamma ? k.gamma[c0 + e] : 1.f;
        bet[e] = k.gamma ? k.beta[c0 + e] : 0.f;
    }
    const long long base = (long long)g * k.R, n = (long long)k.R << lc, st = (long long)gridDim.x * 256;
    auto ldx = [&](long long i, float (&xs)[V]) __attribute__((always_inline)) {
        const long long r = base + (i >> lc);
        if (VEC) { const float4 xv = hv_ld4(k.x, r * k.x_ld + k.x_coff + c0, H); xs[0] = xv.x; xs[V > 1 ? 1 : 0] = xv.y; xs[V > 2 ? 2 : 0] = xv.z; xs[V > 3 ? 3 : 0] = xv.w; }
        else xs[0] = hv_ld1(k.x, r * k.x_ld + k.x_coff + c0, H);
    };
    auto sto = [&](long long i, const float (&xs)[V]) __attribute__((always_inline)) {
        const long long r = base + (i >> lc);
        float o[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float v = (xs[e] - mean[e]) * rstd[e];
            if (k.gamma) v = v * gam[e] + bet[e];
            o[e] = norm_act_fwd(v, k.act, k.post_sigmoid);
        }
        if (VEC) hv_st4(k.out, r * k.o_ld + k.o_coff + c0, make_float4(o[0], o[V > 1 ? 1 : 0], o[V > 2 ? 2 : 0], o[V > 3 ? 3 : 0]), H);
        else hv_st1(k.out, r * k.o_ld + k.o_coff + c0, o[0], H);
    };
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * st < n; i += 4 * st) {        // four items in flight per lane (one per iteration was a chain of load latencies: 11 us for 27 MB)
        float x0[V], x1[V], x2[V], x3[V];
        ldx(i, x0); ldx(i + st, x1); ldx(i + 2 * st, x2); ldx(i + 3 * st, x3);
        sto(i, x0); sto(i + st, x1); sto(i + 2 * st, x2); sto(i + 3 * st, x3);
    }
    for (; i < n; i += st) {
        float x0[V];
        ldx(i, x0);
        sto(i, x0);
    }
}

static int apply_grid(long long n, int G) { static const int ab = getenv("HV_NORM_APPLY_BLOCKS") ? atoi(getenv("HV_NORM_APPLY_BLOCKS")) : 1024;   /* step-level A/B: 15.12 ms at 1024, 15.19 at 2048, 15.42 at 4096 */ long long b = (n + 255) / 256, cap = ab / (G > 0 ? G : 1); if (cap < 8) cap = 8; return (int)(b > cap ? cap : (b < 1 ? 1 : b)); }
static int n_log2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

extern "C" int hv_norm_act_forward(const hv_norm_desc* d, void* stream) {
    if (!d || !d->x || !d->stats || d->B <= 0 || d->HW <= 0 || d->C <= 0) return HV_ERR_ARG;      // (y == NULL: statistics only -- the consumer normalises at its staging)
    if (d->norm != HV_NORM_BATCH && d->norm != HV_NORM_INSTANCE) return HV_ERR_ARG;
    if (!n_shape_ok(d->C)) return HV_ERR_UNSUPPORTED;
    const bool aligned = !(d->x_ld & 3) && !(d->x_coff & 3) && !(d->y_ld & 3) && !(d->y_coff & 3) && !((uintptr_t)d->x & 15) && !((uintptr_t)d->y & 15);
    const bool vec = n_vec_ok(d->C) && aligned;
    if (!vec && !(n_pow2(d->C) && d->C <= 256)) return HV_ERR_UNSUPPORTED;
    if (d->norm == HV_NORM_BATCH && (!d->gamma || !d->beta)) return HV_ERR_ARG;
    if (d->norm == HV_NORM_BATCH && d->groups > 1 && d->B % d->groups) return HV_ERR_ARG;
    NormPlan pl = norm_plan(d->B, d->HW, d->C, d->norm, d->groups, vec);
    const bool use_running = d->norm == HV_NORM_BATCH && !d->training;
    if (use_running && (!d->running_mean || !d->running_var)) return HV_ERR_ARG;
    const size_t need = (size_t)pl.G * pl.nchunk * 2 * d->C * sizeof(double);
    // statistics handed over by the producing conv (hv_conv_desc.stats): no reduction pass over x
    // (groups > 1: the partials of whole images, in image order -- every group is a contiguous, equal share of them)
    const bool handed = d->partials && d->n_partials > 0 && d->norm == HV_NORM_BATCH && !use_running && pl.G >= 1 && d->n_partials % pl.G == 0 &&
                        (d->n_partials / pl.G) % (d->B / pl.G) == 0;
    if (d->partials && d->n_partials > 0 && !use_running && !handed) return HV_ERR_ARG;      // partials that do not divide into the groups: never a silent reduction into an unchecked workspace
    if (!use_running && !handed && (!d->workspace || d->workspace_bytes < need || ((uintptr_t)d->workspace & 7))) return HV_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    NormK k = {};
    k.x = d->x; k.out = d->y; k.x_ld = d->x_ld; k.x_coff = d->x_coff; k.o_ld = d->y_ld; k.o_coff = d->y_coff;
    k.C = d->C; k.R = pl.R; k.rpb = pl.rpb; k.nchunk = pl.nchunk; k.act = d->act; k.post_sigmoid = d->post_sigmoid;
    k.CB = pl.CB; k.lc = n_log2(vec ? d->C / 4 : d->C);
    k.stats = d->stats; k.gamma = d->norm == HV_NORM_BATCH ? d->gamma : nullptr; k.beta = d->beta;
    double* part = (double*)d->workspace;
    if (!use_running && !handed) {
        dim3 grid(pl.nchunk, pl.G, pl.slices);
        if (d->f16) {
            if (vec) hipLaunchKernelGGL((norm_reduce_kernel<0, true, true>), grid, dim3(256), 0, s, k, part);
            else hipLaunchKernelGGL((norm_reduce_kernel<0, false, true>), grid, dim3(256), 0, s, k, part);
        } else {
            if (vec) hipLaunchKernelGGL((norm_reduce_kernel<0, true, false>), grid, dim3(256), 0, s, k, part);
            else hipLaunchKernelGGL((norm_reduce_kernel<0, false, false>), grid, dim3(256), 0, s, k, part);
        }
        HV_LAUNCH_CHECK();
    }
    const int update = d->norm == HV_NORM_BATCH && d->training && d->running_mean && d->running_var;
    static const int skip_fin = getenv("HV_DIAG_SKIP") && strstr(getenv("HV_DIAG_SKIP"), "norm_finalize") ? 1 : 0;     // timing-only diagnostic (stale statistics): what the finalize launches cost the step
    static int fin_calls = 0;
    if (!skip_fin || ++fin_calls <= 36)      // (the first two steps' statistics stay in place: sane operands for everything downstream)
    hipLaunchKernelGGL(norm_fwd_finalize_kernel, dim3(d->C), dim3(64), 0, s, part, pl.G, handed ? d->n_partials : pl.nchunk, d->C, pl.R,
                       d->eps, d->momentum, d->stats, d->running_mean, d->running_var, d->num_batches_tracked, use_running ? 1 : 0, update,
                       handed ? d->partials : nullptr);
    HV_LAUNCH_CHECK();
    const long long n = (long long)pl.R * (vec ? d->C / 4 : d->C);
    const dim3 agrid(apply_grid(n, pl.G), pl.G);
    static const int skip_apply = getenv("HV_DIAG_SKIP") && strstr(getenv("HV_DIAG_SKIP"), "norm_apply") ? 1 : 0;     // timing-only diagnostic (wrong results)
    if (skip_apply || !d->y) return HV_OK;
    if (d->f16) {
        if (vec) hipLaunchKernelGGL((norm_apply_kernel<true, true>), agrid, dim3(256), 0, s, k);
        else hipLaunchKernelGGL((norm_apply_kernel<false, true>), agrid, dim3(256), 0, s, k);
    } else {
        if (vec) hipLaunchKernelGGL((norm_apply_kernel<true, false>), agrid, dim3(256), 0, s, k);
        else hipLaunchKernelGGL((norm_apply_kernel<false, false>), agrid, dim3(256), 0, s, k);
    }
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// sums over chunks -> ab[g][2][C] (floats); dgamma/dbeta over groups
// fpart != NULL: the sums come from the epilogue of the data gradient that produced dy (hv_conv_desc.bstats: [nchunk][C][2] floats, group g = the g-th of G
// equal ranges of rows) instead of norm_reduce_kernel<1>
__global__ void norm_bwd_finalize_kernel(const double* __restrict__ part, int G, int nchunk, int C, float* __restrict__ ab,
                                         float* dgamma, float* dbeta, int accumulate, const float* __restrict__ fpart) {
    const int c = blockIdx.x;   // one wave per channel
    if (c >= C) return;
    double ta = 0, tb = 0;
    for (int g = 0; g < G; ++g) {
        double a = 0, b = 0;
        if (fpart) {
            const int per = nchunk / G;
            for (int k = g * per + threadIdx.x; k < (g + 1) * per; k += 64) {
                const float2 v = *reinterpret_cast<const float2*>(fpart + ((long long)k * C + c) * 2);
                a += v.x; b += v.y;
            }
        } else
        for (int k = threadIdx.x; k < nchunk; k += 64) {
            a += part[((long long)g * nchunk + k) * 2 * C + c];
            b += part[((long long)g * nchunk + k) * 2 * C + C + c];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
        if (threadIdx.x == 0) {
            ab[(long long)g * 2 * C + c] = (float)a;
            ab[(long long)g * 2 * C + C + c] = (float)b;
        }
        ta += a; tb += b;
    }
    if (threadIdx.x != 0) return;
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)tb : (float)tb;
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)ta : (float)ta;
}

template <bool VEC, bool H>
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(const NormK k, const float* __restrict__ ab, int batch_stats) {
    constexpr int V = VEC ? 4 : 1;
    const int C = k.C, g = blockIdx.y, lc = k.lc;
    const float invR = 1.f / (float)k.R;
    const int cg = threadIdx.x & ((1 << lc) - 1), c0 = cg * V;
    float mean[V], rstd[V], gam[V], sa[V], sb[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        mean[e] = k.stats[(long long)g * 2 * C + c0 + e];
        rstd[e] = k.stats[(long long)g * 2 * C + C + c0 + e];
        gam[e] = k.gamma ? k.gamma[c0 + e] : 1.f;
        sa[e] = ab[(long long)g * 2 * C + c0 + e];
        sb[e] = ab[(long long)g * 2 * C + C + c0 + e];
    }
    const bool need_y = k.act != HV_ACT_NONE || k.post_sigmoid;      // (see norm_reduce_kernel)
    const long long base = (long long)g * k.R, n = (long long)k.R << lc, st = (long long)gridDim.x * 256;
    struct Item { float xs[V], ds[V], ys[V]; };
    auto ld = [&](long long i, Item& t) __attribute__((always_inline)) {
        const long long r = base + (i >> lc);
        if (VEC) {
            const float4 xv = hv_ld4(k.x, r * k.x_ld + k.x_coff + c0, H);
            const float4 dv = hv_ld4(k.dy, r * k.dy_ld + k.dy_coff + c0, H);
            const float4 yv = need_y ? hv_ld4(k.y, r * k.y_ld + k.y_coff + c0, H) : make_float4(0.f, 0.f, 0.f, 0.f);
            t.xs[0] = xv.x; t.xs[V > 1 ? 1 : 0] = xv.y; t.xs[V > 2 ? 2 : 0] = xv.z; t.xs[V > 3 ? 3 : 0] = xv.w;
            t.ds[0] = dv.x; t.ds[V > 1 ? 1 : 0] = dv.y; t.ds[V > 2 ? 2 : 0] = dv.z; t.ds[V > 3 ? 3 : 0] = dv.w;
            t.ys[0] = yv.x; t.ys[V > 1 ? 1 : 0] = yv.y; t.ys[V > 2 ? 2 : 0] = yv.z; t.ys[V > 3 ? 3 : 0] = yv.w;
        } else {
            t.xs[0] = hv_ld1(k.x, r * k.x_ld + k.x_coff + c0, H);
            t.ds[0] = hv_ld1(k.dy, r * k.dy_ld + k.dy_coff + c0, H);
            t.ys[0] = need_y ? hv_ld1(k.y, r * k.y_ld + k.y_coff + c0, H) : 0.f;
        }
    };
    auto st1 = [&](long long i, const Item& t) __attribute__((always_inline)) {
        const long long r = base + (i >> lc);
        float o[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float gq = need_y ? t.ds[e] * norm_act_bwd(t.ys[e], k.act, k.post_sigmoid) : t.ds[e];
            float v;
            if (batch_stats) {
                const float xhat = (t.xs[e] - mean[e]) * rstd[e];
                v = gam[e] * rstd[e] * (gq - sa[e] * invR - xhat * sb[e] * invR);
            } else {
                v = gam[e] * rstd[e] * gq;
            }
            o[e] = v;
        }
        if (VEC) hv_st4(k.out, r * k.o_ld + k.o_coff + c0, make_float4(o[0], o[V > 1 ? 1 : 0], o[V > 2 ? 2 : 0], o[V > 3 ? 3 : 0]), H);
        else hv_st1(k.out, r * k.o_ld + k.o_coff + c0, o[0], H);
    };
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + st < n; i += 2 * st) {           // two items (four to six loads) in flight per lane
        Item t0, t1;
        ld(i, t0); ld(i + st, t1);
        st1(i, t0); st1(i + st, t1);
    }
    for (; i < n; i += st) {
        Item t0;
        ld(i, t0);
        st1(i, t0);
    }
}

extern "C" int hv_norm_act_backward(const hv_norm_bwd_desc* d, void* stream) {
    if (!d || !d->dy || !d->x || !d->dx || !d->stats || d->B <= 0 || d->HW <= 0 || d->C <= 0) return HV_ERR_ARG;
    const bool need_y = d->act != HV_ACT_NONE || d->post_sigmoid;     // act none: dy is the gradient at the normalisation's output, y is not read
    if (need_y && !d->y) return HV_ERR_ARG;
    if (d->norm != HV_NORM_BATCH && d->norm != HV_NORM_INSTANCE) return HV_ERR_ARG;
    if (!n_shape_ok(d->C)) return HV_ERR_UNSUPPORTED;
    const bool aligned = !((d->x_ld | d->x_coff | d->dy_ld | d->dy_coff | d->dx_ld | d->dx_coff) & 3) && (!need_y || !((d->y_ld | d->y_coff) & 3)) &&
                         !(((uintptr_t)d->x | (uintptr_t)(need_y ? d->y : nullptr) | (uintptr_t)d->dy | (uintptr_t)d->dx) & 15);
    const bool vec = n_vec_ok(d->C) && aligned;
    if (!vec && !(n_pow2(d->C) && d->C <= 256)) return HV_ERR_UNSUPPORTED;
    NormPlan pl = norm_plan(d->B, d->HW, d->C, d->norm, d->groups, vec);
    const size_t need_part = (size_t)pl.G * pl.nchunk * 2 * d->C * sizeof(double);
    const size_t need = need_part + (size_t)pl.G * 2 * d->C * sizeof(float);
    if (!d->workspace || d->workspace_bytes < need || ((uintptr_t)d->workspace & 7)) return HV_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    NormK k = {};
    k.x = d->x; k.y = d->y; k.dy = d->dy; k.out = d->dx;
    k.x_ld = d->x_ld; k.x_coff = d->x_coff; k.y_ld = d->y_ld; k.y_coff = d->y_coff; k.dy_ld = d->dy_ld; k.dy_coff = d->dy_coff;
    k.o_ld = d->dx_ld; k.o_coff = d->dx_coff;
    k.C = d->C; k.R = pl.R; k.rpb = pl.rpb; k.nchunk = pl.nchunk; k.act = d->act; k.post_sigmoid = d->post_sigmoid;
    k.CB = pl.CB; k.lc = n_log2(vec ? d->C / 4 : d->C);
    k.stats = d->stats; k.gamma = d->norm == HV_NORM_BATCH ? d->gamma : nullptr;
    double* part = (double*)d->workspace;
    float* ab = (float*)((char*)d->workspace + need_part);
    // sums handed over by the data gradient that wrote dy (hv_conv_desc.bstats): no reduction pass over dy and x
    const bool handed = d->partials && d->n_partials > 0;
    if (handed && (d->norm != HV_NORM_BATCH || !d->training || need_y || d->n_partials % pl.G)) return HV_ERR_ARG;
    if (!handed) {
        dim3 grid(pl.nchunk, pl.G, pl.slices);
        if (d->f16) {
            if (vec) hipLaunchKernelGGL((norm_reduce_kernel<1, true, true>), grid, dim3(256), 0, s, k, part);
            else hipLaunchKernelGGL((norm_reduce_kernel<1, false, true>), grid, dim3(256), 0, s, k, part);
        } else {
            if (vec) hipLaunchKernelGGL((norm_reduce_kernel<1, true, false>), grid, dim3(256), 0, s, k, part);
            else hipLaunchKernelGGL((norm_reduce_kernel<1, false, false>), grid, dim3(256), 0, s, k, part);
        }
        HV_LAUNCH_CHECK();
    }
    static const int skip_bfin = getenv("HV_DIAG_SKIP") && strstr(getenv("HV_DIAG_SKIP"), "norm_finalize") ? 1 : 0;     // (timing-only diagnostic)
    static int bfin_calls = 0;
    if (!skip_bfin || ++bfin_calls <= 30)
    hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3(d->C), dim3(64), 0, s, part, pl.G, handed ? d->n_partials : pl.nchunk, d->C, ab, d->dgamma, d->dbeta,
                       d->param_accumulate, handed ? d->partials : nullptr);
    HV_LAUNCH_CHECK();
    const int batch_stats = (d->norm == HV_NORM_INSTANCE || d->training) ? 1 : 0;
    const long long n = (long long)pl.R * (vec ? d->C / 4 : d->C);
    const dim3 agrid(apply_grid(n, pl.G), pl.G);
    static const int skip_bapply = getenv("HV_DIAG_SKIP") && strstr(getenv("HV_DIAG_SKIP"), "norm_bwd") ? 1 : 0;     // timing-only diagnostic (wrong results)
    if (skip_bapply) return HV_OK;
    if (d->f16) {
        if (vec) hipLaunchKernelGGL((norm_bwd_apply_kernel<true, true>), agrid, dim3(256), 0, s, k, ab, batch_stats);
        else hipLaunchKernelGGL((norm_bwd_apply_kernel<false, true>), agrid, dim3(256), 0, s, k, ab, batch_stats);
    } else {
        if (vec) hipLaunchKernelGGL((norm_bwd_apply_kernel<true, false>), agrid, dim3(256), 0, s, k, ab, batch_stats);
        else hipLaunchKernelGGL((norm_bwd_apply_kernel<false, false>), agrid, dim3(256), 0, s, k, ab, batch_stats);
    }
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// BatchNorm2d / InstanceNorm2d + LeakyReLU/ReLU(+sigmoid), forward and backward, NHWC fp32.
// HBM-bound: each pass streams the activation once with 16 B per lane; statistics are reduced in
// fp64 (per-thread partials -> per-block -> finalize) so var = E[x^2]-E[x]^2 is safe.
#include "hv_common.h"

static bool n_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static bool n_vec_ok(int C) { return (C % 4 == 0) && n_pow2(C / 4) && C / 4 <= 256; }
static bool n_shape_ok(int C) { return n_vec_ok(C) || (n_pow2(C) && C <= 256); }

struct NormPlan { int G, R, nchunk, rpb, rstep; };
static NormPlan norm_plan(int B, int HW, int C, int norm, int groups = 1) {
    NormPlan p;
    p.G = norm == HV_NORM_INSTANCE ? B : (groups > 0 ? groups : 1);
    p.R = norm == HV_NORM_INSTANCE ? HW : (B / p.G) * HW;
    p.rstep = n_vec_ok(C) ? 256 / (C / 4) : 256 / C;
    long long rpb = (long long)p.rstep * 32;
    long long nch = (p.R + rpb - 1) / rpb;
    const long long cap = p.G > 1 ? 32 : 256;
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
    const NormPlan a = norm_plan(B, HW, C, HV_NORM_BATCH), b = norm_plan(B, HW, C, HV_NORM_INSTANCE);
    const size_t pa = (size_t)a.G * a.nchunk, pb = (size_t)b.G * b.nchunk;
    return (pa > pb ? pa : pb) * 2 * C * sizeof(double) + (size_t)B * 2 * C * sizeof(float) + 64;
}

struct NormK {
    const float* x; const float* y; const float* dy; float* out;
    int x_ld, x_coff, y_ld, y_coff, dy_ld, dy_coff, o_ld, o_coff;
    int C, R, rpb, nchunk, act, post_sigmoid;
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
template <int MODE, bool VEC>
__global__ __launch_bounds__(256) void norm_reduce_kernel(const NormK k, double* __restrict__ part) {
    __shared__ double sh[256 * 2 * (VEC ? 4 : 1)];
    const int tid = threadIdx.x, g = blockIdx.y;
    const long long r0 = (long long)g * k.R + (long long)blockIdx.x * k.rpb;
    const long long r1 = min((long long)(g + 1) * k.R, r0 + k.rpb);
    const int C = k.C;
    double* dst = part + ((long long)g * k.nchunk + blockIdx.x) * 2 * C;
    if (VEC) {
        const int C4 = C >> 2, cg = tid % C4, rp = tid / C4, rstep = 256 / C4;
        double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
        float mean[4], rstd[4];
        if (MODE == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { mean[e] = k.stats[(long long)g * 2 * C + cg * 4 + e]; rstd[e] = k.stats[(long long)g * 2 * C + C + cg * 4 + e]; }
        }
        for (long long r = r0 + rp; r < r1; r += rstep) {
            const float4 xv = *reinterpret_cast<const float4*>(k.x + r * k.x_ld + k.x_coff + cg * 4);
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
            if (MODE == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { a[e] += xs[e]; b[e] += (double)xs[e] * xs[e]; }
            } else {
                const float4 dv = *reinterpret_cast<const float4*>(k.dy + r * k.dy_ld + k.dy_coff + cg * 4);
                const float4 yv = *reinterpret_cast<const float4*>(k.y + r * k.y_ld + k.y_coff + cg * 4);
                const float ds[4] = {dv.x, dv.y, dv.z, dv.w}, ys[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gq = ds[e] * norm_act_bwd(ys[e], k.act, k.post_sigmoid);
                    a[e] += gq;
                    b[e] += (double)gq * ((xs[e] - mean[e]) * rstd[e]);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { sh[(tid * 4 + e) * 2] = a[e]; sh[(tid * 4 + e) * 2 + 1] = b[e]; }
        __syncthreads();
        if (tid < C4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double sa = 0, sb = 0;
                for (int t = tid; t < 256; t += C4) { sa += sh[(t * 4 + e) * 2]; sb += sh[(t * 4 + e) * 2 + 1]; }
                dst[tid * 4 + e] = sa;
                dst[C + tid * 4 + e] = sb;
            }
        }
    } else {
        const int c = tid % C, rp = tid / C, rstep = 256 / C;
        double a = 0, b = 0;
        float mean = 0.f, rstd = 1.f;
        if (MODE == 1) { mean = k.stats[(long long)g * 2 * C + c]; rstd = k.stats[(long long)g * 2 * C + C + c]; }
        for (long long r = r0 + rp; r < r1; r += rstep) {
            const float xv = k.x[r * k.x_ld + k.x_coff + c];
            if (MODE == 0) { a += xv; b += (double)xv * xv; }
            else {
                const float gq = k.dy[r * k.dy_ld + k.dy_coff + c] * norm_act_bwd(k.y[r * k.y_ld + k.y_coff + c], k.act, k.post_sigmoid);
                a += gq;
                b += (double)gq * ((xv - mean) * rstd);
            }
        }
        sh[tid * 2] = a; sh[tid * 2 + 1] = b;
        __syncthreads();
        if (tid < C) {
            double sa = 0, sb = 0;
            for (int t = tid; t < 256; t += C) { sa += sh[t * 2]; sb += sh[t * 2 + 1]; }
            dst[tid] = sa;
            dst[C + tid] = sb;
        }
    }
}

// one wave per channel; groups are visited in order so that several batch-norm groups in one launch (fake | real halves of a
// discriminator batch) update the running statistics exactly like consecutive forward calls
__global__ void norm_fwd_finalize_kernel(const double* __restrict__ part, int G, int nchunk, int C, int R, float eps, float momentum,
                                         float* __restrict__ stats, float* running_mean, float* running_var, long long* nbt,
                                         int use_running, int update_running) {
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
        for (int k = threadIdx.x; k < nchunk; k += 64) {
            s += part[((long long)g * nchunk + k) * 2 * C + c];
            q += part[((long long)g * nchunk + k) * 2 * C + C + c];
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

template <bool VEC>
__global__ __launch_bounds__(256) void norm_apply_kernel(const NormK k, int G) {
    const int C = k.C;
    const long long total = (long long)G * k.R;
    if (VEC) {
        const int C4 = C >> 2;
        long long i = (long long)blockIdx.x * 256 + threadIdx.x;
        const long long n = total * C4, st = (long long)gridDim.x * 256;
        for (; i < n; i += st) {
            const long long r = i / C4;
            const int cg = (int)(i - r * C4), g = (int)(r / k.R);
            const float4 xv = *reinterpret_cast<const float4*>(k.x + r * k.x_ld + k.x_coff + cg * 4);
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = cg * 4 + e;
                float v = (xs[e] - k.stats[(long long)g * 2 * C + c]) * k.stats[(long long)g * 2 * C + C + c];
                if (k.gamma) v = v * k.gamma[c] + k.beta[c];
                o[e] = norm_act_fwd(v, k.act, k.post_sigmoid);
            }
            *reinterpret_cast<float4*>(k.out + r * k.o_ld + k.o_coff + cg * 4) = make_float4(o[0], o[1], o[2], o[3]);
        }
    } else {
        long long i = (long long)blockIdx.x * 256 + threadIdx.x;
        const long long n = total * C, st = (long long)gridDim.x * 256;
        for (; i < n; i += st) {
            const long long r = i / C;
            const int c = (int)(i - r * C), g = (int)(r / k.R);
            float v = (k.x[r * k.x_ld + k.x_coff + c] - k.stats[(long long)g * 2 * C + c]) * k.stats[(long long)g * 2 * C + C + c];
            if (k.gamma) v = v * k.gamma[c] + k.beta[c];
            k.out[r * k.o_ld + k.o_coff + c] = norm_act_fwd(v, k.act, k.post_sigmoid);
        }
    }
}

static int apply_grid(long long n) { long long b = (n + 255) / 256; return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

extern "C" int hv_norm_act_forward(const hv_norm_desc* d, void* stream) {
    if (!d || !d->x || !d->y || !d->stats || d->B <= 0 || d->HW <= 0 || d->C <= 0) return HV_ERR_ARG;
    if (d->norm != HV_NORM_BATCH && d->norm != HV_NORM_INSTANCE) return HV_ERR_ARG;
    if (!n_shape_ok(d->C)) return HV_ERR_UNSUPPORTED;
    const bool aligned = !(d->x_ld & 3) && !(d->x_coff & 3) && !(d->y_ld & 3) && !(d->y_coff & 3) && !((uintptr_t)d->x & 15) && !((uintptr_t)d->y & 15);
    const bool vec = n_vec_ok(d->C) && aligned;
    if (!vec && !(n_pow2(d->C) && d->C <= 256)) return HV_ERR_UNSUPPORTED;
    if (d->norm == HV_NORM_BATCH && (!d->gamma || !d->beta)) return HV_ERR_ARG;
    if (d->norm == HV_NORM_BATCH && d->groups > 1 && d->B % d->groups) return HV_ERR_ARG;
    NormPlan pl = norm_plan(d->B, d->HW, d->C, d->norm, d->groups);
    if (!vec) { pl.rstep = 256 / d->C; pl.rpb = (pl.rpb + pl.rstep - 1) / pl.rstep * pl.rstep; pl.nchunk = hv_cdiv(pl.R, pl.rpb); }
    const bool use_running = d->norm == HV_NORM_BATCH && !d->training;
    if (use_running && (!d->running_mean || !d->running_var)) return HV_ERR_ARG;
    const size_t need = (size_t)pl.G * pl.nchunk * 2 * d->C * sizeof(double);
    if (!use_running && (!d->workspace || d->workspace_bytes < need || ((uintptr_t)d->workspace & 7))) return HV_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    NormK k = {};
    k.x = d->x; k.out = d->y; k.x_ld = d->x_ld; k.x_coff = d->x_coff; k.o_ld = d->y_ld; k.o_coff = d->y_coff;
    k.C = d->C; k.R = pl.R; k.rpb = pl.rpb; k.nchunk = pl.nchunk; k.act = d->act; k.post_sigmoid = d->post_sigmoid;
    k.stats = d->stats; k.gamma = d->norm == HV_NORM_BATCH ? d->gamma : nullptr; k.beta = d->beta;
    double* part = (double*)d->workspace;
    if (!use_running) {
        dim3 grid(pl.nchunk, pl.G);
        if (vec) hipLaunchKernelGGL((norm_reduce_kernel<0, true>), grid, dim3(256), 0, s, k, part);
        else hipLaunchKernelGGL((norm_reduce_kernel<0, false>), grid, dim3(256), 0, s, k, part);
        HV_LAUNCH_CHECK();
    }
    const int update = d->norm == HV_NORM_BATCH && d->training && d->running_mean && d->running_var;
    hipLaunchKernelGGL(norm_fwd_finalize_kernel, dim3(d->C), dim3(64), 0, s, part, pl.G, pl.nchunk, d->C, pl.R,
                       d->eps, d->momentum, d->stats, d->running_mean, d->running_var, d->num_batches_tracked, use_running ? 1 : 0, update);
    HV_LAUNCH_CHECK();
    const long long n = (long long)pl.G * pl.R * (vec ? d->C / 4 : d->C);
    if (vec) hipLaunchKernelGGL((norm_apply_kernel<true>), dim3(apply_grid(n)), dim3(256), 0, s, k, pl.G);
    else hipLaunchKernelGGL((norm_apply_kernel<false>), dim3(apply_grid(n)), dim3(256), 0, s, k, pl.G);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// sums over chunks -> ab[g][2][C] (floats); dgamma/dbeta over groups
__global__ void norm_bwd_finalize_kernel(const double* __restrict__ part, int G, int nchunk, int C, float* __restrict__ ab,
                                         float* dgamma, float* dbeta, int accumulate) {
    const int c = blockIdx.x;   // one wave per channel
    if (c >= C) return;
    double ta = 0, tb = 0;
    for (int g = 0; g < G; ++g) {
        double a = 0, b = 0;
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

template <bool VEC>
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(const NormK k, int G, const float* __restrict__ ab, int batch_stats) {
    const int C = k.C;
    const long long total = (long long)G * k.R;
    const float invR = 1.f / (float)k.R;
    const int V = VEC ? 4 : 1, CV = C / V;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n = total * CV, st = (long long)gridDim.x * 256;
    for (; i < n; i += st) {
        const long long r = i / CV;
        const int cg = (int)(i - r * CV), g = (int)(r / k.R);
        float xs[4], ds[4], ys[4], o[4];
        if (VEC) {
            const float4 xv = *reinterpret_cast<const float4*>(k.x + r * k.x_ld + k.x_coff + cg * 4);
            const float4 dv = *reinterpret_cast<const float4*>(k.dy + r * k.dy_ld + k.dy_coff + cg * 4);
            const float4 yv = *reinterpret_cast<const float4*>(k.y + r * k.y_ld + k.y_coff + cg * 4);
            xs[0] = xv.x; xs[1] = xv.y; xs[2] = xv.z; xs[3] = xv.w;
            ds[0] = dv.x; ds[1] = dv.y; ds[2] = dv.z; ds[3] = dv.w;
            ys[0] = yv.x; ys[1] = yv.y; ys[2] = yv.z; ys[3] = yv.w;
        } else {
            xs[0] = k.x[r * k.x_ld + k.x_coff + cg];
            ds[0] = k.dy[r * k.dy_ld + k.dy_coff + cg];
            ys[0] = k.y[r * k.y_ld + k.y_coff + cg];
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int c = cg * V + e;
            const float mean = k.stats[(long long)g * 2 * C + c], rstd = k.stats[(long long)g * 2 * C + C + c];
            const float gq = ds[e] * norm_act_bwd(ys[e], k.act, k.post_sigmoid);
            const float gam = k.gamma ? k.gamma[c] : 1.f;
            float v;
            if (batch_stats) {
                const float xhat = (xs[e] - mean) * rstd;
                v = gam * rstd * (gq - ab[(long long)g * 2 * C + c] * invR - xhat * ab[(long long)g * 2 * C + C + c] * invR);
            } else {
                v = gam * rstd * gq;
            }
            o[e] = v;
        }
        if (VEC) *reinterpret_cast<float4*>(k.out + r * k.o_ld + k.o_coff + cg * 4) = make_float4(o[0], o[1], o[2], o[3]);
        else k.out[r * k.o_ld + k.o_coff + cg] = o[0];
    }
}

extern "C" int hv_norm_act_backward(const hv_norm_bwd_desc* d, void* stream) {
    if (!d || !d->dy || !d->y || !d->x || !d->dx || !d->stats || d->B <= 0 || d->HW <= 0 || d->C <= 0) return HV_ERR_ARG;
    if (d->norm != HV_NORM_BATCH && d->norm != HV_NORM_INSTANCE) return HV_ERR_ARG;
    if (!n_shape_ok(d->C)) return HV_ERR_UNSUPPORTED;
    const bool aligned = !((d->x_ld | d->x_coff | d->y_ld | d->y_coff | d->dy_ld | d->dy_coff | d->dx_ld | d->dx_coff) & 3) &&
                         !(((uintptr_t)d->x | (uintptr_t)d->y | (uintptr_t)d->dy | (uintptr_t)d->dx) & 15);
    const bool vec = n_vec_ok(d->C) && aligned;
    if (!vec && !(n_pow2(d->C) && d->C <= 256)) return HV_ERR_UNSUPPORTED;
    NormPlan pl = norm_plan(d->B, d->HW, d->C, d->norm, d->groups);
    if (!vec) { pl.rstep = 256 / d->C; pl.rpb = (pl.rpb + pl.rstep - 1) / pl.rstep * pl.rstep; pl.nchunk = hv_cdiv(pl.R, pl.rpb); }
    const size_t need_part = (size_t)pl.G * pl.nchunk * 2 * d->C * sizeof(double);
    const size_t need = need_part + (size_t)pl.G * 2 * d->C * sizeof(float);
    if (!d->workspace || d->workspace_bytes < need || ((uintptr_t)d->workspace & 7)) return HV_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    NormK k = {};
    k.x = d->x; k.y = d->y; k.dy = d->dy; k.out = d->dx;
    k.x_ld = d->x_ld; k.x_coff = d->x_coff; k.y_ld = d->y_ld; k.y_coff = d->y_coff; k.dy_ld = d->dy_ld; k.dy_coff = d->dy_coff;
    k.o_ld = d->dx_ld; k.o_coff = d->dx_coff;
    k.C = d->C; k.R = pl.R; k.rpb = pl.rpb; k.nchunk = pl.nchunk; k.act = d->act; k.post_sigmoid = d->post_sigmoid;
    k.stats = d->stats; k.gamma = d->norm == HV_NORM_BATCH ? d->gamma : nullptr;
    double* part = (double*)d->workspace;
    float* ab = (float*)((char*)d->workspace + need_part);
    dim3 grid(pl.nchunk, pl.G);
    if (vec) hipLaunchKernelGGL((norm_reduce_kernel<1, true>), grid, dim3(256), 0, s, k, part);
    else hipLaunchKernelGGL((norm_reduce_kernel<1, false>), grid, dim3(256), 0, s, k, part);
    HV_LAUNCH_CHECK();
    hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3(d->C), dim3(64), 0, s, part, pl.G, pl.nchunk, d->C, ab, d->dgamma, d->dbeta,
                       d->param_accumulate);
    HV_LAUNCH_CHECK();
    const int batch_stats = (d->norm == HV_NORM_INSTANCE || d->training) ? 1 : 0;
    const long long n = (long long)pl.G * pl.R * (vec ? d->C / 4 : d->C);
    if (vec) hipLaunchKernelGGL((norm_bwd_apply_kernel<true>), dim3(apply_grid(n)), dim3(256), 0, s, k, pl.G, ab, batch_stats);
    else hipLaunchKernelGGL((norm_bwd_apply_kernel<false>), dim3(apply_grid(n)), dim3(256), 0, s, k, pl.G, ab, batch_stats);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

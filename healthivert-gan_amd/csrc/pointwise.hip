// HBM-bound operators around the convolutions: layout changes, channel copies with nearest
// resampling, generator input assembly, pooled height head, Sobel, SHRM compositing, losses and
// their gradient seeds.  One pass over the data each, 16 B per lane where the layout allows.
#include "hv_common.h"

static int pw_grid(long long n, int cap = 8192) { long long b = (n + 255) / 256; return (int)(b > cap ? cap : (b < 1 ? 1 : b)); }
#define PW_LOOP(i, n) for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)
// flat index / small divisor: every tensor of the step has < 2^31 elements, where one 32-bit division (~20 instructions) stands for the 64-bit one (~100: the flat-index
// kernels spent most of their instructions there)
__device__ __forceinline__ long long pw_div(long long i, int d) { return (i >> 31) == 0 ? (long long)((unsigned)i / (unsigned)d) : i / d; }

// ------------------------------------------------------------------------------------------------ layout
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, void* __restrict__ dst, int dh, int C, int HW, int dst_ld, int dst_coff, long long n) {
    PW_LOOP(i, n) {  // i over dst order (b, p, c)
        const int c = (int)(i % C);
        const long long bp = i / C;
        const long long b = bp / HW, p = bp - b * HW;
        hv_st1(dst, bp * dst_ld + dst_coff + c, src[(b * C + c) * HW + p], dh);
    }
}
__global__ void nhwc_to_nchw_kernel(const void* __restrict__ src, int sh, float* __restrict__ dst, int C, int HW, int src_ld, int src_coff, int acc, long long n) {
    PW_LOOP(i, n) {  // i over dst order (b, c, p)
        const long long p = i % HW;
        const long long bc = i / HW;
        const long long b = bc / C;
        const int c = (int)(bc - b * C);
        const float v = hv_ld1(src, (b * HW + p) * src_ld + src_coff + c, sh);
        dst[i] = acc ? dst[i] + v : v;
    }
}
extern "C" int hv_nchw_to_nhwc(const float* src, void* dst, int dst_f16, int B, int C, int H, int W, int dst_ld, int dst_coff, void* stream) {
    if (!src || !dst || B <= 0 || C <= 0 || H <= 0 || W <= 0 || dst_ld < dst_coff + C) return HV_ERR_ARG;
    const long long n = (long long)B * C * H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(pw_grid(n)), dim3(256), 0, (hipStream_t)stream, src, dst, dst_f16, C, H * W, dst_ld, dst_coff, n);
    HV_LAUNCH_CHECK();
    return HV_OK;
}
extern "C" int hv_nhwc_to_nchw(const void* src, int src_f16, float* dst, int B, int C, int H, int W, int src_ld, int src_coff, int accumulate, void* stream) {
    if (!src || !dst || B <= 0 || C <= 0 || H <= 0 || W <= 0 || src_ld < src_coff + C) return HV_ERR_ARG;
    const long long n = (long long)B * C * H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(pw_grid(n)), dim3(256), 0, (hipStream_t)stream, src, src_f16, dst, C, H * W, src_ld, src_coff, accumulate, n);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// mode 0 same size; 1 src half size (nearest x2 up); 2 src double size (nearest x1/2: even indices);
// 3 dst(half) (+)= sum of the 2x2 block of src(full)  [adjoint of 1];  4 dst(full) (+)= src(half) at even idx else 0 [adjoint of 2]
// grid (pieces of a row / 256, rows): the row (image, h) is a scalar and the position in the row a 32-bit index -- the flat-index form spent five 64-bit
// divisions per element (12 us for a 22-MB copy that the memory system moves in 5)
template <int V>
__global__ __launch_bounds__(256) void copy_channels_kernel(const void* __restrict__ src, int sh, void* __restrict__ dst, int dh, int H, int W, int C, int src_ld, int src_coff,
                                     int dst_ld, int dst_coff, int mode, int acc, int rows) {
    const int CV = C / V, per_row = W * CV;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= per_row) return;
    const int w = CV == 1 ? e : e / CV, cg = e - w * CV;
    for (int row = blockIdx.y; row < rows; row += gridDim.y) {
        const int b = row / H, h = row - b * H;      // (scalar)
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        auto rd = [&](long long pix) {
            const long long si = pix * src_ld + src_coff + cg * V;
            if (V == 4) { const float4 t = hv_ld4(src, si, sh); v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w; }
            else v[0] += hv_ld1(src, si, sh);
        };
        if (mode == 0) rd(((long long)b * H + h) * W + w);
        else if (mode == 1) rd(((long long)b * (H >> 1) + (h >> 1)) * (W >> 1) + (w >> 1));
        else if (mode == 2) rd(((long long)b * (H * 2) + 2 * h) * (W * 2) + 2 * w);
        else if (mode == 3) {
            const long long p = ((long long)b * (H * 2) + 2 * h) * (W * 2) + 2 * w;
            rd(p); rd(p + 1); rd(p + 2 * W); rd(p + 2 * W + 1);
        } else {
            if (!((h | w) & 1)) rd(((long long)b * (H >> 1) + (h >> 1)) * (W >> 1) + (w >> 1));
        }
        const long long di = (((long long)b * H + h) * W + w) * dst_ld + dst_coff + cg * V;
        if (V == 4) {
            float4 o = make_float4(v[0], v[1], v[2], v[3]);
            if (acc) { const float4 t = hv_ld4(dst, di, dh); o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w; }
            hv_st4(dst, di, o, dh);
        } else {
            hv_st1(dst, di, acc ? hv_ld1(dst, di, dh) + v[0] : v[0], dh);
        }
    }
}
extern "C" int hv_copy_channels(const void* src, int src_f16, void* dst, int dst_f16, int B, int H, int W, int C, int src_ld, int src_coff, int dst_ld,
                                int dst_coff, int mode, int accumulate, void* stream) {
    if (!src || !dst || B <= 0 || H <= 0 || W <= 0 || C <= 0 || mode < 0 || mode > 4) return HV_ERR_ARG;
    if (src_ld < src_coff + C || dst_ld < dst_coff + C) return HV_ERR_ARG;
    if ((mode == 1 || mode == 4) && ((H | W) & 1)) return HV_ERR_UNSUPPORTED;
    const bool vec = !(C & 3) && !(src_ld & 3) && !(src_coff & 3) && !(dst_ld & 3) && !(dst_coff & 3) && !((uintptr_t)src & 15) && !((uintptr_t)dst & 15);
    const long long per_row = (long long)W * (vec ? C / 4 : C), rows = (long long)B * H;
    if (per_row >= (1ll << 30) || rows >= (1ll << 31)) return HV_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)((per_row + 255) / 256), (unsigned)(rows < 32768 ? rows : 32768));
    if (vec) hipLaunchKernelGGL((copy_channels_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, src, src_f16, dst, dst_f16, H, W, C, src_ld, src_coff, dst_ld, dst_coff, mode, accumulate, (int)rows);
    else hipLaunchKernelGGL((copy_channels_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, src, src_f16, dst, dst_f16, H, W, C, src_ld, src_coff, dst_ld, dst_coff, mode, accumulate, (int)rows);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// dst = a + b over C channels of same-size tensors (each with its own storage, channel stride and offset): the two-launch "copy, then accumulate" of a gradient
// that has two contributions (x_stage1 and coarse_seg feed both a loss and the refinement generator) as one pass
__global__ __launch_bounds__(256) void add_channels_kernel(const void* __restrict__ a, int ah, int a_ld, int a_coff, const void* __restrict__ b, int bh, int b_ld, int b_coff,
                                                           void* __restrict__ dst, int dh, int d_ld, int d_coff, int C, long long npix) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= npix * C) return;
    const long long pix = C == 1 ? e : e / C;
    const int c = (int)(e - pix * C);
    hv_st1(dst, pix * d_ld + d_coff + c, hv_ld1(a, pix * a_ld + a_coff + c, ah) + hv_ld1(b, pix * b_ld + b_coff + c, bh), dh);
}
extern "C" int hv_add_channels(const void* a, int a_f16, int a_ld, int a_coff, const void* b, int b_f16, int b_ld, int b_coff, void* dst, int dst_f16, int dst_ld,
                               int dst_coff, long long npix, int C, void* stream) {
    if (!a || !b || !dst || npix <= 0 || C <= 0 || a_ld < a_coff + C || b_ld < b_coff + C || dst_ld < dst_coff + C) return HV_ERR_ARG;
    if (npix * C >= (1ll << 40)) return HV_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(add_channels_kernel, dim3((unsigned)((npix * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, a_f16, a_ld, a_coff, b, b_f16, b_ld, b_coff, dst,
                       dst_f16, dst_ld, dst_coff, C, npix);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// ------------------------------------------------------------------------------------------------ generator input
__global__ void gen_input_kernel(const float* __restrict__ x, const float* __restrict__ seg, const float* __restrict__ mask,
                                 const double* __restrict__ ratio, void* __restrict__ dst, int dh, int HW, int CP, int order, long long n) {
    PW_LOOP(i, n) {
        const long long b = pw_div(i, HW);
        const float r = (float)ratio[b];
        float c0 = x[i], c1, c2, c3 = 0.f;
        if (order == 0) { c1 = r; c2 = mask[i]; }
        else { c1 = seg[i]; c2 = mask[i]; c3 = r; }
        hv_st4(dst, i * CP, make_float4(c0, c1, c2, c3), dh);
        for (int c = 4; c < CP; ++c) hv_st1(dst, i * CP + c, 0.f, dh);
    }
}
extern "C" int hv_gen_input(const float* x, const float* seg, const float* mask, const double* slice_ratio, void* dst, int dst_f16, int B, int H,
                            int W, int CP, int order, void* stream) {
    if (!x || !mask || !slice_ratio || !dst || B <= 0 || H <= 0 || W <= 0 || CP < 4 || (CP & 3) || (order == 1 && !seg)) return HV_ERR_ARG;
    const long long n = (long long)B * H * W;
    hipLaunchKernelGGL(gen_input_kernel, dim3(pw_grid(n)), dim3(256), 0, (hipStream_t)stream, x, seg, mask, slice_ratio, dst, dst_f16, H * W, CP, order, n);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// ------------------------------------------------------------------------------------------------ pooled height head
#define GAP_CHUNKS 32
__global__ __launch_bounds__(256) void gap_partial_kernel(const void* __restrict__ x, int xh, int HW, int C, int x_ld, float* __restrict__ part) {
    __shared__ float sh[256];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int c = tid % C, rp = tid / C, rstep = 256 / C;
    const int rows = (HW + GAP_CHUNKS - 1) / GAP_CHUNKS;
    const int r0 = blockIdx.x * rows, r1 = min(HW, r0 + rows);
    float s = 0.f;
    for (int r = r0 + rp; r < r1; r += rstep) s += hv_ld1(x, ((long long)b * HW + r) * x_ld + c, xh);
    sh[tid] = s;
    __syncthreads();
    if (tid < C) {
        float t = 0.f;
        for (int k = tid; k < 256; k += C) t += sh[k];
        part[((long long)b * GAP_CHUNKS + blockIdx.x) * C + tid] = t;
    }
}
__global__ void gap_fc_kernel(const float* __restrict__ part, int HW, int C, const float* __restrict__ w, const float* __restrict__ bias,
                              float* __restrict__ pooled, float* __restrict__ pred) {
    __shared__ float red[20];
    const int b = blockIdx.x, c = threadIdx.x;
    float t = 0.f;
    if (c < C) {
        for (int k = 0; k < GAP_CHUNKS; ++k) t += part[((long long)b * GAP_CHUNKS + k) * C + c];
        t /= (float)HW;
        pooled[(long long)b * C + c] = t;
        t *= w[c];
    }
    const float s = hv_block_sum(t, red);
    if (c == 0) pred[b] = 1.f / (1.f + expf(-(s + bias[0])));
}
extern "C" size_t hv_gap_fc_workspace_bytes(int B, int C) { return (size_t)B * GAP_CHUNKS * C * sizeof(float); }
extern "C" int hv_gap_fc_sigmoid(const void* x, int x_f16, int B, int HW, int C, int x_ld, const float* fc_w, const float* fc_b, float* pooled,
                                 float* pred, float* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !fc_w || !fc_b || !pooled || !pred || B <= 0 || HW <= 0 || C <= 0 || x_ld < C) return HV_ERR_ARG;
    if (C > 256 || (C & (C - 1))) return HV_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < hv_gap_fc_workspace_bytes(B, C)) return HV_ERR_WORKSPACE;
    hipLaunchKernelGGL(gap_partial_kernel, dim3(GAP_CHUNKS, B), dim3(256), 0, (hipStream_t)stream, x, x_f16, HW, C, x_ld, workspace);
    HV_LAUNCH_CHECK();
    hipLaunchKernelGGL(gap_fc_kernel, dim3(B), dim3(C < 64 ? 64 : C), 0, (hipStream_t)stream, workspace, HW, C, fc_w, fc_b, pooled, pred);
    HV_LAUNCH_CHECK();
    return HV_OK;
}
__global__ void gap_bwd_dx_kernel(const float* __restrict__ dpred, const float* __restrict__ pred, const float* __restrict__ w, void* __restrict__ dx, int dxh,
                                  int HW, int C, int dx_ld, long long n, const void* __restrict__ mul, int mulh, int mul_ld, int mul_act) {
    PW_LOOP(i, n) {
        const long long bp = pw_div(i, C);
        const int c = (int)(i - bp * C);
        const long long b = pw_div(bp, HW);
        const float p = pred[b];
        float g = dpred[b] * p * (1.f - p) * w[c] / (float)HW;
        if (mul) g *= hv_act_grad_from_out(hv_ld1(mul, bp * mul_ld + c, mulh), mul_act);      // dx holds PRE-activation gradients (its other writers applied act' too)
        hv_st1(dx, bp * dx_ld + c, hv_ld1(dx, bp * dx_ld + c, dxh) + g, dxh);
    }
}
__global__ void gap_bwd_param_kernel(const float* __restrict__ dpred, const float* __restrict__ pred, const float* __restrict__ pooled, int B, int C,
                                     float* dw, float* db, int acc) {
    const int c = threadIdx.x;
    float sw = 0.f, sb = 0.f;
    for (int b = 0; b < B; ++b) {
        const float p = pred[b], dl = dpred[b] * p * (1.f - p);
        sb += dl;
        if (c < C) sw += dl * pooled[(long long)b * C + c];
    }
    if (c < C) dw[c] = acc ? dw[c] + sw : sw;
    if (c == 0) db[0] = acc ? db[0] + sb : sb;
}
extern "C" int hv_gap_fc_sigmoid_backward(const float* dpred, const float* pred, const float* pooled, const float* fc_w, void* dx, int dx_f16, int B,
                                          int HW, int C, int dx_ld, float* dw, float* db, int accumulate, const void* mul_src, int mul_f16, int mul_ld,
                                          int mul_act, void* stream) {
    if (!dpred || !pred || !pooled || !fc_w || !dx || !dw || !db || B <= 0 || HW <= 0 || C <= 0 || C > 1024) return HV_ERR_ARG;
    if (mul_src && (mul_ld < C || mul_act < HV_ACT_NONE || mul_act > HV_ACT_CLAMP)) return HV_ERR_ARG;
    const long long n = (long long)B * HW * C;
    hipLaunchKernelGGL(gap_bwd_dx_kernel, dim3(pw_grid(n)), dim3(256), 0, (hipStream_t)stream, dpred, pred, fc_w, dx, dx_f16, HW, C, dx_ld, n, mul_src, mul_f16, mul_ld,
                       mul_act);
    HV_LAUNCH_CHECK();
    hipLaunchKernelGGL(gap_bwd_param_kernel, dim3(1), dim3(C < 64 ? 64 : C), 0, (hipStream_t)stream, dpred, pred, pooled, B, C, dw, db, accumulate);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// ------------------------------------------------------------------------------------------------ Sobel
__global__ void sobel_kernel(const float* __restrict__ img, float* __restrict__ out, int H, int W, long long n) {
    PW_LOOP(i, n) {
        const long long r = pw_div(i, W);
        const int w = (int)(i - r * W);
        const int h = (int)(r - pw_div(r, H) * H);
        const float* p = img + (r - h) * W;  // image base
        const int hm = max(h - 1, 0), hp = min(h + 1, H - 1), wm = max(w - 1, 0), wp = min(w + 1, W - 1);
        const float a = p[hm * W + wm], b = p[hm * W + w], c = p[hm * W + wp];
        const float d = p[h * W + wm], f = p[h * W + wp];
        const float g = p[hp * W + wm], hh = p[hp * W + w], k = p[hp * W + wp];
        const float gx = (c - a) + 2.f * (f - d) + (k - g);
        const float gy = (a + 2.f * b + c) - (g + 2.f * hh + k);
        out[i] = fminf(sqrtf(gx * gx + gy * gy), 1.f);
    }
}
extern "C" int hv_sobel(const float* img, float* out, int B, int H, int W, void* stream) {
    if (!img || !out || B <= 0 || H <= 0 || W <= 0) return HV_ERR_ARG;
    const long long n = (long long)B * H * W;
    hipLaunchKernelGGL(sobel_kernel, dim3(pw_grid(n)), dim3(256), 0, (hipStream_t)stream, img, out, H, W, n);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// ------------------------------------------------------------------------------------------------ SHRM compositing
struct Rows { int xu, xb, sh; };
__device__ __forceinline__ Rows shrm_rows(float pred_scaled, long long height, long long x1) {
    long long h = (long long)ceilf(pred_scaled);
    if (h < height) h = height;
    const long long d = h - height;
    Rows r;
    r.sh = (int)(d / 2);
    r.xu = (int)(x1 - d / 2);
    r.xb = r.xu + (int)h;
    return r;
}
__device__ __forceinline__ float shrm_pick(const float* gen_img, const float* real_img, int row, int col, int W, const Rows& r, int x2) {
    if (row < r.xu) return real_img[(row + r.sh) * W + col];
    if (row < r.xb) return gen_img[row * W + col];
    return real_img[(x2 + row - r.xb) * W + col];
}

__global__ void post_generator_kernel(const hv_postg_desc d, long long n) {
    const int HW = d.H * d.W, c0 = d.W / 2 - d.half_band, c1 = d.W / 2 + d.half_band;
    PW_LOOP(i, n) {
        const int b = (int)pw_div(i, HW), p = (int)(i - (long long)b * HW);
        const int row = p / d.W, col = p - row * d.W;
        const float mh = (float)d.maxheight[b];
        const float p1 = d.pred1[b] * mh, p2 = d.pred2[b] * mh;
        const Rows r2 = shrm_rows(p2, d.height[b], d.x1[b]), r1 = shrm_rows(p1, d.height[b], d.x1[b]);
        const int x2 = (int)d.x2[b];
        const float* real = d.real_B + (long long)b * HW;
        const float fb = shrm_pick(d.x_stage2 + (long long)b * HW, real, row, col, d.W, r2, x2);
        const float fc = shrm_pick(d.x_stage1 + (long long)b * HW, real, row, col, d.W, r1, x2);
        d.fake_B[i] = fb;
        d.fake_B_coarse[i] = fc;
        const float band = (col >= c0 && col < c1) ? 1.f : 0.f;
        const float m = d.mask[i];
        d.fake_B_local[i] = m * fb * band;
        d.real_B_local[i] = m * real[p] * band;
        d.fine_bin[i] = d.fine_seg[i] > 0.5f ? 1.f : 0.f;
        d.coarse_bin[i] = d.coarse_seg[i] > 0.5f ? 1.f : 0.f;
        if (p == 0) {
            d.pred1_h[b] = p1;
            d.pred2_h[b] = p2;
            d.rows[b * 4 + 0] = r2.xu; d.rows[b * 4 + 1] = r2.xb; d.rows[b * 4 + 2] = r1.xu; d.rows[b * 4 + 3] = r1.xb;
        }
    }
}
extern "C" int hv_post_generator(const hv_postg_desc* d, void* stream) {
    if (!d || !d->real_B || !d->mask || !d->x_stage1 || !d->x_stage2 || !d->fine_seg || !d->coarse_seg || !d->pred1 || !d->pred2 ||
        !d->height || !d->x1 || !d->x2 || !d->maxheight || !d->fake_B || !d->fake_B_coarse || !d->fake_B_local || !d->real_B_local ||
        !d->fine_bin || !d->coarse_bin || !d->pred1_h || !d->pred2_h || !d->rows || d->B <= 0 || d->H <= 0 || d->W <= 0)
        return HV_ERR_ARG;
    const long long n = (long long)d->B * d->H * d->W;
    hipLaunchKernelGGL(post_generator_kernel, dim3(pw_grid(n)), dim3(256), 0, (hipStream_t)stream, *d, n);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

__global__ void shrm_composite_kernel(const float* __restrict__ gen, const float* __restrict__ real, const float* __restrict__ pred_scaled,
                                      const long long* __restrict__ height, const long long* __restrict__ x1, const long long* __restrict__ x2,
                                      float* __restrict__ out, int* rows, int H, int W, long long n) {
    const int HW = H * W;
    PW_LOOP(i, n) {
        const int b = (int)pw_div(i, HW), p = (int)(i - (long long)b * HW);
        const int row = p / W, col = p - row * W;
        const Rows r = shrm_rows(pred_scaled[b], height[b], x1[b]);
        out[i] = shrm_pick(gen + (long long)b * HW, real + (long long)b * HW, row, col, W, r, (int)x2[b]);
        if (p == 0 && rows) { rows[b * 2] = r.xu; rows[b * 2 + 1] = r.xb; }
    }
}
extern "C" int hv_shrm_composite(const float* gen, const float* real, const float* pred_scaled, const long long* height, const long long* x1,
                                 const long long* x2, float* out, int* rows, int B, int H, int W, void* stream) {
    if (!gen || !real || !pred_scaled || !height || !x1 || !x2 || !out || B <= 0 || H <= 0 || W <= 0) return HV_ERR_ARG;
    const long long n = (long long)B * H * W;
    hipLaunchKernelGGL(shrm_composite_kernel, dim3(pw_grid(n)), dim3(256), 0, (hipStream_t)stream, gen, real, pred_scaled, height, x1, x2, out, rows, H, W, n);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

__global__ void shrm_backward_kernel(const float* __restrict__ d_fake, const float* __restrict__ d_local, const float* __restrict__ mask,
                                     const int* __restrict__ rows, int which, float* __restrict__ d_gen, int H, int W, int half_band, int acc, long long n) {
    const int HW = H * W, c0 = W / 2 - half_band, c1 = W / 2 + half_band;
    PW_LOOP(i, n) {
        const int b = (int)pw_div(i, HW), p = (int)(i - (long long)b * HW);
        const int row = p / W, col = p - row * W;
        const int xu = rows[b * 4 + which * 2], xb = rows[b * 4 + which * 2 + 1];
        float g = 0.f;
        if (row >= xu && row < xb) {
            g = d_fake ? d_fake[i] : 0.f;
            if (d_local && col >= c0 && col < c1) g += d_local[i] * mask[i];
        }
        d_gen[i] = acc ? d_gen[i] + g : g;
    }
}
extern "C" int hv_shrm_backward(const float* d_fake, const float* d_local, const float* mask, const int* rows, int which, float* d_gen,
                                int B, int H, int W, int half_band, int accumulate, void* stream) {
    if (!rows || !d_gen || (d_local && !mask) || which < 0 || which > 1 || B <= 0 || H <= 0 || W <= 0) return HV_ERR_ARG;
    const long long n = (long long)B * H * W;
    hipLaunchKernelGGL(shrm_backward_kernel, dim3(pw_grid(n)), dim3(256), 0, (hipStream_t)stream, d_fake, d_local, mask, rows, which, d_gen, H, W, half_band, accumulate, n);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// ------------------------------------------------------------------------------------------------ GAN loss (PatchGAN logits, small n)
__global__ __launch_bounds__(1024) void gan_loss_kernel(const float* __restrict__ z, long long n, float t, int mode, float lw, float* loss, int lacc,
                                                        float gw, float* __restrict__ dz) {
    __shared__ float red[20];
    float s = 0.f;
    auto term = [&](float v, float& g) {
        if (mode == 0) {   // one exponential serves both: e = exp(-|v|); sigmoid(v) = 1/(1+e) or e/(1+e); softplus tail = log1p(e)
            const float e = expf(-fabsf(v)), r = 1.f / (1.f + e);
            g = (v >= 0.f ? r : e * r) - t;
            return fmaxf(v, 0.f) - v * t + log1pf(e);
        }
        g = 2.f * (v - t);
        return (v - t) * (v - t);
    };
    if (n <= 16 * 1024) {   // PatchGAN logits (B x 30 x 30): all of a lane's loads in flight at once -- the rolled loop below pays one
        float v[16];        // memory round trip per 1024 elements (19 us for 14 400 logits); the summation order is the same
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const long long i = threadIdx.x + k * 1024;
            v[k] = i < n ? z[i] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const long long i = threadIdx.x + k * 1024;
            if (i < n) {
                float g;
                s += term(v[k], g);
                if (dz) dz[i] = gw * g / (float)n;
            }
        }
    } else {
        for (long long i = threadIdx.x; i < n; i += 1024) {
            float g;
            s += term(z[i], g);
            if (dz) dz[i] = gw * g / (float)n;
        }
    }
    s = hv_block_sum(s, red);
    if (threadIdx.x == 0 && loss) {
        const float v = lw * s / (float)n;
        loss[0] = lacc ? loss[0] + v : v;
    }
}
extern "C" int hv_gan_loss(const float* z, long long n, int target_is_real, int mode, float loss_weight, float* loss, int loss_accumulate,
                           float grad_weight, float* dz, void* stream) {
    if (!z || n <= 0 || mode < 0 || mode > 1) return HV_ERR_ARG;
    hipLaunchKernelGGL(gan_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, z, n, target_is_real ? 1.f : 0.f, mode, loss_weight, loss,
                       loss_accumulate, grad_weight, dz);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// The same over many workgroups (one CU evaluates the 14 400 logits' exp / log1p in ~19 us): 256 elements per workgroup, per-workgroup sums
// to the caller's scratch, the last stage adds them in index order (deterministic).
__global__ __launch_bounds__(256) void gan_loss_part_kernel(const float* __restrict__ z, long long n, float t, int mode, float gw, float* __restrict__ dz,
                                                            float* __restrict__ part) {
    __shared__ float red[20];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    float l = 0.f;
    if (i < n) {
        const float v = z[i];
        float g;
        if (mode == 0) {
            const float e = expf(-fabsf(v)), r = 1.f / (1.f + e);
            g = (v >= 0.f ? r : e * r) - t;
            l = fmaxf(v, 0.f) - v * t + log1pf(e);
        } else {
            g = 2.f * (v - t);
            l = (v - t) * (v - t);
        }
        if (dz) dz[i] = gw * g / (float)n;
    }
    l = hv_block_sum(l, red);
    if (threadIdx.x == 0) part[blockIdx.x] = l;
}
__global__ __launch_bounds__(256) void gan_loss_final_kernel(const float* __restrict__ part, int nparts, long long n, float lw, float* loss, int lacc) {
    __shared__ float red[20];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
    s = hv_block_sum(s, red);
    if (threadIdx.x == 0) {
        const float v = lw * s / (float)n;
        loss[0] = lacc ? loss[0] + v : v;
    }
}
extern "C" size_t hv_gan_loss_workspace_bytes(long long n) { return (size_t)((n + 255) / 256) * sizeof(float); }
extern "C" int hv_gan_loss_ws(const float* z, long long n, int target_is_real, int mode, float loss_weight, float* loss, int loss_accumulate,
                              float grad_weight, float* dz, void* workspace, size_t workspace_bytes, void* stream) {
    if (!z || n <= 0 || mode < 0 || mode > 1) return HV_ERR_ARG;
    if (!workspace || workspace_bytes < hv_gan_loss_workspace_bytes(n) || ((uintptr_t)workspace & 3)) return HV_ERR_WORKSPACE;
    const int nparts = (int)((n + 255) / 256);
    float* part = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(gan_loss_part_kernel, dim3(nparts), dim3(256), 0, (hipStream_t)stream, z, n, target_is_real ? 1.f : 0.f, mode, grad_weight, dz, part);
    HV_LAUNCH_CHECK();
    if (loss) {
        hipLaunchKernelGGL(gan_loss_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, part, nparts, n, loss_weight, loss, loss_accumulate);
        HV_LAUNCH_CHECK();
    }
    return HV_OK;
}

// The PatchGAN loss head in two launches: loss, d loss / d logit written STRAIGHT into the logits layer's padded fp16 gradient carrier ([n][4], channels
// 1-3 zero) and the logits layer's bias gradient (= the sum of the stored values).  The separate passes (fp32 dz -> carrier copy, bias column sums and
// their finalize) were four more small launches in the chain between a discriminator's forward and its backward, three times per step.
__global__ __launch_bounds__(256) void gan_loss_head_part_kernel(const float* __restrict__ z, long long n, float t, int mode, float gw, float* __restrict__ dz,
                                                                 _Float16* __restrict__ carrier, float* __restrict__ part, float* __restrict__ gpart) {
    __shared__ float red[20];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    float l = 0.f, gs = 0.f;
    if (i < n) {
        const float v = z[i];
        float g;
        if (mode == 0) {
            const float e = expf(-fabsf(v)), r = 1.f / (1.f + e);
            g = (v >= 0.f ? r : e * r) - t;
            l = fmaxf(v, 0.f) - v * t + log1pf(e);
        } else {
            g = 2.f * (v - t);
            l = (v - t) * (v - t);
        }
        g = gw * g / (float)n;
        if (dz) dz[i] = g;
        const _Float16 h = (_Float16)g;
        *reinterpret_cast<f16x4*>(carrier + i * 4) = (f16x4){h, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        gs = (float)h;
    }
    l = hv_block_sum(l, red);
    if (threadIdx.x == 0) part[blockIdx.x] = l;
    if (gpart) {
        __syncthreads();
        gs = hv_block_sum(gs, red);
        if (threadIdx.x == 0) gpart[blockIdx.x] = gs;
    }
}
__global__ __launch_bounds__(256) void gan_loss_head_final_kernel(const float* __restrict__ part, const float* __restrict__ gpart, int nparts, long long n, float lw,
                                                                  float* loss, int lacc, float* dbias, int bacc) {
    __shared__ float red[20];
    float s = 0.f, g = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) { s += part[i]; if (gpart) g += gpart[i]; }
    s = hv_block_sum(s, red);
    __syncthreads();
    g = hv_block_sum(g, red);
    if (threadIdx.x == 0) {
        if (loss) { const float v = lw * s / (float)n; loss[0] = lacc ? loss[0] + v : v; }
        if (dbias) dbias[0] = bacc ? dbias[0] + g : g;
    }
}
extern "C" size_t hv_gan_loss_head_workspace_bytes(long long n) { return 2 * (size_t)((n + 255) / 256) * sizeof(float); }
extern "C" int hv_gan_loss_head(const float* z, long long n, int target_is_real, int mode, float loss_weight, float* loss, int loss_accumulate, float grad_weight,
                                float* dz, void* carrier_f16, float* dbias, int dbias_accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    if (!z || !carrier_f16 || n <= 0 || mode < 0 || mode > 1) return HV_ERR_ARG;
    if (!workspace || workspace_bytes < hv_gan_loss_head_workspace_bytes(n) || ((uintptr_t)workspace & 3) || ((uintptr_t)carrier_f16 & 7)) return HV_ERR_WORKSPACE;
    const int nparts = (int)((n + 255) / 256);
    float* part = reinterpret_cast<float*>(workspace);
    float* gpart = dbias ? part + nparts : nullptr;
    hipLaunchKernelGGL(gan_loss_head_part_kernel, dim3(nparts), dim3(256), 0, (hipStream_t)stream, z, n, target_is_real ? 1.f : 0.f, mode, grad_weight, dz,
                       reinterpret_cast<_Float16*>(carrier_f16), part, gpart);
    HV_LAUNCH_CHECK();
    if (loss || dbias) {
        hipLaunchKernelGGL(gan_loss_head_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, part, gpart, nparts, n, loss_weight, loss, loss_accumulate, dbias,
                           dbias_accumulate);
        HV_LAUNCH_CHECK();
    }
    return HV_OK;
}

// Both ranges in ONE launch of ceil(n / 256) workgroups: a workgroup writes its block sums (loss, stored gradient) to `part`, takes a ticket, and the workgroup
// that draws the last one folds the partials of each range in their fixed order (deterministic) into the loss slots and the bias gradient, then puts the
// ticket back to zero for the next launch.  Hand-off per the MI355X guide: every wave's stores drained, workgroup barrier, ONE agent-scope release, relaxed
// agent-scope fetch_add; the last arriver: ONE agent-scope acquire, barrier, plain loads.  (A single 1 024-thread workgroup walking all 28 800 logits of a
// bs-32 pass took 29 us: 28 dependent load -> exp / log1p -> store rounds per lane.)
__global__ __launch_bounds__(256) void gan_loss_head_pair_kernel(const float* __restrict__ z0, int n0, float t0, float* loss0, _Float16* __restrict__ c0,
                                                                 const float* __restrict__ z1, int n1, float t1, float* loss1, _Float16* __restrict__ c1, int mode,
                                                                 float lw, int lacc, float gw, float* dbias, int bacc, float* part, unsigned* ticket) {
    __shared__ float red[20];
    __shared__ int last;
    const int nb0 = (n0 + 255) >> 8, nb = (int)gridDim.x;
    const int r = (int)blockIdx.x >= nb0 ? 1 : 0;                      // block-uniform: which range this workgroup belongs to
    const float* z = r ? z1 : z0;
    const int n = r ? n1 : n0;
    const float t = r ? t1 : t0;
    _Float16* carrier = r ? c1 : c0;
    const int i = ((int)blockIdx.x - (r ? nb0 : 0)) * 256 + (int)threadIdx.x;
    float l = 0.f, gs = 0.f;
    if (i < n) {
        const float v = z[i];
        float g;
        if (mode == 0) {
            const float e = expf(-fabsf(v)), q = 1.f / (1.f + e);
            g = (v >= 0.f ? q : e * q) - t;
            l = fmaxf(v, 0.f) - v * t + log1pf(e);
        } else {
            g = 2.f * (v - t);
            l = (v - t) * (v - t);
        }
        g = gw * g / (float)n;
        const _Float16 h = (_Float16)g;
        *reinterpret_cast<f16x4*>(carrier + (long long)i * 4) = (f16x4){h, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        gs = (float)h;
    }
    l = hv_block_sum(l, red);
    __syncthreads();
    gs = hv_block_sum(gs, red);
    if (threadIdx.x == 0) {      // (hv_common.h: publish / ticket / collect without agent-scope fences)
        hv_publish(part + 2 * blockIdx.x, l);
        hv_publish(part + 2 * blockIdx.x + 1, gs);
        hv_stores_done();
        last = hv_take_ticket_is_last(ticket, (unsigned)nb);
    }
    __syncthreads();
    if (!last) return;
    hv_acquire_once();
    // the last arriver: per range, the partials in block order (lane k takes blocks k, k + 256, ..; block sum in hv_block_sum's fixed order)
    float gtot = 0.f;
    for (int rr = 0; rr < 2; ++rr) {
        const int b0 = rr ? nb0 : 0, b1 = rr ? nb : nb0, nn = rr ? n1 : n0;
        if (b1 <= b0) continue;
        float sl = 0.f, sg = 0.f;
        for (int k = b0 + (int)threadIdx.x; k < b1; k += 256) { sl += part[2 * k]; sg += part[2 * k + 1]; }
        __syncthreads();
        sl = hv_block_sum(sl, red);
        __syncthreads();
        sg = hv_block_sum(sg, red);
        float* loss = rr ? loss1 : loss0;
        if (threadIdx.x == 0 && loss) { const float v = lw * sl / (float)nn; loss[0] = lacc ? loss[0] + v : v; }
        gtot = (rr && nb0 > 0) ? gtot + sg : sg;
    }
    if (threadIdx.x == 0) {
        if (dbias) dbias[0] = bacc ? dbias[0] + gtot : gtot;
        hv_ticket_reset(ticket);      // (the next launch of the stream starts behind this kernel)
    }
}
extern "C" size_t hv_gan_loss_head_pair_workspace_bytes(long long n0, long long n1) { return 2 * (size_t)((n0 + 255) / 256 + (n1 + 255) / 256) * sizeof(float) + 64; }
extern "C" int hv_gan_loss_head_pair(const float* z0, long long n0, int real0, float* loss0, void* carrier0_f16, const float* z1, long long n1, int real1, float* loss1,
                                     void* carrier1_f16, int mode, float loss_weight, int loss_accumulate, float grad_weight, float* dbias, int dbias_accumulate,
                                     void* workspace, size_t workspace_bytes, unsigned* ticket, void* stream) {
    if (!z0 || !carrier0_f16 || n0 <= 0 || n1 < 0 || (n1 > 0 && (!z1 || !carrier1_f16)) || mode < 0 || mode > 1 || !ticket) return HV_ERR_ARG;
    if (((uintptr_t)carrier0_f16 | (uintptr_t)carrier1_f16) & 7) return HV_ERR_ARG;
    if (n0 + n1 >= (1ll << 30)) return HV_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < hv_gan_loss_head_pair_workspace_bytes(n0, n1) || ((uintptr_t)workspace & 3)) return HV_ERR_WORKSPACE;
    const int nb = (int)((n0 + 255) / 256 + (n1 + 255) / 256);
    hipLaunchKernelGGL(gan_loss_head_pair_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, z0, (int)n0, real0 ? 1.f : 0.f, loss0,
                       reinterpret_cast<_Float16*>(carrier0_f16), z1, (int)n1, real1 ? 1.f : 0.f, loss1, reinterpret_cast<_Float16*>(carrier1_f16), mode, loss_weight,
                       loss_accumulate, grad_weight, dbias, dbias_accumulate, reinterpret_cast<float*>(workspace), ticket);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// ------------------------------------------------------------------------------------------------ generator losses
#define GL_CHUNKS 32
#define GL_NQ 10   // S1,S2,cnt,tpf,spf,sgf,tpc,spc,sgc,E
__global__ __launch_bounds__(256) void gloss_partial_kernel(const hv_gloss_desc d, double* __restrict__ part) {
    __shared__ double sh[4][GL_NQ];
    const int b = blockIdx.y, HW = d.H * d.W, tid = threadIdx.x;
    const int per = (HW + GL_CHUNKS - 1) / GL_CHUNKS;
    const int p0 = blockIdx.x * per, p1 = min(HW, p0 + per);
    float q[GL_NQ];
#pragma unroll
    for (int k = 0; k < GL_NQ; ++k) q[k] = 0.f;
#pragma unroll 4      // (four pixels' 10 loads in flight per thread: the rolled loop was a chain of eight memory round trips; same order of every thread's sums)
    for (int p = p0 + tid; p < p1; p += 256) {
        const long long i = (long long)b * HW + p;
        const float rb = d.real_B[i];
        q[0] += fabsf(d.fake_B[i] - rb);
        q[1] += fabsf(d.fake_B_coarse[i] - rb);
        q[2] += d.mask[i] != 0.f ? 1.f : 0.f;
        const float pf = d.fine_seg[i], gf = d.real_B_mask[i], pc = d.coarse_seg[i], gc = d.normal_vert[i];
        q[3] += gf * pf; q[4] += pf; q[5] += gf;
        q[6] += gc * pc; q[7] += pc; q[8] += gc;
        const float e = d.fake_edges[i] - d.real_edges[i];
        q[9] += e * e;
    }
#pragma unroll
    for (int k = 0; k < GL_NQ; ++k) {
        const float s = hv_wave_sum(q[k]);
        if ((tid & 63) == 0) sh[tid >> 6][k] = (double)s;
    }
    __syncthreads();
    if (tid < GL_NQ) part[((long long)b * GL_CHUNKS + blockIdx.x) * GL_NQ + tid] = sh[0][tid] + sh[1][tid] + sh[2][tid] + sh[3][tid];
}
// coef layout (floats): [0] coefL1; [1+4b..] per sample {fine: A=(sp+sg+eps), T=(2tp+eps); coarse: A, T}
// one workgroup: thread (b, k) sums quantity k of sample b over its GL_CHUNKS partials in a fixed order (was: one thread walking all
// B*GL_CHUNKS*GL_NQ values from global memory, 72 us at B = 16); thread 0 then combines the samples in order as before -- same result.
#define GL_MAXB 64
__global__ __launch_bounds__(256) void gloss_finalize_kernel(const hv_gloss_desc d, const double* __restrict__ part, float* __restrict__ coef) {
    __shared__ double qs[GL_MAXB][GL_NQ];
    const int B = d.B;
    for (int b0 = 0; b0 < B; b0 += GL_MAXB) {       // (B <= GL_MAXB in practice: one pass)
        __syncthreads();
        for (int e = threadIdx.x; e < min(B - b0, GL_MAXB) * GL_NQ; e += 256) {
            const int b = e / GL_NQ, k = e - b * GL_NQ;
            double v[GL_CHUNKS];      // every load of the thread in flight before the first add (the rolled loop was a chain of 32 round trips); same order of the sum
#pragma unroll
            for (int c = 0; c < GL_CHUNKS; ++c) v[c] = part[((long long)(b0 + b) * GL_CHUNKS + c) * GL_NQ + k];
            double q = 0;
#pragma unroll
            for (int c = 0; c < GL_CHUNKS; ++c) q += v[c];
            qs[b][k] = q;
        }
        __syncthreads();
        if (b0 + GL_MAXB < B && threadIdx.x == 0)     // spill the finished rows back (only for B > GL_MAXB)
            for (int b = 0; b < GL_MAXB; ++b)
                for (int k = 0; k < GL_NQ; ++k) const_cast<double*>(part)[((long long)(b0 + b) * GL_CHUNKS) * GL_NQ + k] = qs[b][k];
    }
    const double N = (double)B * d.H * d.W, eps = 1e-5;
    const int last0 = ((B - 1) / GL_MAXB) * GL_MAXB;
    // per-sample terms (the two double divisions, the coefficient stores, the height-loss terms) by one thread per sample; thread 0 then adds them in sample order as
    // before -- the same values in the same order (one thread doing all of it took most of this kernel's 18 us, between the discriminator passes and the backward)
    __shared__ double dterm[GL_MAXB][3];
    __shared__ double qsum[GL_MAXB][4];
    if (B <= GL_MAXB && threadIdx.x < B) {
        const int b = threadIdx.x;
        const double* q = qs[b];
        const double Af = q[4] + q[5] + eps, Tf = 2 * q[3] + eps, Ac = q[7] + q[8] + eps, Tc = 2 * q[6] + eps;
        dterm[b][0] = Tf / Af; dterm[b][1] = Tc / Ac;
        qsum[b][0] = q[0]; qsum[b][1] = q[1]; qsum[b][2] = q[2]; qsum[b][3] = q[9];
        coef[1 + 4 * b + 0] = (float)Af; coef[1 + 4 * b + 1] = (float)Tf; coef[1 + 4 * b + 2] = (float)Ac; coef[1 + 4 * b + 3] = (float)Tc;
        const float h = (float)d.height[b], mh = (float)d.maxheight[b];
        const float a1 = d.pred1_h[b] - h, a2 = d.pred2_h[b] - h;
        dterm[b][2] = (double)(fabsf(a1) / h * 40.f + fabsf(a2) / h * 40.f);
        const float s1 = a1 > 0.f ? 1.f : (a1 < 0.f ? -1.f : 0.f), s2 = a2 > 0.f ? 1.f : (a2 < 0.f ? -1.f : 0.f);
        const float gs = d.grad_scale > 0.f ? d.grad_scale : 1.f;
        if (d.d_pred1) d.d_pred1[b] = gs * (40.f * s1 / h * mh / (float)B);
        if (d.d_pred2) d.d_pred2[b] = gs * (40.f * s2 / h * mh / (float)B);
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    double S1 = 0, S2 = 0, cnt = 0, E = 0, dice_f = 0, dice_c = 0;
    if (B <= GL_MAXB) {
        for (int b = 0; b < B; ++b) { S1 += qsum[b][0]; S2 += qsum[b][1]; cnt += qsum[b][2]; E += qsum[b][3]; dice_f += dterm[b][0]; dice_c += dterm[b][1]; }
    } else
    for (int b = 0; b < B; ++b) {
        double q[GL_NQ];
        for (int k = 0; k < GL_NQ; ++k) q[k] = b >= last0 ? qs[b - last0][k] : part[((long long)b * GL_CHUNKS) * GL_NQ + k];
        S1 += q[0]; S2 += q[1]; cnt += q[2]; E += q[9];
        const double Af = q[4] + q[5] + eps, Tf = 2 * q[3] + eps, Ac = q[7] + q[8] + eps, Tc = 2 * q[6] + eps;
        dice_f += Tf / Af;
        dice_c += Tc / Ac;
        coef[1 + 4 * b + 0] = (float)Af; coef[1 + 4 * b + 1] = (float)Tf; coef[1 + 4 * b + 2] = (float)Ac; coef[1 + 4 * b + 3] = (float)Tc;
    }
    const double scale = 0.5 * d.lambda_L1 * ((double)d.W * d.W / cnt) * 2.0;
    const float l1 = (float)((S1 / N + S2 / N) * scale);
    coef[0] = (float)(scale / N);
    const float ldice = (float)((1.0 - dice_f / B) * 15.0), lcd = (float)((1.0 - dice_c / B) * 10.0);
    const float ledge = (float)(E / N * 800.0);
    double hsum = 0;
    if (B <= GL_MAXB) {
        for (int b = 0; b < B; ++b) hsum += dterm[b][2];
    } else
    for (int b = 0; b < B; ++b) {
        const float h = (float)d.height[b], mh = (float)d.maxheight[b];
        const float a1 = d.pred1_h[b] - h, a2 = d.pred2_h[b] - h;
        hsum += (double)(fabsf(a1) / h * 40.f + fabsf(a2) / h * 40.f);
        const float s1 = a1 > 0.f ? 1.f : (a1 < 0.f ? -1.f : 0.f), s2 = a2 > 0.f ? 1.f : (a2 < 0.f ? -1.f : 0.f);
        const float gs = d.grad_scale > 0.f ? d.grad_scale : 1.f;
        if (d.d_pred1) d.d_pred1[b] = gs * (40.f * s1 / h * mh / (float)B);
        if (d.d_pred2) d.d_pred2[b] = gs * (40.f * s2 / h * mh / (float)B);
    }
    const float lh = (float)(hsum / B);
    d.losses[0] = l1; d.losses[1] = ldice; d.losses[2] = lcd; d.losses[3] = ledge; d.losses[4] = lh;
    d.losses[5] = l1 + ldice + lcd + ledge + lh;
    if (d.gan_terms && d.loss_G_GAN) {
        float lg = d.gan_terms[0];
        for (int k = 1; k < d.n_gan_terms; ++k) lg += d.gan_terms[k];
        d.loss_G_GAN[0] = lg;
        if (d.loss_G) d.loss_G[0] = d.losses[5] + lg;
    }
}
__global__ void gloss_seed_kernel(const hv_gloss_desc d, const float* __restrict__ coef, long long n) {
    const int HW = d.H * d.W;
    const float gs = d.grad_scale > 0.f ? d.grad_scale : 1.f;      // power of two: the scaled seeds are exact multiples
    const float cl1 = coef[0] * gs, invB = gs / (float)d.B;
    PW_LOOP(i, n) {
        const int b = (int)pw_div(i, HW);
        const float rb = d.real_B[i];
        const float e1 = d.fake_B[i] - rb, e2 = d.fake_B_coarse[i] - rb;
        const float s1 = cl1 * (e1 > 0.f ? 1.f : (e1 < 0.f ? -1.f : 0.f));
        d.d_fake_B[i] = d.add_d_fake_B ? s1 + d.add_d_fake_B[i] : s1;
        d.d_fake_B_coarse[i] = cl1 * (e2 > 0.f ? 1.f : (e2 < 0.f ? -1.f : 0.f));
        const float Af = coef[1 + 4 * b], Tf = coef[2 + 4 * b], Ac = coef[3 + 4 * b], Tc = coef[4 + 4 * b];
        d.d_fine_seg[i] = -15.f * invB * (2.f * d.real_B_mask[i] * Af - Tf) / (Af * Af);
        d.d_coarse_seg[i] = -10.f * invB * (2.f * d.normal_vert[i] * Ac - Tc) / (Ac * Ac);
    }
}
extern "C" size_t hv_generator_losses_workspace_bytes(int B) {
    return (size_t)B * GL_CHUNKS * GL_NQ * sizeof(double) + (size_t)(1 + 4 * B) * sizeof(float) + 64;
}
extern "C" int hv_generator_losses(const hv_gloss_desc* d, void* stream) {
    if (!d || !d->fake_B || !d->fake_B_coarse || !d->real_B || !d->mask || !d->fine_seg || !d->coarse_seg || !d->real_B_mask ||
        !d->normal_vert || !d->fake_edges || !d->real_edges || !d->pred1_h || !d->pred2_h || !d->height || !d->maxheight || !d->losses ||
        d->B <= 0 || d->H <= 0 || d->W <= 0)
        return HV_ERR_ARG;
    if (!d->workspace || d->workspace_bytes < hv_generator_losses_workspace_bytes(d->B) || ((uintptr_t)d->workspace & 7)) return HV_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    double* part = (double*)d->workspace;
    float* coef = (float*)((char*)d->workspace + (size_t)d->B * GL_CHUNKS * GL_NQ * sizeof(double));
    hipLaunchKernelGGL(gloss_partial_kernel, dim3(GL_CHUNKS, d->B), dim3(256), 0, s, *d, part);
    HV_LAUNCH_CHECK();
    hipLaunchKernelGGL(gloss_finalize_kernel, dim3(1), dim3(256), 0, s, *d, part, coef);
    HV_LAUNCH_CHECK();
    if (d->d_fake_B && d->d_fake_B_coarse && d->d_fine_seg && d->d_coarse_seg) {
        const long long n = (long long)d->B * d->H * d->W;
        hipLaunchKernelGGL(gloss_seed_kernel, dim3(pw_grid(n)), dim3(256), 0, s, *d, coef, n);
        HV_LAUNCH_CHECK();
    }
    return HV_OK;
}

// ------------------------------------------------------------------------------------------------ threshold (seg > 0.5 -> label id)
__global__ void threshold_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, float thr, float value) {
    PW_LOOP(i, n) y[i] = x[i] > thr ? value : 0.f;
}
extern "C" int hv_threshold(const float* x, float* y, long long n, float thr, float value, void* stream) {
    if (!x || !y || n <= 0) return HV_ERR_ARG;
    hipLaunchKernelGGL(threshold_kernel, dim3(pw_grid(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, thr, value);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

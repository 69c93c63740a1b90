// In-training evaluation metrics on the device: the per-sample arithmetic of the reference's
// evaluate_model (train.py:101-141) -- masked SSIM and PSNR of the composited image against the ground truth, Dice of the coarse
// segmentation against the normal-vertebra mask (dice_score, train.py:37-41), IoU of the fine segmentation against the label
// (iou_score, :43-48) and the relative height error -- for a whole batch in three small launches, no host round trip per sample.
//
// SSIM / PSNR are scikit-image's (requirements.txt pins 0.22.0; absent from the build container, so this follows the PUBLISHED algorithm:
// structural_similarity defaults = 7x7 uniform window, sample covariance (49/48), K1 0.01, K2 0.03, border of 3 cropped, float32 images,
// float64 mean; peak_signal_noise_ratio = 10 log10(R^2 / mean((a-b)^2, float64))).  Operation order and precisions follow skimage so the
// CPU restatement (oracle/restate.py: eval_ssim / eval_psnr, built on scipy.ndimage.uniform_filter) agrees to float32 rounding.
#include "hv_common.h"

namespace {

constexpr int STAT_N = 12;   // per sample: max_inp, min_inp, min_gt, sse, n_cb*nv, n_cb, n_nv, n_fb*lab, n_fb, n_lab, (2 spare)
constexpr int SSIM_TILE_ROWS = 8;

__device__ __forceinline__ double block_reduce(double v, double* red, int op /*0 sum, 1 max, 2 min*/) {
    for (int o = 32; o > 0; o >>= 1) {
        const double w = __shfl_xor(v, o, 64);
        v = op == 0 ? v + w : (op == 1 ? fmax(v, w) : fmin(v, w));
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = op == 0 ? r + red[i] : (op == 1 ? fmax(r, red[i]) : fmin(r, red[i]));
    return r;
}

// one workgroup per sample: extrema of the (unmasked) images, squared error of the masked ones, overlap counts of the binary maps
__global__ void __launch_bounds__(1024) eval_stats_kernel(const float* __restrict__ inp, const float* __restrict__ gt, const float* __restrict__ mask,
                                                         const float* __restrict__ cb, const float* __restrict__ nv, const float* __restrict__ fb,
                                                         const float* __restrict__ lab, int HW, double* __restrict__ stats) {
    __shared__ double red[16];
    const long long base = (long long)blockIdx.x * HW;
    double mx = -INFINITY, mn = INFINITY, mg = INFINITY, sse = 0, a = 0, b = 0, c = 0, d = 0, e = 0, f = 0;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) {
        const float x = inp[base + i], g = gt[base + i], m = mask[base + i];
        mx = fmax(mx, (double)x); mn = fmin(mn, (double)x); mg = fmin(mg, (double)g);
        const float df = g * m - x * m;        // float32 images, float32 difference and square (skimage), float64 accumulation
        sse += (double)(df * df);
        const float vcb = cb[base + i], vnv = nv[base + i], vfb = fb[base + i], vl = lab[base + i];
        a += (double)(vcb * vnv); b += vcb; c += vnv;
        d += (double)(vfb * vl); e += vfb; f += vl;
    }
    const double r[10] = {block_reduce(mx, red, 1), block_reduce(mn, red, 2), block_reduce(mg, red, 2), block_reduce(sse, red, 0),
                          block_reduce(a, red, 0), block_reduce(b, red, 0), block_reduce(c, red, 0), block_reduce(d, red, 0),
                          block_reduce(e, red, 0), block_reduce(f, red, 0)};
    if (threadIdx.x < 10) stats[(long long)blockIdx.x * STAT_N + threadIdx.x] = r[threadIdx.x];
}

// grid (row tiles, B): partial sums of the SSIM map over the cropped region, one output pixel per thread and window position
__global__ void __launch_bounds__(256) eval_ssim_kernel(const float* __restrict__ inp, const float* __restrict__ gt, const float* __restrict__ mask, int H,
                                                       int W, const double* __restrict__ stats, double* __restrict__ partial) {
    __shared__ double red[4];
    const int b = blockIdx.y, r0 = 3 + blockIdx.x * SSIM_TILE_ROWS;
    const long long base = (long long)b * H * W;
    const float R = (float)stats[(long long)b * STAT_N + 0] - (float)stats[(long long)b * STAT_N + 1];   // data_range = inp.max() - inp.min()
    const float C1 = (0.01f * R) * (0.01f * R), C2 = (0.03f * R) * (0.03f * R);
    const float cov_norm = 49.0f / 48.0f;
    const int wv = W - 6;
    double acc = 0;
    for (int t = threadIdx.x; t < SSIM_TILE_ROWS * wv; t += blockDim.x) {
        const int i = r0 + t / wv, j = 3 + t % wv;
        if (i >= H - 3) break;
        double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
        for (int di = -3; di <= 3; ++di) {
            const long long row = base + (long long)(i + di) * W + j;
#pragma unroll
            for (int dj = -3; dj <= 3; ++dj) {
                const float m = mask[row + dj];
                const float x = gt[row + dj] * m, y = inp[row + dj] * m;     // im1 = ground truth * mask, im2 = result * mask
                sx += x; sy += y; sxx += (double)(x * x); syy += (double)(y * y); sxy += (double)(x * y);
            }
        }
        const float ux = (float)(sx / 49.0), uy = (float)(sy / 49.0), uxx = (float)(sxx / 49.0), uyy = (float)(syy / 49.0), uxy = (float)(sxy / 49.0);
        const float vx = cov_norm * (uxx - ux * ux), vy = cov_norm * (uyy - uy * uy), vxy = cov_norm * (uxy - ux * uy);
        const float A1 = 2.f * ux * uy + C1, A2 = 2.f * vxy + C2, B1 = ux * ux + uy * uy + C1, B2 = vx + vy + C2;
        const float D = B1 * B2;
        acc += (double)((A1 * A2) / D);
    }
    const double s = block_reduce(acc, red, 0);
    if (threadIdx.x == 0) partial[(long long)b * gridDim.x + blockIdx.x] = s;
}

__global__ void eval_final_kernel(const double* __restrict__ stats, const double* __restrict__ partial, int tiles, int H, int W,
                                  const float* __restrict__ pred_h, const long long* __restrict__ height, int B, float* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double* s = stats + (long long)b * STAT_N;
    double ss = 0;
    for (int t = 0; t < tiles; ++t) ss += partial[(long long)b * tiles + t];
    const double ssim = ss / ((double)(H - 6) * (double)(W - 6));
    const float Rp = (float)s[0] - (float)s[2];                                // data_range = inp.max() - ground_truth.min()
    const double err = s[3] / ((double)H * (double)W);
    const double psnr = 10.0 * log10((double)(Rp * Rp) / err);
    const float smooth = 1e-5f;
    const float dice = (2.f * (float)s[4] + smooth) / ((float)s[5] + (float)s[6] + smooth);
    const float uni = (float)s[8] + (float)s[9] - (float)s[7];
    const float iou = ((float)s[7] + smooth) / (uni + smooth);
    const float hh = (float)height[b];
    const float dh = fabsf(pred_h[b] - hh) / hh * 100.f;
    float* o = out + (long long)b * 5;
    o[0] = (float)ssim; o[1] = (float)psnr; o[2] = dice; o[3] = iou; o[4] = dh;
}

}  // namespace

extern "C" size_t hv_eval_metrics_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 6 || W <= 6) return 0;
    const int tiles = hv_cdiv(H - 6, SSIM_TILE_ROWS);
    return ((size_t)B * STAT_N + (size_t)B * tiles) * sizeof(double);
}

extern "C" int hv_eval_metrics(const float* inpainted, const float* gt, const float* mask, const float* coarse_bin, const float* normal_vert,
                               const float* fine_bin, const float* label, const float* pred_h, const long long* height, int B, int H, int W, float* out,
                               void* workspace, size_t workspace_bytes, void* stream) {
    if (!inpainted || !gt || !mask || !coarse_bin || !normal_vert || !fine_bin || !label || !pred_h || !height || !out || B <= 0) return HV_ERR_ARG;
    if (H <= 6 || W <= 6) return HV_ERR_UNSUPPORTED;      // the 7x7 SSIM window needs a larger image (skimage raises, too)
    if (!workspace || workspace_bytes < hv_eval_metrics_workspace_bytes(B, H, W) || ((uintptr_t)workspace & 7)) return HV_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    double* stats = (double*)workspace;
    double* partial = stats + (size_t)B * STAT_N;
    const int tiles = hv_cdiv(H - 6, SSIM_TILE_ROWS);
    hipLaunchKernelGGL(eval_stats_kernel, dim3(B), dim3(1024), 0, s, inpainted, gt, mask, coarse_bin, normal_vert, fine_bin, label, H * W, stats);
    HV_LAUNCH_CHECK();
    hipLaunchKernelGGL(eval_ssim_kernel, dim3(tiles, B), dim3(256), 0, s, inpainted, gt, mask, H, W, stats, partial);
    HV_LAUNCH_CHECK();
    hipLaunchKernelGGL(eval_final_kernel, dim3(hv_cdiv(B, 64)), dim3(64), 0, s, stats, partial, tiles, H, W, pred_h, height, B, out);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

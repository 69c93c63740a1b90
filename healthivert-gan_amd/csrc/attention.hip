// Memory-bound stages of the contextual-attention block (reference models/inpaint_networks.py:247-410).
// The two large contractions (patch matching, patch pasting) and their gradients run through
// hv_conv2d with per-sample filters; the kernels here build those filters from the feature map
// (patch matrices), fuse / mask / soft-max the L x L score matrix, and scatter gradients back.
// Score matrices are stored [b][p][l]: p = foreground position (row), l = background patch (contiguous).
#include "hv_common.h"

static int at_grid(long long n, int cap = 16384) { long long b = (n + 255) / 256; return (int)(b > cap ? cap : (b < 1 ? 1 : b)); }
#define AT_LOOP(i, n) for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)

// ---- downsample + 3x3 patch matrix (+ transposed copy) ------------------------------------------------
__global__ void ca_down_kernel(const void* __restrict__ f, int fh, float* __restrict__ fd, int H, int W, int C, int f_ld, long long n) {
    const int h = H / 2, w = W / 2;
    AT_LOOP(i, n) {
        const int c = (int)(i % C);
        long long r = i / C;
        const int x = (int)(r % w); r /= w;
        const int y = (int)(r % h);
        const long long b = r / h;
        fd[i] = hv_ld1(f, ((b * H + 2 * y) * W + 2 * x) * f_ld + c, fh);          // fh: the feature map is stored as fp16
    }
}
__global__ void ca_wp_kernel(const float* __restrict__ fd, float* __restrict__ wp, float* __restrict__ wpT, int h, int w, int C, long long n) {
    const int L = h * w;
    AT_LOOP(i, n) {  // i over wp order (b, l, tap, c)
        const int c = (int)(i % C);
        long long r = i / C;
        const int tap = (int)(r % 9); r /= 9;
        const int l = (int)(r % L);
        const long long b = r / L;
        const int y = l / w + tap / 3 - 1, x = l % w + tap % 3 - 1;
        float v = 0.f;
        if ((unsigned)y < (unsigned)h && (unsigned)x < (unsigned)w) v = fd[((b * h + y) * w + x) * C + c];
        wp[i] = v;
        if (wpT) wpT[(b * 9 * C + tap * C + c) * L + l] = v;
    }
}
__global__ __launch_bounds__(64) void ca_norm_kernel(const float* __restrict__ wp, float* __restrict__ norm, float* __restrict__ rnorm, int K) {
    const long long row = blockIdx.x;
    float s = 0.f;
    for (int k = threadIdx.x; k < K; k += 64) { const float v = wp[row * K + k]; s += v * v; }
    s = hv_wave_sum(s);
    if (threadIdx.x == 0) {
        const float nv = fmaxf(sqrtf(s), 1e-4f);
        norm[row] = nv;
        rnorm[row] = 1.f / nv;
    }
}
// 4x4 stride-2 'same' patches (pad 1 top/left): raw[b][l][tap][c] and/or rawT[b][c][tap][l]
__global__ void ca_raw_kernel(const float* __restrict__ f, float* __restrict__ raw, float* __restrict__ rawT, int H, int W, int C, int f_ld, long long n) {
    const int h = H / 2, w = W / 2, L = h * w;
    AT_LOOP(i, n) {  // i over rawT order (b, c, tap, l): coalesced writes of the larger-stride copy
        const int l = (int)(i % L);
        long long r = i / L;
        const int tap = (int)(r % 16); r /= 16;
        const int c = (int)(r % C);
        const long long b = r / C;
        const int y = 2 * (l / w) - 1 + tap / 4, x = 2 * (l % w) - 1 + tap % 4;
        float v = 0.f;
        if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) v = f[((b * H + y) * W + x) * f_ld + c];
        if (rawT) rawT[i] = v;
        if (raw) raw[((b * L + l) * 16 + tap) * C + c] = v;
    }
}

extern "C" int hv_transpose_batched(const float* src, float* dst, int B, int R, int C, void* stream);

// Both patch layouts from one LDS-staged tile (the element-wise kernels above pay 64-bit divisions per element and write one of the two
// layouts 4 bytes at a 4-KB stride: 81 / 93 us for 38 / 134 MB).  One workgroup = 64 patch positions x 64 channels of one tap:
// rows are read (and the [l][tap][c] copy written) 16 bytes per lane along the channels, the [c][tap][l] copy is written along l from
// the transposed tile.  KS x KS taps, stride ST, pad 1 (3x3 / 1 on the down-sampled map, 4x4 / 2 on the full map).
template <int KS, int ST, typename OT = float, typename ST_ = float>       // OT = _Float16: both copies written as fp16 (operands of the batched GEMMs only); ST_: storage of the source map
__global__ __launch_bounds__(256) void ca_patch_tile_kernel(const ST_* __restrict__ src, OT* __restrict__ lt, OT* __restrict__ tl,
                                                            int Hs, int Ws, int w, int L, int C, int s_ld, _Float16* __restrict__ lt_h) {
    __shared__ float tile[64][65];
    constexpr int T = KS * KS;
    const int t = threadIdx.x, l0 = blockIdx.x * 64, tap = blockIdx.y % T, c0 = (blockIdx.y / T) * 64;
    const long long b = blockIdx.z;
    const int dy = tap / KS - 1, dx = tap % KS - 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ll = (t >> 4) + 16 * k, l = l0 + ll, c = c0 + (t & 15) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (l < L && c < C) {
            const int y = ST * (l / w) + dy, x = ST * (l % w) + dx;
            if ((unsigned)y < (unsigned)Hs && (unsigned)x < (unsigned)Ws) {
                if constexpr (sizeof(ST_) == 2) {
                    const f16x4 h4 = *reinterpret_cast<const f16x4*>(src + ((b * Hs + y) * Ws + x) * s_ld + c);
                    v = make_float4((float)h4[0], (float)h4[1], (float)h4[2], (float)h4[3]);
                } else {
                    v = *reinterpret_cast<const float4*>(src + ((b * Hs + y) * Ws + x) * s_ld + c);
                }
            }
            if (lt) {
                if constexpr (sizeof(OT) == 2) *reinterpret_cast<f16x4*>(lt + ((b * L + l) * T + tap) * C + c) = (f16x4){(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
                else *reinterpret_cast<float4*>(lt + ((b * L + l) * T + tap) * C + c) = v;
            }
            // the same rows once more as fp16: the operand copy of the batched GEMMs (same rounding as their staging of an fp32 operand)
            if (lt_h) *reinterpret_cast<f16x4*>(lt_h + ((b * L + l) * T + tap) * C + c) = (f16x4){(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        }
        tile[ll][(t & 15) * 4 + 0] = v.x; tile[ll][(t & 15) * 4 + 1] = v.y; tile[ll][(t & 15) * 4 + 2] = v.z; tile[ll][(t & 15) * 4 + 3] = v.w;
    }
    __syncthreads();
    if (!tl) return;
    const int ll = t & 63;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int cc = (t >> 6) + 4 * k;
        if (l0 + ll < L && c0 + cc < C) tl[((b * C + c0 + cc) * T + tap) * L + l0 + ll] = (OT)tile[ll][cc];
    }
}
static bool ca_tile_ok(const float* src, int C, int s_ld, const float* lt) {
    static const int enabled = getenv("HV_CA_TILE") ? atoi(getenv("HV_CA_TILE")) : 1;   // A/B knob
    return enabled && (C & 3) == 0 && (s_ld & 3) == 0 && !((uintptr_t)src & 15) && !((uintptr_t)lt & 15);
}

static int ca_patches_impl(const void* f, int f_f16, int B, int H, int W, int C, int f_ld, float* fd, float* wp, float* wpT, float* norm,
                           float* rnorm, _Float16* wp_h, void* stream) {
    if (!f || !fd || !wp || !norm || !rnorm || B <= 0 || H <= 0 || W <= 0 || C <= 0 || ((H | W) & 1) || f_ld < C) return HV_ERR_ARG;
    if (wp_h && (!(ca_tile_ok(fd, C, C, wp) && B <= 65535) || ((uintptr_t)wp_h & 7))) return HV_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int h = H / 2, w = W / 2;
    long long n = (long long)B * h * w * C;
    hipLaunchKernelGGL(ca_down_kernel, dim3(at_grid(n)), dim3(256), 0, s, f, f_f16, fd, H, W, C, f_ld, n);
    HV_LAUNCH_CHECK();
    n *= 9;
    if (ca_tile_ok(fd, C, C, wp) && B <= 65535)    // wpT layout: [b][(tap, c)][l] -- as [c][tap][l] it would need c outermost: only wp here, wpT below
        hipLaunchKernelGGL((ca_patch_tile_kernel<3, 1>), dim3(hv_cdiv(h * w, 64), 9 * hv_cdiv(C, 64), B), dim3(256), 0, s, fd, wp, (float*)nullptr, h, w, w, h * w, C, C, wp_h);
    else
        hipLaunchKernelGGL(ca_wp_kernel, dim3(at_grid(n)), dim3(256), 0, s, fd, wp, (float*)nullptr, h, w, C, n);
    HV_LAUNCH_CHECK();
    if (wpT) {   // [b][tap*C + c][l] = transpose of wp[b][l][tap*C + c]
        const int rc = hv_transpose_batched(wp, wpT, B, h * w, 9 * C, stream);
        if (rc != HV_OK) return rc;
    }
    hipLaunchKernelGGL(ca_norm_kernel, dim3(B * h * w), dim3(64), 0, s, wp, norm, rnorm, 9 * C);
    HV_LAUNCH_CHECK();
    return HV_OK;
}
extern "C" int hv_ca_patches(const void* f, int f_f16, int B, int H, int W, int C, int f_ld, float* fd, float* wp, float* wpT, float* norm,
                             float* rnorm, void* stream) {
    return ca_patches_impl(f, f_f16, B, H, W, C, f_ld, fd, wp, wpT, norm, rnorm, nullptr, stream);
}
extern "C" int hv_ca_patches_h(const void* f, int f_f16, int B, int H, int W, int C, int f_ld, float* fd, float* wp, void* wp_h, float* norm,
                               float* rnorm, void* stream) {
    if (!wp_h) return HV_ERR_ARG;
    return ca_patches_impl(f, f_f16, B, H, W, C, f_ld, fd, wp, nullptr, norm, rnorm, (_Float16*)wp_h, stream);
}
extern "C" int hv_ca_raw_patches(const float* f, int B, int H, int W, int C, int f_ld, float* raw, float* rawT, void* stream) {
    if (!f || (!raw && !rawT) || B <= 0 || H <= 0 || W <= 0 || C <= 0 || ((H | W) & 1) || f_ld < C) return HV_ERR_ARG;
    const long long n = (long long)B * C * 16 * (H / 2) * (W / 2);
    if (ca_tile_ok(f, C, f_ld, raw) && B <= 65535)
        hipLaunchKernelGGL((ca_patch_tile_kernel<4, 2>), dim3(hv_cdiv((H / 2) * (W / 2), 64), 16 * hv_cdiv(C, 64), B), dim3(256), 0, (hipStream_t)stream,
                           f, raw, rawT, H, W, W / 2, (H / 2) * (W / 2), C, f_ld, (_Float16*)nullptr);
    else
        hipLaunchKernelGGL(ca_raw_kernel, dim3(at_grid(n)), dim3(256), 0, (hipStream_t)stream, f, raw, rawT, H, W, C, f_ld, n);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// the same two patch tables stored as fp16 (operands of hv_bgemm_nt only; the fp32 tables feed the per-sample-filter convolutions of the fp32 mode)
extern "C" int hv_ca_raw_patches_f16(const void* f, int f_f16, int B, int H, int W, int C, int f_ld, void* raw_h, void* rawT_h, void* stream) {
    if (!f || (!raw_h && !rawT_h) || B <= 0 || H <= 0 || W <= 0 || C <= 0 || ((H | W) & 1) || f_ld < C) return HV_ERR_ARG;
    if (!ca_tile_ok((const float*)f, C, f_ld, (const float*)raw_h) || B > 65535 || ((uintptr_t)raw_h & 7)) return HV_ERR_UNSUPPORTED;
    const dim3 grid(hv_cdiv((H / 2) * (W / 2), 64), 16 * hv_cdiv(C, 64), B);
    if (f_f16)
        hipLaunchKernelGGL((ca_patch_tile_kernel<4, 2, _Float16, _Float16>), grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)f, (_Float16*)raw_h,
                           (_Float16*)rawT_h, H, W, W / 2, (H / 2) * (W / 2), C, f_ld, (_Float16*)nullptr);
    else
    hipLaunchKernelGGL((ca_patch_tile_kernel<4, 2, _Float16>), grid, dim3(256), 0, (hipStream_t)stream,
                       (const float*)f, (_Float16*)raw_h, (_Float16*)rawT_h, H, W, W / 2, (H / 2) * (W / 2), C, f_ld, (_Float16*)nullptr);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// ---- mask of valid background patches (sample 0 only) ------------------------------------------------
__global__ void ca_mask_kernel(const float* __restrict__ mask, int Himg, int Wimg, int h, int w, float* __restrict__ mm, long long mask_bs) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= h * w) return;
    mask += blockIdx.y * mask_bs;      // blockIdx.y = sample (hv_ca_mask_batched); the one-sample form launches a single row
    mm += (long long)blockIdx.y * h * w;
    const int sy = Himg / h, sx = Wimg / w;
    float s = 0.f;
    for (int t = 0; t < 9; ++t) {
        const int y = l / w + t / 3 - 1, x = l % w + t % 3 - 1;
        if ((unsigned)y < (unsigned)h && (unsigned)x < (unsigned)w) s += mask[(long long)(y * sy) * Wimg + x * sx];
    }
    mm[l] = (s / 9.f == 0.f) ? 1.f : 0.f;
}
extern "C" int hv_ca_mask(const float* mask, int Himg, int Wimg, int h, int w, float* mm, void* stream) {
    if (!mask || !mm || h <= 0 || w <= 0 || Himg % h || Wimg % w) return HV_ERR_ARG;
    hipLaunchKernelGGL(ca_mask_kernel, dim3(hv_cdiv(h * w, 256)), dim3(256), 0, (hipStream_t)stream, mask, Himg, Wimg, h, w, mm, 0ll);
    HV_LAUNCH_CHECK();
    return HV_OK;
}
// every sample's own mask: mm[B][h*w] (a batch that stands for B independent single-sample calls, e.g. the z-slices of one inference stage)
extern "C" int hv_ca_mask_batched(const float* mask, int B, long long mask_bstride, int Himg, int Wimg, int h, int w, float* mm, void* stream) {
    if (!mask || !mm || B <= 0 || B > 65535 || h <= 0 || w <= 0 || Himg % h || Wimg % w) return HV_ERR_ARG;
    hipLaunchKernelGGL(ca_mask_kernel, dim3(hv_cdiv(h * w, 256), B), dim3(256), 0, (hipStream_t)stream, mask, Himg, Wimg, h, w, mm, mask_bstride);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// ---- score fusion ----------------------------------------------------------------------------------------
__device__ __forceinline__ int tr_idx(int l, int h, int w) { return (l % w) * h + l / w; }  // (lh,lw) -> lw*h+lh
__global__ void ca_fuse_kernel(const float* __restrict__ S, float* __restrict__ out, int h, int w, int adjoint, long long n) {
    const int L = h * w;
    AT_LOOP(i, n) {
        const int l = (int)(i % L);
        long long r = i / L;
        const int p = (int)(r % L);
        const float* Sb = S + (r / L) * (long long)L * L;
        float acc = 0.f;
        if (!adjoint) {
            const int tp = tr_idx(p, h, w), tl = tr_idx(l, h, w);
#pragma unroll
            for (int d = -1; d <= 1; ++d) {
                const int a = tp + d, b = tl + d;
                if ((unsigned)a >= (unsigned)L || (unsigned)b >= (unsigned)L) continue;
                const int pa = tr_idx(a, w, h), lb = tr_idx(b, w, h);  // inverse transpose
#pragma unroll
                for (int e = -1; e <= 1; ++e) {
                    const int pp = pa + e, ll = lb + e;
                    if ((unsigned)pp < (unsigned)L && (unsigned)ll < (unsigned)L) acc += Sb[(long long)pp * L + ll];
                }
            }
        } else {
#pragma unroll
            for (int e = -1; e <= 1; ++e) {
                const int pe = p + e, le = l + e;
                if ((unsigned)pe >= (unsigned)L || (unsigned)le >= (unsigned)L) continue;
                const int tp = tr_idx(pe, h, w), tl = tr_idx(le, h, w);
#pragma unroll
                for (int d = -1; d <= 1; ++d) {
                    const int a = tp + d, b = tl + d;
                    if ((unsigned)a < (unsigned)L && (unsigned)b < (unsigned)L) acc += Sb[(long long)tr_idx(a, w, h) * L + tr_idx(b, w, h)];
                }
            }
        }
        out[i] = acc;
    }
}
// the same for h, w powers of two (the 32 x 32 attention map): all index arithmetic is shifts and masks, 32-bit inside a sample
template <bool ADJ>
__global__ __launch_bounds__(256) void ca_fuse_p2_kernel(const float* __restrict__ S, float* __restrict__ out, int lh, int lw) {
    const int h = 1 << lh, w = 1 << lw, L = h * w, lL = lh + lw;
    const float* Sb = S + (long long)blockIdx.y * L * L;
    float* ob = out + (long long)blockIdx.y * L * L;
    auto tr = [&](int l) { return ((l & (w - 1)) << lh) + (l >> lw); };      // (lh,lw) -> lw*h+lh
    auto itr = [&](int a) { return ((a & (h - 1)) << lw) + (a >> lh); };     // inverse
    const int n = L * L;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int l = i & (L - 1), p = i >> lL;
        float acc = 0.f;
        if (!ADJ) {
            const int tp = tr(p), tl = tr(l);
#pragma unroll
            for (int d = -1; d <= 1; ++d) {
                const int a = tp + d, b = tl + d;
                if ((unsigned)a >= (unsigned)L || (unsigned)b >= (unsigned)L) continue;
                const int pa = itr(a), lb = itr(b);
#pragma unroll
                for (int e = -1; e <= 1; ++e) {
                    const int pp = pa + e, ll = lb + e;
                    if ((unsigned)pp < (unsigned)L && (unsigned)ll < (unsigned)L) acc += Sb[(pp << lL) + ll];
                }
            }
        } else {
#pragma unroll
            for (int e = -1; e <= 1; ++e) {
                const int pe = p + e, le = l + e;
                if ((unsigned)pe >= (unsigned)L || (unsigned)le >= (unsigned)L) continue;
                const int tp = tr(pe), tl = tr(le);
#pragma unroll
                for (int d = -1; d <= 1; ++d) {
                    const int a = tp + d, b = tl + d;
                    if ((unsigned)a < (unsigned)L && (unsigned)b < (unsigned)L) acc += Sb[(itr(a) << lL) + itr(b)];
                }
            }
        }
        ob[i] = acc;
    }
}
// Forward fuse for the 32 x 32 attention map, tiled: an output tile = one grid row of p (32 consecutive rows) x one grid row of l (32 columns).
// For each column-order shift d the source rows itr(tr(p) + d) of the tile are CONSECUTIVE again (p + d*w inside the map; 1 + px when the last
// grid row wraps to the top of the next column; (h-1)*w - 1 + px for the first one), and so are the columns: three 34 x 34 pieces of S (one halo
// row / column for the row-order shift e) are staged in LDS once and every output is nine LDS reads -- the kernel above reads each source row of S
// nine times through L2 (250 MB of fabric traffic for 128 MB of operands, 110 us).  Terms are added in the same (d, e) order: the same bits.
// Tile order (round 3): the three pieces of a tile are the S blocks (py0 + d - 1, ly0 + d - 1), so each block of S is read by the three tiles of one
// (circular) tile DIAGONAL ly0 - py0 = const.  In (ly0, py0, sample) launch order those three run on different XCDs at different times and every block
// came over the fabric three times (PMC: 404 MB per launch for 128 MB of operands -- the kernel ran at the fabric's 7 TB/s).  ca_tile_of() deals whole
// diagonals to the XCD that the hardware's round-robin gives a workgroup (linear id & 7), consecutive workgroups of an XCD walking along one diagonal:
// the re-reads hit that XCD's L2.
__device__ __forceinline__ void ca_tile_of(int id, int xcd_order, int& py0, int& ly0, int& b) {
    if (xcd_order) {
        const int xcd = id & 7;
        int q = id >> 3;
        const int k = q & 31;
        q >>= 5;
        const int delta = (q & 3) * 8 + xcd;
        b = q >> 2;
        py0 = k;
        ly0 = (k + delta) & 31;
    } else {
        ly0 = id & 31; py0 = (id >> 5) & 31; b = id >> 10;
    }
}
__global__ __launch_bounds__(256) void ca_fuse_tile32_kernel(const float* __restrict__ S, float* __restrict__ out, int xcd_order) {
    constexpr int W = 32, HH = 32, L = W * HH, TS = 34, LDT = 35;
    __shared__ float T[3][TS * LDT];
    int py0, ly0, bb;
    ca_tile_of((int)blockIdx.x, xcd_order, py0, ly0, bb);
    const float* Sb = S + (long long)bb * L * L;
    float* ob = out + (long long)bb * L * L;
    const int p0 = py0 * W, l0 = ly0 * W;
    int pb[3], lb[3];
    pb[0] = py0 >= 1 ? p0 - W : (HH - 1) * W - 1;  pb[1] = p0;  pb[2] = py0 < HH - 1 ? p0 + W : 1;
    lb[0] = ly0 >= 1 ? l0 - W : (HH - 1) * W - 1;  lb[1] = l0;  lb[2] = ly0 < HH - 1 ? l0 + W : 1;
    {   // all of a lane's loads are issued before the first LDS store (the pieces are 3 x 34 rows of 136 bytes: latency, not bandwidth)
        constexpr int NIT = (TS * TS + 255) / 256;
        float v[3][NIT];
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int e = threadIdx.x + it * 256, i = e / TS, j = e - i * TS;
                const int pr = pb[d] - 1 + i, lc = lb[d] - 1 + j;
                v[d][it] = (e < TS * TS && (unsigned)pr < (unsigned)L && (unsigned)lc < (unsigned)L) ? Sb[(long long)pr * L + lc] : 0.f;
            }
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int e = threadIdx.x + it * 256, i = e / TS, j = e - i * TS;
                if (e < TS * TS) T[d][i * LDT + j] = v[d][it];
            }
    }
    __syncthreads();
    const int r = threadIdx.x >> 3, c0 = (threadIdx.x & 7) * 4;
    float o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = c0 + u;
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            // tr(p) + d and tr(l) + d inside [0, L): only the first / last grid row can fall out, at px = 0 / w - 1 (resp. lx)
            const bool vp = d == 1 || (d == 0 ? (py0 >= 1 || r >= 1) : (py0 < HH - 1 || r + 1 < W));
            const bool vl = d == 1 || (d == 0 ? (ly0 >= 1 || c >= 1) : (ly0 < HH - 1 || c + 1 < W));
            if (!(vp && vl)) continue;
            acc += T[d][r * LDT + c];
            acc += T[d][(r + 1) * LDT + c + 1];
            acc += T[d][(r + 2) * LDT + c + 2];
        }
        o[u] = acc;
    }
    *reinterpret_cast<float4*>(ob + (long long)(p0 + r) * L + l0 + c0) = make_float4(o[0], o[1], o[2], o[3]);
}
// The adjoint on the same tiles: out[p][l] = sum_e U[p+e][l+e] with U[q][m] = sum_d S[itr(tr(q)+d)][itr(tr(m)+d)].  The 34 rows q = p0-1 .. p0+32
// (the two halo rows belong to the neighbouring grid rows, whose wrap cases differ) get their three source rows from a small index table built by
// the first lanes; columns alike.  Terms are added in the (e, d) order of ca_fuse_p2_kernel<true>: the same bits.
__global__ __launch_bounds__(256) void ca_fuse_adj_tile32_kernel(const float* __restrict__ S, float* __restrict__ out, int xcd_order) {
    constexpr int W = 32, HH = 32, L = W * HH, TS = 34, LDT = 35;
    __shared__ float T[3][TS * LDT];
    __shared__ int prow[3][TS], pcol[3][TS];           // source row / column of (d, i), -1 = outside
    int py0, ly0, bb;
    ca_tile_of((int)blockIdx.x, xcd_order, py0, ly0, bb);
    const float* Sb = S + (long long)bb * L * L;
    float* ob = out + (long long)bb * L * L;
    const int p0 = py0 * W, l0 = ly0 * W;
    if (threadIdx.x < 2 * 3 * TS) {
        const int which = threadIdx.x / (3 * TS), e = threadIdx.x % (3 * TS), d = e / TS, i = e % TS;
        const int q = (which ? l0 : p0) - 1 + i;
        int src = -1;
        if ((unsigned)q < (unsigned)L) {
            const int a = (q & (W - 1)) * HH + (q >> 5) + (d - 1);          // tr(q) + d
            if ((unsigned)a < (unsigned)L) src = (a & (HH - 1)) * W + (a >> 5);     // itr
        }
        (which ? pcol : prow)[d][i] = src;
    }
    __syncthreads();
    {
        constexpr int NIT = (TS * TS + 255) / 256;
        float v[3][NIT];
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int e = threadIdx.x + it * 256, i = e / TS, j = e - i * TS;
                const int pr = e < TS * TS ? prow[d][i] : -1, lc = e < TS * TS ? pcol[d][j] : -1;
                v[d][it] = (pr >= 0 && lc >= 0) ? Sb[(long long)pr * L + lc] : 0.f;
            }
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int e = threadIdx.x + it * 256, i = e / TS, j = e - i * TS;
                if (e < TS * TS) T[d][i * LDT + j] = v[d][it];
            }
    }
    __syncthreads();
    const int r = threadIdx.x >> 3, c0 = (threadIdx.x & 7) * 4;
    float o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = c0 + u;
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 3; ++e)
#pragma unroll
            for (int d = 0; d < 3; ++d) acc += T[d][(r + e) * LDT + c + e];
        o[u] = acc;
    }
    *reinterpret_cast<float4*>(ob + (long long)(p0 + r) * L + l0 + c0) = make_float4(o[0], o[1], o[2], o[3]);
}
// (A variant with the nine index maps tabulated in LDS -- nine table reads instead of ~150 integer instructions per element -- measured
// 204 / 220 us against 109 / 117 us: the dependent LDS read in front of every global load costs more than the arithmetic.  Not kept.)
static bool at_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
extern "C" int hv_ca_fuse(const float* S, float* out, int B, int h, int w, int adjoint, void* stream) {
    if (!S || !out || S == out || B <= 0 || h <= 0 || w <= 0) return HV_ERR_ARG;
    if (at_pow2(h) && at_pow2(w) && (long long)h * w <= 32768) {
        static const int tile32 = getenv("HV_CA_FUSE_TILE") ? atoi(getenv("HV_CA_FUSE_TILE")) : 1;      // A/B knob
        if (tile32 && h == 32 && w == 32 && B <= 65535 && !((uintptr_t)out & 15)) {
            static const int xcd_order = getenv("HV_CA_FUSE_XCD") ? atoi(getenv("HV_CA_FUSE_XCD")) : 1;      // A/B knob (same bits either way)
            if (adjoint) hipLaunchKernelGGL(ca_fuse_adj_tile32_kernel, dim3(1024 * B), dim3(256), 0, (hipStream_t)stream, S, out, xcd_order);
            else hipLaunchKernelGGL(ca_fuse_tile32_kernel, dim3(1024 * B), dim3(256), 0, (hipStream_t)stream, S, out, xcd_order);
            HV_LAUNCH_CHECK();
            return HV_OK;
        }
        const int lh = __builtin_ctz(h), lw = __builtin_ctz(w);
        const dim3 grid(at_grid((long long)h * w * h * w, 4096), B);
        if (adjoint) hipLaunchKernelGGL((ca_fuse_p2_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, S, out, lh, lw);
        else hipLaunchKernelGGL((ca_fuse_p2_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, S, out, lh, lw);
        HV_LAUNCH_CHECK();
        return HV_OK;
    }
    const long long L = (long long)h * w, n = (long long)B * L * L;
    hipLaunchKernelGGL(ca_fuse_kernel, dim3(at_grid(n, 65536)), dim3(256), 0, (hipStream_t)stream, S, out, h, w, adjoint, n);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// ---- masked scaled softmax over l (one 256-thread block per row) -------------------------------------
template <typename AT = float>       // AT = _Float16: the attention matrix stored as fp16 (the GEMM route: its consumers are the batched GEMMs and the soft-max backward)
__global__ __launch_bounds__(256) void ca_softmax_kernel(const float* __restrict__ S, const float* __restrict__ mm, AT* __restrict__ A, int L,
                                                         float scale, int* __restrict__ argmax, long long mm_bs) {
    __shared__ float red[8];
    __shared__ int redi[8];
    const long long row = blockIdx.x;
    mm += (row / L) * mm_bs;           // per-sample masks (hv_ca_softmax_batched) or one shared mask (stride 0)
    const float* s = S + row * L;
    AT* a = A + row * L;
    const int tid = threadIdx.x;
    if (L <= 2048 && (L & 255) == 0) {   // the row fits the workgroup's registers: S is read once instead of three times
        const int ept = L >> 8;
        float v[8], m[8];
        float mx = -3.0e38f;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < ept) { m[k] = mm[tid + k * 256]; v[k] = s[tid + k * 256] * m[k] * scale; mx = fmaxf(mx, v[k]); }
        mx = hv_wave_max(mx);
        if ((tid & 63) == 0) red[tid >> 6] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < ept) { v[k] = expf(v[k] - mx); sum += v[k]; }
        sum = hv_wave_sum(sum);
        __syncthreads();
        if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
        __syncthreads();
        sum = red[4] + red[5] + red[6] + red[7];
        float best = -1.f;
        int bi = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < ept) {
                const float o = v[k] / sum * m[k];
                a[tid + k * 256] = (AT)o;
                if (o > best) { best = o; bi = tid + k * 256; }
            }
        if (argmax) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ob = __shfl_xor(best, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            __syncthreads();
            if ((tid & 63) == 0) { red[tid >> 6] = best; redi[tid >> 6] = bi; }
            __syncthreads();
            if (tid == 0) {
                for (int k = 1; k < 4; ++k)
                    if (red[k] > best || (red[k] == best && redi[k] < bi)) { best = red[k]; bi = redi[k]; }
                argmax[row] = bi;
            }
        }
        return;
    }
    float mx = -3.0e38f;
    for (int l = tid; l < L; l += 256) mx = fmaxf(mx, s[l] * mm[l] * scale);
    mx = hv_wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int l = tid; l < L; l += 256) sum += expf(s[l] * mm[l] * scale - mx);
    sum = hv_wave_sum(sum);
    __syncthreads();
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
    __syncthreads();
    sum = red[4] + red[5] + red[6] + red[7];
    float best = -1.f;
    int bi = 0x7fffffff;
    for (int l = tid; l < L; l += 256) {
        const float v = expf(s[l] * mm[l] * scale - mx) / sum * mm[l];
        a[l] = (AT)v;
        if (v > best) { best = v; bi = l; }
    }
    if (argmax) {  // first index of the maximum (torch.argmax tie rule on CPU)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        __syncthreads();
        if ((tid & 63) == 0) { red[tid >> 6] = best; redi[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int k = 1; k < 4; ++k)
                if (red[k] > best || (red[k] == best && redi[k] < bi)) { best = red[k]; bi = redi[k]; }
            argmax[row] = bi;
        }
    }
}
// One WAVE per row (round 3): a lane keeps NV x 4 consecutive-in-groups elements of its row (16-byte loads, NV of them in flight), the maximum, the sum
// and the arg-max are wave reductions -- no LDS, no workgroup barrier (the block-per-row kernel above spends three __syncthreads on 4 elements per
// thread at L = 1024).  Same formulas; the fp32 sum of the exponentials is taken in another order.  L = 256 * NV.
template <typename AT, int NV>
__global__ __launch_bounds__(256) void ca_softmax_wave_kernel(const float* __restrict__ S, const float* __restrict__ mm, AT* __restrict__ A, float scale,
                                                              int* __restrict__ argmax, long long mm_bs, long long rows) {
    constexpr int L = 256 * NV;
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* mrow = mm + (row / L) * mm_bs;
    const float* s = S + row * L;
    float4 v[NV], m[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) { v[k] = *reinterpret_cast<const float4*>(s + k * 256 + lane * 4); m[k] = *reinterpret_cast<const float4*>(mrow + k * 256 + lane * 4); }
    float mx = -3.0e38f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        v[k].x = v[k].x * m[k].x * scale; v[k].y = v[k].y * m[k].y * scale; v[k].z = v[k].z * m[k].z * scale; v[k].w = v[k].w * m[k].w * scale;
        mx = fmaxf(fmaxf(mx, fmaxf(v[k].x, v[k].y)), fmaxf(v[k].z, v[k].w));
    }
    mx = hv_wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        v[k].x = expf(v[k].x - mx); v[k].y = expf(v[k].y - mx); v[k].z = expf(v[k].z - mx); v[k].w = expf(v[k].w - mx);
        sum += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    sum = hv_wave_sum(sum);
    float best = -1.f;
    int bi = 0x7fffffff;
    AT* a = A + row * L;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const float o[4] = {v[k].x / sum * m[k].x, v[k].y / sum * m[k].y, v[k].z / sum * m[k].z, v[k].w / sum * m[k].w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (o[e] > best) { best = o[e]; bi = k * 256 + lane * 4 + e; }       // increasing index inside a lane: the first maximum wins
        if constexpr (sizeof(AT) == 2) *reinterpret_cast<f16x4*>(a + k * 256 + lane * 4) = (f16x4){(_Float16)o[0], (_Float16)o[1], (_Float16)o[2], (_Float16)o[3]};
        else *reinterpret_cast<float4*>(a + k * 256 + lane * 4) = make_float4(o[0], o[1], o[2], o[3]);
    }
    if (argmax) {      // first index of the maximum (torch.argmax tie rule on CPU)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) argmax[row] = bi;
    }
}
// dS[p][l] = scale*mm[l]*A[p][l]*(dA[p][l] - sum_l' dA[p][l']*A[p][l']) with one wave per row
template <typename AT, int NV>
__global__ __launch_bounds__(256) void ca_softmax_bwd_wave_kernel(const float* __restrict__ dA, const AT* __restrict__ A, const float* __restrict__ mm,
                                                                  float* __restrict__ dS, float scale, long long rows) {
    constexpr int L = 256 * NV;
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* g = dA + row * L;
    const AT* a = A + row * L;
    float4 gv[NV], av[NV], mv[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        gv[k] = *reinterpret_cast<const float4*>(g + k * 256 + lane * 4);
        mv[k] = *reinterpret_cast<const float4*>(mm + k * 256 + lane * 4);
        if constexpr (sizeof(AT) == 2) {
            const f16x4 h = *reinterpret_cast<const f16x4*>(a + k * 256 + lane * 4);
            av[k] = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
        } else {
            av[k] = *reinterpret_cast<const float4*>(a + k * 256 + lane * 4);
        }
    }
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) dot += (gv[k].x * av[k].x + gv[k].y * av[k].y) + (gv[k].z * av[k].z + gv[k].w * av[k].w);
    dot = hv_wave_sum(dot);
#pragma unroll
    for (int k = 0; k < NV; ++k)
        *reinterpret_cast<float4*>(dS + row * L + k * 256 + lane * 4) =
            make_float4(scale * mv[k].x * av[k].x * (gv[k].x - dot), scale * mv[k].y * av[k].y * (gv[k].y - dot), scale * mv[k].z * av[k].z * (gv[k].z - dot),
                        scale * mv[k].w * av[k].w * (gv[k].w - dot));
}
static const int ca_sm_wave = getenv("HV_CA_SOFTMAX_WAVE") ? atoi(getenv("HV_CA_SOFTMAX_WAVE")) : 1;      // A/B knob
template <typename AT>
static bool ca_softmax_wave_launch(const float* S, const float* mm, long long mm_bs, AT* A, int B, int L, float scale, int* argmax, hipStream_t s) {
    if (!ca_sm_wave || (((uintptr_t)S | (uintptr_t)mm | (uintptr_t)A) & 15) || (mm_bs & 3)) return false;
    const long long rows = (long long)B * L;
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (L == 1024) hipLaunchKernelGGL((ca_softmax_wave_kernel<AT, 4>), grid, dim3(256), 0, s, S, mm, A, scale, argmax, mm_bs, rows);
    else if (L == 4096) hipLaunchKernelGGL((ca_softmax_wave_kernel<AT, 16>), grid, dim3(256), 0, s, S, mm, A, scale, argmax, mm_bs, rows);
    else if (L == 256) hipLaunchKernelGGL((ca_softmax_wave_kernel<AT, 1>), grid, dim3(256), 0, s, S, mm, A, scale, argmax, mm_bs, rows);
    else return false;
    return true;
}
template <typename AT>
static bool ca_softmax_bwd_wave_launch(const float* dA, const AT* A, const float* mm, float* dS, int B, int L, float scale, hipStream_t s) {
    if (!ca_sm_wave || (((uintptr_t)dA | (uintptr_t)mm | (uintptr_t)A | (uintptr_t)dS) & 15)) return false;
    const long long rows = (long long)B * L;
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (L == 1024) hipLaunchKernelGGL((ca_softmax_bwd_wave_kernel<AT, 4>), grid, dim3(256), 0, s, dA, A, mm, dS, scale, rows);
    else if (L == 4096) hipLaunchKernelGGL((ca_softmax_bwd_wave_kernel<AT, 16>), grid, dim3(256), 0, s, dA, A, mm, dS, scale, rows);
    else if (L == 256) hipLaunchKernelGGL((ca_softmax_bwd_wave_kernel<AT, 1>), grid, dim3(256), 0, s, dA, A, mm, dS, scale, rows);
    else return false;
    return true;
}
extern "C" int hv_ca_softmax(const float* S, const float* mm, float* A, int B, int L, float scale, int* argmax, void* stream) {
    if (!S || !mm || !A || B <= 0 || L <= 0) return HV_ERR_ARG;
    if (ca_softmax_wave_launch<float>(S, mm, 0ll, A, B, L, scale, argmax, (hipStream_t)stream)) { HV_LAUNCH_CHECK(); return HV_OK; }
    hipLaunchKernelGGL(ca_softmax_kernel<float>, dim3(B * L), dim3(256), 0, (hipStream_t)stream, S, mm, A, L, scale, argmax, 0ll);
    HV_LAUNCH_CHECK();
    return HV_OK;
}
extern "C" int hv_ca_softmax_batched(const float* S, const float* mm, long long mm_bstride, float* A, int B, int L, float scale, int* argmax,
                                     void* stream) {
    if (!S || !mm || !A || B <= 0 || L <= 0 || mm_bstride < 0) return HV_ERR_ARG;
    if (ca_softmax_wave_launch<float>(S, mm, mm_bstride, A, B, L, scale, argmax, (hipStream_t)stream)) { HV_LAUNCH_CHECK(); return HV_OK; }
    hipLaunchKernelGGL(ca_softmax_kernel<float>, dim3(B * L), dim3(256), 0, (hipStream_t)stream, S, mm, A, L, scale, argmax, mm_bstride);
    HV_LAUNCH_CHECK();
    return HV_OK;
}
extern "C" int hv_ca_softmax_f16(const float* S, const float* mm, long long mm_bstride, void* A_h, int B, int L, float scale, int* argmax, void* stream) {
    if (!S || !mm || !A_h || B <= 0 || L <= 0 || mm_bstride < 0) return HV_ERR_ARG;
    if (ca_softmax_wave_launch<_Float16>(S, mm, mm_bstride, (_Float16*)A_h, B, L, scale, argmax, (hipStream_t)stream)) { HV_LAUNCH_CHECK(); return HV_OK; }
    hipLaunchKernelGGL(ca_softmax_kernel<_Float16>, dim3(B * L), dim3(256), 0, (hipStream_t)stream, S, mm, (_Float16*)A_h, L, scale, argmax, mm_bstride);
    HV_LAUNCH_CHECK();
    return HV_OK;
}
// (Measured and not kept, round 3: score fusion + softmax as ONE kernel -- a workgroup per grid row of p walking the 32 column tiles with the fused
// rows held in LDS ([32][1024] floats), then the softmax from LDS: 128 us against 58 + 41 us for the two kernels.  The 131 KB row buffer leaves one
// 4-wave workgroup per CU, and the tile loop (15 four-byte loads, 15 LDS stores, 36 LDS reads per lane and tile) is then fully exposed; the two
// separate kernels run at 8 workgroups per CU and the extra 128 MB round trip of the fused scores costs less than that.)
// dS[p][l] = scale*mm[l]*A[p][l]*(dA[p][l] - sum_l' dA[p][l']*A[p][l'])   (A already carries the mask)
template <typename AT = float>
__global__ __launch_bounds__(256) void ca_softmax_bwd_kernel(const float* __restrict__ dA, const AT* __restrict__ A, const float* __restrict__ mm,
                                                             float* __restrict__ dS, int L, float scale) {
    __shared__ float red[20];
    const long long row = blockIdx.x;
    const AT* a = A + row * L;
    const float* g = dA + row * L;
    float dot = 0.f;
    for (int l = threadIdx.x; l < L; l += 256) dot += g[l] * (float)a[l];
    dot = hv_block_sum(dot, red);
    for (int l = threadIdx.x; l < L; l += 256) dS[row * L + l] = scale * mm[l] * (float)a[l] * (g[l] - dot);
}
extern "C" int hv_ca_softmax_backward(const float* dA, const float* A, const float* mm, float* dS, int B, int L, float scale, void* stream) {
    if (!dA || !A || !mm || !dS || B <= 0 || L <= 0) return HV_ERR_ARG;
    if (ca_softmax_bwd_wave_launch<float>(dA, A, mm, dS, B, L, scale, (hipStream_t)stream)) { HV_LAUNCH_CHECK(); return HV_OK; }
    hipLaunchKernelGGL(ca_softmax_bwd_kernel<float>, dim3(B * L), dim3(256), 0, (hipStream_t)stream, dA, A, mm, dS, L, scale);
    HV_LAUNCH_CHECK();
    return HV_OK;
}
extern "C" int hv_ca_softmax_backward_f16(const float* dA, const void* A_h, const float* mm, float* dS, int B, int L, float scale, void* stream) {
    if (!dA || !A_h || !mm || !dS || B <= 0 || L <= 0) return HV_ERR_ARG;
    if (ca_softmax_bwd_wave_launch<_Float16>(dA, (const _Float16*)A_h, mm, dS, B, L, scale, (hipStream_t)stream)) { HV_LAUNCH_CHECK(); return HV_OK; }
    hipLaunchKernelGGL(ca_softmax_bwd_kernel<_Float16>, dim3(B * L), dim3(256), 0, (hipStream_t)stream, dA, (const _Float16*)A_h, mm, dS, L, scale);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// ---- batched transpose (32x32 LDS tiles) ----------------------------------------------------------------
template <typename OT = float, typename IT = float>
__global__ __launch_bounds__(256) void transpose_kernel(const IT* __restrict__ src, OT* __restrict__ dst, int R, int C) {
    __shared__ float t[32][33];
    const long long b = blockIdx.z;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8)
        if (r0 + k < R && c0 + tx < C) t[k][tx] = (float)src[(b * R + r0 + k) * C + c0 + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
        if (c0 + k < C && r0 + tx < R) dst[(b * C + c0 + k) * R + r0 + tx] = (OT)t[tx][k];
}
// fp16-output transposes on 64 x 64 tiles with 16-byte accesses on both sides (R, C multiples of 8, 16-byte aligned bases): the 32 x 32 kernel above moves
// 2 (4) bytes per lane and instruction -- 32.4 us for the 64 MB of the attention matrix, 20.6 us for wp.  Exact (a conversion and a copy).
template <typename IT>
__global__ __launch_bounds__(256) void transpose64_h_kernel(const IT* __restrict__ src, _Float16* __restrict__ dst, int R, int C) {
    constexpr int LDT = 72;                                       // halfs per tile row (144 B)
    __shared__ __attribute__((aligned(16))) _Float16 t[64 * LDT];
    const long long b = blockIdx.z;
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64, tid = threadIdx.x;
    const IT* sb = src + b * (long long)R * C;
    _Float16* db = dst + b * (long long)R * C;
#pragma unroll
    for (int k = 0; k < 2; ++k) {                                 // 64 rows x 8 pieces of 8 elements
        const int e = tid + 256 * k, r = e >> 3, p8 = (e & 7) * 8;
        f16x8 h = {0, 0, 0, 0, 0, 0, 0, 0};
        if (r0 + r < R && c0 + p8 < C) {
            if constexpr (sizeof(IT) == 2) {
                h = __builtin_bit_cast(f16x8, *reinterpret_cast<const hv_u32x4*>(sb + (long long)(r0 + r) * C + c0 + p8));
            } else {
                const float4 a = *reinterpret_cast<const float4*>(sb + (long long)(r0 + r) * C + c0 + p8);
                const float4 c = *reinterpret_cast<const float4*>(sb + (long long)(r0 + r) * C + c0 + p8 + 4);
                h = f16x8{(_Float16)a.x, (_Float16)a.y, (_Float16)a.z, (_Float16)a.w, (_Float16)c.x, (_Float16)c.y, (_Float16)c.z, (_Float16)c.w};
            }
        }
        *reinterpret_cast<hv_u32x4*>(t + r * LDT + p8) = __builtin_bit_cast(hv_u32x4, h);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {                                 // 64 output rows (source columns) x 8 pieces of 8 source rows
        const int e = tid + 256 * k, c = e & 63, r8 = (e >> 6) * 8;
        if (c0 + c >= C || r0 + r8 >= R) continue;
        f16x8 h;
#pragma unroll
        for (int q = 0; q < 8; ++q) h[q] = t[(r8 + q) * LDT + c];
        *reinterpret_cast<hv_u32x4*>(db + (long long)(c0 + c) * R + r0 + r8) = __builtin_bit_cast(hv_u32x4, h);
    }
}
template <typename IT>
static bool transpose64_launch(const IT* src, _Float16* dst, int B, int R, int C, hipStream_t s) {
    static const int on = getenv("HV_TRANSPOSE64") ? atoi(getenv("HV_TRANSPOSE64")) : 1;      // A/B knob
    if (!on || (R & 7) || (C & 7) || (((uintptr_t)src | (uintptr_t)dst) & 15) || B > 65535) return false;
    hipLaunchKernelGGL(transpose64_h_kernel<IT>, dim3(hv_cdiv(C, 64), hv_cdiv(R, 64), B), dim3(256), 0, s, src, dst, R, C);
    return true;
}
extern "C" int hv_transpose_batched(const float* src, float* dst, int B, int R, int C, void* stream) {
    if (!src || !dst || B <= 0 || R <= 0 || C <= 0) return HV_ERR_ARG;
    hipLaunchKernelGGL(transpose_kernel<float>, dim3(hv_cdiv(C, 32), hv_cdiv(R, 32), B), dim3(256), 0, (hipStream_t)stream, src, dst, R, C);
    HV_LAUNCH_CHECK();
    return HV_OK;
}
extern "C" int hv_transpose_batched_f16(const float* src, void* dst_h, int B, int R, int C, void* stream) {      // the transpose stored as fp16
    if (!src || !dst_h || B <= 0 || R <= 0 || C <= 0) return HV_ERR_ARG;
    if (transpose64_launch<float>(src, (_Float16*)dst_h, B, R, C, (hipStream_t)stream)) { HV_LAUNCH_CHECK(); return HV_OK; }
    hipLaunchKernelGGL(transpose_kernel<_Float16>, dim3(hv_cdiv(C, 32), hv_cdiv(R, 32), B), dim3(256), 0, (hipStream_t)stream, src, (_Float16*)dst_h, R, C);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

extern "C" int hv_transpose_batched_h2h(const void* src_h, void* dst_h, int B, int R, int C, void* stream) {        // fp16 in, fp16 out (exact)
    if (!src_h || !dst_h || B <= 0 || R <= 0 || C <= 0) return HV_ERR_ARG;
    if (transpose64_launch<_Float16>((const _Float16*)src_h, (_Float16*)dst_h, B, R, C, (hipStream_t)stream)) { HV_LAUNCH_CHECK(); return HV_OK; }
    hipLaunchKernelGGL((transpose_kernel<_Float16, _Float16>), dim3(hv_cdiv(C, 32), hv_cdiv(R, 32), B), dim3(256), 0, (hipStream_t)stream, (const _Float16*)src_h,
                       (_Float16*)dst_h, R, C);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// ---- backward of the matching scores -------------------------------------------------------------------
// Gs[b][i][j] = dS[b][j][i]*rnorm[b][i] + dS[b][i][j]*rnorm[b][j]
// xcd_order (L / 32 a multiple of 8): tile (i, j) reads the blocks (j, i) and (i, j) of dS, tile (j, i) the same two -- both are dealt to the XCD
// (i + j) & 7 (linear workgroup id & 7 is the XCD the hardware's round-robin gives it), where an XCD's 128 tiles of a sample run side by side: every
// block of dS crosses the fabric once instead of twice (PMC: 126 MB fetched per launch for 64 MB, the kernel ran at 5.7 TB/s of fabric traffic).
__global__ __launch_bounds__(256) void ca_gs_kernel(const float* __restrict__ dS, const float* __restrict__ rnorm, float* __restrict__ Gs, int L, int xcd_order) {
    __shared__ float t[32][33];
    long long b;
    int it, jt;
    if (xcd_order) {
        const int nt = L >> 5, per = nt * (nt >> 3), id = (int)blockIdx.x, xcd = id & 7;
        int q = id >> 3;
        b = q / per;
        q -= (int)b * per;
        it = q / (nt >> 3);
        jt = ((xcd - it) & 7) + 8 * (q % (nt >> 3));
    } else {
        b = blockIdx.z; jt = blockIdx.x; it = blockIdx.y;
    }
    const int j0 = jt * 32, i0 = it * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* D = dS + b * (long long)L * L;
    for (int k = ty; k < 32; k += 8) t[k][tx] = D[(long long)(j0 + k) * L + i0 + tx];   // t[jj][ii] = dS[j0+jj][i0+ii]
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int i = i0 + k, j = j0 + tx;
        Gs[(b * L + i) * L + j] = t[tx][k] * rnorm[b * L + i] + D[(long long)i * L + j] * rnorm[b * L + j];
    }
}
// coef[b][l] = -(sum_p dS[p][l]*S0[p][l]) / norm[l]^2   (0 where the norm was clamped)
// grid (L/64, B, CA_PCH): 64 columns x 4 row lanes per block, the rows split into CA_PCH chunks -> part[b][chunk][l];
// a second tiny kernel folds the chunks in fixed order
#define CA_PCH 16
__global__ __launch_bounds__(256) void ca_coef_part_kernel(const float* __restrict__ dS, const float* __restrict__ S0, float* __restrict__ part, int L) {
    __shared__ float sh[4][64];
    const long long b = blockIdx.y;
    const int lx = threadIdx.x & 63, pr = threadIdx.x >> 6;
    const int l = blockIdx.x * 64 + lx;
    const int rows = (L + CA_PCH - 1) / CA_PCH, p0 = blockIdx.z * rows, p1 = min(L, p0 + rows);
    const float* D = dS + b * (long long)L * L;
    const float* S = S0 + b * (long long)L * L;
    float s0 = 0.f, s1 = 0.f;
    if (l < L) {
        int p = p0 + pr;
        for (; p + 4 < p1; p += 8) {
            s0 += D[(long long)p * L + l] * S[(long long)p * L + l];
            s1 += D[(long long)(p + 4) * L + l] * S[(long long)(p + 4) * L + l];
        }
        for (; p < p1; p += 4) s0 += D[(long long)p * L + l] * S[(long long)p * L + l];
    }
    sh[pr][lx] = s0 + s1;
    __syncthreads();
    if (pr == 0 && l < L) part[(b * CA_PCH + blockIdx.z) * L + l] = (sh[0][lx] + sh[1][lx]) + (sh[2][lx] + sh[3][lx]);
}
__global__ void ca_coef_final_kernel(const float* __restrict__ part, const float* __restrict__ norm, float* __restrict__ coef, int L) {
    const long long b = blockIdx.y;
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CA_PCH; ++c) s += part[(b * CA_PCH + c) * L + l];
    const float nv = norm[b * L + l];
    coef[b * L + l] = nv > 1e-4f ? -s / (nv * nv) : 0.f;
}
extern "C" int hv_ca_score_backward_prep(const float* dS, const float* S0, const float* norm, const float* rnorm, float* Gs, float* coef,
                                         int B, int L, void* stream) {
    if (!dS || !S0 || !norm || !rnorm || !Gs || !coef || B <= 0 || L <= 0 || (L & 31)) return HV_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    static const int gs_xcd = getenv("HV_CA_GS_XCD") ? atoi(getenv("HV_CA_GS_XCD")) : 1;      // A/B knob (same bits either way)
    if (gs_xcd && (L / 32) % 8 == 0 && (long long)(L / 32) * (L / 32) * B < (1ll << 31))
        hipLaunchKernelGGL(ca_gs_kernel, dim3((L / 32) * (L / 32) * B), dim3(256), 0, s, dS, rnorm, Gs, L, 1);
    else
        hipLaunchKernelGGL(ca_gs_kernel, dim3(L / 32, L / 32, B), dim3(256), 0, s, dS, rnorm, Gs, L, 0);
    HV_LAUNCH_CHECK();
    // row-chunk partials live behind the result: coef holds B*L*(1 + 16) floats (include/hvgan.h)
    float* part = coef + (long long)B * L;
    hipLaunchKernelGGL(ca_coef_part_kernel, dim3(hv_cdiv(L, 64), B, CA_PCH), dim3(256), 0, s, dS, S0, part, L);
    HV_LAUNCH_CHECK();
    hipLaunchKernelGGL(ca_coef_final_kernel, dim3(hv_cdiv(L, 128), B), dim3(128), 0, s, part, norm, coef, L);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// The adjoint of the score fusion and the two reductions above in one pass (attention map 32 x 32): Gs needs the tile (it, jt) of dS0 AND its mirror
// (jt, it), so a workgroup owns the PAIR -- it builds both tiles exactly as ca_fuse_adj_tile32_kernel does (same term order, same bits), keeps them in
// LDS, writes the two Gs tiles (term order of ca_gs_kernel: same bits) and the column sums of dS0 * S0 of both tiles as partials part[b][row block][l]
// (32 row blocks; ca_coef_final_kernel<32> folds them in block order -- a different summation order from the 16 row chunks above, same tolerance).
// dS0 never reaches HBM (64 MB written, 2 x 64 + 64 MB read back by the two consumers).  Tile pairs are dealt by circular diagonal delta = jt - it
// (0 .. 16; -delta is the mirror) to the XCD the round-robin gives the workgroup, as ca_tile_of() does: delta = 8 * group + XCD, group 0 .. 2, the
// ids whose delta exceeds 16 (and the second half of delta = 16, which mirrors the first) leave at once; every XCD gets two diagonals' worth of work.
__device__ __forceinline__ void ca_adj_tile32(const float* __restrict__ Sb, int p0, int l0, float* T, int (*prow)[34], int (*pcol)[34], float (&o)[4]) {
    constexpr int W = 32, HH = 32, L = W * HH, TS = 34, LDT = 35;
    if (threadIdx.x < 2 * 3 * TS) {
        const int which = threadIdx.x / (3 * TS), e = threadIdx.x % (3 * TS), d = e / TS, i = e % TS;
        const int q = (which ? l0 : p0) - 1 + i;
        int src = -1;
        if ((unsigned)q < (unsigned)L) {
            const int a = (q & (W - 1)) * HH + (q >> 5) + (d - 1);
            if ((unsigned)a < (unsigned)L) src = (a & (HH - 1)) * W + (a >> 5);
        }
        (which ? pcol : prow)[d][i] = src;
    }
    __syncthreads();
    {
        constexpr int NIT = (TS * TS + 255) / 256;
        float v[3][NIT];
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int e = threadIdx.x + it * 256, i = e / TS, j = e - i * TS;
                const int pr = e < TS * TS ? prow[d][i] : -1, lc = e < TS * TS ? pcol[d][j] : -1;
                v[d][it] = (pr >= 0 && lc >= 0) ? Sb[(long long)pr * L + lc] : 0.f;
            }
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int e = threadIdx.x + it * 256, i = e / TS, j = e - i * TS;
                if (e < TS * TS) T[d * TS * LDT + i * LDT + j] = v[d][it];
            }
    }
    __syncthreads();
    const int r = threadIdx.x >> 3, c0 = (threadIdx.x & 7) * 4;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = c0 + u;
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 3; ++e)
#pragma unroll
            for (int d = 0; d < 3; ++d) acc += T[d * TS * LDT + (r + e) * LDT + c + e];
        o[u] = acc;
    }
}
__global__ __launch_bounds__(256) void ca_fuse_adj_prep32_kernel(const float* __restrict__ dS1, const float* __restrict__ S0, const float* __restrict__ rnorm,
                                                                 float* __restrict__ Gs, float* __restrict__ part) {
    constexpr int L = 1024, TS = 34, LDT = 35, LP = 33;
    __shared__ float T[3 * TS * LDT];                  // the three source pieces of a tile; afterwards the two 32 x 33 product tiles
    __shared__ float tA[32 * LP], tB[32 * LP];         // dS0 tiles (it, jt) and (jt, it)
    __shared__ int prow[3][TS], pcol[3][TS];
    const int id = (int)blockIdx.x, xcd = id & 7;
    int q = id >> 3;
    const int k = q & 31;
    q >>= 5;
    const int delta = (q % 3) * 8 + xcd, bb = q / 3;
    if (delta > 16 || (delta == 16 && k >= 16)) return;
    const bool self = delta == 0;
    const int it = k, jt = (k + delta) & 31, i0 = it * 32, j0 = jt * 32;
    const float* Db = dS1 + (long long)bb * L * L;
    const float* Sb = S0 + (long long)bb * L * L;
    const float* rn = rnorm + (long long)bb * L;
    float* Gb = Gs + (long long)bb * L * L;
    const int r = threadIdx.x >> 3, c0 = (threadIdx.x & 7) * 4;
    float oA[4], oB[4] = {0.f, 0.f, 0.f, 0.f};
    const float4 sA = *reinterpret_cast<const float4*>(Sb + (long long)(i0 + r) * L + j0 + c0);
    float4 sB = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!self) sB = *reinterpret_cast<const float4*>(Sb + (long long)(j0 + r) * L + i0 + c0);
    ca_adj_tile32(Db, i0, j0, T, prow, pcol, oA);
    if (!self) {
        __syncthreads();                               // T and the index tables are rebuilt
        ca_adj_tile32(Db, j0, i0, T, prow, pcol, oB);
    }
    __syncthreads();
    float* PA = T;
    float* PB = T + 32 * LP;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        tA[r * LP + c0 + u] = oA[u];
        tB[r * LP + c0 + u] = oB[u];
    }
    PA[r * LP + c0 + 0] = oA[0] * sA.x;  PA[r * LP + c0 + 1] = oA[1] * sA.y;  PA[r * LP + c0 + 2] = oA[2] * sA.z;  PA[r * LP + c0 + 3] = oA[3] * sA.w;
    PB[r * LP + c0 + 0] = oB[0] * sB.x;  PB[r * LP + c0 + 1] = oB[1] * sB.y;  PB[r * LP + c0 + 2] = oB[2] * sB.z;  PB[r * LP + c0 + 3] = oB[3] * sB.w;
    __syncthreads();
    const float* tM = self ? tA : tB;                  // the mirror tile
    {   // Gs(it, jt)[r][c] = dS0[j0 + c][i0 + r] * rnorm[i0 + r] + dS0[i0 + r][j0 + c] * rnorm[j0 + c]
        const float ri = rn[i0 + r];
        const float4 rj = *reinterpret_cast<const float4*>(rn + j0 + c0);
        *reinterpret_cast<float4*>(Gb + (long long)(i0 + r) * L + j0 + c0) =
            make_float4(tM[(c0 + 0) * LP + r] * ri + oA[0] * rj.x, tM[(c0 + 1) * LP + r] * ri + oA[1] * rj.y,
                        tM[(c0 + 2) * LP + r] * ri + oA[2] * rj.z, tM[(c0 + 3) * LP + r] * ri + oA[3] * rj.w);
    }
    if (!self) {   // Gs(jt, it)[r][c] = dS0[i0 + c][j0 + r] * rnorm[j0 + r] + dS0[j0 + r][i0 + c] * rnorm[i0 + c]
        const float rj = rn[j0 + r];
        const float4 ri = *reinterpret_cast<const float4*>(rn + i0 + c0);
        *reinterpret_cast<float4*>(Gb + (long long)(j0 + r) * L + i0 + c0) =
            make_float4(tA[(c0 + 0) * LP + r] * rj + oB[0] * ri.x, tA[(c0 + 1) * LP + r] * rj + oB[1] * ri.y,
                        tA[(c0 + 2) * LP + r] * rj + oB[2] * ri.z, tA[(c0 + 3) * LP + r] * rj + oB[3] * ri.w);
    }
    if (threadIdx.x < (self ? 32 : 64)) {              // column sums of dS0 * S0, rows in order
        const int half = threadIdx.x >> 5, c = threadIdx.x & 31;
        const float* P = half ? PB : PA;
        float s = 0.f;
#pragma unroll
        for (int rr = 0; rr < 32; ++rr) s += P[rr * LP + c];
        part[((long long)bb * 32 + (half ? jt : it)) * L + (half ? i0 : j0) + c] = s;
    }
}
// (Measured and not kept: every sample's last workgroup folding the 32 partials itself instead of the launch below -- 8 448 ticket atomics on 16 addresses
// serialise at the memory side: the kernel went from ~85 to 124 us, with agent-scope release fences per workgroup to +0.4 ms per step; hv_common.h.)
template <int NCH>
__global__ void ca_coef_final_n_kernel(const float* __restrict__ part, const float* __restrict__ norm, float* __restrict__ coef, int L) {
    const long long b = blockIdx.y;
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) s += part[(b * NCH + c) * L + l];
    const float nv = norm[b * L + l];
    coef[b * L + l] = nv > 1e-4f ? -s / (nv * nv) : 0.f;
}
extern "C" int hv_ca_fuse_backward_prep(const float* dS1, const float* S0, const float* norm, const float* rnorm, float* Gs, float* coef, int B, int h, int w,
                                        void* stream) {
    if (!dS1 || !S0 || !norm || !rnorm || !Gs || !coef || dS1 == Gs || B <= 0) return HV_ERR_ARG;
    if (h != 32 || w != 32 || B > 65535 || (((uintptr_t)S0 | (uintptr_t)Gs | (uintptr_t)rnorm) & 15)) return HV_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int L = 1024;
    float* part = coef + (long long)B * L;             // 32 row-block partials behind the result: coef holds 33 * B * L floats
    hipLaunchKernelGGL(ca_fuse_adj_prep32_kernel, dim3(768 * B), dim3(256), 0, s, dS1, S0, rnorm, Gs, part);
    HV_LAUNCH_CHECK();
    hipLaunchKernelGGL(ca_coef_final_n_kernel<32>, dim3(hv_cdiv(L, 128), B), dim3(128), 0, s, part, norm, coef, L);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// ---- col2im of the patch-matrix gradient, scattered to the even positions of the full map ------------
__global__ void ca_patches_bwd_kernel(const float* __restrict__ dwp, const float* __restrict__ wp, const float* __restrict__ coef,
                                      float* __restrict__ df, int H, int W, int C, int df_ld, int acc, long long n) {
    const int h = H / 2, w = W / 2, L = h * w;
    AT_LOOP(i, n) {
        const int c = (int)(i % C);
        long long r = i / C;
        const int X = (int)(r % W); r /= W;
        const int Y = (int)(r % H);
        const long long b = r / H;
        float v = 0.f;
        if (!((X | Y) & 1)) {
            const int y = Y / 2, x = X / 2;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ly = y - (tap / 3 - 1), lx = x - (tap % 3 - 1);
                if ((unsigned)ly < (unsigned)h && (unsigned)lx < (unsigned)w) {
                    const long long l = (long long)ly * w + lx;
                    const long long o = ((b * L + l) * 9 + tap) * C + c;
                    v += dwp[o] + coef[b * L + l] * wp[o];
                }
            }
        }
        float* d = df + ((b * H + Y) * W + X) * df_ld + c;
        *d = acc ? *d + v : v;
    }
}
// The accumulate form on four channels per thread: only the even positions of the full map receive anything, so the work items are the h x w
// patch positions (a quarter of the map), the nine taps' 16-byte loads are all issued before the first add, indices are 32-bit.  Same operation
// order per element as the kernel above: the same bits (that kernel walks the whole map with three 64-bit divisions and 18 four-byte loads per element).
__global__ __launch_bounds__(256) void ca_patches_bwd_vec_kernel(const float* __restrict__ dwp, const float* __restrict__ wp, const float* __restrict__ coef,
                                                                 float* __restrict__ df, int H, int W, int C, int df_ld, int n) {
    const int h = H / 2, w = W / 2, L = h * w, C4 = C >> 2;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int c = (i % C4) * 4;
        int r = i / C4;
        const int x = r % w;
        r /= w;
        const int y = r % h, b = r / h;
        float4 dv[9], wv[9];
        float cf[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ly = y - (tap / 3 - 1), lx = x - (tap % 3 - 1);
            const bool ok = (unsigned)ly < (unsigned)h && (unsigned)lx < (unsigned)w;
            const int l = ok ? ly * w + lx : 0;
            const long long o = (((long long)b * L + l) * 9 + tap) * C + c;
            dv[tap] = ok ? *reinterpret_cast<const float4*>(dwp + o) : make_float4(0.f, 0.f, 0.f, 0.f);
            wv[tap] = ok ? *reinterpret_cast<const float4*>(wp + o) : make_float4(0.f, 0.f, 0.f, 0.f);
            cf[tap] = ok ? coef[(long long)b * L + l] : 0.f;
        }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ly = y - (tap / 3 - 1), lx = x - (tap % 3 - 1);
            if ((unsigned)ly < (unsigned)h && (unsigned)lx < (unsigned)w) {
                v.x += dv[tap].x + cf[tap] * wv[tap].x; v.y += dv[tap].y + cf[tap] * wv[tap].y;
                v.z += dv[tap].z + cf[tap] * wv[tap].z; v.w += dv[tap].w + cf[tap] * wv[tap].w;
            }
        }
        float4* d = reinterpret_cast<float4*>(df + (((long long)b * H + 2 * y) * W + 2 * x) * df_ld + c);
        const float4 o4 = *d;
        *d = make_float4(o4.x + v.x, o4.y + v.y, o4.z + v.z, o4.w + v.w);
    }
}
extern "C" int hv_ca_patches_backward(const float* dwp, const float* wp, const float* coef, float* df, int B, int H, int W, int C, int df_ld,
                                      int accumulate, void* stream) {
    if (!dwp || !wp || !coef || !df || B <= 0 || H <= 0 || W <= 0 || C <= 0 || ((H | W) & 1) || df_ld < C) return HV_ERR_ARG;
    static const int pb_vec = getenv("HV_CA_PBWD_VEC") ? atoi(getenv("HV_CA_PBWD_VEC")) : 1;      // A/B knob (same bits either way)
    const long long n4 = (long long)B * (H / 2) * (W / 2) * (C / 4);
    if (pb_vec && accumulate && !(C & 3) && !(df_ld & 3) && !(((uintptr_t)dwp | (uintptr_t)wp | (uintptr_t)df) & 15) && n4 < (1ll << 31)) {
        long long blocks = (n4 + 255) / 256;
        if (blocks > 65536) blocks = 65536;
        hipLaunchKernelGGL(ca_patches_bwd_vec_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dwp, wp, coef, df, H, W, C, df_ld, (int)n4);
        HV_LAUNCH_CHECK();
        return HV_OK;
    }
    const long long n = (long long)B * H * W * C;
    hipLaunchKernelGGL(ca_patches_bwd_kernel, dim3(at_grid(n)), dim3(256), 0, (hipStream_t)stream, dwp, wp, coef, df, H, W, C, df_ld, accumulate, n);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// ------------------------------------------------------------------------------------------------ offset_flow (visualisation slot)
// The reference colours the arg-max offsets with the Middlebury wheel on the host (inpaint_networks.py:368,389-410; inpaint_tools.py:73-100,
// 181-211,245-280), one device->host->device round trip per forward.  Same arithmetic here in double precision, in the reference's
// operation order: offsets (row, col) = (arg // w - y, arg % w - x); the radius is normalised by the RUNNING maximum over samples 0..b
// (flow_to_image keeps maxrad across the batch loop); colour = wheel interpolation, floor(255*col) as uint8, /255, nearest x`up`.
__device__ __forceinline__ double flow_wheel(int k, int ch) {   // make_color_wheel(): 55 rows (RY 15, YG 6, GC 4, CB 11, BM 13, MR 6)
    const int seg_n[6] = {15, 6, 4, 11, 13, 6}, full[6] = {0, 1, 1, 2, 2, 0}, ramp[6] = {1, 0, 2, 1, 0, 2}, up[6] = {1, 0, 1, 0, 1, 0};
    int s = 0, base = 0;
    while (k >= base + seg_n[s]) { base += seg_n[s]; ++s; }
    if (ch == full[s]) return 255.0;
    if (ch != ramp[s]) return 0.0;
    const double t = floor(255.0 * (double)(k - base) / (double)seg_n[s]);
    return up[s] ? t : 255.0 - t;
}

__global__ void __launch_bounds__(256) ca_flow_kernel(const int* __restrict__ argmax, int B, int h, int w, int up, float* __restrict__ flow) {
    __shared__ double red[4];
    const int b = blockIdx.x, L = h * w;
    // running maximum radius over samples 0..b (initial value -1, like the reference)
    double mx = -1.0;
    for (long long i = threadIdx.x; i < (long long)(b + 1) * L; i += blockDim.x) {
        const int p = (int)(i % L), a = argmax[i];
        const double u = (double)(a / w - p / w), v = (double)(a % w - p % w);
        mx = fmax(mx, sqrt(u * u + v * v));
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    const double den = mx + 2.220446049250313e-16;   // np.finfo(float).eps
    const int Ho = h * up, Wo = w * up;
    for (int p = threadIdx.x; p < L; p += blockDim.x) {
        const int a = argmax[(long long)b * L + p];
        const double u = (double)(a / w - p / w) / den, v = (double)(a % w - p % w) / den;
        const double rad = sqrt(u * u + v * v);
        const double ang = atan2(-v, -u) / 3.141592653589793;
        const double fk = (ang + 1.0) / 2.0 * 54.0 + 1.0;
        const int k0 = (int)floor(fk);
        const int k1 = (k0 + 1 == 56) ? 1 : k0 + 1;
        const double f = fk - (double)k0;
        for (int ch = 0; ch < 3; ++ch) {
            double col = (1.0 - f) * (flow_wheel(k0 - 1, ch) / 255.0) + f * (flow_wheel(k1 - 1, ch) / 255.0);
            col = (rad <= 1.0) ? 1.0 - rad * (1.0 - col) : col * 0.75;
            const float val = (float)(unsigned char)floor(255.0 * col) / 255.f;
            float* dst = flow + (((long long)b * 3 + ch) * Ho + (long long)(p / w) * up) * Wo + (long long)(p % w) * up;
            for (int yy = 0; yy < up; ++yy)
                for (int xx = 0; xx < up; ++xx) dst[(long long)yy * Wo + xx] = val;
        }
    }
}
extern "C" int hv_ca_flow(const int* argmax, int B, int h, int w, int up, float* flow, void* stream) {
    if (!argmax || !flow || B <= 0 || h <= 0 || w <= 0 || up <= 0) return HV_ERR_ARG;
    hipLaunchKernelGGL(ca_flow_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, argmax, B, h, w, up, flow);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

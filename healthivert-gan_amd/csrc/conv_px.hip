// Thin full-resolution layers of the generators (3x3 / 5x5, stride 1, at most 16 channels on either side: the 5x5 stems, the 16 <-> 8 channel layers, the
// 1-channel heads and all their data gradients), fp16 mode: ONE LANE = ONE OUTPUT PIXEL on v_mfma_f32_4x4x4_16B_f16.
//
// These layers are 16-100 MB of traffic and 1-13 GFLOP at 256^2, bs 16 -- memory work -- but ran at 2-4x their memory time: the 16-wide MFMA tiles of the
// other kernels turn 16 pixels into one wave-instruction, so the per-instruction work around the MFMAs (addresses, edge masks, fragment packing, the
// epilogue: ~130 vector instructions per 16 pixels in the heads' data gradient, PMC SQ_INSTS_VALU) is paid four times per 64 pixels; the lane-per-pixel
// VALU kernels pay a multiply-add and a conversion per product.  The 4x4x4 MFMA is sixteen independent 4x4 blocks: block b = lanes 4b .. 4b+3,
//   A (4 rows x 4 k): lane 4b + i holds row i      -> four output channels x four consecutive operand channels of one tap: the same for every block
//   B (4 k x 4 cols): lane 4b + j holds column j   -> the lane's OWN pixel: four consecutive channels of one window pixel, straight from the load registers
//   D (4 x 4):        lane 4b + j holds column j   -> the lane's own pixel, channels 4m .. 4m+3 in acc[m]
// so every lane feeds and receives its own pixel (lane = 4b + j), fragments need no packing, edges are handled by the loads' range check (one load per window
// pixel piece, out-of-image -> zero), and the epilogue works on the lane's whole channel row (16-byte stores).  A comes from an LDS table
// [co quad][k quad][row i][4 halfs] filled once per workgroup (the 4 rows are the only distinct addresses of a read: broadcast).
// MFMA work per 64 pixels: (taps * Cin / 4) * (Cout / 4) instructions of 8 cycles -- 0.6-0.8k cycles for the widest of these layers.
#include <stdlib.h>
#include <type_traits>

#include "conv_halo.h"

typedef unsigned int u32x2p __attribute__((ext_vector_type(2)));

struct PxK {
    const _Float16* x; const _Float16* w; const float* bias; void* y; const void* mul;
    int B, H, W, x_ld, x_coff, Cin, Cout, y_ld, y_coff, mul_ld, mul_coff, mul_act, mul_half, y_half, act, accumulate, pad;
    float alpha;
    unsigned x_bytes;
    int tiles, segs, rblocks;      // tiles (TH rows x 256 pixels) in all, 256-pixel segments per row, row blocks per image
};

// KS: filter size; CI: operand channels per pixel as loaded (4, 8, 12 or 16; channels >= Cin read whatever the tensor holds there and meet zero filters);
// CQ: output channel quads; TR: the gather form of the transposed convolution (data gradients); Y1: one fp32 output channel (the heads); TH: output rows per tile.
// A workgroup takes tiles of TH rows x 256 pixels: the TH + KS - 1 input rows x (256 + KS - 1) pixels are staged ONCE in LDS as stored (16-byte pieces, zeros
// outside the image), then wave w walks the TH rows of its 64-pixel segment: a lane reads its window row by row from LDS (its own pixel row is contiguous:
// 16-byte reads at a lane stride of one pixel -- conflict-free) and every 8-byte piece is a B fragment as it is.  (With the window fetched from global memory
// per lane -- nine to twenty-five loads per pixel through L1 -- the wider layers took 44-59 us.)
template <int KS, int CI, int CQ, bool TR, bool Y1, int TH>
__global__ __launch_bounds__(256) void conv_px_kernel(const PxK p) {
    constexpr int QI = CI / 4, NKQ = KS * KS * QI;                       // k quads: (staged row r', window pixel j, channel quad)
    constexpr int PW = 256 + KS - 1, PR = TH + KS - 1, PB = CI * 2;      // staged pixels per row, rows, bytes per pixel
    constexpr int VB = (CI & 7) ? 8 : 16, VPP = PB / VB;                 // staging piece: bytes, pieces per pixel
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* wl = reinterpret_cast<_Float16*>(smem);                    // [CQ][NKQ][4 rows][4 halfs]
    char* xs = smem + CQ * NKQ * 32;                                     // [PR][PW][PB]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < CQ * NKQ * 4; e += 256) {
        const int i = e & 3, kq = (e >> 2) % NKQ, m = (e >> 2) / NKQ;
        const int q = kq % QI, j = (kq / QI) % KS, rr = kq / (QI * KS);
        // staged row rr / window pixel j of an output pixel is tap (r, s): forward (rr, j), gather form (KS-1 - rr, KS-1 - j)
        const int r = TR ? KS - 1 - rr : rr, s = TR ? KS - 1 - j : j, co = 4 * m + i, c0 = 4 * q;
        f16x4v v = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        if (co < p.Cout) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c0 + c < p.Cin) v[c] = p.w[((long long)co * KS * KS + r * KS + s) * p.Cin + c0 + c];
        }
        *reinterpret_cast<f16x4v*>(wl + e * 4) = v;
    }
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.x), 0, p.x_bytes, 0x00020000);
    const _Float16* wrow = wl + (lane & 3) * 4;
    float bv[CQ][4];                                                     // (uniform: scalar registers)
#pragma unroll
    for (int m = 0; m < CQ; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[m][r] = (p.bias && 4 * m + r < p.Cout) ? p.bias[4 * m + r] : 0.f;
    for (int t = blockIdx.x; t < p.tiles; t += gridDim.x) {
        const int seg = t % p.segs, rb = t / p.segs, yb = rb % p.rblocks, b = rb / p.rblocks;      // (scalar)
        const int y0 = yb * TH, x0 = seg * 256;
        __syncthreads();                                                 // the previous tile's readers are done (first pass: the filter table is written)
        // ---- stage rows y0 - pad .. y0 + TH - 1 + pad, pixels x0 - pad .. x0 + 255 + pad: the pieces of a row are dealt to the threads (piece -> pixel is a
        // shift or a small constant division; the row's validity and base are scalars)
        constexpr int RP = PW * VPP, IT = (RP + 255) / 256;                // pieces per row, rounds per row
        typedef typename std::conditional<VB == 16, u32x4, u32x2p>::type SV;
        SV sv[PR][IT];
        unsigned poff[IT];
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int e = tid + i * 256, px = e / VPP, pc = e - px * VPP, wi = x0 - p.pad + px;
            poff[i] = (e < RP && (unsigned)wi < (unsigned)p.W) ? (unsigned)((wi * p.x_ld) * 2 + pc * VB) : HV_OOB;
        }
#pragma unroll
        for (int r = 0; r < PR; ++r) {
            const int hi = y0 - p.pad + r;                               // (scalar)
            const bool rok = (unsigned)hi < (unsigned)p.H;
            const unsigned rbase = rok ? (unsigned)((((b * p.H + hi) * p.W) * p.x_ld + p.x_coff) * 2) : HV_OOB;
#pragma unroll
            for (int i = 0; i < IT; ++i) {
                const unsigned off = (rok && poff[i] != HV_OOB) ? rbase + poff[i] : HV_OOB;
                if constexpr (VB == 16) sv[r][i] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, off, 0, 0);
                else sv[r][i] = __builtin_amdgcn_raw_buffer_load_b64(xsrc, off, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < PR; ++r)
#pragma unroll
            for (int i = 0; i < IT; ++i) {
                const int e = tid + i * 256;
                if (e < RP) *reinterpret_cast<SV*>(xs + (r * RP + e) * VB) = sv[r][i];
            }
        __syncthreads();
        const int lx = wave * 64 + lane, x = x0 + lx;
#pragma unroll 1
        for (int ty = 0; ty < TH; ++ty) {
            const int y = y0 + ty;
            if (y >= p.H) break;                                         // (scalar)
            f32x4 acc[CQ];
#pragma unroll
            for (int m = 0; m < CQ; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rr = 0; rr < KS; ++rr) {
                // the lane's KS window pixels of this staged row: KS * PB contiguous bytes
                const char* rp = xs + ((ty + rr) * PW + lx) * PB;
                u32x2p wv[KS * QI];
                if constexpr (VB == 16) {
#pragma unroll
                    for (int i = 0; i < KS * QI / 2; ++i) {
                        const u32x4 v = *reinterpret_cast<const u32x4*>(rp + i * 16);
                        wv[2 * i] = u32x2p{v.x, v.y}; wv[2 * i + 1] = u32x2p{v.z, v.w};
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < KS * QI; ++i) wv[i] = *reinterpret_cast<const u32x2p*>(rp + i * 8);
                }
#pragma unroll
                for (int i = 0; i < KS * QI; ++i) {
                    const f16x4v bf = __builtin_bit_cast(f16x4v, wv[i]);
#pragma unroll
                    for (int m = 0; m < CQ; ++m) {
                        const f16x4v af = *reinterpret_cast<const f16x4v*>(wrow + (m * NKQ + rr * KS * QI + i) * 16);
                        acc[m] = __builtin_amdgcn_mfma_f32_4x4x4f16(af, bf, acc[m], 0, 0, 0);
                    }
                }
            }
            if (x >= p.W) continue;
            // ---- epilogue on the lane's own channel row: alpha, bias, (pre-activation accumulate), activation, act' multiplier, accumulate
            const long long pix = ((long long)b * p.H + y) * p.W + x;
            if constexpr (Y1) {
                float v = acc[0][0] * p.alpha + bv[0][0];
                float* yp = reinterpret_cast<float*>(p.y) + pix * p.y_ld + p.y_coff;
                if (p.accumulate == 2) v += *yp;
                v = hv_act(v, p.act);      // (one value per pixel: the exact forms, as the VALU head kernels)
                if (p.accumulate == 1) v += *yp;
                *yp = v;
            } else {
                _Float16* yp = reinterpret_cast<_Float16*>(p.y) + pix * p.y_ld + p.y_coff;
                const _Float16* mp = reinterpret_cast<const _Float16*>(p.mul) + pix * p.mul_ld + p.mul_coff;
                // whole quads; the activation / multiplier switches are wave-uniform and taken once per quad, not per element
#pragma unroll
                for (int m = 0; m < CQ; ++m) {
                    if (4 * m >= p.Cout) break;
                    f16x4v old = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f}, m4 = old;
                    if (p.accumulate) old = *reinterpret_cast<const f16x4v*>(yp + 4 * m);
                    if (p.mul) m4 = *reinterpret_cast<const f16x4v*>(mp + 4 * m);
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = acc[m][r] * p.alpha + bv[m][r];
                    if (p.accumulate == 2) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += (float)old[r];
                    }
                    if (p.act == HV_ACT_ELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = hv_act_fast(v[r], HV_ACT_ELU);
                    } else if (p.act != HV_ACT_NONE) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = hv_act_fast(v[r], p.act);
                    }
                    if (p.mul) {
                        if (p.mul_act == HV_ACT_ELU) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] *= (float)m4[r] > 0.f ? 1.f : (float)m4[r] + 1.f;
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] *= hv_act_grad_from_out((float)m4[r], p.mul_act);
                        }
                    }
                    if (p.accumulate == 1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += (float)old[r];
                    }
                    *reinterpret_cast<f16x4v*>(yp + 4 * m) = f16x4v{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                }
            }
        }
    }
}

template <int KS, int CI, int CQ, bool TR, bool Y1, int TH>
static int launch_px(PxK& k, hipStream_t s) {
    constexpr int NKQ = KS * KS * (CI / 4);
    const size_t lds = (size_t)CQ * NKQ * 32 + (size_t)(TH + KS - 1) * (256 + KS - 1) * CI * 2;
    k.segs = hv_cdiv(k.W, 256); k.rblocks = hv_cdiv(k.H, TH); k.tiles = k.B * k.rblocks * k.segs;
    static const int want = getenv("HV_PX_WGS") ? atoi(getenv("HV_PX_WGS")) : 4096;      // tuning knob: workgroups (each walks tiles of TH rows x 256 pixels)
    int blocks = k.tiles < want ? k.tiles : want;
    HV_KNAME("conv_px_kernel<%d, %d, %d, %s, %s, %d>", KS, CI, CQ, TR ? "true" : "false", Y1 ? "true" : "false", TH);
    if (hv_probe_only) return HV_OK;
    hipLaunchKernelGGL((conv_px_kernel<KS, CI, CQ, TR, Y1, TH>), dim3(blocks), dim3(256), lds, s, k);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// hv_conv2d: KH = KW in {3, 5}, stride 1, dilation 1, fp16 operands and fp16 filter rows; the operand's pixel row holds CI = 4 / 8 / 12 / 16 channels (x_ld a
// multiple of 4, the view starting on a multiple of 4), Cout <= 16 stored as fp16 rows of whole quads, or ONE fp32 output channel (the heads)
int hv_conv2d_px(const hv_conv_desc* d, hipStream_t s) {
    // A/B knob, bits: 1 heads forward (13.4 / 9.4 us against 22.0 / 15.3 for 12 / 8 channels at 256^2, bs 16), 2 heads' data gradient (16.6 / 12.8 against 38.7 / 26.7),
    // 4 the 8 <-> 16 channel 3x3 layer (26.9 against 39.9 us data gradient, 27.4 against 38.8 forward).  Measured and not kept (the tiled kernels are as fast or
    // faster there): the 16 -> 8 channel 3x3 layer and its data gradient (28.1 / 29.4 against 25.9 / 26.3 us), the 5x5 stems (30.5 against 30.3 us) and their data
    // gradient (16 -> 4: 34.5 against 34.0 us -- 25 taps x 32 B of window plus as many bytes of filter fragments per pixel: LDS-read bound either way)
    static const int on = getenv("HV_CONV_PX") ? atoi(getenv("HV_CONV_PX")) : 7;
    if (!on || d->precision != HV_F16 || !d->w_f16 || !d->x_f16 || d->KH != d->KW || (d->KH != 3 && d->KH != 5) || d->stride != 1 || d->dil != 1 || d->in_shift ||
        d->w_bstride || d->ch_scale || d->x1 || d->pool2 || d->stats || d->bstats || d->xn_stats)
        return HV_ERR_UNSUPPORTED;
    if (d->Ho != d->H || d->Wo != d->W || 2 * d->pad != d->KH - 1) return HV_ERR_UNSUPPORTED;      // 'same' layers only
    if ((d->x_ld & 3) || (d->x_coff & 3) || ((uintptr_t)d->x & 15) || d->Cin > 16 || d->Cout > 16) return HV_ERR_UNSUPPORTED;
    const int CI = (d->Cin + 3) & ~3;
    if (d->x_coff + CI > d->x_ld) return HV_ERR_UNSUPPORTED;            // (the padded quad must lie inside the pixel row)
    if ((CI & 7) == 0 && ((d->x_ld & 7) || (d->x_coff & 7))) return HV_ERR_UNSUPPORTED;      // 16-byte pieces
    const bool y1 = d->Cout == 1 && !d->y_f16;
    if (!y1 && (!d->y_f16 || (d->Cout & 3) || (d->y_ld & 3) || (d->y_coff & 3) || ((uintptr_t)d->y & 7))) return HV_ERR_UNSUPPORTED;
    if (y1 && d->mul_src) return HV_ERR_UNSUPPORTED;
    if (d->mul_src && (!d->mul_f16 || (d->mul_ld & 3) || (d->mul_coff & 3) || ((uintptr_t)d->mul_src & 7))) return HV_ERR_UNSUPPORTED;
    if (d->accumulate > 2 || (long long)d->B * d->H * d->W * d->x_ld >= (1ll << 30)) return HV_ERR_UNSUPPORTED;
    PxK k;
    k.x = reinterpret_cast<const _Float16*>(d->x); k.w = reinterpret_cast<const _Float16*>(d->w_f16); k.bias = d->bias; k.y = d->y; k.mul = d->mul_src;
    k.B = d->B; k.H = d->H; k.W = d->W; k.x_ld = d->x_ld; k.x_coff = d->x_coff; k.Cin = d->Cin; k.Cout = d->Cout; k.y_ld = d->y_ld; k.y_coff = d->y_coff;
    k.mul_ld = d->mul_ld; k.mul_coff = d->mul_coff; k.mul_act = d->mul_act; k.mul_half = 1; k.y_half = d->y_f16 ? 1 : 0; k.act = d->act; k.accumulate = d->accumulate;
    k.pad = d->pad; k.alpha = d->alpha;
    k.x_bytes = (unsigned)((long long)d->B * d->H * d->W * d->x_ld * 2);
    const bool tr = d->transposed != 0;
    const int cq = y1 ? 1 : d->Cout / 4;
    hv_path_note = 14;
    HV_WUSE(2);
#define PX(KS_, CI_, CQ_, TR_, Y1_) return launch_px<KS_, CI_, CQ_, TR_, Y1_, (CI_ >= 12 ? 2 : 4)>(k, s)
    if (d->KH == 3) {
        if (y1 && !tr && (on & 1)) { if (CI == 12) PX(3, 12, 1, false, true); if (CI == 8) PX(3, 8, 1, false, true); }
        if (!y1 && tr && CI == 4 && (on & 2)) { if (cq == 3) PX(3, 4, 3, true, false); if (cq == 2) PX(3, 4, 2, true, false); if (cq == 4) PX(3, 4, 4, true, false); }
        if (!y1 && CI == 8 && cq == 4 && (on & 4)) { if (tr) PX(3, 8, 4, true, false); PX(3, 8, 4, false, false); }
    }
#undef PX
    return HV_ERR_UNSUPPORTED;
}

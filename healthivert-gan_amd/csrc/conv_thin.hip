// Data gradient of a 3x3 stride-1 convolution with <= 4 gradient channels and <= 16 output channels: the generators' 1-channel heads handing their
// gradient back to the 8 / 12-channel layer below (hv_conv2d, transposed = 1, Cin <= 4 -- the head's output channel padded to 4).
//
// These launches are 1.2 GFLOP and 25-33 MB: pure HBM work that ran through conv_halo2_kernel's ragged 16-channel MFMA tiles at 34-36 us.  Here a
// lane owns one output pixel: the 3x3 neighbourhood of the gradient (nine 8-byte loads, out-of-image -> zero through the buffer range check), the
// filters in LDS as floats (broadcast reads), Cout fp32 accumulators, then the act' multiplier / accumulate forms of the shared epilogue on the
// lane's own 16- or 24-byte channel row and one store.  fp32 arithmetic on the fp16-stored operands (exact products, fp32 sums).
// (Round 5: the step's own head gradients -- 256^2, 4-wide carriers -- go to conv_px_kernel first (conv_px.hip: 21.5 / 16.0 us against 38.7 / 26.7 us here, which is
// bound by its CO x 9 x 4 multiply-adds and as many LDS filter reads per pixel); a 16x16x16-MFMA form with the nine taps as the contraction -- 16 pixels per
// wave step -- reached 22 / 18 us and was dropped for it: ~130 vector instructions per 16 pixels around one MFMA, PMC SQ_INSTS_VALU.)  This kernel keeps
// the shapes conv_px_kernel does not take.
#include "conv_halo.h"

struct ThinK {
    const _Float16* g; const float* w; void* y; const _Float16* mul;
    int B, H, W, g_ld, g_coff, Cin, Cout, w_row;       // w: [Cout][9][w_row / 9] floats (the data-gradient table)
    int y_ld, y_coff, mul_ld, mul_coff, mul_act, accumulate, pad;
    float alpha;
    unsigned g_bytes;
    // logits_dgrad_kernel only: the batch-norm backward sums of the layer whose output gradient it writes (hv_conv_desc.bstats; at most two groups)
    const _Float16* bn_x; const float* bn_stats; float* bstats;
    int bn_x_ld, bn_x_coff, bn_ipg, bn_groups;
};

template <int CO>        // output channels per lane: 8, 12 or 16
__global__ __launch_bounds__(256) void thin_dgrad_kernel(const ThinK p) {
    __shared__ float wl[CO * 9 * 4];
    const int CinP = p.w_row / 9;
    for (int i = threadIdx.x; i < CO * 9 * 4; i += 256) {
        const int co = i / 36, r = i - co * 36, tap = r >> 2, ci = r & 3;
        wl[i] = (co < p.Cout && ci < p.Cin) ? p.w[co * p.w_row + tap * CinP + ci] : 0.f;
    }
    __syncthreads();
    const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long npix = (long long)p.B * p.H * p.W;
    if (pix >= npix) return;
    const int x = (int)(pix % p.W);
    const long long t = pix / p.W;
    const int y = (int)(t % p.H), n = (int)(t / p.H);
    const __amdgpu_buffer_rsrc_t gsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.g), 0, p.g_bytes, 0x00020000);
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    u32x2 gv[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {      // gather form of the transposed conv: input pixel = output pixel + pad - tap offset
        const int hi = y + p.pad - tap / 3, wi = x + p.pad - tap % 3;
        const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        gv[tap] = __builtin_amdgcn_raw_buffer_load_b64(gsrc, ok ? (unsigned)((((long long)n * p.H + hi) * p.W + wi) * p.g_ld + p.g_coff) * 2u : HV_OOB, 0, 0);
    }
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = 0.f;
#pragma unroll 1      // (fully unrolled the compiler hoists all 9 x CO filter reads: 256 registers + scratch)
    for (int tap = 0; tap < 9; ++tap) {
        const f16x4v h = __builtin_bit_cast(f16x4v, gv[tap]);
        const float g0 = (float)h[0], g1 = (float)h[1], g2 = (float)h[2], g3 = (float)h[3];
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            const float4 w4 = *reinterpret_cast<const float4*>(wl + c * 36 + tap * 4);
            acc[c] += g0 * w4.x + g1 * w4.y + g2 * w4.z + g3 * w4.w;
        }
    }
    _Float16* yp = reinterpret_cast<_Float16*>(p.y) + pix * p.y_ld + p.y_coff;
    const _Float16* mp = p.mul ? p.mul + pix * p.mul_ld + p.mul_coff : nullptr;
#pragma unroll
    for (int c0 = 0; c0 < CO; c0 += 4) {
        if (c0 >= p.Cout) break;
        f16x4v o;
        f16x4v m4 = {(_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0}, y4 = m4;
        if (mp) m4 = *reinterpret_cast<const f16x4v*>(mp + c0);
        if (p.accumulate) y4 = *reinterpret_cast<const f16x4v*>(yp + c0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = acc[c0 + r] * p.alpha;
            v = (float)(_Float16)v;                                           // the tile value is rounded to fp16 before the multiplier (as the MFMA kernels' second stage does)
            if (mp) v = (float)(_Float16)(v * hv_act_grad_from_out((float)m4[r], p.mul_act));
            if (p.accumulate) v += (float)y4[r];
            o[r] = (_Float16)v;
        }
        *reinterpret_cast<f16x4v*>(yp + c0) = o;
    }
}

// Data gradient of the PatchGAN logits layer (4x4, stride 1, pad 1; the 1-channel logit gradient, stored with a channel stride of 4, back to 512
// channels): 4 GFLOP, 15.7 MB written + 15.7 MB of act' multiplier read -- the gather kernel took 27 us, eight times per step at the head of every
// discriminator backward.  The 16 taps are the contraction of ONE v_mfma_f32_16x16x16_f16 per (16 pixels, 16 channels, gradient channel): A = the
// filters [channel][tap] (from an LDS copy [gradient channel][channel][tap], transposed once per workgroup), B = the lane's row of the 4x4 window
// (lane = (pixel, filter row): four 2-byte loads); a wave turns 16 pixels into all Cout channels (32 MFMAs per gradient channel at 512), stages
// them in LDS and writes whole channel rows as 16-byte pieces with the act' multiplier / accumulate applied there.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <int NP>        // a workgroup's channel slice CQ = 32 * NP (128, or all of a 32/64-channel layer); blockIdx & (nq - 1) picks the slice
__global__ __launch_bounds__(256) void logits_dgrad_kernel(const ThinK p, const _Float16* __restrict__ wh, int nq, int nq_log) {
    constexpr int CQ = 32 * NP, LDW = CQ * 16, LDS_ROW = CQ + 8, NTL = CQ / 16, PPR = CQ / 8;   // PPR: 16-byte pieces per pixel row of the slice
    __shared__ __attribute__((aligned(16))) _Float16 wl[4 * LDW];        // [gradient channel][channel][tap]
    __shared__ __attribute__((aligned(16))) _Float16 st[4 * 16 * LDS_ROW];   // [wave][16 pixels][CQ + 8]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, gq = lane >> 4;
    const int cb = (blockIdx.x & (nq - 1)) * CQ;                         // first channel of the slice
    {                                                                    // source [channel][tap][4]: one 16-byte item = 2 taps x 4 gradient channels
        f16x8 v[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) v[j] = *reinterpret_cast<const f16x8*>(wh + (long long)cb * 64 + (long long)(threadIdx.x + 256 * j) * 8);
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int i = threadIdx.x + 256 * j, co = i >> 3, t0 = (i & 7) * 2;
#pragma unroll
            for (int c = 0; c < 4; ++c) *reinterpret_cast<f16x2*>(wl + c * LDW + co * 16 + t0) = f16x2{v[j][c], v[j][4 + c]};
        }
    }
    __syncthreads();
    // batch-norm backward sums (hv_conv_desc.bstats): a lane's pieces always carry the same 8 channels (piece lane % PPR of the slice), so it keeps
    // sum g and sum g * x of those channels per normalisation group over its whole grid-stride loop (two multiply-adds per element; the mean / rstd of
    // xhat = (x - mean) * rstd enter once, when the lanes' sums are folded); one part per (group, workgroup of the slice)
    const bool bst = p.bstats != nullptr;       // scalar
    const int pcl = lane % PPR;
    float bs1[2][8], bs2[2][8];
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
        for (int e = 0; e < 8; ++e) bs1[g2][e] = bs2[g2][e] = 0.f;
    const int Hi = p.H - 1, Wi = p.W - 1;                                // gradient (input) size; p.H, p.W = output size
    const int gpr = (p.W + 15) >> 4;                                     // 16-pixel groups per output row
    const int ngroups = p.B * p.H * gpr;
    _Float16* mine = st + wave * 16 * LDS_ROW;
    for (int grp = (blockIdx.x >> nq_log) * 4 + wave; grp < ngroups; grp += (gridDim.x >> nq_log) * 4) {
        const int xg = grp % gpr, t = grp / gpr, y = t % p.H, b = t / p.H;
        const int x = xg * 16 + n;
        // B fragments: k = 4 * kh + kw with kh = gq: g[y + 1 - kh][x + 1 - kw][c]
        const int hi = y + 1 - gq;
        f16x4 g4[4];
#pragma unroll
        for (int kw = 0; kw < 4; ++kw) {
            const int wi = x + 1 - kw;
            g4[kw] = f16x4{0, 0, 0, 0};
            if ((unsigned)hi < (unsigned)Hi && (unsigned)wi < (unsigned)Wi && x < p.W)
                g4[kw] = *reinterpret_cast<const f16x4*>(p.g + (((long long)b * Hi + hi) * Wi + wi) * p.g_ld + p.g_coff);
        }
        // second-stage operands, all in flight before the MFMA section: piece it = lane + 64 j -> pixel q = it / PPR, piece pc = it % PPR
        const long long pix0 = ((long long)b * p.H + y) * p.W + xg * 16;
        f16x8 m8[NP], y8[NP], x8[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int it = lane + 64 * j, q = it / PPR, pc = it % PPR;
            m8[j] = f16x8{0, 0, 0, 0, 0, 0, 0, 0}; y8[j] = m8[j]; x8[j] = m8[j];
            if (xg * 16 + q < p.W) {
                if (p.mul) m8[j] = *reinterpret_cast<const f16x8*>(p.mul + (pix0 + q) * p.mul_ld + p.mul_coff + cb + pc * 8);
                if (p.accumulate) y8[j] = *reinterpret_cast<const f16x8*>(reinterpret_cast<const _Float16*>(p.y) + (pix0 + q) * p.y_ld + p.y_coff + cb + pc * 8);
                if (bst) x8[j] = *reinterpret_cast<const f16x8*>(p.bn_x + (pix0 + q) * p.bn_x_ld + p.bn_x_coff + cb + pc * 8);
            }
        }
        const int bgi = bst ? (b / p.bn_ipg) & 1 : 0;                    // the image's normalisation group (wave-uniform)
        f16x4 bfr[4];
        bool live[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bfr[c] = f16x4{g4[0][c], g4[1][c], g4[2][c], g4[3][c]};
            const bool nz = (float)bfr[c][0] != 0.f || (float)bfr[c][1] != 0.f || (float)bfr[c][2] != 0.f || (float)bfr[c][3] != 0.f;
            live[c] = __builtin_amdgcn_ballot_w64(nz) != 0;             // wave-uniform: an all-zero gradient channel (the padding of a 1-channel head) adds nothing
        }
#pragma unroll
        for (int tl = 0; tl < NTL; ++tl) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (!live[c]) continue;
                const f16x4 a4 = *reinterpret_cast<const f16x4*>(wl + c * LDW + (tl * 16 + n) * 16 + gq * 4);
                acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, bfr[c], acc, 0, 0, 0);
            }
            // D: row (channel) = 4 gq + r, column (pixel) = n
            *reinterpret_cast<f16x4*>(mine + n * LDS_ROW + tl * 16 + gq * 4) =
                f16x4{(_Float16)(acc[0] * p.alpha), (_Float16)(acc[1] * p.alpha), (_Float16)(acc[2] * p.alpha), (_Float16)(acc[3] * p.alpha)};
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                              // lgkmcnt(0): the wave's own LDS writes (no other wave reads them)
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int it = lane + 64 * j, q = it / PPR, pc = it % PPR;
            if (xg * 16 + q >= p.W) continue;
            f16x8 v8 = *reinterpret_cast<const f16x8*>(mine + q * LDS_ROW + pc * 8);
            if (p.mul) {
                {
                    float f0[4] = {(float)m8[j][0], (float)m8[j][1], (float)m8[j][2], (float)m8[j][3]}, f1[4] = {(float)m8[j][4], (float)m8[j][5], (float)m8[j][6], (float)m8[j][7]};
                    hv_act_grad4(f0, p.mul_act); hv_act_grad4(f1, p.mul_act);      // (one switch per quad, not per element)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v8[e] = (_Float16)((float)v8[e] * f0[e]); v8[4 + e] = (_Float16)((float)v8[4 + e] * f1[e]); }
                }
            }
            if (p.accumulate) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v8[e] = (_Float16)((float)v8[e] + (float)y8[j][e]);
            }
            *reinterpret_cast<f16x8*>(reinterpret_cast<_Float16*>(p.y) + (pix0 + q) * p.y_ld + p.y_coff + cb + pc * 8) = v8;
            if (bst) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float gv = (float)v8[e], gx = gv * (float)x8[j][e];
                    if (bgi == 0) { bs1[0][e] += gv; bs2[0][e] += gx; }
                    else { bs1[1][e] += gv; bs2[1][e] += gx; }
                }
            }
        }
    }
    if (bst) {      // fold the 256 / PPR lanes that share a piece, group by group, in a fixed order; row = group * (workgroups per slice) + this workgroup
        // the fold buffer: 256 x 16 floats.  With 128-channel slices the filter copy `wl` has exactly that size and nobody reads it any more
        __shared__ float red_small[NP == 4 ? 1 : 256 * 16];
        float* red = NP == 4 ? reinterpret_cast<float*>(wl) : red_small;
        static_assert(NP != 4 || sizeof(wl) >= 256 * 16 * sizeof(float), "fold buffer aliases the filter copy");
        const int gbw = gridDim.x >> nq_log, wgi = blockIdx.x >> nq_log;
        for (int g2 = 0; g2 < p.bn_groups; ++g2) {
            __syncthreads();
            float* mine2 = red + ((threadIdx.x / PPR) * PPR + pcl) * 16;
#pragma unroll
            for (int e = 0; e < 8; ++e) { mine2[e] = g2 == 0 ? bs1[0][e] : bs1[1][e]; mine2[8 + e] = g2 == 0 ? bs2[0][e] : bs2[1][e]; }
            __syncthreads();
            if (threadIdx.x < PPR * 8) {      // one thread per channel of the slice: sum g, and sum g * xhat = rstd * (sum g x - mean * sum g)
                const int pq = threadIdx.x >> 3, e = threadIdx.x & 7;
                float sg = 0.f, sgx = 0.f;
                for (int r = 0; r < 256 / PPR; ++r) { sg += red[(r * PPR + pq) * 16 + e]; sgx += red[(r * PPR + pq) * 16 + 8 + e]; }
                const int ch = cb + pq * 8 + e;
                const float mean = p.bn_stats[(long long)g2 * 2 * p.Cout + ch], rstd = p.bn_stats[(long long)g2 * 2 * p.Cout + p.Cout + ch];
                float* o = p.bstats + (((long long)g2 * gbw + wgi) * p.Cout + ch) * 2;
                o[0] = sg;
                o[1] = rstd * (sgx - mean * sg);
            }
        }
    }
}

// workgroups per channel slice of a launch (= parts per normalisation group of hv_conv_desc.bstats)
static int logits_gb(int Cout, int groups) {
    const int NP = Cout == 32 ? 1 : Cout == 64 ? 2 : 4, nq = Cout / (32 * NP);
    int gb = (groups + 3) / 4;                                           // 4 pixel groups each, at most ~8 workgroups per CU in all
    if (gb * nq > 2048) gb = 2048 / nq;
    return gb;
}

template <int NP>
static int launch_logits(const ThinK& k, const _Float16* wh, int groups, hipStream_t s) {
    const int nq = k.Cout / (32 * NP);                                   // a power of two (checked by the caller)
    int nq_log = 0;
    while ((1 << nq_log) < nq) ++nq_log;
    const int gb = logits_gb(k.Cout, groups);
    HV_KNAME("logits_dgrad_kernel");
    hipLaunchKernelGGL(logits_dgrad_kernel<NP>, dim3(gb * nq), dim3(256), 0, s, k, wh, nq, nq_log);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// hv_conv2d: transposed, 4x4, stride 1, pad 1, gradient channel stride 4 (<= 4 live channels), Cout % 16 == 0 (<= 512), fp16 views + fp16 filter copy
static bool logits_dgrad_eligible(const hv_conv_desc* d) {
    static const int on = getenv("HV_LOGITS_DGRAD") ? atoi(getenv("HV_LOGITS_DGRAD")) : 1;
    if (!on || !d->transposed || d->KH != 4 || d->KW != 4 || d->stride != 1 || d->pad != 1 || d->dil != 1 || d->in_shift || d->w_bstride || d->ch_scale || !d->w_f16)
        return false;
    if (d->Cin != 4 || d->x_ld != 4 || (d->x_coff & 3) || !d->x_f16 || !d->y_f16 || d->bias || d->act != HV_ACT_NONE || d->accumulate > 1) return false;
    if (d->Cout > 512 || (d->Cout & 31) || (d->Cout & (d->Cout - 1)) || ((uintptr_t)d->w_f16 & 15) || (d->y_ld & 7) || (d->y_coff & 7) || ((uintptr_t)d->y & 15) || ((uintptr_t)d->x & 7) ||
        d->Ho != d->H + 1 || d->Wo != d->W + 1)
        return false;
    if (d->mul_src && (!d->mul_f16 || (d->mul_ld & 7) || (d->mul_coff & 7) || ((uintptr_t)d->mul_src & 15))) return false;
    if ((long long)d->B * d->Ho * d->Wo * d->y_ld >= (1ll << 31)) return false;
    return true;
}

// parts of hv_conv_desc.bstats this kernel writes: (normalisation groups, at most two) x (workgroups per channel slice); 0 = not served
size_t hv_conv2d_logits_bstats_parts(const hv_conv_desc* d) {
    if (!logits_dgrad_eligible(d) || d->accumulate) return 0;
    const int G = d->bn_groups > 0 ? d->bn_groups : 1;
    if (G > 2 || d->B % G || !d->bn_x || !d->bn_stats || (d->bn_x_ld & 7) || (d->bn_x_coff & 7) || ((uintptr_t)d->bn_x & 15)) return 0;
    if ((long long)d->B * d->Ho * d->Wo * d->bn_x_ld >= (1ll << 31)) return 0;
    return (size_t)G * logits_gb(d->Cout, d->B * d->Ho * ((d->Wo + 15) / 16));
}

int hv_conv2d_logits_dgrad(const hv_conv_desc* d, hipStream_t s) {
    if (!logits_dgrad_eligible(d)) return HV_ERR_UNSUPPORTED;
    ThinK k;
    HV_WUSE(1 | 2);
    k.g = reinterpret_cast<const _Float16*>(d->x); k.w = d->w; k.y = d->y; k.mul = reinterpret_cast<const _Float16*>(d->mul_src);
    k.B = d->B; k.H = d->Ho; k.W = d->Wo; k.g_ld = d->x_ld; k.g_coff = d->x_coff; k.Cin = d->Cin; k.Cout = d->Cout; k.w_row = 16 * d->Cin;
    k.y_ld = d->y_ld; k.y_coff = d->y_coff; k.mul_ld = d->mul_ld; k.mul_coff = d->mul_coff; k.mul_act = d->mul_act; k.accumulate = d->accumulate; k.pad = d->pad;
    k.alpha = d->alpha; k.g_bytes = 0;
    const int groups = d->B * d->Ho * ((d->Wo + 15) / 16);
    k.bn_x = nullptr; k.bn_stats = nullptr; k.bstats = nullptr; k.bn_x_ld = k.bn_x_coff = 0; k.bn_ipg = 1; k.bn_groups = 0;
    if (d->bstats) {
        if (!hv_conv2d_logits_bstats_parts(d)) return HV_ERR_UNSUPPORTED;
        k.bn_x = reinterpret_cast<const _Float16*>(d->bn_x); k.bn_stats = d->bn_stats; k.bstats = d->bstats; k.bn_x_ld = d->bn_x_ld; k.bn_x_coff = d->bn_x_coff;
        k.bn_groups = d->bn_groups > 0 ? d->bn_groups : 1; k.bn_ipg = d->B / k.bn_groups;
    }
    const _Float16* wh = reinterpret_cast<const _Float16*>(d->w_f16);
    hv_path_note = 9;
    if (d->Cout == 32) return launch_logits<1>(k, wh, groups, s);
    if (d->Cout == 64) return launch_logits<2>(k, wh, groups, s);
    return launch_logits<4>(k, wh, groups, s);                           // 128-channel slices
}

// (Measured and not kept, round 3: the same lane-per-pixel form for the PatchGAN logits layer's data gradient (4x4, 1 -> 512 channels; fp16 filters in
// LDS as [tap][ci][Cout], a lane = 8 channels of 4 pixels, persistent workgroups): step 9.04 -> 9.15 ms against the gather kernel's 26.7 us launches --
// 64 LDS filter reads per lane and pixel group cost more than the gather kernel's MFMA tiles save.)
// hv_conv2d: transposed, 3x3, stride 1, dilation 1, Cin <= 4 (channel stride 4), Cout in {8, 12, 16}, fp16 views, no bias / activation of its own
int hv_conv2d_thin_dgrad(const hv_conv_desc* d, hipStream_t s) {
    static const int on = getenv("HV_THIN_DGRAD") ? atoi(getenv("HV_THIN_DGRAD")) : 1;
    if (!on || !d->transposed || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->dil != 1 || d->in_shift || d->w_bstride || d->ch_scale) return HV_ERR_UNSUPPORTED;
    if (d->Cin > 4 || (d->x_ld & 3) || (d->x_coff & 3) || !d->x_f16 || !d->y_f16 || d->bias || d->act != HV_ACT_NONE || d->accumulate > 1) return HV_ERR_UNSUPPORTED;
    if (d->Cout > 16 || (d->Cout & 3) || (d->y_ld & 3) || (d->y_coff & 3) || ((uintptr_t)d->y & 7) || ((uintptr_t)d->x & 7) || d->Ho != d->H || d->Wo != d->W) return HV_ERR_UNSUPPORTED;
    if (d->mul_src && (!d->mul_f16 || (d->mul_ld & 3) || (d->mul_coff & 3) || ((uintptr_t)d->mul_src & 7))) return HV_ERR_UNSUPPORTED;
    const long long npix = (long long)d->B * d->H * d->W;
    if (npix * d->x_ld >= (1ll << 30) || npix >= (1ll << 31) - 256) return HV_ERR_UNSUPPORTED;
    ThinK k;
    HV_WUSE(1 | 2);
    k.g = reinterpret_cast<const _Float16*>(d->x); k.w = d->w; k.y = d->y; k.mul = reinterpret_cast<const _Float16*>(d->mul_src);
    k.B = d->B; k.H = d->H; k.W = d->W; k.g_ld = d->x_ld; k.g_coff = d->x_coff; k.Cin = d->Cin; k.Cout = d->Cout; k.w_row = 9 * d->Cin;
    k.y_ld = d->y_ld; k.y_coff = d->y_coff; k.mul_ld = d->mul_ld; k.mul_coff = d->mul_coff; k.mul_act = d->mul_act; k.accumulate = d->accumulate; k.pad = d->pad;
    k.alpha = d->alpha;
    k.g_bytes = (unsigned)(npix * d->x_ld * 2);
    const dim3 grid((unsigned)((npix + 255) / 256));
    hv_path_note = 9;
    if (d->Cout <= 8) { HV_KNAME("thin_dgrad_kernel<8>"); hipLaunchKernelGGL(thin_dgrad_kernel<8>, grid, dim3(256), 0, s, k); }
    else if (d->Cout <= 12) { HV_KNAME("thin_dgrad_kernel<12>"); hipLaunchKernelGGL(thin_dgrad_kernel<12>, grid, dim3(256), 0, s, k); }
    else { HV_KNAME("thin_dgrad_kernel<16>"); hipLaunchKernelGGL(thin_dgrad_kernel<16>, grid, dim3(256), 0, s, k); }
    HV_LAUNCH_CHECK();
    return HV_OK;
}

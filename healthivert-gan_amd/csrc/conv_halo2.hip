// Halo-tiled implicit-GEMM convolution, second form: weights go straight from L2 into MFMA operand registers.
//
// conv_halo_kernel (conv_halo.hip) streams the filter taps through a double-buffered LDS tile: one barrier and one
// LDS round trip per TG taps, ~50 KB of LDS that caps a CU at two workgroups.  The layers of this model are short
// (tens of microseconds), so those barriers and the low occupancy cost more than the MFMAs.  Here
//   * the input patch is the only thing in LDS (fp16 [patch pixel][CK channels], double-buffered over channel chunks):
//     ONE barrier per 32-channel chunk;
//   * every wave owns BN/WN output channels and fetches their filter rows itself: lane (row = lane & 15, k-group =
//     lane >> 4) loads the 16 B of  w[cout0 + row][tap][c0 + 8*kgroup ..]  it needs as MFMA A-operand, a ring of D
//     taps ahead of the MFMAs (weights are L2-resident; the ring index is static because the tap count is a template
//     parameter and D divides it);
//   * staging is branch-free: offsets are computed once, taps / chunks enter as scalar buffer offsets, out-of-image
//     and out-of-range lanes carry an offset beyond the descriptor's range (the load returns zeros).
// Shapes: the same as conv_halo_kernel with a uniform tap count of 4 (4x4 stride-2 data gradient classes), 9 (3x3) or
// 16 (4x4); everything else stays on conv_halo_kernel.
#include <stdlib.h>
#include <type_traits>

#include "conv_halo.h"

template <int CK> struct WLoad;
template <> struct WLoad<32> {
    static __device__ __forceinline__ f16x8 ld(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
        return __builtin_bit_cast(f16x8, v);
    }
};
template <> struct WLoad<16> {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ f16x4v ld(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
        return __builtin_bit_cast(f16x4v, v);
    }
};

// SPAN = filter extent in input pixels of one class (2: stride-2 data-gradient class of a 4x4 filter, 3, 4)
// CK = channels per staged chunk: 16 (one 16x16x16 MFMA step), 32 (one 16x16x32 step) or 64 (two steps: a lane's two
// 16-B weight loads then consume a whole 128-B line of its filter row, and barriers halve)
template <int TH, int TW, int BN, int WM, int WN, int CK, int BSTEP, int SPAN, int D, bool XH>
__global__ __launch_bounds__(256, 2) void conv_halo2_kernel(const HaloK p) {   // two workgroups per CU: at most 256 registers per lane
    typedef HvSt<XH> XS;                         // storage of the input tensor: fp32 (converted when staged) or fp16 (staged as it is)
    constexpr int NTAPS = SPAN * SPAN;
    constexpr int BM = TH * TW;
    constexpr int FK = CK == 16 ? 16 : 32;                          // channels per MFMA step
    constexpr int KS = CK / FK;                                     // MFMA steps per chunk
    // patch row stride (halfs): 96 B (CK 32) and 160 B (CK 64) rows are conflict-free for unit-step b128 reads; 80 B for stride 2
    constexpr int LDP = CK + ((CK == 32 && BSTEP == 1) ? 16 : 8);     // 96-B / 48-B / 144-B pixel rows: 16 lanes x 16 B land on distinct banks
    constexpr int MT = BM / WM / 16, NT = BN / WN / 16, GX = TW / 16;
    constexpr int PHM = (TH - 1) * BSTEP + SPAN, PWM = (TW - 1) * BSTEP + SPAN;   // patch extent
    // staging items: 4 channels of a patch pixel (16 B of fp32 / 8 B of fp16), or 8 channels (16 B) of an fp16 tensor with whole 32-channel chunks
    constexpr int IW = (XH && CK >= 32) ? 8 : 4;
    constexpr int PV = CK / IW;                                                   // items per patch pixel and chunk
    constexpr int PMAX = (PHM * PWM * PV + 255) / 256;
    static_assert(WM * WN == 4 && MT >= 1 && NT >= 1 && TW % 16 == 0 && NTAPS % D == 0, "bad tile");
    typedef typename HFrag<FK>::V V;
    // stride-2 patches are 4x the output tile: ONE LDS buffer (the next chunk waits in the prefetch registers -- half as many since the
    // tensors are stored as fp16 -- and is written between two barriers), so that two or three workgroups still fit a CU
    constexpr bool DBUF = BSTEP == 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* patch = reinterpret_cast<_Float16*>(smem);                          // [2 or 1][PHM*PWM][LDP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int ci = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < p.ncls && (int)blockIdx.x >= p.cls[i].t0) ci = i;
    const HaloCls& C = p.cls[ci];
    const int PW = C.PW, npatch = C.PH * C.PW;
    int t = (int)blockIdx.x - C.t0;
    // dilation: residue class (ry, rx) of this workgroup's sub-grid (see hv_conv2d_halo); all classes have the same tiling
    int ry = 0, rx = 0;
    if (p.dil > 1) {
        const int per = C.tiles * p.B, rid = t / per;
        t -= rid * per;
        ry = rid / p.dil; rx = rid - ry * p.dil;
    }
    const int n_img = t / C.tiles;
    t -= n_img * C.tiles;
    const int tile_y = t / C.tiles_x, tile_x = t - tile_y * C.tiles_x;
    const int i0 = tile_y * TH, j0 = tile_x * TW;
    const int n_base = blockIdx.y * BN;
    const int h0 = i0 * p.bstep + p.boff + C.dh_min, w0 = j0 * p.bstep + p.boff + C.dw_min;
    const int wm = wave / WN, wn = wave % WN;

    // scalar tap table: LDS offset of the tap inside the patch, filter index
    int toff[NTAPS], widx[NTAPS];
#pragma unroll
    for (int q = 0; q < NTAPS; ++q) {
        const uint32_t e = C.taps[q];
        toff[q] = ((int)(e & 0xff) * PW + (int)((e >> 8) & 0xff)) * LDP;
        widx[q] = (int)(e >> 16);
    }
    int poff[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int g = wm * MT + m, ty = g / GX, tx = (g % GX) * 16 + (lane & 15);
        poff[m] = (ty * BSTEP * PW + tx * BSTEP) * LDP;
    }
    f32x4 acc[NT][MT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // this lane's bias values, fetched now: read in the epilogue they were four dependent global loads per fragment at the kernel's very end
    // (only where registers are plentiful: with 16 accumulator fragments, or the stride-2 patch prefetch, the extra live registers spilled)
    constexpr bool BPF = MT * NT <= 8 && BSTEP == 1;
    f32x4 bias_r[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int ch0 = n_base + wn * (BN / WN) + n * 16 + (lane >> 4) * 4;
        bias_r[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (BPF && p.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (ch0 + r < p.Cout) bias_r[n][r] = p.bias[ch0 + r];
        }
    }
    // filters: plain rows [Cout][taps][Cin], or (p.wt) MFMA-fragment order: a fragment = 16 rows x FK channels is one contiguous piece, lane l
    // owns bytes [l * FK/2, (l+1) * FK/2) of it (include/hvgan.h, hv_weight_tile_f16).  Plain rows make a wave's fragment load 16 rows x 64 B that
    // lie a whole filter row (8 KB at 4x4x256) apart -- 16 half-used cache lines on ONE L2 channel; measured 93 -> 76 us on the 256 -> 512 layer
    // (the launcher has put the tiled table in p.w / p.w_bytes: a select between two pointers here lands the descriptor in vector registers
    // and every filter load in a waterfall loop)
    const bool tiledw = p.wt != nullptr;                                  // scalar
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.w), 0, p.w_bytes, 0x00020000);
    const int wsc = tiledw ? 32 : 2;                                      // bytes per (tap, channel) step: a 16-row fragment column vs one row
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    unsigned wvo[NT];            // byte offset of this lane's filter row / k-group, without tap and chunk
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int row = n_base + wn * (BN / WN) + n * 16 + (lane & 15);
        wvo[n] = row < p.Cout ? (unsigned)((row * p.w_row + (lane >> 4) * (FK / 4)) * 2) : HV_OOB;
        if (tiledw) wvo[n] = row < ((p.Cout + 15) & ~15) ? (unsigned)((row >> 4) * (16 * p.w_row * 2) + lane * (FK / 2)) : HV_OOB;
    }
    typedef typename std::conditional<IW == 8, u32x4, typename XS::R>::type PR;
    PR preg[PMAX];
    unsigned pvo[PMAX];          // byte offset of (patch pixel, channel group) in x, HV_OOB outside the image / patch
    int plo[PMAX];               // LDS offset (halfs), -1 = no element
    const unsigned xbase = (unsigned)(n_img * p.img_stride + p.x_coff) * XS::B;
#pragma unroll
    for (int i = 0; i < PMAX; ++i) {
        const int e = tid + i * 256;
        const int c4 = e % PV, pix = e / PV;
        const int py = pix / PW, px = pix - py * PW;
        const int hi = h0 + py, wi = w0 + px;
        const bool in = e < npatch * PV;
        plo[i] = in ? pix * LDP + c4 * IW : -1;
        pvo[i] = (in && (unsigned)hi < (unsigned)p.Hl && (unsigned)wi < (unsigned)p.Wl)
                     ? xbase + (unsigned)((((ry + hi * p.dil) >> p.in_shift) * p.Wp + ((rx + wi * p.dil) >> p.in_shift)) * p.x_ld + c4 * IW) * XS::B : HV_OOB;
    }
    // ragged channel counts (Cin % CK != 0, CK == 16 only: a lane's 4 channels are all inside or all outside): lanes beyond Cin
    // read zeros through the range check, for the patch and for the filter rows alike
    const bool ragged = (p.Cin % CK) != 0;
    auto ppref = [&](int c0) __attribute__((always_inline)) {
        if constexpr (IW == 8) {
#pragma unroll
            for (int i = 0; i < PMAX; ++i) preg[i] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, pvo[i], c0 * 2, 0);
        } else if (ragged) {
#pragma unroll
            for (int i = 0; i < PMAX; ++i) {
                const int c4 = (tid + i * 256) % PV;
                preg[i] = XS::ld(xsrc, c0 + c4 * 4 < p.Cin ? pvo[i] : HV_OOB, c0 * (int)XS::B);
            }
        } else {
#pragma unroll
            for (int i = 0; i < PMAX; ++i) preg[i] = XS::ld(xsrc, pvo[i], c0 * (int)XS::B);
        }
    };
    const int kgc = (lane >> 4) * (FK / 4);          // first channel of this lane's k-group inside an MFMA step
    auto wld = [&](int n, int widx_, int c0, int ks) __attribute__((always_inline)) {
        const unsigned vo = (ragged && c0 + ks * FK + kgc >= p.Cin) ? HV_OOB : wvo[n];
        return WLoad<FK>::ld(wsrc, vo, (widx_ * p.Cin + c0 + ks * FK) * wsc);
    };
    auto pflush = [&](_Float16* dst) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PMAX; ++i) {
            if (plo[i] < 0) continue;
            if constexpr (IW == 8) *reinterpret_cast<u32x4*>(dst + plo[i]) = preg[i];
            else *reinterpret_cast<f16x4v*>(dst + plo[i]) = XS::h4(preg[i]);
        }
    };

    const int nchunks = (p.Cin + CK - 1) / CK;
    V wf[D][NT][KS];
    ppref(0);
#pragma unroll
    for (int j = 0; j < D; ++j)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wf[j][n][ks] = wld(n, widx[j], 0, ks);
    pflush(patch);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const _Float16* pb = patch + (DBUF ? (c & 1) : 0) * (PHM * PWM * LDP);
        if (c + 1 < nchunks) ppref((c + 1) * CK);     // next chunk's patch rides behind this chunk's MFMAs
        // The B-fragments of tap q+1 are read from LDS while the MFMAs of tap q run (two register sets): with the reads issued
        // right before their use a wave waited ~100 cycles per pair of MFMAs (PMC: 64 % of the wave cycles in s_waitcnt).
        // (only where the second register set is affordable: up to 8 fragments per tap)
        constexpr bool DB = KS * MT <= 8 && BSTEP == 1;      // (stride 2: the patch prefetch registers take the room of the second fragment set)
        V xf[DB ? 2 : 1][KS][MT];
        if (DB) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int m = 0; m < MT; ++m) xf[0][ks][m] = HFrag<FK>::ld(pb + poff[m] + toff[0] + ks * FK, lane);
        }
#pragma unroll
        for (int q = 0; q < NTAPS; ++q) {
            const int cur = DB ? (q & 1) : 0;
            if (DB ? (q + 1 < NTAPS) : true) {
                const int qq = DB ? q + 1 : q;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int m = 0; m < MT; ++m) xf[DB ? ((q + 1) & 1) : 0][ks][m] = HFrag<FK>::ld(pb + poff[m] + toff[qq] + ks * FK, lane);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[n][m] = HFrag<FK>::mma(wf[q % D][n][ks], xf[cur][ks][m], acc[n][m]);
            // keep the next tap's LDS reads ahead of this tap's MFMAs in the instruction stream (the scheduler would sink them)
            if (DB && q + 1 < NTAPS) {
#pragma unroll
                for (int i = 0; i < KS * MT; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // one ds_read
                    __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);      // NT MFMAs
                }
            }
            // refill the ring slot with the tap D ahead (possibly the next chunk's)
            const int qn = (q + D) % NTAPS;
            const int cn = (q + D >= NTAPS) ? c + 1 : c;
            if (cn < nchunks) {
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) wf[q % D][n][ks] = wld(n, widx[qn], cn * CK, ks);
            }
        }
        if (!DBUF) __syncthreads();      // every wave is done with this chunk's patch
        if (c + 1 < nchunks) pflush(patch + (DBUF ? ((c + 1) & 1) : 0) * (PHM * PWM * LDP));
        __syncthreads();
    }

    // ---- epilogue (same contract as conv_halo_kernel)
    // (alpha == 1, every biased layer of the networks: the prefetched bias is added to the accumulators here, exactly what the shared epilogue code
    // would compute from the pointer; other alphas keep the pointer)
    const bool bias_folded = BPF && p.bias && p.alpha == 1.f;
    const HvEpi epi = {p.alpha, p.act, p.accumulate, p.vec_store, p.Cout, bias_folded ? nullptr : p.bias, nullptr, p.mul_act, p.mul_vec, p.y_half, p.mul_half};
    if (bias_folded) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[n][m][r] += bias_r[n][r];
    }
    if (p.ep16) {
        // fp16 output (assigned, or added to y: accumulate == 1): the tile goes through LDS (the patch buffers are free: the main loop ends behind a barrier) and leaves as
        // 16-byte pieces, a pixel's channel row contiguous.  A lane's own values are 4 channels = 8 bytes of one pixel, so the direct stores below
        // write 32-byte fragments of every 128-byte pixel row from four different waves at four different times: timing-only builds put the
        // direct epilogue at 7.4 of the 17.1 us of a 64 -> 64 channel 3x3 layer at 64x64, bs 16 (main loop 4.4 us).
        constexpr int LDO = BN + 8;                                   // halfs per pixel row in LDS: 16-byte pieces, rows 16 bytes apart in banks
        _Float16* ot = patch;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int g = wm * MT + m, ty = g / GX, tx = (g % GX) * 16 + (lane & 15);
            const int i = i0 + ty, j = j0 + tx;
            const bool inside = i < C.Hc && j < C.Wc;
            const int ho = C.ph + ry + i * p.ostep, wo = C.pw + rx + j * p.ostep;
            const long long opix = (long long)(n_img * p.Ho + ho) * p.Wo + wo;
            // (ep16 == 2: the act' multiplier is applied in the second stage from 16-byte loads)
            const void* mp = (p.mul_src && inside && p.ep16 == 1) ? hv_eptr(p.mul_src, opix * p.mul_ld + p.mul_coff, p.mul_half) : nullptr;
#pragma unroll
            for (int nn = 0; nn < NT; ++nn) {
                const int cl = wn * (BN / WN) + nn * 16 + (lane >> 4) * 4;
                const f32x4 v = hv_conv_value4<true>(epi, acc[nn][m], n_base + cl, mp);
                *reinterpret_cast<f16x4v*>(ot + (g * 16 + (lane & 15)) * LDO + cl) = (f16x4v){(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            }
        }
        __syncthreads();
        constexpr int PIECES = BN / 8;
        _Float16* yb = reinterpret_cast<_Float16*>(p.y);
        for (int it = tid; it < TH * TW * PIECES; it += 256) {
            const int q = it / PIECES, pc = it - q * PIECES;
            const int g = q >> 4, ty = g / GX, tx = (g % GX) * 16 + (q & 15);
            const int i = i0 + ty, j = j0 + tx, ch = n_base + pc * 8;
            if (i >= C.Hc || j >= C.Wc || ch >= p.Cout) continue;                  // Cout % 8 == 0 (host): a piece is wholly inside or outside
            const int ho = C.ph + ry + i * p.ostep, wo = C.pw + rx + j * p.ostep;
            const long long opix = (long long)(n_img * p.Ho + ho) * p.Wo + wo;
            u32x4 o = *reinterpret_cast<const u32x4*>(ot + q * LDO + pc * 8);
            if (p.ep16 == 2) {      // v * act'(m), m = the producer's fp16 output at the same pixel / channels (the tile value was rounded to fp16 once before)
                const f16x8 m8 = *reinterpret_cast<const f16x8*>(reinterpret_cast<const _Float16*>(p.mul_src) + opix * p.mul_ld + p.mul_coff + ch);
                f16x8 v8 = __builtin_bit_cast(f16x8, o);
                {
                    float f0[4] = {(float)m8[0], (float)m8[1], (float)m8[2], (float)m8[3]}, f1[4] = {(float)m8[4], (float)m8[5], (float)m8[6], (float)m8[7]};
                    hv_act_grad4(f0, p.mul_act); hv_act_grad4(f1, p.mul_act);      // (one switch per quad, not per element)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v8[e] = (_Float16)((float)v8[e] * f0[e]); v8[4 + e] = (_Float16)((float)v8[4 + e] * f1[e]); }
                }
                o = __builtin_bit_cast(u32x4, v8);
            }
            if (p.accumulate) {     // y += v (a gradient buffer's later writers)
                const f16x8 y8 = *reinterpret_cast<const f16x8*>(yb + opix * p.y_ld + p.y_coff + ch);
                f16x8 v8 = __builtin_bit_cast(f16x8, o);
#pragma unroll
                for (int e = 0; e < 8; ++e) v8[e] = (_Float16)((float)v8[e] + (float)y8[e]);
                o = __builtin_bit_cast(u32x4, v8);
            }
            *reinterpret_cast<u32x4*>(yb + opix * p.y_ld + p.y_coff + ch) = o;
        }
        return;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int g = wm * MT + m, ty = g / GX, tx = (g % GX) * 16 + (lane & 15);
        const int i = i0 + ty, j = j0 + tx;
        if (i >= C.Hc || j >= C.Wc) continue;
        const int ho = C.ph + ry + i * p.ostep, wo = C.pw + rx + j * p.ostep;
        const long long opix = (long long)(n_img * p.Ho + ho) * p.Wo + wo;
        void* yp = hv_eptr(p.y, opix * p.y_ld + p.y_coff, p.y_half);
        const void* mp = p.mul_src ? hv_eptr(p.mul_src, opix * p.mul_ld + p.mul_coff, p.mul_half) : nullptr;
#pragma unroll
        for (int nn = 0; nn < NT; ++nn) {
            const int ch0 = n_base + wn * (BN / WN) + nn * 16 + (lane >> 4) * 4;
            hv_conv_epilogue4<true>(epi, acc[nn][m], ch0, yp, mp);
        }
    }
}

template <int TH, int TW, int BN, int WM, int WN, int CK, int BSTEP, int SPAN, int D>
static int launch2(HaloK& k, hipStream_t s, int th0 = 8, int tw0 = TW) {
    constexpr int LDP = CK + ((CK == 32 && BSTEP == 1) ? 16 : 8);     // 96-B / 48-B / 144-B pixel rows: 16 lanes x 16 B land on distinct banks
    constexpr int PHM = (TH - 1) * BSTEP + SPAN, PWM = (TW - 1) * BSTEP + SPAN;
    int tiles = 0;
    for (int c = 0; c < k.ncls; ++c) {
        HaloCls& C = k.cls[c];
        // the caller sized the patch for its th0 x tw0 tile: re-size it for this instantiation's tile
        const int ph = C.PH - (th0 - TH) * k.bstep, pw = C.PW - (tw0 - TW) * k.bstep;
        if (ph > PHM || pw > PWM || C.ntaps != SPAN * SPAN) return HV_ERR_UNSUPPORTED;
        C.tiles_x = hv_cdiv(C.Wc, TW);
        C.tiles = C.tiles_x * hv_cdiv(C.Hc, TH);
        C.t0 = tiles;
        tiles += C.tiles * k.B;
    }
    for (int c = 0; c < k.ncls; ++c) {
        k.cls[c].PH -= (th0 - TH) * k.bstep;
        k.cls[c].PW -= (tw0 - TW) * k.bstep;
    }
    // the tiled filter table is cut in fragments of 32 channels when Cin % 32 == 0, else 16: usable only by the matching instantiation
    if (k.wt && (k.Cin % CK != 0 || (k.Cin % 32 == 0 ? 32 : 16) != (CK == 16 ? 16 : 32))) k.wt = nullptr;
    const size_t lds = (size_t)(BSTEP == 1 ? 2 : 1) * PHM * PWM * LDP * sizeof(_Float16);
    auto kern = conv_halo2_kernel<TH, TW, BN, WM, WN, CK, BSTEP, SPAN, D, true>;      // fp16 storage (hv_conv2d_halo refuses fp32 inputs)
    static bool raised[2] = {false, false};   // per instantiation: raise the dynamic-LDS cap once (not a stream operation)
    if (lds > 48 * 1024 && !raised[k.x_half]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return -1000 - (int)e;
        raised[k.x_half] = true;
    }
    // (an XCD-aware 1-D launch that runs a pixel tile's channel blocks back to back on one XCD -- the input patch fetched into one L2 -- measured
    // 98.9 vs 100.5 us on the 256 -> 512 layer and no step-level gain: this kernel is not bound by the patch fetch.  Not kept.)
    dim3 grid(tiles * (k.dil > 1 ? k.dil * k.dil : 1), hv_cdiv(k.Cout, BN));
    hv_path_note = 3;
    HV_KNAME("conv_halo2_kernel<%d, %d, %d, %d, %d, %d, %d, %d, %d, %s>", TH, TW, BN, WM, WN, CK, BSTEP, SPAN, D, k.x_half ? "true" : "false");
    HaloK kk = k;       // the kernel's view: the tiled table IS its filter table (k itself stays as it is for a fallback kernel)
    {   // coalesced fp16 epilogue through LDS (HV_HALO2_EP16=0: direct 8-byte stores)
        static const int ep16 = getenv("HV_HALO2_EP16") ? atoi(getenv("HV_HALO2_EP16")) : 1;
        kk.ep16 = (ep16 && kk.y_half && kk.accumulate <= 1 && !(kk.Cout & 7) && !(kk.y_ld & 7) && !(kk.y_coff & 7) && !((uintptr_t)kk.y & 15) &&
                   lds >= (size_t)TH * TW * (BN + 8) * sizeof(_Float16)) ? 1 : 0;
        // the act' multiplier read as 16-byte pieces in the second stage (a lane's own 8-byte loads are 32-byte fragments of the rows, like its stores)
        if (kk.ep16 && kk.mul_src && kk.mul_half && !(kk.mul_ld & 7) && !(kk.mul_coff & 7) && !((uintptr_t)kk.mul_src & 15)) kk.ep16 = 2;
    }
    if (kk.wt) { kk.w = kk.wt; kk.w_bytes = kk.wt_bytes; }
    HV_WUSE(kk.wt ? 4 : 2);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, kk);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

int hv_halo2_launch(HaloK& k, int TW, int KH, int KW, int maxpatch, hipStream_t s) {
    (void)maxpatch;
    if (k.Cin % 4) return HV_ERR_UNSUPPORTED;
    const bool whole16 = k.Cin % 16 == 0;
    int ntaps = k.cls[0].ntaps;
    for (int c = 1; c < k.ncls; ++c)
        if (k.cls[c].ntaps != ntaps) return HV_ERR_UNSUPPORTED;
    // PatchGAN layers: 4x4 filters, Cout >= 128
    // 5x5 stems of the generators (4 input channels) and their data gradient (4 output channels): Cout <= 16, 256x256 maps
    if (ntaps == 25 && KH == 5 && KW == 5 && k.bstep == 1 && TW == 32 && k.Cout <= 16) return launch2<8, 32, 16, 4, 1, 16, 1, 5, 5>(k, s);
    if (ntaps == 25 && KH == 5 && KW == 5 && k.bstep == 1 && TW == 16 && k.Cout <= 16) return launch2<8, 16, 16, 4, 1, 16, 1, 5, 5>(k, s);
    if (ntaps == 16 && KH == 4 && KW == 4 && TW == 16 && k.Cout > 64 && k.bstep == 1 && k.Cin % 32 == 0) {
        // 64-channel blocks when 128-channel blocks would leave a CU with a single workgroup (512 -> 256 data gradient: 102 vs 112 us)
        const long long wgs128 = (long long)k.B * hv_cdiv(k.cls[0].Hc, 8) * hv_cdiv(k.cls[0].Wc, 16) * hv_cdiv(k.Cout, 128);
        // weight-ring depth (taps in flight): vmcnt retires in order, so the first wait on a filter row issued AFTER the next chunk's patch
        // prefetch also waits for that prefetch (an HBM round trip); a deeper ring moves that wait further behind the prefetch.  HV_HALO2_RING
        // (alone a ring of 8 is 3-4 % faster than 4; inside the step, three discriminators in flight: 11.47 vs 11.54 ms -- default stays 4)
        static const int ring = getenv("HV_HALO2_RING") ? atoi(getenv("HV_HALO2_RING")) : 4;
        // wave arrangement (HV_HALO2_WM: bit 0 -> 64-channel blocks as 2 x 2 waves, bit 1 -> 128-channel blocks as 2 x 2).  With 1 x 4 waves every
        // wave reads the whole 128-pixel patch for its 16 channels: at 64-channel blocks that is 8 LDS fragment reads per 8 MFMAs, the LDS peak.
        // Measured with fragment-ordered filters (256 -> 512 forward @32^2 / 512 -> 256 data gradient @31^2): ring 4: 74.9 / 82.2 us (2 x 2; 85.7 as
        // 1 x 4), ring 8: 72.0 / 79.9 us, ring 16 (64-channel blocks, 1 x 4): 78.0 us; 128-channel blocks as 2 x 2: no difference
        static const int wm = getenv("HV_HALO2_WM") ? atoi(getenv("HV_HALO2_WM")) : 1;
        // (4x16-pixel tiles for these layers: 75.6 -> 87.3 us forward, 80.0 -> 102.9 us data gradient -- twice the filter fetches per MFMA.  Not kept)
        static const int thr128 = getenv("HV_HALO2_THR128") ? atoi(getenv("HV_HALO2_THR128")) : 512;     // (256: 128-channel blocks for the 512 -> 256 data gradient: alone 89 -> 97 us, step 10.71 vs 10.69 ms)
        if (wgs128 < thr128) {
            if (wm & 1) return ring >= 8 ? launch2<8, 16, 64, 2, 2, 32, 1, 4, 8>(k, s) : launch2<8, 16, 64, 2, 2, 32, 1, 4, 4>(k, s);
            if (ring == 16) return launch2<8, 16, 64, 1, 4, 32, 1, 4, 16>(k, s);
            return ring == 8 ? launch2<8, 16, 64, 1, 4, 32, 1, 4, 8>(k, s) : launch2<8, 16, 64, 1, 4, 32, 1, 4, 4>(k, s);
        }
        if (wm & 2) return launch2<8, 16, 128, 2, 2, 32, 1, 4, 4>(k, s);
        return ring >= 8 ? launch2<8, 16, 128, 1, 4, 32, 1, 4, 8>(k, s) : launch2<8, 16, 128, 1, 4, 32, 1, 4, 4>(k, s);
    }
    // 4x4 stride-2 forward with a single-buffered patch (see the kernel).  Measured alone, same device: 128 -> 256 @64^2 41.8 -> 34.9 us,
    // 64 -> 128 @128^2 47.5 -> 49.3 us: taken from 128 input channels (HV_HALO2_S2F: 0 never, 2 always)
    static const int s2f = getenv("HV_HALO2_S2F") ? atoi(getenv("HV_HALO2_S2F")) : 1;
    if (s2f && ntaps == 16 && KH == 4 && KW == 4 && TW == 16 && k.bstep == 2 && k.Cin % 32 == 0 && k.Cout >= 64 && (k.Cin >= 128 || s2f == 2))
        return launch2<8, 16, 64, 1, 4, 32, 2, 4, 4>(k, s);      // (128-channel blocks spill: 10 prefetch items + 64 accumulators + the weight ring)
    // PatchGAN logits layer (512 -> 1): the single output channel rides in a 16-channel MFMA tile, the input is staged once
    if (ntaps == 16 && KH == 4 && KW == 4 && TW == 16 && k.Cout <= 16 && k.bstep == 1 && k.Cin % 32 == 0) {
        // small maps: 4-row tiles double the workgroup count (31 x 31 logits: 128 -> 256 workgroups)
        if ((long long)k.B * hv_cdiv(k.cls[0].Hc, 8) * hv_cdiv(k.cls[0].Wc, 16) < 256) return launch2<4, 16, 16, 4, 1, 32, 1, 4, 4>(k, s, 8, 16);
        return launch2<8, 16, 16, 4, 1, 32, 1, 4, 4>(k, s);
    }
    // (4x4 stride-2 FORWARD stays on conv_halo_kernel: its 18 x 34-pixel patch leaves this kernel either 16-channel chunks or 4-row tiles
    // to keep two workgroups per CU -- measured 60-76 us against 48 us on 64 -> 128 @128^2 and 48-66 us against 43 us on 128 -> 256 @64^2)
    // data gradient of the 4x4 stride-2 layers: four output-parity classes of 2x2 taps each
    static const int m4 = getenv("HV_HALO2_S2T") ? atoi(getenv("HV_HALO2_S2T")) : 1;
    if (m4 && ntaps == 4 && KH == 4 && KW == 4 && k.bstep == 1 && k.Cin % 32 == 0 && k.Cout > 32) {
        // Tile and wave arrangement, measured alone (HV_HALO2_T2=0: 8x16 tiles with 1 x 4 waves everywhere; 2: also the 4x16 tiles below -- inside the
        // step the three discriminators' kernels share the GPU, the small grids are filled anyway and the 4x16 tiles' doubled filter traffic shows no
        // gain there: 11.17 vs 11.22 ms, so only the LDS-saving 2 x 2 arrangement is on by default):
        //   64 <- 128 @64^2 (64-channel blocks): 8x16 tiles, 2 x 2 waves 42.7 -> 36.3 us (1 x 4 waves read 8 LDS fragments per 8 MFMAs)
        //   128 <- 256 @64^2 (512 workgroups of 8x16 pixels): 4x16 tiles 36.3 -> 27.7 us (64-channel blocks as 2 x 2: 28.5 us); 256 <- 512 @32^2: 47.3 -> 32.0 us
        static const int t2 = getenv("HV_HALO2_T2") ? atoi(getenv("HV_HALO2_T2")) : 1;
        if (TW == 16 && t2) {
            if (k.Cout <= 64) return launch2<8, 16, 64, 2, 2, 32, 1, 2, 4>(k, s);
            long long wgs = 0;
            for (int c = 0; c < k.ncls; ++c) wgs += (long long)k.B * hv_cdiv(k.cls[c].Hc, 8) * hv_cdiv(k.cls[c].Wc, 16) * hv_cdiv(k.Cout, 128);
            if (wgs < 1024 && t2 == 2) return launch2<4, 16, 128, 1, 4, 32, 1, 2, 4>(k, s, 8, 16);
        }
        if (TW == 32) return k.Cout <= 64 ? launch2<8, 32, 64, 1, 4, 32, 1, 2, 4>(k, s) : launch2<8, 32, 128, 1, 4, 32, 1, 2, 4>(k, s);
        return k.Cout <= 64 ? launch2<8, 16, 64, 1, 4, 32, 1, 2, 4>(k, s) : launch2<8, 16, 128, 1, 4, 32, 1, 2, 4>(k, s);
    }
    // 3x3 layers of the generator.  Waves split the output channels (WN = 4) wherever a wave still gets 16 of them, so
    // no two waves fetch the same filter rows.  HV_HALO2_MASK (bit per Cout class 16/32/64/128) is an A/B knob.
    static const int mask = getenv("HV_HALO2_MASK") ? atoi(getenv("HV_HALO2_MASK")) : 15;
    if (ntaps == 9 && KH == 3 && KW == 3 && k.bstep == 1) {
        const bool ck32 = k.Cin % 32 == 0;
        const int cls = k.Cout <= 16 ? 1 : k.Cout <= 32 ? 2 : k.Cout <= 64 ? 4 : 8;
        if (!(mask & cls)) return HV_ERR_UNSUPPORTED;
        if (TW == 32) {
            if (cls == 1) return ck32 ? launch2<8, 32, 16, 4, 1, 32, 1, 3, 3>(k, s) : launch2<8, 32, 16, 4, 1, 16, 1, 3, 3>(k, s);
            if (cls == 2) return ck32 ? launch2<8, 32, 32, 2, 2, 32, 1, 3, 3>(k, s) : launch2<8, 32, 32, 2, 2, 16, 1, 3, 3>(k, s);
            if (cls == 4) return ck32 ? launch2<8, 32, 64, 1, 4, 32, 1, 3, 3>(k, s) : launch2<8, 32, 64, 1, 4, 16, 1, 3, 3>(k, s);
            // 128 channels x 256 pixels per workgroup needs 128 accumulator registers per lane and wastes half of them on the
            // 68-channel layer of this model (measured 2.5x slower than conv_halo_kernel): not taken
            return HV_ERR_UNSUPPORTED;
        }
        if (!ck32) {   // 16-channel chunks (whole or ragged) on 8x16 tiles
            if (cls == 1) return launch2<8, 16, 16, 4, 1, 16, 1, 3, 3>(k, s);
            if (cls == 2) return launch2<8, 16, 32, 2, 2, 16, 1, 3, 3>(k, s);
            if (cls == 4) return launch2<8, 16, 64, 1, 4, 16, 1, 3, 3>(k, s);
            return HV_ERR_UNSUPPORTED;
        }
        if (cls == 1) return launch2<8, 16, 16, 4, 1, 32, 1, 3, 3>(k, s);
        if (cls == 2) return launch2<8, 16, 32, 2, 2, 32, 1, 3, 3>(k, s);
        // 64-channel blocks: 4x16-pixel tiles (HV_HALO2_T3=0: 8x16; 2: 8x16 with 2 x 2 waves; 3: 4x16 with 2 x 2).  Alone: 64 -> 64 @64^2 15.0 -> 12.9 us (2: 14.6,
        // 3: 13.6), 128 -> 64 @64^2 20.7 -> 18.2 us, 32 -> 64 @128^2 33.4 -> 29.9 us: these layers are latency chains of 1-4 chunks, more and smaller
        // workgroups overlap them better
        static const int t3 = getenv("HV_HALO2_T3") ? atoi(getenv("HV_HALO2_T3")) : 1;
        // (64-channel chunks -- the whole K of a 64-channel layer behind one barrier -- measured 16.9 -> 21.6 us: not kept)
        if (cls == 4 && t3 == 1) return launch2<4, 16, 64, 1, 4, 32, 1, 3, 3>(k, s, 8, 16);
        if (cls == 4 && t3 == 2) return launch2<8, 16, 64, 2, 2, 32, 1, 3, 3>(k, s);
        if (cls == 4 && t3 == 3) return launch2<4, 16, 64, 2, 2, 32, 1, 3, 3>(k, s, 8, 16);
        if (cls == 4) return launch2<8, 16, 64, 1, 4, 32, 1, 3, 3>(k, s);
        return launch2<8, 16, 128, 1, 4, 32, 1, 3, 3>(k, s);
    }
    return HV_ERR_UNSUPPORTED;
}

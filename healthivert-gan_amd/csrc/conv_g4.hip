// 4x4 stride-2 convolution and its data gradient (the PatchGAN / U-Net down-sampling layers) as a pipelined implicit GEMM.
//
// These layers are GEMM-sized (64 -> 128 @128^2: M = 65 536 output pixels, N = 128, K = 1 024; 17.2 GFLOP per launch at bs 16) but ran at 12-21 % of
// the fp16 MFMA peak: conv_halo_kernel / conv_halo2_kernel stage a stride-2 patch (18 x 34 pixels for 8 x 16 outputs, 4x the pixels of a stride-1
// patch) per 32-channel chunk, read MFMA B fragments at a pixel stride of 2 and fetch a 64-128 KB filter slice per 128-pixel tile.  Here
//   forward (MODE 0): the convolution is read as a 2x2-tap STRIDE-1 convolution over the space-to-depth view x'[r][c][(dy, dx, ch)] =
//     x[2r - 1 + dy][2c - 1 + dx][ch] (never materialised: a 32-channel chunk of x' is a 64-byte piece of one real pixel), so a K-chunk's patch is
//     (TH + 1) x (TW + 1) pixels and fragments are unit-stride;
//   data gradient (MODE 1): the four output-parity classes are 2x2-tap stride-1 convolutions over the SAME gradient patch; a workgroup computes two
//     (class, 64-channel) slots, every wave one slot;
//   both: a workgroup of 8 waves owns TH x 16 pixels x 128 GEMM columns (8 row blocks of the fragment-ordered filter table); K runs in chunks of
//     32 operand channels x 4 taps; per chunk the 32 KB filter slice and the patch go global -> registers -> LDS, double-buffered, the loads of chunk
//     c + 1 in flight behind the MFMAs of chunk c, one barrier per chunk; a wave computes (TH/4 rows x 16 pixels) x 64 columns: 4 A + TH/4 B fragment
//     reads per TH MFMAs; the fp16 output tile leaves through LDS as 16-byte pieces.
// Optional epilogue: per-channel sum / sum of squares of the (fp16-rounded) outputs of the workgroup's tile (hv_conv_desc.stats), the BatchNorm
// statistics the separate reduction pass used to re-read the tensor for.
#include <stdlib.h>
#include <type_traits>

#include "conv_halo.h"

struct G4K {
    const void* x; const _Float16* w; const float* bias; void* y; const void* mul_src;
    float* stats;                              // [tiles_total][Cstat][2] partial sums (or NULL)
    // data-gradient forms: the sums of the batch normalisation whose output gradient this launch writes (hv_conv_desc.bstats) -- bn_x its raw input (fp16, same
    // pixels / channels as y), bn_stats its [groups][2][Cout] mean / rstd, bn_ipg images per group; bstats[(part * Cout + c) * 2 + {sum g, sum g * xhat}]
    const void* bn_x; const float* bn_stats; float* bstats;
    int bn_x_ld, bn_x_coff, bn_ipg;
    int B, H, W, x_ld, x_coff, Cin;            // operand tensor (forward: x, data gradient: g) and its channel count (the GEMM's K source)
    int Ho, Wo, y_ld, y_coff, Cout;
    int mul_ld, mul_coff, mul_act, act, accumulate;
    float alpha;
    int tiles_x, tiles;                        // tiles per image on the low-resolution grid (forward: outputs; data gradient: g pixels)
    int Hc, Wc;                                // extent of that grid
    unsigned x_bytes, w_bytes, w_rb;           // w_rb: bytes of one 16-row block of the filter table
    int dbg;                                   // diagnostic builds (G4_STAMPS): bit 0 no loads / LDS writes in the loop, bit 1 no barriers, bit 2 no MFMAs, bit 3 no fragment reads
};

typedef __attribute__((address_space(3))) void* lds_ptr;
// LDS-DMA: 16 bytes per lane, global -> LDS at dst + 16 * lane (out-of-range lanes write zeros).  In its own device-only body: with the builtin inside the
// kernel template hipcc's host pass silently drops the kernel's host stub (undefined symbol at load time).
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t r, lds_ptr dst, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, voff, soff, 0, 0);
#endif
}

// this lane's bias values (channels cob + 16 n + 4 (lane >> 4) ..), requested in the kernel's prologue: fetched in the epilogue they were a memory round
// trip in front of the activation pass (no bias: the descriptor's range is empty, the loads return zeros)
__device__ __forceinline__ void g4_load_bias(const G4K& p, int cob, int lane, f32x4 (&bias_r)[4]) {
    const __amdgpu_buffer_rsrc_t bsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias ? (unsigned)p.Cout * 4u : 0u, 0x00020000);
#pragma unroll
    for (int n = 0; n < 4; ++n)
        bias_r[n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bsrc, (unsigned)(cob + n * 16 + (lane >> 4) * 4) * 4u, 0, 0));
}

// Shared epilogue of the pipelined-GEMM kernels: the workgroup's TH x 16 pixels x 128 columns (two 64-column slots) -> (alpha, +bias, activation) ->
// fp16 tile in LDS -> 16-byte pieces with the act' multiplier / accumulate forms; optional per-channel statistics of the stored tile.
// MODE 1 (stride-2 data gradient): slot s is the output-parity class cls_s[s] -- pixel (i, j) of the tile lands at (2 i + py, 2 j + px).
template <int MODE, int MT>
__device__ __forceinline__ void g4_epilogue(const G4K& p, f32x4 (&acc)[4][MT], char* smem, const int (&cls_s)[2], const int (&cob_s)[2], int n_img, int i0,
                                            int j0, const f32x4 (&bias_r)[4]) {
    constexpr int TW = 16, TH = 4 * MT, NTHR = 512, LDO = 128 + 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    _Float16* As = reinterpret_cast<_Float16*>(smem);
    // ---- epilogue: this thread's output pieces (act' multiplier, old gradient) requested first, then (alpha, +bias, activation) -> fp16 tile in
    // LDS -> 16-byte pieces
    constexpr int OITEMS = TH * TW * 16 / NTHR;
    const __amdgpu_buffer_rsrc_t msrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.mul_src), 0, p.mul_src ? 0x7ffffff0u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ysrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.accumulate ? 0x7ffffff0u : 0u, 0x00020000);
    u32x4 mreg[OITEMS], yreg[OITEMS];
    long long ooff[OITEMS];
#pragma unroll
    for (int k = 0; k < OITEMS; ++k) {
        const int it = tid + k * NTHR;
        const int q = it >> 4, pc = it & 15, s = pc >> 3;
        const int i = i0 + (q >> 4), j = j0 + (q & 15);
        const int ch = cob_s[s] + (pc & 7) * 8;
        const bool ok = i < p.Hc && j < p.Wc && ch < p.Cout;
        int ho = i, wo = j;
        if (MODE == 1) { ho = 2 * i + (cls_s[s] >> 1); wo = 2 * j + (cls_s[s] & 1); }
        const long long opix = (long long)(n_img * p.Ho + ho) * p.Wo + wo;
        ooff[k] = ok ? opix * p.y_ld + p.y_coff + ch : -1;
        mreg[k] = __builtin_amdgcn_raw_buffer_load_b128(msrc, ok ? (unsigned)((opix * p.mul_ld + p.mul_coff + ch) * 2) : HV_OOB, 0, 0);
        yreg[k] = __builtin_amdgcn_raw_buffer_load_b128(ysrc, ok ? (unsigned)(ooff[k] * 2) : HV_OOB, 0, 0);
    }
    _Float16* ot = As;         // the whole LDS is free behind the loop's last barrier: [TH * TW][LDO]
    auto stage = [&](auto actf) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int q = (wm * MT + m) * 16 + (lane & 15);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                f16x4v h;
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = (_Float16)actf(acc[n][m][r] * p.alpha + bias_r[n][r]);
                *reinterpret_cast<f16x4v*>(ot + q * LDO + wn * 64 + n * 16 + (lane >> 4) * 4) = h;
            }
        }
    };
    switch (p.act) {
        case HV_ACT_ELU:
            stage([](float v) { const float e = __builtin_amdgcn_exp2f(v * 1.44269504f) - 1.f, sm = v + 0.5f * v * v; return v > 0.f ? v : (v > -0.00390625f ? sm : e); });
            break;
        case HV_ACT_RELU: stage([](float v) { return v > 0.f ? v : 0.f; }); break;
        case HV_ACT_LRELU: stage([](float v) { return v > 0.f ? v : 0.2f * v; }); break;
        case HV_ACT_SIGMOID: stage([](float v) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-v * 1.44269504f)); }); break;
        case HV_ACT_CLAMP: stage([](float v) { return fminf(fmaxf(v, -1.f), 1.f); }); break;
        default: stage([](float v) { return v; }); break;
    }
    __syncthreads();
    _Float16* yb = reinterpret_cast<_Float16*>(p.y);
    u32x4 o[OITEMS];
#pragma unroll
    for (int k = 0; k < OITEMS; ++k) {
        const int it = tid + k * NTHR;
        o[k] = *reinterpret_cast<const u32x4*>(ot + (it >> 4) * LDO + (it & 15) * 8);
    }
    if (MODE != 1 && p.stats) {
        // BatchNorm statistics of this tile: thread (q-lane, piece) sums its OITEMS pixels' 8 channels (piece = it & 15 is the same for all of a
        // thread's items), then the 32 threads that share a piece are folded through LDS; invalid pixels / channels contribute zeros
        float s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
#pragma unroll
        for (int k = 0; k < OITEMS; ++k) {
            if (ooff[k] < 0) continue;
            const f16x8 v8 = __builtin_bit_cast(f16x8, o[k]);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float v = (float)v8[e]; s1[e] += v; s2[e] += v * v; }
        }
        __syncthreads();                       // the staging tile has been read into registers by every thread
        float* red = reinterpret_cast<float*>(smem);       // [32 rows][16 pieces][16]
        float* mine = red + ((tid >> 4) * 16 + (tid & 15)) * 16;
#pragma unroll
        for (int e = 0; e < 8; ++e) { mine[e] = s1[e]; mine[8 + e] = s2[e]; }
        __syncthreads();
        if (tid < 256) {      // 128 channels x {sum, sum of squares}
            const int pc = tid >> 4, e = tid & 15;
            float s = 0.f;
#pragma unroll 8
            for (int r = 0; r < 32; ++r) s += red[(r * 16 + pc) * 16 + e];
            const int ch = (int)blockIdx.y * 128 + pc * 8 + (e & 7);
            if (ch < p.Cout) p.stats[((long long)blockIdx.x * p.Cout + ch) * 2 + (e >> 3)] = s;
        }
    }
    if (p.mul_src) {
        auto mulf = [&](auto gradf) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < OITEMS; ++k) {
                const f16x8 m8 = __builtin_bit_cast(f16x8, mreg[k]);
                f16x8 v8 = __builtin_bit_cast(f16x8, o[k]);
#pragma unroll
                for (int e = 0; e < 8; ++e) v8[e] = (_Float16)((float)v8[e] * gradf((float)m8[e]));
                o[k] = __builtin_bit_cast(u32x4, v8);
            }
        };
        switch (p.mul_act) {
            case HV_ACT_ELU: mulf([](float y) { return y > 0.f ? 1.f : y + 1.f; }); break;
            case HV_ACT_RELU: mulf([](float y) { return y > 0.f ? 1.f : 0.f; }); break;
            case HV_ACT_LRELU: mulf([](float y) { return y > 0.f ? 1.f : 0.2f; }); break;
            case HV_ACT_SIGMOID: mulf([](float y) { return y * (1.f - y); }); break;
            case HV_ACT_CLAMP: mulf([](float y) { return (y > -1.f && y < 1.f) ? 1.f : 0.f; }); break;
            default: break;
        }
    }
    if (p.accumulate) {
#pragma unroll
        for (int k = 0; k < OITEMS; ++k) {
            const f16x8 y8 = __builtin_bit_cast(f16x8, yreg[k]);
            f16x8 v8 = __builtin_bit_cast(f16x8, o[k]);
#pragma unroll
            for (int e = 0; e < 8; ++e) v8[e] = (_Float16)((float)v8[e] + (float)y8[e]);
            o[k] = __builtin_bit_cast(u32x4, v8);
        }
    }
#pragma unroll
    for (int k = 0; k < OITEMS; ++k)
        if (ooff[k] >= 0) *reinterpret_cast<u32x4*>(yb + ooff[k]) = o[k];
    if (p.bstats) {
        // Batch-norm backward sums of this tile (the reduction pass norm_reduce_kernel<1> re-read dy and x for): per channel sum g and sum g * xhat, g = the value
        // just stored, xhat = (x - mean) * rstd of the normalisation's raw input at the same pixel.  A thread's OITEMS pieces share one 8-channel piece
        // (it & 15): its x pieces are requested now (the multiplier / old-gradient registers are dead), folded like the forward statistics above.
        const int pc = tid & 15, s = pc >> 3;
        const int ch0 = cob_s[s] + (pc & 7) * 8;
        const int grp = n_img / p.bn_ipg;
        const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.bn_x), 0, 0x7ffffff0u, 0x00020000);
        u32x4 xreg[OITEMS];
#pragma unroll
        for (int k = 0; k < OITEMS; ++k) {
            // the x piece of output piece k: same pixel and channels, the normalisation input's own strides
            const int q = (tid + k * NTHR) >> 4;
            int ho = i0 + (q >> 4), wo = j0 + (q & 15);
            if (MODE == 1) { ho = 2 * ho + (cls_s[s] >> 1); wo = 2 * wo + (cls_s[s] & 1); }
            const long long opix = (long long)(n_img * p.Ho + ho) * p.Wo + wo;
            xreg[k] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, ooff[k] >= 0 ? (unsigned)((opix * p.bn_x_ld + p.bn_x_coff + ch0) * 2) : HV_OOB, 0, 0);
        }
        float s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
#pragma unroll
        for (int k = 0; k < OITEMS; ++k) {
            if (ooff[k] < 0) continue;
            const f16x8 v8 = __builtin_bit_cast(f16x8, o[k]), x8 = __builtin_bit_cast(f16x8, xreg[k]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {      // sum g and sum g * x: the mean / rstd of xhat = (x - mean) * rstd enter once, when the tile's sums are folded
                const float g = (float)v8[e];
                s1[e] += g; s2[e] += g * (float)x8[e];
            }
        }
        __syncthreads();                       // the staging tile has been read into registers by every thread
        float* red = reinterpret_cast<float*>(smem);       // [32 rows][16 pieces][16]
        float* mine = red + ((tid >> 4) * 16 + (tid & 15)) * 16;
#pragma unroll
        for (int e = 0; e < 8; ++e) { mine[e] = s1[e]; mine[8 + e] = s2[e]; }
        __syncthreads();
        if (tid < 128) {      // one thread per column (two 64-channel slots): sum g, and sum g * xhat = rstd * (sum g x - mean * sum g)
            const int pq = tid >> 3, e = tid & 7;
            float sg = 0.f, sgx = 0.f;
#pragma unroll 8
            for (int r = 0; r < 32; ++r) { sg += red[(r * 16 + pq) * 16 + e]; sgx += red[(r * 16 + pq) * 16 + 8 + e]; }
            const int sl = pq >> 3, ch = cob_s[sl] + (pq & 7) * 8 + e;
            // MODE 1: the four output-parity classes of a tile are four parts (two per workgroup of the class pair); else one part per workgroup
            const long long part = MODE == 1 ? (long long)blockIdx.x * 4 + cls_s[sl] : (long long)blockIdx.x;
            if (ch < p.Cout) {
                const float mean = p.bn_stats[(long long)grp * 2 * p.Cout + ch], rstd = p.bn_stats[(long long)grp * 2 * p.Cout + p.Cout + ch];
                float* o2 = p.bstats + (part * p.Cout + ch) * 2;
                o2[0] = sg;
                o2[1] = rstd * (sgx - mean * sg);
            }
        }
    }
}

template <int MODE, int MT>
__global__ __launch_bounds__(512, 2) void conv_g4_kernel(const G4K p) {
    constexpr int TW = 16, TH = 4 * MT, NTHR = 512;
    constexpr int PH = TH + 1 + MODE, PW = TW + 1 + MODE;
    // LDS: three buffers of {A: 32 fragments of 1 KB, linear; B: the patch as 64-byte pixel rows, 16 pixels per 1-KB piece}, both filled by LDS-DMA
    // (buffer_load ... lds: 1 KB per wave instruction, no registers, no ds_write pass).  A piece's LDS image is lane-linear, so the patch rows cannot be
    // padded; instead slot s of pixel pp holds its 16-byte channel piece s ^ 2((pp >> 2) & 1) (the permutation goes on the SOURCE address): the 16
    // lanes of every ds_read_b128 lane group of a B-fragment read -- 16 consecutive pixels from any start, two adjacent channel pieces -- then hit 16
    // different 16-byte columns of the 256-byte LDS line (checked exhaustively over start offsets, tools/lds_swizzle_check.py).
    constexpr int NBP = (PH * PW + 15) / 16;                                  // 1-KB pieces of the patch
    constexpr int BPW = (NBP + 7) / 8;                                        // ... per wave
    constexpr int ABUF = 32 * 512;                                            // halfs: 8 row blocks x 4 taps x (16 x 32) fragment
    constexpr int BBUF = NBP * 512;                                           // halfs
    constexpr int NBUF = 3;
    constexpr int LDO = 128 + 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* As = reinterpret_cast<_Float16*>(smem);                         // [3][ABUF]
    _Float16* Bs = As + NBUF * ABUF;                                          // [3][BBUF]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef G4_STAMPS     // diagnostic build only (tools/g4_stamps.py): phase times of one workgroup, written over the first bytes of y
    unsigned long long st[8];
#define G4_STAMP(i) st[i] = __builtin_amdgcn_s_memrealtime()
#else
#define G4_STAMP(i)
#endif
    G4_STAMP(0);
    const int wm = wave >> 1, wn = wave & 1;
    int t = (int)blockIdx.x;
    const int n_img = t / p.tiles;
    t -= n_img * p.tiles;
    const int tile_y = t / p.tiles_x, tile_x = t - tile_y * p.tiles_x;
    const int i0 = tile_y * TH, j0 = tile_x * TW;
    const int Cin = p.Cin, KC = Cin >> 5;
    const int NCH = MODE == 0 ? 4 * KC : KC;

    // the two 64-column slots of the workgroup: forward = output channels n_base + 64 wn; data gradient = (parity class, channel block) pairs
    int cls_s[2], cob_s[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (MODE == 0) { cls_s[s] = 0; cob_s[s] = (int)blockIdx.y * 128 + s * 64; }
        else {
            const int nco = p.Cout >> 6, sl = (int)blockIdx.y * 2 + s;
            cls_s[s] = sl / nco; cob_s[s] = (sl - cls_s[s] * nco) * 64;
        }
    }
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.w), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    f32x4 bias_r[4];
    g4_load_bias(p, cob_s[wn], lane, bias_r);

    // ---- this lane's part of the wave's patch pieces (chunk-independent): piece j = wave + 8 i covers pixels 16 j .. 16 j + 15, lane -> (pixel, slot)
    int pbase[BPW], pval[BPW];
    const int xbase = (n_img * p.H * p.W * p.x_ld + p.x_coff) * 2;
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
        const int pix = (wave + 8 * i) * 16 + (lane >> 2);
        const int c8 = (lane & 3) ^ (((pix >> 2) & 1) << 1);                  // the channel piece this LDS slot holds
        const int pr = pix / PW, pc = pix - pr * PW;
        const bool in = pix < PH * PW && wave + 8 * i < NBP;
        if (MODE == 0) {
            const int r0 = 2 * (i0 + pr) - 1, c0 = 2 * (j0 + pc) - 1;
            pbase[i] = xbase + ((r0 * p.W + c0) * p.x_ld + c8 * 8) * 2;
            int v = 0;
#pragma unroll
            for (int dd = 0; dd < 4; ++dd)
                if (in && (unsigned)(r0 + (dd >> 1)) < (unsigned)p.H && (unsigned)(c0 + (dd & 1)) < (unsigned)p.W) v |= 1 << dd;
            pval[i] = v;
        } else {
            const int r0 = i0 - 1 + pr, c0 = j0 - 1 + pc;
            pbase[i] = xbase + ((r0 * p.W + c0) * p.x_ld + c8 * 8) * 2;
            pval[i] = (in && (unsigned)r0 < (unsigned)p.H && (unsigned)c0 < (unsigned)p.W) ? 15 : 0;
        }
    }
    // every wave issues the same number of LDS-DMA instructions per chunk (4 filter fragments + BPW patch pieces; a piece index beyond the patch is
    // sent with every lane out of range and lands in a spare KB), so that the counted vmcnt below means the same thing in every wave
    constexpr int DMA_PER_CHUNK = 4 + BPW;
    auto issue = [&](int c, int b) __attribute__((always_inline)) {
        int dy = 0, dx = 0, kc = c;
        if (MODE == 0) { dy = c / (2 * KC); const int rem = c - dy * 2 * KC; dx = rem / KC; kc = rem - dx * KC; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = wave + 8 * i, rbw = f >> 2, tt = f & 3, s = rbw >> 2;
            const int rb = (cob_s[s] >> 4) + (rbw & 3);
            int tap;
            if (MODE == 0) tap = (2 * (tt >> 1) + dy) * 4 + 2 * (tt & 1) + dx;
            else { const int py = cls_s[s] >> 1, px = cls_s[s] & 1; tap = (1 - py + 2 * (tt >> 1)) * 4 + (1 - px + 2 * (tt & 1)); }
            lds_dma16(wsrc, (lds_ptr)(As + b * ABUF + f * 512), (unsigned)lane * 16u, rb * (int)p.w_rb + (tap * Cin + kc * 32) * 32);
        }
        const int coff = MODE == 0 ? ((dy * p.W + dx) * p.x_ld + kc * 32) * 2 : kc * 64;
        const int vb = MODE == 0 ? dy * 2 + dx : 0;
#pragma unroll
        for (int i = 0; i < BPW; ++i) {
            _Float16* dst = wave + 8 * i < NBP ? Bs + b * BBUF + (wave + 8 * i) * 512 : Bs + NBUF * BBUF;      // (the spare KB behind the three buffers)
            lds_dma16(xsrc, (lds_ptr)dst, ((pval[i] >> vb) & 1) ? (unsigned)(pbase[i] + coff) : HV_OOB, 0);
        }
    };

    // B-fragment addresses of this lane: pixel row m of the wave, tap tt (the slot permutation depends on the pixel, i.e. on the tap)
    int boffs[4][MT];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
        int dh, dw;
        if (MODE == 0) { dh = tt >> 1; dw = tt & 1; }
        else { const int py = cls_s[wn] >> 1, px = cls_s[wn] & 1; dh = py - (tt >> 1) + 1; dw = px - (tt & 1) + 1; }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int pix = (wm * MT + m + dh) * PW + (lane & 15) + dw;
            boffs[tt][m] = pix * 32 + (((lane >> 4) ^ (((pix >> 2) & 1) << 1)) << 3);      // halfs
        }
    }
    f32x4 acc[4][MT];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int aoff = (wn * 16) * 512 + lane * 8;

    f16x8 a[2][4], bf[2][MT];
    auto frags = [&](int b, int tt, int buf) __attribute__((always_inline)) {
        const _Float16* Ab = As + b * ABUF + aoff;
        const _Float16* Bb = Bs + b * BBUF;
#pragma unroll
        for (int n = 0; n < 4; ++n) a[buf][n] = *reinterpret_cast<const f16x8*>(Ab + (n * 4 + tt) * 512);
#pragma unroll
        for (int m = 0; m < MT; ++m) bf[buf][m] = *reinterpret_cast<const f16x8*>(Bb + boffs[tt][m]);
    };
    auto mfmas = [&](int buf) __attribute__((always_inline)) {
#ifdef G4_STAMPS
        if (p.dbg & 4) return;
#endif
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[buf][n], bf[buf][m], acc[n][m], 0, 0, 0);
    };
    // Chunk c computes on LDS buffer c % 3 while the DMA of chunks c + 1 and c + 2 is in flight.  The barrier sits BEFORE the last tap's MFMAs: by then
    // the wave holds all its fragments of chunk c in registers (lgkmcnt(0)) and has waited for its OWN pieces of chunk c + 1 (requested a whole chunk
    // earlier), so behind the barrier chunk c + 1 is complete in LDS and buffer c % 3 is free; the first fragments of chunk c + 1 are then read behind
    // the last tap's MFMAs instead of at the head of the next chunk with every wave of the workgroup waiting for LDS at once.
    auto chunk = [&](int c, int b) __attribute__((always_inline)) {
        const int bn = b == 2 ? 0 : b + 1;
        // (the pieces of chunk c + 2 one per tap instead of all here -- what conv_g4s1_kernel gains 0-15 % from -- measured 0.95-1.03x on these shapes: not kept)
#ifdef G4_STAMPS
        if (c + 2 < NCH && !(p.dbg & 1)) issue(c + 2, b == 0 ? 2 : b - 1);
#else
        if (c + 2 < NCH) issue(c + 2, b == 0 ? 2 : b - 1);
#endif
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
            __builtin_amdgcn_sched_barrier(0);
            frags(b, tt + 1, (tt + 1) & 1);
            mfmas(tt & 1);
            // one fragment read per two MFMAs: a burst of 8 reads from each of the 8 barrier-aligned waves is 64 KB the LDS serves in 256 cycles
            // during which no wave can issue an MFMA
#pragma unroll
            for (int i = 0; i < 4 + MT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, (4 * MT) / (4 + MT), 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c + 2 < NCH) __builtin_amdgcn_s_waitcnt(0x0070 | (DMA_PER_CHUNK & 15) | ((DMA_PER_CHUNK >> 4) << 14));     // lgkmcnt(0), vmcnt(DMA_PER_CHUNK): chunk c + 1 landed, c + 2 in flight
        else __builtin_amdgcn_s_waitcnt(0x0070);                                                                     // lgkmcnt(0), vmcnt(0)
        __builtin_amdgcn_s_barrier();
        if (c + 1 < NCH) frags(bn, 0, 0);
        mfmas(1);
#pragma unroll
        for (int i = 0; i < 4 + MT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, (4 * MT) / (4 + MT), 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    issue(0, 0);
    if (NCH > 1) issue(1, 1);
    G4_STAMP(1);
    if (NCH > 1) __builtin_amdgcn_s_waitcnt(0x0F70 | (DMA_PER_CHUNK & 15) | ((DMA_PER_CHUNK >> 4) << 14));
    else __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();
    G4_STAMP(2);
    frags(0, 0, 0);
    for (int c = 0; c < NCH; c += 3) {
        chunk(c, 0);
        if (c + 1 < NCH) chunk(c + 1, 1);
        if (c + 2 < NCH) chunk(c + 2, 2);
    }
    __syncthreads();
    G4_STAMP(3);

    g4_epilogue<MODE, MT>(p, acc, smem, cls_s, cob_s, n_img, i0, j0, bias_r);
#ifdef G4_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
    G4_STAMP(5);
    if (tid == 0 && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0) {
        unsigned long long* d = reinterpret_cast<unsigned long long*>(p.y);
        for (int i = 0; i < 6; ++i) d[i] = st[i];
    }
#endif
}

// 4x4 STRIDE-1 convolution (DG = 0: pad 1, output (H - 1) x (W - 1)) and its data gradient (DG = 1: the same correlation with the taps mirrored,
// output (H + 1) x (W + 1)) -- the PatchGAN 256 <-> 512 layers, 64.5 GFLOP per launch -- on the same pipelined-GEMM core: TH x 16 pixels x 128
// columns per workgroup, K in sub-chunks of 32 channels x ONE filter row (its 4 taps): the 32 KB filter slice of a sub-chunk arrives by LDS-DMA into
// a ring of three buffers, the (TH + 3) x 20-pixel patch of a 32-channel chunk arrives ONCE for its four filter rows (two buffers).  conv_halo2_kernel
// fetched 1 MB of filters per 128-pixel x 64/128-channel tile straight into registers (537 MB of L2 -> CU traffic per launch, 2x the tensors'
// HBM bytes); a 256-pixel x 128-channel tile halves that and both operands come from LDS.
template <int DG, int MT, bool SPREAD>
__device__ __forceinline__ void conv_g4s1_body(const G4K& p) {
    constexpr int TW = 16, TH = 4 * MT, PH = TH + 3, PW = 20;                // patch rows of 20 pixels (19 used): a row shift moves the swizzle phase by its parity only
    constexpr int NBP = (PH * PW + 15) / 16, BPW = (NBP + 7) / 8;
    static_assert(BPW <= 4, "one patch piece per tap");
    constexpr int ABUF = 32 * 512, BBUF = NBP * 512;                          // halfs
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* As = reinterpret_cast<_Float16*>(smem);                         // [3][ABUF]
    _Float16* Bs = As + 3 * ABUF;                                             // [2][BBUF] + one spare KB
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
#ifdef G4_STAMPS     // diagnostic build only (tools/g4_stamps.py): phase times of one workgroup, written over the first bytes of y
    unsigned long long st[8];
    st[0] = __builtin_amdgcn_s_memrealtime();
#endif
    int t = (int)blockIdx.x;
    const int n_img = t / p.tiles;
    t -= n_img * p.tiles;
    const int tile_y = t / p.tiles_x, tile_x = t - tile_y * p.tiles_x;
    const int i0 = tile_y * TH, j0 = tile_x * TW;
    const int Cin = p.Cin, KC = Cin >> 5, NS = 4 * KC;
    int cls_s[2] = {0, 0}, cob_s[2] = {(int)blockIdx.y * 128, (int)blockIdx.y * 128 + 64};
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.w), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    f32x4 bias_r[4];
    g4_load_bias(p, cob_s[wn], lane, bias_r);
    // patch origin: forward reads x[i - 1 + kh], the data gradient g[i + 1 - kh] = g[i - 2 + (3 - kh)]
    constexpr int ORG = DG ? 2 : 1;
    unsigned pvo[BPW];
    const int xbase = (n_img * p.H * p.W * p.x_ld + p.x_coff) * 2;
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
        const int pix = (wave + 8 * i) * 16 + (lane >> 2);
        const int c8 = (lane & 3) ^ (((pix >> 2) & 1) << 1);
        const int pr = pix / PW, pc = pix - pr * PW;
        const int r0 = i0 - ORG + pr, c0 = j0 - ORG + pc;
        const bool ok = wave + 8 * i < NBP && pr < PH && pc < PW - 1 && (unsigned)r0 < (unsigned)p.H && (unsigned)c0 < (unsigned)p.W;
        pvo[i] = ok ? (unsigned)(xbase + ((r0 * p.W + c0) * p.x_ld + c8 * 8) * 2) : HV_OOB;
    }
    auto issueA = [&](int kc, int kh, int ab) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = wave + 8 * i, rbw = f >> 2, kw = f & 3;
            const int rb = (cob_s[rbw >> 2] >> 4) + (rbw & 3);
            lds_dma16(wsrc, (lds_ptr)(As + ab * ABUF + f * 512), (unsigned)lane * 16u, rb * (int)p.w_rb + ((kh * 4 + kw) * Cin + kc * 32) * 32);
        }
    };
    // SPREAD: one piece at a time, in front of each tap's MFMAs (below); `live` = false sends the piece out of range (zeros into a free buffer), so that the
    // instruction count a wave's vmcnt waits rely on is the same for every sub-chunk
    auto issueA1 = [&](int kc, int kh, int ab, int i, bool live) __attribute__((always_inline)) {
        const int f = wave + 8 * i, rbw = f >> 2, kw = f & 3;
        const int rb = (cob_s[rbw >> 2] >> 4) + (rbw & 3);
        lds_dma16(wsrc, (lds_ptr)(As + ab * ABUF + f * 512), live ? (unsigned)lane * 16u : HV_OOB, rb * (int)p.w_rb + ((kh * 4 + kw) * Cin + kc * 32) * 32);
    };
    auto issueB1 = [&](int kc, int bb, int i, bool live) __attribute__((always_inline)) {
        _Float16* dst = wave + 8 * i < NBP ? Bs + bb * BBUF + (wave + 8 * i) * 512 : Bs + 2 * BBUF;
        lds_dma16(xsrc, (lds_ptr)dst, live ? pvo[i] : HV_OOB, kc * 64);
    };
    auto issueB = [&](int kc, int bb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < BPW; ++i) {
            _Float16* dst = wave + 8 * i < NBP ? Bs + bb * BBUF + (wave + 8 * i) * 512 : Bs + 2 * BBUF;
            lds_dma16(xsrc, (lds_ptr)dst, pvo[i], kc * 64);
        }
    };
    // B-fragment addresses (halfs) of this lane without the filter row's shift, for even and odd row shifts (the slot permutation flips with the
    // parity of the shift: 20 pixels per row = 5 groups of 4); the shift itself is an immediate (kh is unrolled)
    int bo[2][4][MT];
#pragma unroll
    for (int kw = 0; kw < 4; ++kw)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int pix = (wm * MT + m) * PW + (lane & 15) + (DG ? 3 - kw : kw);
            const int s0 = (pix >> 2) & 1;
            bo[0][kw][m] = pix * 32 + (((lane >> 4) ^ (s0 << 1)) << 3);
            bo[1][kw][m] = pix * 32 + (((lane >> 4) ^ ((s0 ^ 1) << 1)) << 3);
        }
    f32x4 acc[4][MT];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int aoff = (wn * 16) * 512 + lane * 8;
    f16x8 a[2][4], bf[2][MT];
    // one sub-chunk = (32-channel chunk kc, filter row kh): 4 taps x (4 x MT) MFMAs per wave.  DMA of sub-chunk s + 2 goes out at the head of s; before
    // the barrier at the end of s every wave waits for its own pieces of s + 1 (counted vmcnt: the newer pieces of s + 2 stay in flight).
    auto sub = [&](int kc, auto KH, int ab) __attribute__((always_inline)) {
        constexpr int kh = decltype(KH)::value;
        constexpr int rs = DG ? 3 - kh : kh;                                  // row shift of this filter row inside the patch
        const int s_ = kc * 4 + kh;
        // what goes out now: filters of sub-chunk s + 2, and (kh == 2) the patch of chunk kc + 1
        constexpr int kh2 = (kh + 2) & 3;
        const int kc2 = kc + (kh >= 2 ? 1 : 0);
        const int ab2 = ab == 0 ? 2 : ab - 1;                                 // (s + 2) % 3
        const bool live = s_ + 2 < NS;
        if (!SPREAD && live) {
            issueA(kc2, kh2, ab2);
            if (kh == 2) issueB(kc + 1, (kc + 1) & 1);
        }
        const _Float16* Ab = As + ab * ABUF + aoff;
        const _Float16* Bb = Bs + (kc & 1) * BBUF + rs * PW * 32;
        auto frags = [&](int kw, int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int n = 0; n < 4; ++n) a[buf][n] = *reinterpret_cast<const f16x8*>(Ab + (n * 4 + kw) * 512);
#pragma unroll
            for (int m = 0; m < MT; ++m) bf[buf][m] = *reinterpret_cast<const f16x8*>(Bb + bo[rs & 1][kw][m]);
        };
        frags(0, 0);
#pragma unroll
        for (int kw = 0; kw < 4; ++kw) {
            __builtin_amdgcn_sched_barrier(0);
            if (SPREAD) {      // eight waves issuing their 4 (+ BPW) pieces at once stall in the issue: the texture path takes one wave-instruction at a time
                issueA1(kc2, kh2, ab2, kw, live);
                if (kh == 2 && kw < BPW) issueB1(kc + 1, (kc + 1) & 1, kw, live);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (kw + 1 < 4) frags(kw + 1, (kw + 1) & 1);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[kw & 1][n], bf[kw & 1][m], acc[n][m], 0, 0, 0);
            if (kw + 1 < 4) {
#pragma unroll
                for (int i = 0; i < 4 + MT; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, (4 * MT) / (4 + MT), 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // newest DMA instructions of this wave that may stay in flight: those of sub-chunk s + 2 (issued above)
        if (SPREAD || live) {
            if (kh == 2) __builtin_amdgcn_s_waitcnt(0x0070 | ((4 + BPW) & 15));
            else __builtin_amdgcn_s_waitcnt(0x0070 | 4);
        } else {
            __builtin_amdgcn_s_waitcnt(0x0070);
        }
        __builtin_amdgcn_s_barrier();
    };
    // prologue: sub-chunks 0 and 1 (+ the patch of chunk 0) in flight; wait for sub-chunk 0's
    issueB(0, 0);
    issueA(0, 0, 0);
    issueA(0, 1, 1);
    __builtin_amdgcn_s_waitcnt(0x0F70 | 4);
    __builtin_amdgcn_s_barrier();
#ifdef G4_STAMPS
    st[1] = __builtin_amdgcn_s_memrealtime();
#endif
    int ab = 0;
    for (int kc = 0; kc < KC; ++kc) {
        sub(kc, std::integral_constant<int, 0>(), ab); ab = ab == 2 ? 0 : ab + 1;
        sub(kc, std::integral_constant<int, 1>(), ab); ab = ab == 2 ? 0 : ab + 1;
        sub(kc, std::integral_constant<int, 2>(), ab); ab = ab == 2 ? 0 : ab + 1;
        sub(kc, std::integral_constant<int, 3>(), ab); ab = ab == 2 ? 0 : ab + 1;
    }
    if (SPREAD) __builtin_amdgcn_s_waitcnt(0x0F70);      // the last sub-chunks' out-of-range pieces (zeros) land before the epilogue reuses the buffers
    __syncthreads();
#ifdef G4_STAMPS
    st[2] = __builtin_amdgcn_s_memrealtime();
#endif
    g4_epilogue<0, MT>(p, acc, smem, cls_s, cob_s, n_img, i0, j0, bias_r);
#ifdef G4_STAMPS
    st[3] = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0);
    st[4] = __builtin_amdgcn_s_memrealtime();
    if (tid == 0 && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0) {
        unsigned long long* d = reinterpret_cast<unsigned long long*>(p.y);
        for (int i = 0; i < 5; ++i) d[i] = st[i];
    }
#endif
}

// (two kernels over one body so that the profiled name of the production form stays conv_g4s1_kernel<DG, MT>)
template <int DG, int MT>
__global__ __launch_bounds__(512, 2) void conv_g4s1_kernel(const G4K p) { conv_g4s1_body<DG, MT, true>(p); }
template <int DG, int MT>
__global__ __launch_bounds__(512, 2) void conv_g4s1_headissue_kernel(const G4K p) { conv_g4s1_body<DG, MT, false>(p); }      // HV_G4S1_SPREAD=0: every piece at the sub-chunk's head

template <int DG, int MT>
static int launch_g4s1(G4K& k, int ny, hipStream_t s) {
    constexpr int TH = 4 * MT, PH = TH + 3, NBP = (PH * 20 + 15) / 16;
    constexpr size_t lds_loop = (size_t)(3 * 32 * 512 + (2 * NBP + 1) * 512) * 2, lds_out = (size_t)TH * 16 * 136 * 2, lds_red = 32 * 16 * 16 * 4;
    constexpr size_t lds = lds_loop > lds_out ? (lds_loop > lds_red ? lds_loop : lds_red) : (lds_out > lds_red ? lds_out : lds_red);
    static_assert(lds <= 160 * 1024, "LDS");
    k.tiles_x = hv_cdiv(k.Wc, 16);
    k.tiles = k.tiles_x * hv_cdiv(k.Hc, TH);
    static const int spread = getenv("HV_G4S1_SPREAD") ? atoi(getenv("HV_G4S1_SPREAD")) : 1;      // A/B knob (same bits): LDS-DMA pieces issued tap by tap
    auto kern = spread ? conv_g4s1_kernel<DG, MT> : conv_g4s1_headissue_kernel<DG, MT>;
    static bool raised = false;
    if (!raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return -1000 - (int)e;
        raised = true;
    }
    hv_path_note = 8;
    HV_KNAME("conv_g4s1_kernel<%d, %d>", DG, MT);
    HV_WUSE(4);
    hipLaunchKernelGGL(kern, dim3(k.tiles * k.B, ny), dim3(512), lds, s, k);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

template <int MODE, int MT>
static int launch_g4(G4K& k, int ny, hipStream_t s) {
    constexpr int TH = 4 * MT, PH = TH + 1 + MODE, PW = 17 + MODE;
    constexpr size_t lds_loop = (size_t)(3 * 32 * 512 + (3 * ((PH * PW + 15) / 16) + 1) * 512) * 2, lds_out = (size_t)TH * 16 * 136 * 2, lds_red = 32 * 16 * 16 * 4;
    constexpr size_t lds = lds_loop > lds_out ? (lds_loop > lds_red ? lds_loop : lds_red) : (lds_out > lds_red ? lds_out : lds_red);
    static_assert(lds <= 160 * 1024, "LDS");
    k.tiles_x = hv_cdiv(k.Wc, 16);
    k.tiles = k.tiles_x * hv_cdiv(k.Hc, TH);
    auto kern = conv_g4_kernel<MODE, MT>;
    static bool raised = false;
    if (!raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return -1000 - (int)e;
        raised = true;
    }
    hv_path_note = 8;
    HV_KNAME("conv_g4_kernel<%d, %d>", MODE, MT);
    HV_WUSE(4);
    hipLaunchKernelGGL(kern, dim3(k.tiles * k.B, ny), dim3(512), lds, s, k);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// workgroups of a launch and floats of its statistics output (hv_conv2d_stats_floats)
static int g4_tile_rows(int B, int Hc, int Wc, int ny) {
    static const int force = getenv("HV_G4_MT") ? atoi(getenv("HV_G4_MT")) : 0;
    if (force == 2 || force == 4) return 4 * force;
    return (long long)B * hv_cdiv(Hc, 16) * hv_cdiv(Wc, 16) * ny >= 200 ? 16 : 8;
}

// The ONE eligibility predicate of the pipelined 4x4 kernels: 4x4, stride 2 or 1, pad 1, dilation 1, fp16 NHWC views with 16-byte aligned channel rows,
// fragment-ordered filters.  hv_conv2d_g4 launches exactly when it holds, and hv_conv2d_g4_stats_floats promises a statistics epilogue only then.
size_t hv_conv2d_g4_bstats_parts(const hv_conv_desc* d);
static bool g4_eligible(const hv_conv_desc* d) {
    static const int on = getenv("HV_CONV_G4") ? atoi(getenv("HV_CONV_G4")) : 3;      // bit 0: forward, bit 1: data gradient
    if (!(on & (d->transposed ? 2 : 1))) return false;
    if (d->KH != 4 || d->KW != 4 || (d->stride != 2 && d->stride != 1) || d->pad != 1 || d->dil != 1 || d->in_shift || d->w_bstride || d->ch_scale) return false;
    if (d->precision != HV_F16 || !d->w_f16_tiled || !d->x_f16 || !d->y_f16 || d->accumulate > 1) return false;
    if (d->stride == 1) {
        static const int s1 = getenv("HV_CONV_G4S1") ? atoi(getenv("HV_CONV_G4S1")) : 3;      // bit 0: forward, bit 1: data gradient
        if (!(s1 & (d->transposed ? 2 : 1)) || (d->Cout & 127) || (d->Cin & 31)) return false;
    }
    if ((d->Cin & 31) || (d->Cout & 63) || (!d->transposed && (d->Cout & 127))) return false;
    if ((d->x_ld & 7) || (d->x_coff & 7) || ((uintptr_t)d->x & 15) || (d->y_ld & 7) || (d->y_coff & 7) || ((uintptr_t)d->y & 15) || ((uintptr_t)d->w_f16_tiled & 15))
        return false;
    if (d->mul_src && (!d->mul_f16 || (d->mul_ld & 7) || (d->mul_coff & 7) || ((uintptr_t)d->mul_src & 15))) return false;
    if (d->stride == 2 && !d->transposed && (d->Ho != d->H / 2 || d->Wo != d->W / 2 || (d->H & 1) || (d->W & 1))) return false;
    if (d->stride == 2 && d->transposed && (d->Ho != 2 * d->H || d->Wo != 2 * d->W)) return false;
    if (d->stride == 1 && (d->Ho != d->H + (d->transposed ? 1 : -1) || d->Wo != d->W + (d->transposed ? 1 : -1) || d->Ho < 1 || d->Wo < 1)) return false;
    if ((long long)d->B * d->H * d->W * d->x_ld >= (1ll << 30) || (long long)d->B * d->Ho * d->Wo * d->y_ld >= (1ll << 30)) return false;
    return true;
}

int hv_conv2d_g4(const hv_conv_desc* d, hipStream_t s) {
    if (!g4_eligible(d)) return HV_ERR_UNSUPPORTED;
    G4K k;
    k.dbg = getenv("HV_G4_DBG") ? atoi(getenv("HV_G4_DBG")) : 0;
    k.x = d->x; k.w = reinterpret_cast<const _Float16*>(d->w_f16_tiled); k.bias = d->bias; k.y = d->y; k.mul_src = d->mul_src;
    k.stats = d->transposed ? nullptr : d->stats;
    k.bn_x = nullptr; k.bn_stats = nullptr; k.bstats = nullptr; k.bn_x_ld = k.bn_x_coff = 0; k.bn_ipg = 1;
    if (d->bstats) {
        if (!hv_conv2d_g4_bstats_parts(d)) return HV_ERR_UNSUPPORTED;
        k.bn_x = d->bn_x; k.bn_stats = d->bn_stats; k.bstats = d->bstats; k.bn_x_ld = d->bn_x_ld; k.bn_x_coff = d->bn_x_coff;
        k.bn_ipg = d->B / (d->bn_groups > 0 ? d->bn_groups : 1);
    }
    k.B = d->B; k.H = d->H; k.W = d->W; k.x_ld = d->x_ld; k.x_coff = d->x_coff; k.Cin = d->Cin;
    k.Ho = d->Ho; k.Wo = d->Wo; k.y_ld = d->y_ld; k.y_coff = d->y_coff; k.Cout = d->Cout;
    k.mul_ld = d->mul_ld; k.mul_coff = d->mul_coff; k.mul_act = d->mul_act; k.act = d->act; k.accumulate = d->accumulate; k.alpha = d->alpha;
    k.x_bytes = (unsigned)((size_t)d->B * d->H * d->W * d->x_ld * 2);
    k.w_rb = (unsigned)(16 * 16 * d->Cin * 2);
    k.w_bytes = (unsigned)((size_t)hv_cdiv(d->Cout, 16) * k.w_rb);
    if (d->stride == 1) {
        k.Hc = d->Ho; k.Wc = d->Wo;
        const int ny = d->Cout / 128;
        k.stats = d->transposed ? nullptr : d->stats;
        if (g4_tile_rows(d->B, k.Hc, k.Wc, ny) == 16) return d->transposed ? launch_g4s1<1, 4>(k, ny, s) : launch_g4s1<0, 4>(k, ny, s);
        return d->transposed ? launch_g4s1<1, 2>(k, ny, s) : launch_g4s1<0, 2>(k, ny, s);
    }
    if (!d->transposed) {
        k.Hc = d->Ho; k.Wc = d->Wo;
        const int ny = d->Cout / 128;
        return g4_tile_rows(d->B, k.Hc, k.Wc, ny) == 16 ? launch_g4<0, 4>(k, ny, s) : launch_g4<0, 2>(k, ny, s);
    }
    k.Hc = d->H; k.Wc = d->W;
    const int ny = 2 * (d->Cout / 64);
    return g4_tile_rows(d->B, k.Hc, k.Wc, ny) == 16 ? launch_g4<1, 4>(k, ny, s) : launch_g4<1, 2>(k, ny, s);
}

// parts of the batch-norm backward sums (hv_conv_desc.bstats) the data-gradient launches of hv_conv2d_g4 write: one per workgroup along x (stride 1) or four
// (the output-parity classes, stride 2); 0 = no such epilogue for this descriptor
size_t hv_conv2d_g4_bstats_parts(const hv_conv_desc* d) {
    if (!d->transposed || d->accumulate || !g4_eligible(d)) return 0;
    if (!d->bn_x || !d->bn_stats || !d->mul_src || (d->bn_x_ld & 7) || (d->bn_x_coff & 7) || ((uintptr_t)d->bn_x & 15) || ((uintptr_t)d->bn_stats & 15) || (d->Cout & 7)) return 0;
    const int G = d->bn_groups > 0 ? d->bn_groups : 1;
    if (d->B % G) return 0;
    if ((long long)d->B * d->Ho * d->Wo * d->bn_x_ld >= (1ll << 30)) return 0;
    if (d->stride == 1) {
        const int th = g4_tile_rows(d->B, d->Ho, d->Wo, d->Cout / 128);
        return (size_t)d->B * hv_cdiv(d->Ho, th) * hv_cdiv(d->Wo, 16);
    }
    const int th = g4_tile_rows(d->B, d->H, d->W, 2 * (d->Cout / 64));
    return (size_t)d->B * hv_cdiv(d->H, th) * hv_cdiv(d->W, 16) * 4;
}

// floats of the statistics partials hv_conv2d writes for this descriptor when d->stats is set: [workgroups along x][Cout][2]; 0 = this shape's
// kernel has no statistics epilogue (the caller runs its reduction pass)
size_t hv_conv2d_g4_stats_floats(const hv_conv_desc* d, int* nparts) {
    hv_conv_desc t = *d;
    // the forward launches of hv_conv2d_g4 (same predicate: alignment, size limits and the act' operand included) that assign their output
    if (t.transposed || t.accumulate || !g4_eligible(&t)) return 0;
    const int th = g4_tile_rows(t.B, t.Ho, t.Wo, t.Cout / 128);
    const int parts = t.B * hv_cdiv(t.Ho, th) * hv_cdiv(t.Wo, 16);
    if (nparts) *nparts = parts;
    return (size_t)parts * t.Cout * 2;
}

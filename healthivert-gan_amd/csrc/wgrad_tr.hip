// Weight gradient on fp16-stored tensors with TRANSPOSED LDS READS (gfx950 ds_read_b64_tr_b16): no register transposes, no conversions,
// the input patch staged once per pixel tile for all taps.
//
//   dW[co][(r,q)][ci] = sum_{n,oy,ox} g[n,oy,ox,co] * x[n, oy*S - pad + r, ox*S - pad + q, ci]
//
// As a GEMM the contraction index k is the PIXEL, while both operands are stored channel-contiguous (NHWC): the 16x16x32 MFMA wants, per
// lane, 8 consecutive k of one channel -- a transposed access.  The gather kernel (conv_igemm.hip: wgrad_kernel) transposes in registers
// (8 loads + 32 moves per staged item, 8.7 VALU instructions per MFMA on the PatchGAN layers) and re-reads x once per tap through L2.
// Here the tiles sit in LDS exactly as they sit in memory,
//     G[8x16 output pixels][BN channels],   X[(7*S+KS) x (15*S+KS) input pixels][BC channels]     (fp16, row = pixel)
// staged by plain 16-byte copies, and every MFMA operand is two ds_read_b64_tr_b16: a 16-lane group reads a block of 4 pixel rows x 16
// channels and each lane receives one channel's 4 pixels.  The instruction takes one ADDRESS PER LANE (block row = pixel), so the tap
// shift (r, q) of the input operand is nothing but another constant added to the lanes' patch addresses: x is staged once for all KS*KS
// taps.  k-step = 2 tile rows x 16 pixels: group g's first read takes pixels (ty, 4g .. 4g+3), its second (ty+1, 4g .. 4g+3); a
// 32-lane half therefore reads 8 consecutive pixel rows, which are conflict-free with a row stride = 32 B x odd (stride 1) or 16 B x odd
// (stride 2: every other patch pixel).
// The taps are dealt round-robin to the four waves (wave w owns taps w, w+4, ...): each wave keeps (BN/16)(BC/16)ceil(KS^2/4)
// accumulator tiles and reads every A fragment once per k-step.  One slab per workgroup, summed by wgrad_reduce_kernel (fixed order).
#include <stdlib.h>

#include <type_traits>

#include "hv_common.h"

typedef __fp16 hv_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct WTrK {
    const _Float16* x; const _Float16* g; float* slabs;
    int Hl, Wl, in_shift, Wp, img_stride, x_ld, x_coff, Cin;
    int Ho, Wo, g_ld, g_coff, Cout, pad;
    int tiles_x, tiles_per_img, ntiles;
    int dil, tiles_per_sub;      // dilation d > 1 (3x3, stride 1, pad d): the layer as d*d undilated convolutions on the residue sub-grids (Ho/d x Wo/d pixels, step d);
                                 // tiles_per_img = d*d * tiles_per_sub, Ho / Wo / Hl / Wl are the SUB-grid extents and Hf / Wf the full ones
    int Hf, Wf;
    long long slab;              // floats per slab = Cout * KS*KS * Cin
    unsigned x_bytes, g_bytes;   // buffer descriptor ranges
    float* bias_out;             // per-workgroup column sums of g (bias gradient), [gridDim.x][Cout], or NULL
    int dbg;                     // diagnostic builds (WT_STAMPS), timing only: bit 0 no DMA after the first tile, bit 1 no MFMAs, bit 2 no fragment reads
    int gx;                      // x extent of the main grid (pixel chunks = slabs); wgrad_tr_kernel: blocks beyond it run the carried fold
    HvFold fold;                 // the previous weight gradient's slab fold, carried along (hv_wgrad_desc.carry; splits == 0: none)
};

__device__ __forceinline__ f16x4 tr_read(const _Float16* lds_addr) {
    typedef hv_fp16x4 __attribute__((address_space(3)))* lp;
    return __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(lds_addr)));
}

// row strides (bytes) of the LDS images: see the bank argument above
__host__ __device__ constexpr int wtr_stride(int ch, int st) {
    const int row = ch * 2;
    if (st == 1) return (row / 32) % 2 ? row : row + 32;                 // 32 B x odd
    return row + 16;                                                     // 16 B x odd (row is a multiple of 32)
}

template <int KS, int ST, int BN, int BC>
__global__ __launch_bounds__(256, 2) void wgrad_tr_kernel(const WTrK p) {
    constexpr int TH = 8, TW = 16, TAPS = KS * KS, SLOTS = (TAPS + 3) / 4;
    constexpr int PH = (TH - 1) * ST + KS, PW = (TW - 1) * ST + KS;
    constexpr int NT = BN / 16, CT = BC / 16;
    constexpr int SG = wtr_stride(BN, 1), SX = wtr_stride(BC, ST);       // bytes per pixel row
    constexpr int GI = TH * TW * (BN / 8), XI = PH * PW * (BC / 8);       // 16-byte staging items
    constexpr int GPT = (GI + 255) / 256, XPT = (XI + 255) / 256;
    static_assert(NT * CT * SLOTS <= 32, "accumulator budget");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Gs = smem;                               // [TH*TW][SG]
    char* Xs = smem + TH * TW * SG;                // [PH*PW][SX]

    if ((int)blockIdx.x >= p.gx) {      // (block-uniform) the carried fold's workgroups
        const int fx = (int)gridDim.x - p.gx;
        hv_fold_blocks(p.fold, ((int)blockIdx.x - p.gx) + fx * ((int)blockIdx.y + (int)gridDim.y * (int)blockIdx.z), fx * (int)gridDim.y * (int)gridDim.z, smem);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int co0 = blockIdx.y * BN, ci0 = blockIdx.z * BC;
    const int grp = lane >> 4, sub = lane & 15, qr = sub >> 2, pc = sub & 3;

    f32x4 acc[SLOTS][NT][CT];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[s][n][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool do_bias = p.bias_out != nullptr && blockIdx.z == 0 && wave == 0;      // wave-uniform
    f32x4 bacc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) bacc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // per-lane LDS addresses of the transposed reads: block row = pixel (4*grp + qr) of a tile row, 4 channels at 8*pc bytes
    const _Float16* ga = reinterpret_cast<const _Float16*>(Gs + (4 * grp + qr) * SG + pc * 8);
    const _Float16* xa[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int t = wave + 4 * s, r = t / KS, q = t - r * KS;       // this wave's tap of slot s (t >= TAPS: unused)
        xa[s] = reinterpret_cast<const _Float16*>(Xs + ((4 * grp + qr) * ST + (t < TAPS ? r * PW + q : 0)) * SX + pc * 8);
    }

    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t gsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.g), 0, p.g_bytes, 0x00020000);
    u32x4 rg[GPT], rx[XPT];
    auto prefetch = [&](int tile) __attribute__((always_inline)) {
        const int n_img = tile / p.tiles_per_img;
        int tr = tile - n_img * p.tiles_per_img;
        // dilation d: tile of residue sub-grid (ry, rx); sub-grid pixel (y, x) is real pixel (ry + d y, rx + d x) of both tensors (d = 1: ry = rx = 0)
        const int d = p.dil, sub = tr / p.tiles_per_sub;
        tr -= sub * p.tiles_per_sub;
        const int ry = sub / d, rxo = sub - ry * d;
        const int oy0 = (tr / p.tiles_x) * TH, ox0 = (tr % p.tiles_x) * TW;
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const int e = tid + i * 256;
            const int c8 = e % (BN / 8), pix = e / (BN / 8), ty = pix / TW, tx = pix - ty * TW;
            const int oy = oy0 + ty, ox = ox0 + tx, co = co0 + c8 * 8;
            const bool ok = e < GI && oy < p.Ho && ox < p.Wo && co < p.Cout;
            rg[i] = __builtin_amdgcn_raw_buffer_load_b128(gsrc, ok ? (unsigned)(((n_img * p.Hf + ry + d * oy) * p.Wf + rxo + d * ox) * p.g_ld + p.g_coff + co) * 2u : 0x80000000u, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int e = tid + i * 256;
            const int c8 = e % (BC / 8), pix = e / (BC / 8), py = pix / PW, px = pix - py * PW;
            const int hi = oy0 * ST - p.pad + py, wi = ox0 * ST - p.pad + px, ci = ci0 + c8 * 8;
            const bool ok = e < XI && (unsigned)hi < (unsigned)p.Hl && (unsigned)wi < (unsigned)p.Wl && ci < p.Cin;
            rx[i] = __builtin_amdgcn_raw_buffer_load_b128(
                xsrc, ok ? (unsigned)(n_img * p.img_stride + (((ry + d * hi) >> p.in_shift) * p.Wp + ((rxo + d * wi) >> p.in_shift)) * p.x_ld + p.x_coff + ci) * 2u : 0x80000000u, 0, 0);
        }
    };
    auto flush = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const int e = tid + i * 256;
            if (e < GI) *reinterpret_cast<u32x4*>(Gs + (e / (BN / 8)) * SG + (e % (BN / 8)) * 16) = rg[i];
        }
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int e = tid + i * 256;
            if (e < XI) *reinterpret_cast<u32x4*>(Xs + (e / (BC / 8)) * SX + (e % (BC / 8)) * 16) = rx[i];
        }
    };
    const f16x8 ones = {1, 1, 1, 1, 1, 1, 1, 1};

    if ((int)blockIdx.x < p.ntiles) prefetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < p.ntiles; tile += p.gx) {
        __syncthreads();   // previous tile's reads are done
        flush();
        __syncthreads();
        if (tile + p.gx < p.ntiles) prefetch(tile + p.gx);   // next tile's loads fly behind this tile's MFMAs
        // software pipeline over the (k-step, tap slot) pairs of the tile: the transposed LDS reads of the next pair are issued before the MFMAs of
        // the current one (with the reads right in front of their MFMAs a wave alternated ~130 cycles of LDS latency with 128 cycles of MFMAs; at
        // one workgroup per CU nothing else filled the gaps)
        auto lda = [&](int ks, f16x8 (&a)[NT]) __attribute__((always_inline)) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const f16x4 lo = tr_read(ga + ((2 * ks) * TW * SG + n * 32) / 2), hi = tr_read(ga + ((2 * ks + 1) * TW * SG + n * 32) / 2);
                a[n] = (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        };
        auto ldb = [&](int ks, int sl, f16x8 (&bb)[CT]) __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const f16x4 lo = tr_read(xa[sl] + ((2 * ks) * ST * PW * SX + c * 32) / 2), hi = tr_read(xa[sl] + ((2 * ks + 1) * ST * PW * SX + c * 32) / 2);
                bb[c] = (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        };
        f16x8 aq[2][NT], bq[2][CT];
        lda(0, aq[0]);
        ldb(0, 0, bq[0]);
#pragma unroll
        for (int ks = 0; ks < TH / 2; ++ks) {
            if (do_bias) {
#pragma unroll
                for (int n = 0; n < NT; ++n) bacc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aq[ks & 1][n], ones, bacc[n], 0, 0, 0);
            }
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
                const int idx = ks * SLOTS + sl;
                __builtin_amdgcn_sched_barrier(0);
                if (sl + 1 < SLOTS) ldb(ks, sl + 1, bq[(idx + 1) & 1]);
                else if (ks + 1 < TH / 2) { lda(ks + 1, aq[(ks + 1) & 1]); ldb(ks + 1, 0, bq[(idx + 1) & 1]); }
                __builtin_amdgcn_sched_barrier(0);
                if (wave + 4 * sl < TAPS) {     // wave-uniform (scalar) branch: MFMA ignores EXEC
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int c = 0; c < CT; ++c) acc[sl][n][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aq[ks & 1][n], bq[idx & 1][c], acc[sl][n][c], 0, 0, 0);
                }
            }
        }
    }
    // ---- one slab per workgroup; D layout: row (= co) = (lane>>4)*4 + r, col (= ci) = lane & 15
    float* out = p.slabs + (long long)blockIdx.x * p.slab;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int t = wave + 4 * s;
        if (t >= TAPS) continue;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const int ci = ci0 + c * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + n * 16 + (lane >> 4) * 4 + r;
                    if (co < p.Cout && ci < p.Cin) out[((long long)co * TAPS + t) * p.Cin + ci] = acc[s][n][c][r];
                }
            }
    }
    if (do_bias && (lane & 15) == 0) {     // every column of the ones-product holds the row sums: take column 0
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + n * 16 + (lane >> 4) * 4 + r;
                if (co < p.Cout) p.bias_out[(long long)blockIdx.x * p.Cout + co] = bacc[n][r];
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The same product with the tiles brought in by LDS-DMA (buffer_load ... lds, 16 bytes per lane) into TWO buffers: no staging registers, no
// flush pass, one barrier per pixel tile, and the next tile's bytes fly behind this tile's MFMAs.  (In-kernel phase stamps of the kernel above
// on PatchGAN 256 -> 512, 32 tiles per workgroup: 76.6 us loop = 48.0 MFMA section (27 of MFMAs) + 16.5 issuing the prefetch (its index
// arithmetic) + 7.2 flush and barrier + 2.2 first barrier; stride 2: 30 us = 14.8 (6.8) + 8.6 + 4.3 + 0.4.)  The DMA writes 64 x 16 bytes
// lane-linear, so the LDS images cannot carry the odd row strides used above; instead the SOURCE side permutes 16-byte pieces:
//   G   [128 pixels][128 B]:  piece j of tile pixel (ty, tx) holds channel piece j ^ 2 ((tx >> 1) & 3)
//   X   stride 1: [PH][24 pixels][64 B], piece j of patch pixel (py, px) holds channel piece j ^ 2 ((px >> 2) & 1)
//       stride 2: [PH][2 column parities][24 pixels][64 B]: the patch row DE-INTERLEAVED by column parity at DMA time (patch column px sits at
//                 (px & 1, px >> 1)), so a tap's stride-2 pixel walk is a unit-stride walk inside one parity half and the stride-1 image's piece key
//                 (on the half's own index) serves it unchanged.  Round 4 -- before: [PH][40 pixels][32 B] with 16-channel input blocks (a 64-byte row
//                 under a 2-pixel step puts a half-wave's eight rows on two bank octets whatever the piece key); the 32-channel block halves the number
//                 of (filter block, channel block) pairs, i.e. the G tiles' re-reads and the DMA instructions per MFMA
// (patch rows padded to a multiple of 8 pixels so that the keys depend on px alone; the pad pixels are out-of-range lanes of the DMA = zeros,
// never read).  Each transposed read's eight pixel rows of a 32-lane half then cover all 64 banks once for every tap and k-step (brute-forced
// over all (tap, k-step, fragment) address sets).  The per-lane address of a fragment is (lane base) ^ (fragment << 5) + an immediate.
// Index arithmetic of the DMA is hoisted: per item a tile-invariant relative offset + its (py, px); a tile adds one scalar base and four bounds.
// Fragments are double-buffered over stages of 8 MFMAs (the next stage's reads are issued before the current stage's MFMAs).
// The DMA is issued from inline assembly: through the builtin the compiler knows that LDS is written and puts s_waitcnt vmcnt(0) in front of
// the next transposed read (it cannot tell that the read goes to the OTHER buffer), i.e. it waited for the next tile's bytes before the first
// MFMA of this one.  Here only the explicit s_waitcnt vmcnt(0) in front of the tile barrier orders the DMA.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wtr_dma16(i32x4 rsrc, unsigned lds_byte, unsigned voff) {      // 16 bytes per lane -> LDS at lds_byte + 16 * lane; out-of-range lanes write zeros
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(__builtin_amdgcn_readfirstlane(lds_byte)), "v"(voff), "s"(rsrc) : "memory", "m0");
}
__device__ __forceinline__ i32x4 wtr_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    return (i32x4){(int)__builtin_amdgcn_readfirstlane((unsigned)a), (int)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu),
                   (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000};
}

template <int KS, int ST>
struct WTrdCfg {
    static constexpr int BN = 64, BC = 32, TH = 8, TW = 16, TAPS = KS * KS, SLOTS = (TAPS + 3) / 4;
    static constexpr int PH = (TH - 1) * ST + KS, PW = (TW - 1) * ST + KS, PWP = ST == 1 ? 24 : 48;      // stride 2: two parity halves of 24
    static constexpr int NT = BN / 16, CT = BC / 16, XPR = BC / 8, XROW = BC * 2;
    static constexpr int XI = PH * PWP * XPR, XINS = (XI + 63) / 64, XPT = (XINS + 3) / 4;
    static constexpr int GBUF = TH * TW * 128, XBUF = XINS * 1024, BUFSZ = GBUF + XBUF;
    static constexpr int SPS = CT == 2 ? 2 : 4;          // tap slots per stage (16 MFMAs)
};

template <int KS, int ST>
__global__ __launch_bounds__(512, 1) void wgrad_trd_kernel(const WTrK p) {
    using C = WTrdCfg<KS, ST>;
    constexpr int BN = C::BN, BC = C::BC, TH = C::TH, TW = C::TW, TAPS = C::TAPS, SLOTS = C::SLOTS, PW = C::PW, PWP = C::PWP;
    constexpr int NT = C::NT, CT = C::CT, XPR = C::XPR, XROW = C::XROW, XI = C::XI, XINS = C::XINS, XPT = C::XPT, GBUF = C::GBUF, BUFSZ = C::BUFSZ;
    constexpr int SPS = SLOTS < C::SPS ? SLOTS : C::SPS, NSTG = (SLOTS + SPS - 1) / SPS;     // stages per k-step
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [2][GBUF | XBUF]
    // waves 0-3 run the MFMAs (one per SIMD, the taps dealt round-robin as above); waves 4-7 are the loaders: an LDS-DMA instruction holds the issuing
    // wave for ~100-200 cycles while the memory pipe takes its 8-16 lines (stamps: 9 instructions per wave and tile = 1.05 us against 1.43 us of
    // MFMAs at the clock the chip holds under this load), so the issue sits in a wave that has nothing else to do
    const int tid = threadIdx.x, lane = tid & 63, wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave8 >= 4;
    const int wave = wave8 & 3;
    const int co0 = blockIdx.y * BN, ci0 = blockIdx.z * BC;
    const int grp = lane >> 4, sub = lane & 15, qr = sub >> 2, pc = sub & 3, tx_l = 4 * grp + qr;

    f32x4 acc[SLOTS][NT][CT];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[s][n][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool do_bias = p.bias_out != nullptr && blockIdx.z == 0 && wave8 == 4;      // wave-uniform: the first loader wave also sums g (the bias gradient)
    f32x4 bacc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) bacc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- per-lane LDS byte addresses of the transposed reads (buffer 0; buffer 1 and the k-step rows are immediates)
    const int ga0 = tx_l * 128 + (((tx_l >> 1) & 3) << 5) + pc * 8;                  // fragment n: ga0 ^ (n << 5)
    int xa0[SLOTS];                                                                  // fragment c: xa0 ^ (c << 5)
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int t = wave + 4 * s, tt = t < TAPS ? t : 0, r = tt / KS, q = tt - r * KS;
        // stride 2: patch column 2 tx + q lies in parity half q & 1 at index u = tx + (q >> 1)
        const int u = ST == 1 ? tx_l + q : tx_l + (q >> 1), half = ST == 1 ? 0 : (q & 1) * 24;
        xa0[s] = GBUF + (r * PWP + half + u) * XROW + (((u >> 2) & 1) << 5) + pc * 8;
    }

    // ---- tile-invariant part of the DMA items: wave w issues G instructions 4w .. 4w+3 and X instructions w, w+4, ...
    constexpr int NOPE = -(1 << 30);
    int grel[4], xrel[XPT], xpp[XPT];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = (wave * 4 + i) * 64 + lane, pix = e >> 3, j = e & 7, ty = pix >> 4, tx = pix & 15;
        const int co = co0 + ((j ^ (((tx >> 1) & 3) << 1)) << 3);
        grel[i] = co < p.Cout ? (ty * p.Wo + tx) * p.g_ld + p.g_coff + co : NOPE;
    }
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
        const int e = (wave + 4 * i) * 64 + lane, row = e / XPR, j = e % XPR;
        const int py = row / PWP, rem = row - py * PWP;
        const int u = ST == 1 ? rem : rem % 24, px = ST == 1 ? rem : 2 * u + rem / 24;      // LDS slot (py, [parity,] u) <- patch pixel (py, px)
        const int ci = ci0 + ((j ^ (((u >> 2) & 1) << 1)) << 3);
        const bool ok = e < XI && px < PW && ci < p.Cin;
        xrel[i] = ok ? (py * p.Wl + px) * p.x_ld + p.x_coff + ci : NOPE;
        xpp[i] = (py << 8) | px;
    }
    const i32x4 xsrc = wtr_rsrc(p.x, p.x_bytes), gsrc = wtr_rsrc(p.g, p.g_bytes);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // the DMA of one tile: 4 + XPT instructions per loader wave
    struct TileAt { int hmax, wmax, gbase, y0, x0, xbase; };
    auto tile_at = [&](int tile) __attribute__((always_inline)) {
        TileAt t;
        const int n_img = tile / p.tiles_per_img, tr = tile - n_img * p.tiles_per_img;
        const int ty0 = tr / p.tiles_x, oy0 = ty0 * TH, ox0 = (tr - ty0 * p.tiles_x) * TW;
        t.hmax = p.Ho - oy0; t.wmax = p.Wo - ox0;
        t.gbase = ((n_img * p.Ho + oy0) * p.Wo + ox0) * p.g_ld;
        t.y0 = oy0 * ST - p.pad; t.x0 = ox0 * ST - p.pad;
        t.xbase = n_img * p.img_stride + (t.y0 * p.Wl + t.x0) * p.x_ld;
        return t;
    };
    constexpr int NPART = 4 + XPT;
    auto issue_part = [&](const TileAt& t, int buf, int part) __attribute__((always_inline)) {
        const unsigned dstb = lds0 + buf * BUFSZ;
        if (part < 4) {
            const int i = part, k = wave * 4 + i, pix = k * 8 + (lane >> 3), ty = pix >> 4, tx = pix & 15;
            const bool ok = grel[i] >= 0 && ty < t.hmax && tx < t.wmax;
            wtr_dma16(gsrc, dstb + k * 1024, ok ? (unsigned)(t.gbase + grel[i]) * 2u : 0x80000000u);
        } else {
            const int i = part - 4, k = wave + 4 * i;
            if (k < XINS) {
                const int hi = t.y0 + (xpp[i] >> 8), wi = t.x0 + (xpp[i] & 255);
                const bool ok = xrel[i] > NOPE / 2 && (unsigned)hi < (unsigned)p.Hl && (unsigned)wi < (unsigned)p.Wl;
                wtr_dma16(xsrc, dstb + GBUF + k * 1024, ok ? (unsigned)(t.xbase + xrel[i]) * 2u : 0x80000000u);
            }
        }
    };

    const f16x8 ones = {1, 1, 1, 1, 1, 1, 1, 1};
    // fragments of one stage: A halves (the k-step's NT fragments are fetched NT / NSTG per stage of the PREVIOUS k-step) and SPS x CT B fragments
    f16x8 aq[2][NT], bq[2][SPS][CT];
    // (one loop body for both buffers: with the body unrolled per buffer the two loop exits cost a shuffle of all accumulators through scratch)
    auto lda = [&](int BUFC, int ks, int n, f16x8 (&a)[NT]) __attribute__((always_inline)) {
        const _Float16* base = reinterpret_cast<const _Float16*>(smem + ((ga0 + BUFC) ^ (n << 5)));
        const f16x4 lo = tr_read(base + ((2 * ks) * TW * 128) / 2), hi = tr_read(base + ((2 * ks + 1) * TW * 128) / 2);
        a[n] = (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    auto ldb = [&](int BUFC, int ks, int sl, f16x8 (&bb)[CT]) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const _Float16* base = reinterpret_cast<const _Float16*>(smem + ((xa0[sl] + BUFC) ^ (c << 5)));
            const f16x4 lo = tr_read(base + ((2 * ks) * ST * PWP * XROW) / 2), hi = tr_read(base + ((2 * ks + 1) * ST * PWP * XROW) / 2);
            bb[c] = (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
    };
    // stage index g = ks * NSTG + j over the tile; loads of stage g + 1 go in front of the MFMAs of stage g
    auto load_stage = [&](int BUFC, int g) __attribute__((always_inline)) {
        const int ks = g / NSTG, j = g - ks * NSTG;
#pragma unroll
        for (int u = 0; u < SPS; ++u)
            if (j * SPS + u < SLOTS) ldb(BUFC, ks, j * SPS + u, bq[g & 1][u]);
    };
    auto load_a_part = [&](int BUFC, int ks, int j) __attribute__((always_inline)) {     // part j of NSTG of k-step ks's A fragments
        constexpr int PER = (NT + NSTG - 1) / NSTG;
#pragma unroll
        for (int n = j * PER; n < (j + 1) * PER && n < NT; ++n) lda(BUFC, ks, n, aq[ks & 1]);
    };
    // One stage = the reads of the next stage's fragments dealt one per MFMA among this stage's 16 MFMAs (issued in a block in front of them they
    // kept the MFMA pipe idle for their ~60-100 issue cycles per stage: MFMAs alone 43 us, with the reads in front 54 us on the 256 -> 512 layer).
#define WTD_MIX()                                                                                                                       \
    do {                                                                                                                                \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                             \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                             \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                             \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                             \
    } while (0)
    auto tile_body = [&](int BUFC) __attribute__((always_inline)) {      // BUFC: byte offset of the tile's buffer
        constexpr int NG = (TH / 2) * NSTG;
#pragma unroll
        for (int j = 0; j < NSTG; ++j) load_a_part(BUFC, 0, j);
        load_stage(BUFC, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int ks = g / NSTG, j = g - ks * NSTG;
            __builtin_amdgcn_sched_barrier(0);
#ifdef WT_STAMPS
            if (!(p.dbg & 4))
#endif
            if (g + 1 < NG) {
                load_stage(BUFC, g + 1);
                if (ks + 1 < TH / 2) load_a_part(BUFC, ks + 1, j);
            }
#pragma unroll
            for (int u = 0; u < SPS; ++u) {
                const int sl = j * SPS + u;
#ifdef WT_STAMPS
                if (p.dbg & 2) continue;
#endif
                if (sl < SLOTS && wave + 4 * sl < TAPS) {     // wave-uniform (scalar) branch (folds away for 4 x 4 taps): MFMA ignores EXEC
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int c = 0; c < CT; ++c) acc[sl][n][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aq[ks & 1][n], bq[g & 1][u][c], acc[sl][n][c], 0, 0, 0);
                }
            }
            WTD_MIX(); WTD_MIX(); WTD_MIX(); WTD_MIX();
        }
    };
    // the bias gradient (column sums of g) by the first loader wave, after it has issued the next tile's DMA: ones-products on the G image
    auto bias_tile = [&](int BUFC) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < TH / 2; ++ks) {
            f16x8 a[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) lda(BUFC, ks, n, a);
#pragma unroll
            for (int n = 0; n < NT; ++n) bacc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[n], ones, bacc[n], 0, 0, 0);
        }
    };

#ifdef WT_STAMPS      // diagnostic build only (tools/wtr_stamps.py): phase sums of one workgroup's wave 0, printed at the end
    unsigned long long wt_t[4], wt_sum[3] = {0, 0, 0}, wt_begin = __builtin_amdgcn_s_memrealtime();
#define WTD_STAMP(i) wt_t[i] = __builtin_amdgcn_s_memrealtime()
#define WTD_ACC() do { __builtin_amdgcn_sched_barrier(0); WTD_STAMP(3); for (int i_ = 0; i_ < 3; ++i_) wt_sum[i_] += wt_t[i_ + 1] - wt_t[i_]; } while (0)
#else
#define WTD_STAMP(i)
#define WTD_ACC()
#endif
    const int step = gridDim.x;
    auto issue_tile = [&](int t, int buf) __attribute__((always_inline)) {
        const TileAt at = tile_at(t);
#pragma unroll
        for (int q = 0; q < NPART; ++q) issue_part(at, buf, q);
    };
    // two disjoint loops with the same barrier count (the loaders' registers and the accumulators never share a path); barrier t: buffer t & 1 is whole
    // (each loader waited for its own DMA) and buffer (t + 1) & 1 is no longer read
    if (loader) {
        int tile = blockIdx.x, buf = 0;
        if (tile < p.ntiles) issue_tile(tile, 0);
        for (; tile < p.ntiles; tile += step, buf ^= 1) {
            __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0)
            __builtin_amdgcn_s_barrier();
#ifdef WT_STAMPS
            if (p.dbg & 1) continue;
#endif
            if (tile + step < p.ntiles) issue_tile(tile + step, buf ^ 1);
            if (do_bias) bias_tile(buf * BUFSZ);
        }
    } else {
        int bufoff = 0;
        for (int tile = blockIdx.x; tile < p.ntiles; tile += step, bufoff ^= BUFSZ) {
            WTD_STAMP(0);
            __syncthreads();
            WTD_STAMP(1);
            WTD_STAMP(2);
            tile_body(bufoff);
            WTD_ACC();
        }
    }
#ifdef WT_STAMPS
    const unsigned long long wt_loop_end = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- one slab per workgroup; D layout: row (= co) = (lane>>4)*4 + r, col (= ci) = lane & 15
    float* out = p.slabs + (long long)blockIdx.x * p.slab;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int t = wave + 4 * s;
        if (t >= TAPS || loader) continue;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const int ci = ci0 + c * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + n * 16 + (lane >> 4) * 4 + r;
                    if (co < p.Cout && ci < p.Cin) out[((long long)co * TAPS + t) * p.Cin + ci] = acc[s][n][c][r];
                }
            }
    }
    if (do_bias && (lane & 15) == 0) {     // every column of the ones-product holds the row sums: take column 0
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + n * 16 + (lane >> 4) * 4 + r;
                if (co < p.Cout) p.bias_out[(long long)blockIdx.x * p.Cout + co] = bacc[n][r];
            }
    }
#ifdef WT_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
    if (tid == 0 && blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2 && blockIdx.z == 0)
        printf("wtrd<%d,%d> grid %d x %d x %d: barrier %.2f  tile scalars %.2f  mfma section (+dma issue) %.2f  | loop %.2f  slab write %.2f us\n", KS, ST,
               (int)gridDim.x, (int)gridDim.y, (int)gridDim.z, wt_sum[0] / 100.0, wt_sum[1] / 100.0, wt_sum[2] / 100.0,
               (wt_loop_end - wt_begin) / 100.0, (__builtin_amdgcn_s_memrealtime() - wt_loop_end) / 100.0);
#endif
}

struct WTrPlan { int BN, BC, gx, dil; size_t lds; bool dma; };

static bool wgrad_tr_plan(const hv_wgrad_desc* d, WTrPlan* pl) {
    static const int enabled = getenv("HV_WGRAD_TR") ? atoi(getenv("HV_WGRAD_TR")) : 1;   // A/B knob
    if (!enabled || d->precision != HV_F16 || !d->x_f16 || !d->g_f16 || d->KH != d->KW) return false;
    if (!((d->KH == 3 || d->KH == 4) && (d->stride == 1 || d->stride == 2))) return false;
    pl->dil = 1;
    if (d->dil != 1) {
        // dilated same-size 3x3 layers (the generators' d = 2, 4, 8): d*d undilated weight gradients on the residue sub-grids, summed in the same accumulators
        // (round 4; they ran in the gather kernel at 24.5 us).  d = 16 leaves 4 x 4-pixel sub-grids in 8 x 16 tiles: stays there
        static const int dilated = getenv("HV_WGRAD_TR_DIL") ? atoi(getenv("HV_WGRAD_TR_DIL")) : 1;      // A/B knob
        if (!dilated || (d->dil != 2 && d->dil != 4 && d->dil != 8) || d->KH != 3 || d->stride != 1 || d->pad != d->dil || d->in_shift) return false;
        if (d->H % d->dil || d->W % d->dil || d->Ho != d->H || d->Wo != d->W) return false;
        pl->dil = d->dil;
    } else if (d->Ho != (d->H + 2 * d->pad - d->KH) / d->stride + 1 || d->Wo != (d->W + 2 * d->pad - d->KW) / d->stride + 1) return false;
    // 16-byte staging items: 8-channel groups must be whole and aligned in both tensors
    if ((d->Cin & 7) || (d->Cout & 7) || (d->x_ld & 7) || (d->x_coff & 7) || (d->g_ld & 7) || (d->g_coff & 7)) return false;
    // narrower layers: wgrad_halo_kernel / the gather kernel -- except the 16 -> 16 stride-2 layers, which the gather kernel served at 24.3 us (bs 16, 256 x 256):
    // as half-empty 32-filter blocks here 14.8 us (round 4; at stride 1 the 16-filter layers are as fast in wgrad_halo_kernel)
    if (d->Cout < (d->stride == 2 ? 16 : 32) || d->Cin < 16) return false;
    pl->BN = d->Cout >= 64 ? 64 : 32;
    pl->BC = d->Cin >= 32 ? 32 : 16;
    {   // layers with a small dW (the generators' 64 x 64 x 9): every workgroup writes a whole slab of its tile pair, so 32-filter blocks halve the slab bytes
        // (and double the pixel tiles per workgroup at the same workgroup count)
        static const int bn32 = getenv("HV_WTR_BN32") ? atoi(getenv("HV_WTR_BN32")) : 1;      // A/B knob (three same-box pairs: 8.85 -> 8.82 ms)
        if (bn32 && d->stride == 1 && d->Cout == 64 && d->Cin <= 64 && d->Cin >= 32) pl->BN = 32;
    }
    // (Measured and not kept, round 4: 16 x 16 blocks -- 16 block pairs x 32 pixel chunks, 9-KB slab tiles, 4.7 MB of slabs instead of 18.9 -- for the
    // generators' 32- / 64-channel 3x3 layers: 64 -> 64 @64^2 21.6 -> 29.1 us, 32 -> 32 @128^2 19.4 -> 27.1, 32 -> 32 @256^2 40.2 -> 91.7 at the best of 256 .. 2048
    // workgroups (kernel + slab fold): one A and one B fragment per MFMA makes the LDS reads the bound; the slab bytes were not.)
    const bool s2_dma = d->stride == 2 && pl->BN == 64 && !(d->Cin & 31) && d->in_shift == 0;      // the LDS-DMA form takes 32-channel blocks at both strides
    if (d->stride == 2 && pl->BN == 64 && !s2_dma) pl->BC = 16;      // register-staged form: the stride-2 patch is 3x larger, its prefetch registers leave room for 16 accumulator tiles
    const int PH = 7 * d->stride + d->KH, PW = 15 * d->stride + d->KW;
    pl->lds = (size_t)128 * wtr_stride(pl->BN, 1) + (size_t)PH * PW * wtr_stride(pl->BC, d->stride);
    // the LDS-DMA form: 64-channel output blocks, its fixed input block (32 channels at stride 1, 16 at stride 2), no fused upsampling
    static const int dma_on = getenv("HV_WGRAD_TRD") ? atoi(getenv("HV_WGRAD_TRD")) : 1;   // A/B knob
    pl->dma = dma_on && pl->BN == 64 && pl->BC == 32 && d->in_shift == 0 && pl->dil == 1;
    const bool dma_candidate = pl->dma;
    if (pl->dma) {
        const int PWP = d->stride == 1 ? 24 : 48, XI = PH * PWP * (pl->BC / 8);
        pl->lds = (size_t)2 * (8 * 16 * 128 + (XI + 63) / 64 * 1024);
    }
    const long long ntiles = (long long)d->B * pl->dil * pl->dil * hv_cdiv(d->Ho / pl->dil, 8) * hv_cdiv(d->Wo / pl->dil, 16);
    // workgroups wanted per launch (split over pixel chunks): every chunk writes a whole slab of dW, so a layer with a small dW tile count (the
    // generators' 64-channel layers: 2 tiles, 256 slabs of 147 KB = 38 MB for 17 MB of operands) is bound by its slab traffic, not by its MFMAs
    // (step-level A/B, round 3, same box: PatchGAN layers 512 -> 256 workgroups 9.43 -> 9.25 ms (their slabs are 8 MB each); 192: 9.23; the
    // generators' layers 512 / 256: no difference, 128: +0.2 ms)
    static const int want_big = getenv("HV_WGRAD_TR_WGS") ? atoi(getenv("HV_WGRAD_TR_WGS")) : 256;
    static const int want_small = getenv("HV_WGRAD_TR_WGS_SMALL") ? atoi(getenv("HV_WGRAD_TR_WGS_SMALL")) : 512;
    auto chunks = [&]() {
        const long long pairs = (long long)hv_cdiv(d->Cout, pl->BN) * hv_cdiv(d->Cin, pl->BC);
        // (the small-dW rule by BYTES since round 4: with 32-channel blocks at stride 2 the PatchGAN 64 -> 128 layer has 4 pairs too, and 128 slabs of its
        // 0.5-MB dW would be 67 MB)
        const int want = (pairs <= 4 && (long long)d->Cout * d->KH * d->KW * d->Cin * 4 <= 256 * 1024) ? want_small : want_big;
        long long gx = (want + pairs - 1) / pairs;
        if (gx > ntiles) gx = ntiles;
        if (gx < 1) gx = 1;
        pl->gx = (int)gx;
    };
    chunks();
    // the DMA form pays a longer prologue (8 waves, hoisted DMA indices): it needs a few tiles per workgroup to win (generator 64 -> 64 at 64 x 64, 2 tiles
    // per workgroup: 22.8 us against 16.1)
    if (dma_candidate && ntiles < 4 * pl->gx) {
        pl->dma = false;
        if (d->stride == 2) { pl->BC = 16; chunks(); }      // (the register-staged form's block at stride 2)
        pl->lds = (size_t)128 * wtr_stride(pl->BN, 1) + (size_t)PH * PW * wtr_stride(pl->BC, d->stride);
    }
    return true;
}

size_t hv_wgrad_tr_workspace_bytes(const hv_wgrad_desc* d) {
    WTrPlan pl;
    if (!wgrad_tr_plan(d, &pl)) return 0;
    return (size_t)pl.gx * ((size_t)d->Cout * d->KH * d->KW * d->Cin + (d->dbias ? d->Cout : 0)) * sizeof(float);
}

template <int KS, int ST, int BN, int BC>
static int launch_wtr(const WTrK& k, const WTrPlan& pl, const hv_wgrad_desc* d, hipStream_t s) {
    auto kern = wgrad_tr_kernel<KS, ST, BN, BC>;
    static int lds_limit = 48 * 1024;
    if ((int)pl.lds > lds_limit) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return -1000 - (int)e;
        lds_limit = 150 * 1024;
    }
    dim3 grid(pl.gx, hv_cdiv(d->Cout, BN), hv_cdiv(d->Cin, BC));
    // the previous weight gradient's fold rides along as extra x-blocks of every (y, z) plane (this kernel leaves LDS and registers for a third workgroup per CU)
    WTrK kk = k;
    kk.fold.splits = 0;
    const int fx = hv_carry_blocks((int)(grid.y * grid.z));
    if (fx > 0) { kk.fold = hv_carry; hv_carry_taken = 1; grid.x += fx; }
    hv_path_note = 12;
    HV_KNAME("wgrad_tr_kernel<%d, %d, %d, %d>", KS, ST, BN, BC);
    HV_TIMING_BEGIN(s);
    hipLaunchKernelGGL(kern, grid, dim3(256), pl.lds, s, kk);
    HV_TIMING_END(s);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

template <int KS, int ST>
static int launch_wtrd(const WTrK& k, const WTrPlan& pl, const hv_wgrad_desc* d, hipStream_t s) {
    auto kern = wgrad_trd_kernel<KS, ST>;
    static bool raised = false;
    if (!raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return -1000 - (int)e;
        raised = true;
    }
    dim3 grid(pl.gx, hv_cdiv(d->Cout, 64), hv_cdiv(d->Cin, WTrdCfg<KS, ST>::BC));
    hv_path_note = 13;
    HV_KNAME("wgrad_trd_kernel<%d, %d>", KS, ST);
    HV_TIMING_BEGIN(s);
    hipLaunchKernelGGL(kern, grid, dim3(512), pl.lds, s, k);
    HV_TIMING_END(s);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Thin INPUT (at most 4 channels: the generators' 5x5 stems, the PatchGAN stem), round 5.  With the input padded to a 16-channel block every tap is an MFMA whose
// B operand is 3/4 zeros and whose fragments are read per tap: wgrad_tr_kernel with the 8-byte pixels staged into 16-channel rows (tried first: 58.9 -> 29.8 us) moves
// 131 KB through LDS per 6-KB tile and runs at one tile per microsecond and CU whatever the occupancy and the load depth (wgrad_halo_kernel, which served these
// layers, also stages them as 8-byte loads of 32-byte segments and transposes in registers: 0.9 TB/s).  Here the GEMM's N is (tap, channel): a transposed read takes one ADDRESS PER LANE, the four lanes (qr = 0 .. 3) of a quad share pc, give
// the addresses of four consecutive pixels and receive channel qr of them -- so pc can be a TAP: quad pc reads the 8-byte pixels at tap pc's shift, and the fragment's
// column lane & 15 = 4 pc + e is (tap 4 j + pc, channel e).  25 taps are 7 N tiles instead of 25, the patch sits in LDS as it sits in memory (8 bytes per pixel), and
// the k-steps of a tile (two tile rows each) are dealt to the four waves, so every fragment is read once: 32 KB of LDS reads per tile.  The waves' accumulators
// are summed in LDS in wave order at the end (deterministic); one slab per workgroup as everywhere.
template <int KS, int ST, int BN>
__global__ __launch_bounds__(256, 2) void wgrad_thinx_kernel(const WTrK p) {
    constexpr int TH = 8, TW = 16, TAPS = KS * KS, NTN = (TAPS + 3) / 4, NT = BN / 16;
    constexpr int PH = (TH - 1) * ST + KS, PW = (TW - 1) * ST + KS;
    constexpr int SG = wtr_stride(BN, 1);
    constexpr int GI = TH * TW * (BN / 8), XI = PH * PW;
    constexpr int GPT = (GI + 255) / 256, XPT = (XI + 255) / 256;
    constexpr int XBYTES = (XI * 8 + 15) / 16 * 16;
    static_assert(NT * NTN <= 16, "accumulator budget");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Gs = smem;                               // [TH*TW][SG]
    char* Xs = smem + TH * TW * SG;                // [PH*PW][8 bytes]
    float* red = reinterpret_cast<float*>(smem);   // [NT][NTN][256] (+ [NT][16] bias) after the tile loop

    if ((int)blockIdx.x >= p.gx) {      // (block-uniform) the carried fold's workgroups
        const int fx = (int)gridDim.x - p.gx;
        hv_fold_blocks(p.fold, ((int)blockIdx.x - p.gx) + fx * ((int)blockIdx.y + (int)gridDim.y * (int)blockIdx.z), fx * (int)gridDim.y * (int)gridDim.z, smem);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int co0 = blockIdx.y * BN;
    const int grp = lane >> 4, sub = lane & 15, qr = sub >> 2, pc = sub & 3;
    f32x4 acc[NT][NTN];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int j = 0; j < NTN; ++j) acc[n][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool do_bias = p.bias_out != nullptr;
    f32x4 bacc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) bacc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // this wave's k-step = tile rows 2 wave, 2 wave + 1
    const _Float16* ga = reinterpret_cast<const _Float16*>(Gs + ((2 * wave) * TW + 4 * grp + qr) * SG + pc * 8);
    const _Float16* xb[NTN];
#pragma unroll
    for (int j = 0; j < NTN; ++j) {
        const int t = 4 * j + pc, tt = t < TAPS ? t : 0, r = tt / KS, q = tt - r * KS;      // (a tap beyond the filter: any address, its columns are not stored)
        xb[j] = reinterpret_cast<const _Float16*>(Xs + ((((2 * wave) * ST + r) * PW) + (4 * grp + qr) * ST + q) * 8);
    }
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t gsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.g), 0, p.g_bytes, 0x00020000);
    constexpr int DEPTH = BN > 16 ? 2 : 3;      // tiles of loads in flight (a tile's MFMA section is a fraction of a microsecond; 64-filter blocks: register budget)
    u32x4 rgs[DEPTH][GPT];
    u32x2 rxs[DEPTH][XPT];
    auto prefetch = [&](int tile, auto SI) __attribute__((always_inline)) {
        u32x4 (&rg)[GPT] = rgs[decltype(SI)::value];
        u32x2 (&rx)[XPT] = rxs[decltype(SI)::value];
        const int n_img = tile / p.tiles_per_img, tr = tile - n_img * p.tiles_per_img;
        const int oy0 = (tr / p.tiles_x) * TH, ox0 = (tr % p.tiles_x) * TW;
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const int e = tid + i * 256;
            const int c8 = e % (BN / 8), pix = e / (BN / 8), ty = pix / TW, tx = pix - ty * TW;
            const int oy = oy0 + ty, ox = ox0 + tx, co = co0 + c8 * 8;
            const bool ok = e < GI && oy < p.Ho && ox < p.Wo && co < p.Cout;
            rg[i] = __builtin_amdgcn_raw_buffer_load_b128(gsrc, ok ? (unsigned)(((n_img * p.Ho + oy) * p.Wo + ox) * p.g_ld + p.g_coff + co) * 2u : 0x80000000u, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int e = tid + i * 256;
            const int py = e / PW, px = e - py * PW;
            const int hi = oy0 * ST - p.pad + py, wi = ox0 * ST - p.pad + px;
            const bool ok = e < XI && (unsigned)hi < (unsigned)p.Hl && (unsigned)wi < (unsigned)p.Wl;
            rx[i] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(xsrc, ok ? (unsigned)(n_img * p.img_stride + (hi * p.Wp + wi) * p.x_ld + p.x_coff) * 2u : 0x80000000u, 0, 0));
        }
    };
    auto flush = [&](auto SI) __attribute__((always_inline)) {
        u32x4 (&rg)[GPT] = rgs[decltype(SI)::value];
        u32x2 (&rx)[XPT] = rxs[decltype(SI)::value];
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const int e = tid + i * 256;
            if (e < GI) *reinterpret_cast<u32x4*>(Gs + (e / (BN / 8)) * SG + (e % (BN / 8)) * 16) = rg[i];
        }
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int e = tid + i * 256;
            if (e < XI) *reinterpret_cast<u32x2*>(Xs + e * 8) = rx[i];
        }
    };
    const f16x8 ones = {1, 1, 1, 1, 1, 1, 1, 1};
    auto mfma_tile = [&]() __attribute__((always_inline)) {
        f16x8 a[NT], b[NTN];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const f16x4 lo = tr_read(ga + (n * 32) / 2), hi = tr_read(ga + (TW * SG + n * 32) / 2);
            a[n] = (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int j = 0; j < NTN; ++j) {
            const f16x4 lo = tr_read(xb[j]), hi = tr_read(xb[j] + (ST * PW * 8) / 2);
            b[j] = (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
        if (do_bias) {
#pragma unroll
            for (int n = 0; n < NT; ++n) bacc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[n], ones, bacc[n], 0, 0, 0);
        }
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int j = 0; j < NTN; ++j) acc[n][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[n], b[j], acc[n][j], 0, 0, 0);
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    typedef std::integral_constant<int, (DEPTH > 2 ? 2 : 0)> S2;
    int tile = blockIdx.x;
    if (tile < p.ntiles) prefetch(tile, S0());
    if (DEPTH > 2 && tile + p.gx < p.ntiles) prefetch(tile + p.gx, S1());
    auto one = [&](auto SA, auto SC) __attribute__((always_inline)) {      // the tile staged in SA; the loads of tile + (DEPTH - 1) gx into SC
        __syncthreads();
        flush(SA);
        __syncthreads();
        if (tile + (DEPTH - 1) * p.gx < p.ntiles) prefetch(tile + (DEPTH - 1) * p.gx, SC);
        mfma_tile();
        tile += p.gx;
    };
    while (tile < p.ntiles) {
        if constexpr (DEPTH > 2) {
            one(S0(), S2());
            if (tile >= p.ntiles) break;
            one(S1(), S0());
            if (tile >= p.ntiles) break;
            one(S2(), S1());
        } else {
            one(S0(), S1());
            if (tile >= p.ntiles) break;
            one(S1(), S0());
        }
    }
    // ---- the four waves' sums in wave order; D layout: row (= co) = 4 (lane >> 4) + r, column lane & 15 = (tap 4 j + (column >> 2), channel column & 3)
    __syncthreads();
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
#pragma unroll
                for (int j = 0; j < NTN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* d = red + ((n * NTN + j) * 16 + 4 * (lane >> 4) + r) * 16 + (lane & 15);
                        *d = wv == 0 ? acc[n][j][r] : *d + acc[n][j][r];
                    }
                if (do_bias && (lane & 15) == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* d = red + NT * NTN * 256 + n * 16 + 4 * (lane >> 4) + r;
                        *d = wv == 0 ? bacc[n][r] : *d + bacc[n][r];
                    }
                }
            }
        }
        __syncthreads();
    }
    float* out = p.slabs + (long long)blockIdx.x * p.slab;
    for (int e = tid; e < NT * NTN * 256; e += 256) {
        const int col = e & 15, row = (e >> 4) & 15, nj = e >> 8, j = nj % NTN, n = nj / NTN;
        const int t = 4 * j + (col >> 2), ci = col & 3, co = co0 + n * 16 + row;
        if (t < TAPS && ci < p.Cin && co < p.Cout) out[((long long)co * TAPS + t) * p.Cin + ci] = red[e];
    }
    if (do_bias && tid < BN && co0 + tid < p.Cout) p.bias_out[(long long)blockIdx.x * p.Cout + co0 + tid] = red[NT * NTN * 256 + tid];
}

// Thin GRADIENT (the 1-channel heads: g is a [pixel][4] carrier), the mirror image: the sum runs over INPUT pixels, dW[co][(r, q)][ci] = sum_{iy, ix}
// x[iy][ix][ci] * g[iy + pad - r][ix + pad - q][co], so the shifted 8-byte operand is g (shift KS - 1 - r inside a patch that starts KS - 1 - pad pixels before
// the tile) and the wide one is the x tile; the MFMA's rows are input channels, its columns (tap, co) -- stored transposed into the slab.  Stride 1 only.
// IB: bytes per staging item of x (16, or 8 where the channel stride is not a multiple of 8: the 12-channel head input).
template <int KS, int BN, int IB>
__global__ __launch_bounds__(256, 2) void wgrad_thing_kernel(const WTrK p) {
    constexpr int TH = 8, TW = 16, TAPS = KS * KS, NTN = (TAPS + 3) / 4, NT = BN / 16;
    constexpr int PH = TH + KS - 1, PW = TW + KS - 1;
    constexpr int SX = wtr_stride(BN, 1);
    constexpr int IPP = BN * 2 / IB;               // staging items per pixel
    constexpr int XI = TH * TW * IPP, GI = PH * PW;
    constexpr int XPT = (XI + 255) / 256, GPT = (GI + 255) / 256;
    static_assert(NT * NTN <= 16 && (IB == 8 || IB == 16), "accumulator budget / item size");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Xs = smem;                               // [TH*TW][SX]   the x tile
    char* Ts = smem + TH * TW * SX;                // [PH*PW][8 bytes]   the g patch
    float* red = reinterpret_cast<float*>(smem);   // [NT][NTN][256] after the tile loop

    if ((int)blockIdx.x >= p.gx) {      // (block-uniform) the carried fold's workgroups
        const int fx = (int)gridDim.x - p.gx;
        hv_fold_blocks(p.fold, ((int)blockIdx.x - p.gx) + fx * ((int)blockIdx.y + (int)gridDim.y * (int)blockIdx.z), fx * (int)gridDim.y * (int)gridDim.z, smem);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ci0 = blockIdx.y * BN;
    const int grp = lane >> 4, sub = lane & 15, qr = sub >> 2, pc = sub & 3;
    f32x4 acc[NT][NTN];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int j = 0; j < NTN; ++j) acc[n][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const _Float16* xa = reinterpret_cast<const _Float16*>(Xs + ((2 * wave) * TW + 4 * grp + qr) * SX + pc * 8);
    const _Float16* tb[NTN];
#pragma unroll
    for (int j = 0; j < NTN; ++j) {
        const int t = 4 * j + pc, tt = t < TAPS ? t : 0, r = tt / KS, q = tt - r * KS;
        tb[j] = reinterpret_cast<const _Float16*>(Ts + (((2 * wave) + (KS - 1 - r)) * PW + (4 * grp + qr) + (KS - 1 - q)) * 8);
    }
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t gsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.g), 0, p.g_bytes, 0x00020000);
    constexpr int DEPTH = BN > 16 ? 2 : 3;
    u32x4 rxs[DEPTH][XPT];
    u32x2 rgs[DEPTH][GPT];
    if constexpr (IB == 8 && (IPP & 3)) {}      // (rows shorter than the block: the tail of a row is zeroed below)
    auto prefetch = [&](int tile, auto SI) __attribute__((always_inline)) {
        u32x4 (&rx)[XPT] = rxs[decltype(SI)::value];
        u32x2 (&rg)[GPT] = rgs[decltype(SI)::value];
        const int n_img = tile / p.tiles_per_img, tr = tile - n_img * p.tiles_per_img;
        const int iy0 = (tr / p.tiles_x) * TH, ix0 = (tr % p.tiles_x) * TW;
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int e = tid + i * 256;
            const int it = e % IPP, pix = e / IPP, ty = pix / TW, tx = pix - ty * TW;
            const int iy = iy0 + ty, ix = ix0 + tx, ci = ci0 + it * (IB / 2);
            const bool ok = e < XI && iy < p.Hl && ix < p.Wl && ci < p.Cin;
            const unsigned off = ok ? (unsigned)(n_img * p.img_stride + (iy * p.Wp + ix) * p.x_ld + p.x_coff + ci) * 2u : 0x80000000u;
            if constexpr (IB == 16) rx[i] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, off, 0, 0);
            else { const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(xsrc, off, 0, 0)); rx[i] = u32x4{v.x, v.y, 0u, 0u}; }
        }
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const int e = tid + i * 256;
            const int py = e / PW, px = e - py * PW;
            const int oy = iy0 + p.pad - (KS - 1) + py, ox = ix0 + p.pad - (KS - 1) + px;
            const bool ok = e < GI && (unsigned)oy < (unsigned)p.Ho && (unsigned)ox < (unsigned)p.Wo;
            rg[i] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(gsrc, ok ? (unsigned)(((n_img * p.Ho + oy) * p.Wo + ox) * p.g_ld + p.g_coff) * 2u : 0x80000000u, 0, 0));
        }
    };
    auto flush = [&](auto SI) __attribute__((always_inline)) {
        u32x4 (&rx)[XPT] = rxs[decltype(SI)::value];
        u32x2 (&rg)[GPT] = rgs[decltype(SI)::value];
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int e = tid + i * 256;
            if (e < XI) {
                if constexpr (IB == 16) *reinterpret_cast<u32x4*>(Xs + (e / IPP) * SX + (e % IPP) * 16) = rx[i];
                else *reinterpret_cast<u32x2*>(Xs + (e / IPP) * SX + (e % IPP) * 8) = u32x2{rx[i].x, rx[i].y};
            }
        }
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const int e = tid + i * 256;
            if (e < GI) *reinterpret_cast<u32x2*>(Ts + e * 8) = rg[i];
        }
    };
    auto mfma_tile = [&]() __attribute__((always_inline)) {
        f16x8 a[NT], b[NTN];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const f16x4 lo = tr_read(xa + (n * 32) / 2), hi = tr_read(xa + (TW * SX + n * 32) / 2);
            a[n] = (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int j = 0; j < NTN; ++j) {
            const f16x4 lo = tr_read(tb[j]), hi = tr_read(tb[j] + (PW * 8) / 2);
            b[j] = (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int j = 0; j < NTN; ++j) acc[n][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[n], b[j], acc[n][j], 0, 0, 0);
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    typedef std::integral_constant<int, (DEPTH > 2 ? 2 : 0)> S2;
    int tile = blockIdx.x;
    if (tile < p.ntiles) prefetch(tile, S0());
    if (DEPTH > 2 && tile + p.gx < p.ntiles) prefetch(tile + p.gx, S1());
    auto one = [&](auto SA, auto SC) __attribute__((always_inline)) {
        __syncthreads();
        flush(SA);
        __syncthreads();
        if (tile + (DEPTH - 1) * p.gx < p.ntiles) prefetch(tile + (DEPTH - 1) * p.gx, SC);
        mfma_tile();
        tile += p.gx;
    };
    while (tile < p.ntiles) {
        if constexpr (DEPTH > 2) {
            one(S0(), S2());
            if (tile >= p.ntiles) break;
            one(S1(), S0());
            if (tile >= p.ntiles) break;
            one(S2(), S1());
        } else {
            one(S0(), S1());
            if (tile >= p.ntiles) break;
            one(S1(), S0());
        }
    }
    // ---- the four waves' sums in wave order; D layout: row (= input channel) = 4 (lane >> 4) + r, column lane & 15 = (tap 4 j + (column >> 2), filter column & 3)
    __syncthreads();
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int j = 0; j < NTN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* d = red + ((n * NTN + j) * 16 + 4 * (lane >> 4) + r) * 16 + (lane & 15);
                        *d = wv == 0 ? acc[n][j][r] : *d + acc[n][j][r];
                    }
        }
        __syncthreads();
    }
    float* out = p.slabs + (long long)blockIdx.x * p.slab;
    for (int e = tid; e < NT * NTN * 256; e += 256) {      // (consecutive lanes = consecutive input channels: the slab's fastest index)
        const int row = e & 15, n = (e >> 4) % NT, cj = e / (16 * NT), col = cj & 15, j = cj >> 4;
        const int t = 4 * j + (col >> 2), co = col & 3, ci = ci0 + n * 16 + row;
        if (t < TAPS && co < p.Cout && ci < p.Cin) out[((long long)co * TAPS + t) * p.Cin + ci] = red[((n * NTN + j) * 16 + row) * 16 + col];
    }
}

// ---- thin input: which layers, how many workgroups
struct WThinPlan { int kind, gx; size_t lds; };      // kind 1: 5x5 stride 1, Cin <= 4, Cout <= 16 (the generators' stems); 2 / 3: Cout <= 4, stride 1 -- 4x4 with Cin % 64 == 0 (the PatchGAN head) / 3x3 with Cin <= 16 (the generators' heads)
static bool wgrad_thin_plan(const hv_wgrad_desc* d, WThinPlan* pl) {
    static const int enabled = getenv("HV_WGRAD_THIN") ? atoi(getenv("HV_WGRAD_THIN")) : 3;   // A/B knob: bit 0 thin input (kind 1), bit 1 thin gradient (kinds 2, 3)
    if (!enabled || d->precision != HV_F16 || !d->x_f16 || !d->g_f16 || d->KH != d->KW || d->dil != 1 || d->in_shift != 0) return false;
    if (d->Ho != (d->H + 2 * d->pad - d->KH) / d->stride + 1 || d->Wo != (d->W + 2 * d->pad - d->KW) / d->stride + 1) return false;
    pl->kind = 0;
    const bool thin_x = d->Cin <= 4 && !(d->x_ld & 3) && !(d->x_coff & 3) && !(d->Cout & 7) && !(d->g_ld & 7) && !(d->g_coff & 7);
    const bool thin_g = (enabled & 2) && d->Cout <= 4 && !(d->g_ld & 3) && !(d->g_coff & 3) && d->stride == 1 && !d->dbias && !(d->x_ld & 3) && !(d->x_coff & 3);
    if ((enabled & 1) && thin_x && d->KH == 5 && d->stride == 1 && d->Cout <= 16) pl->kind = 1;
    else if (thin_g && d->KH == 4 && !(d->Cin & 63) && !(d->x_ld & 7) && !(d->x_coff & 7)) pl->kind = 2;
    else if (thin_g && d->KH == 3 && d->Cin <= 16 && d->Cin > 4) pl->kind = 3;
    // (Measured and not kept: the PatchGAN stem -- 4x4 stride 2, 1 -> 64 -- as wgrad_thinx_kernel<4, 2, 64>: 249 registers, two tiles of loads in flight, 33.3 / 26.9 us
    // at B32 / B16 against the gather kernel's 30.2 / 19.1: its time is the 67 MB of g, which both read once.)
    if (!pl->kind) return false;
    const int BN = pl->kind == 2 ? 64 : 16, TAPS = d->KH * d->KW, NTN = (TAPS + 3) / 4;
    const int PH = 7 * d->stride + d->KH, PW = 15 * d->stride + d->KW;
    const size_t stage = (size_t)128 * wtr_stride(BN, 1) + (size_t)(PH * PW * 8 + 15) / 16 * 16, red = (size_t)(BN / 16) * (NTN * 256 + 16) * 4;
    pl->lds = stage > red ? stage : red;
    if (pl->lds < 4096) pl->lds = 4096;      // (the carried fold's workgroups use the launch's LDS)
    // tiles: output pixels (thin input) / input pixels (thin gradient)
    const long long ntiles = pl->kind == 1 ? (long long)d->B * hv_cdiv(d->Ho, 8) * hv_cdiv(d->Wo, 16) : (long long)d->B * hv_cdiv(d->H, 8) * hv_cdiv(d->W, 16);
    static const int want = getenv("HV_WGRAD_THIN_WGS") ? atoi(getenv("HV_WGRAD_THIN_WGS")) : 512;      // tuning knob (kernel + fold, us at 128 / 256 / 384 / 512 / 768 / 1024: 41.5 / 26.1 / 22.0 / 19.7 / 20.9 / 23.8)
    const long long blocks = pl->kind == 2 ? d->Cin / 64 : 1;      // channel blocks along y
    long long gx = want / blocks;
    if (gx < 32) gx = 32;
    pl->gx = (int)(gx < ntiles ? gx : ntiles);
    return true;
}
size_t hv_wgrad_thin_workspace_bytes(const hv_wgrad_desc* d) {
    WThinPlan pl;
    if (!wgrad_thin_plan(d, &pl)) return 0;
    return (size_t)pl.gx * ((size_t)d->Cout * d->KH * d->KW * d->Cin + (d->dbias ? d->Cout : 0)) * sizeof(float);
}
template <int KS, int ST, int BN>
static int launch_wthinx(const WTrK& k, const WThinPlan& pl, const hv_wgrad_desc* d, hipStream_t s) {
    auto kern = wgrad_thinx_kernel<KS, ST, BN>;
    dim3 grid(pl.gx, hv_cdiv(d->Cout, BN), 1);
    WTrK kk = k;
    kk.fold.splits = 0;
    const int fx = hv_carry_blocks((int)grid.y);
    if (fx > 0) { kk.fold = hv_carry; hv_carry_taken = 1; grid.x += fx; }
    hv_path_note = 12;
    HV_KNAME("wgrad_thinx_kernel<%d, %d, %d>", KS, ST, BN);
    HV_TIMING_BEGIN(s);
    hipLaunchKernelGGL(kern, grid, dim3(256), pl.lds, s, kk);
    HV_TIMING_END(s);
    HV_LAUNCH_CHECK();
    return HV_OK;
}
template <int KS, int BN, int IB>
static int launch_wthing(const WTrK& k, const WThinPlan& pl, const hv_wgrad_desc* d, hipStream_t s) {
    auto kern = wgrad_thing_kernel<KS, BN, IB>;
    dim3 grid(pl.gx, hv_cdiv(d->Cin, BN), 1);
    WTrK kk = k;
    kk.fold.splits = 0;
    const int fx = hv_carry_blocks((int)grid.y);
    if (fx > 0) { kk.fold = hv_carry; hv_carry_taken = 1; grid.x += fx; }
    hv_path_note = 12;
    HV_KNAME("wgrad_thing_kernel<%d, %d, %d>", KS, BN, IB);
    HV_TIMING_BEGIN(s);
    hipLaunchKernelGGL(kern, grid, dim3(256), pl.lds, s, kk);
    HV_TIMING_END(s);
    HV_LAUNCH_CHECK();
    return HV_OK;
}
int hv_wgrad_thin(const hv_wgrad_desc* d, int* nslabs, hipStream_t s) {
    WThinPlan pl;
    if (!wgrad_thin_plan(d, &pl)) return HV_ERR_UNSUPPORTED;
    if (!d->workspace || d->workspace_bytes < hv_wgrad_thin_workspace_bytes(d)) return HV_ERR_WORKSPACE;
    WTrK k;
    k.x = reinterpret_cast<const _Float16*>(d->x); k.g = reinterpret_cast<const _Float16*>(d->g); k.slabs = d->workspace;
    k.Hl = d->H; k.Wl = d->W; k.in_shift = 0; k.Wp = d->W;
    k.img_stride = d->H * d->W * d->x_ld; k.x_ld = d->x_ld; k.x_coff = d->x_coff; k.Cin = d->Cin;
    k.Ho = d->Ho; k.Wo = d->Wo; k.Hf = d->Ho; k.Wf = d->Wo; k.g_ld = d->g_ld; k.g_coff = d->g_coff; k.Cout = d->Cout; k.pad = d->pad;
    k.dil = 1;
    if (pl.kind == 1) { k.tiles_x = hv_cdiv(k.Wo, 16); k.tiles_per_sub = k.tiles_x * hv_cdiv(k.Ho, 8); }
    else { k.tiles_x = hv_cdiv(k.Wl, 16); k.tiles_per_sub = k.tiles_x * hv_cdiv(k.Hl, 8); }
    k.tiles_per_img = k.tiles_per_sub; k.ntiles = k.tiles_per_img * d->B;
    k.slab = (long long)d->Cout * d->KH * d->KW * d->Cin;
    k.bias_out = d->dbias ? d->workspace + (long long)pl.gx * k.slab : nullptr;
    k.dbg = 0;
    k.gx = pl.gx;
    k.fold.splits = 0;
    k.x_bytes = (unsigned)((size_t)d->B * k.img_stride * sizeof(_Float16));
    k.g_bytes = (unsigned)((size_t)d->B * d->Ho * d->Wo * d->g_ld * sizeof(_Float16));
    *nslabs = pl.gx;
    if (pl.kind == 1) return launch_wthinx<5, 1, 16>(k, pl, d, s);
    if (pl.kind == 2) return launch_wthing<4, 64, 16>(k, pl, d, s);
    return launch_wthing<3, 16, 8>(k, pl, d, s);
}

// returns HV_ERR_UNSUPPORTED when the shape does not qualify; on success the slabs (*nslabs of them) are in d->workspace
int hv_wgrad_tr(const hv_wgrad_desc* d, int* nslabs, hipStream_t s) {
    WTrPlan pl;
    if (!wgrad_tr_plan(d, &pl)) return HV_ERR_UNSUPPORTED;
    const size_t need = hv_wgrad_tr_workspace_bytes(d);
    if (!d->workspace || d->workspace_bytes < need) return HV_ERR_WORKSPACE;
    WTrK k;
    k.x = reinterpret_cast<const _Float16*>(d->x); k.g = reinterpret_cast<const _Float16*>(d->g); k.slabs = d->workspace;
    const int dl = pl.dil;
    k.Hl = d->H / dl; k.Wl = d->W / dl; k.in_shift = d->in_shift; k.Wp = d->W >> d->in_shift;
    k.img_stride = (d->H >> d->in_shift) * k.Wp * d->x_ld; k.x_ld = d->x_ld; k.x_coff = d->x_coff; k.Cin = d->Cin;
    k.Ho = d->Ho / dl; k.Wo = d->Wo / dl; k.Hf = d->Ho; k.Wf = d->Wo; k.g_ld = d->g_ld; k.g_coff = d->g_coff; k.Cout = d->Cout; k.pad = dl > 1 ? 1 : d->pad;
    k.dil = dl;
    k.tiles_x = hv_cdiv(k.Wo, 16); k.tiles_per_sub = k.tiles_x * hv_cdiv(k.Ho, 8); k.tiles_per_img = k.tiles_per_sub * dl * dl; k.ntiles = k.tiles_per_img * d->B;
    k.slab = (long long)d->Cout * d->KH * d->KW * d->Cin;
    k.bias_out = d->dbias ? d->workspace + (long long)pl.gx * k.slab : nullptr;
    k.dbg = 0;
    k.gx = pl.gx;
    k.fold.splits = 0;
#ifdef WT_STAMPS
    if (getenv("HV_WTR_DBG")) k.dbg = atoi(getenv("HV_WTR_DBG"));
#endif
    k.x_bytes = (unsigned)((size_t)d->B * k.img_stride * sizeof(_Float16));
    k.g_bytes = (unsigned)((size_t)d->B * d->Ho * d->Wo * d->g_ld * sizeof(_Float16));
    *nslabs = pl.gx;
#define WTR(KS_, ST_)                                                                                     \
    do {                                                                                                  \
        if (pl.dma) return launch_wtrd<KS_, ST_>(k, pl, d, s);                                            \
        if (pl.BN == 64 && pl.BC == 32) return launch_wtr<KS_, ST_, 64, 32>(k, pl, d, s);                 \
        if (pl.BN == 64 && pl.BC == 16) return launch_wtr<KS_, ST_, 64, 16>(k, pl, d, s);                 \
        if (pl.BN == 32 && pl.BC == 32) return launch_wtr<KS_, ST_, 32, 32>(k, pl, d, s);                 \
        return launch_wtr<KS_, ST_, 32, 16>(k, pl, d, s);                                                 \
    } while (0)
    if (d->KH == 3 && d->stride == 1) WTR(3, 1);
    if (d->KH == 3) WTR(3, 2);
    if (d->stride == 1) WTR(4, 1);
    WTR(4, 2);
#undef WTR
}

// Batched "NT" GEMM on fp16 MFMA for the contextual-attention contractions, and the fold (col2im) of 4x4 stride-2 patches.
//
//   C[b][m][n] = alpha * colscale[b][n] * sum_k A[b][m][k] * B[b][n][k]          (A, B, C fp32 in memory; products on v_mfma_f32_16x16x32_f16)
//
// The five big contractions of ContextualAttention (reference models/inpaint_networks.py:327-381 and their autograd) are plain matrix products
// between per-sample matrices that already exist with the contraction index contiguous:
//   scores   S[p][l]    = rnorm[l] * <wp[p][:], wp[l][:]>             K = 9C    (the conv input's 3x3 patches ARE the filters: f == b)
//   paste    O[p][(t,c)] = <A[p][:], rawT[(c,t)][:]>                  K = L     then fold: out[y,x,c] = 1/4 sum of the 4 taps that reach (y, x)
//            (rows of rawT taken in (tap, channel) order -- b_split -- so that the fold reads whole channel rows)
//   dA       dA[p][l]   = 1/4 <dOraw[p][:], raw[l][:]>                K = 16C
//   d raw    dR[l][(t,c)] = <AT[l][:], dOrawT[(c,t)][:]>              K = L     then the same fold into the feature-map gradient
//   d wp     dwp[p][k]  = <Gs[p][:], wpT[k][:]>                       K = L
// Run as convolutions with per-sample filters they went through the gather kernel: every 64-pixel tile re-read its sample's whole filter matrix (the
// paste moved 1.1 GB per launch for 140 MB of operands, 164 us).  Here a workgroup owns a 128 x 128 tile of C, both operand tiles go global ->
// registers (fp32, 16 B per lane, converted) -> LDS as fp16 with 80-byte rows (a 16-lane group's 16-byte fragment reads hit every bank once), double
// buffered with one barrier per 32-deep k-step, and all tiles of one sample run on one XCD (its L2 fetches the sample's operands once).
// (Measured and not kept, round 3: a 64-deep k-step for fp16 x fp16 -- half the barriers, the second half-step's fragments read behind the first
// half-step's MFMAs, 144-byte LDS rows: 57.2 -> 56.0 us.  The kernel is not barrier-bound.)
#include <stdlib.h>

#include "hv_common.h"

struct BgemmK {
    const void* A; const void* B; float* C; const float* colscale;     // A / B: fp32, or fp16 elements with the AH / BH instantiations
    long long sA, sB, sC, sS;      // batch strides (elements)
    int lda, ldb, ldc;
    int M, N, K, batch;
    int tiles_m, tiles_n, swizzle;
    int b_split;                   // > 0: logical row n = t * b_split + c of B is stored as row c * (N / b_split) + t (see hv_bgemm_nt)
    float alpha;
    int c_f16;                     // C stored as fp16 (hv_bgemm_nt_h): the product only feeds a fold / a fp16 consumer
};

// one operand tile's staging: BR rows x 32 k.  fp32 source: 256 threads = 32 rows x 8 float4 per pass, converted when written to LDS;
// fp16 source: 64 rows x 4 sixteen-byte items per pass, copied as they are
template <int BR, bool H> struct BgStage;
template <int BR> struct BgStage<BR, false> {
    static constexpr int P = BR / 32;
    const float* ptr[P];
    float4 r[P];
    __device__ __forceinline__ void init(const void* base, int row0, int rows, int ld, int tid, int split) {
#pragma unroll
        for (int i = 0; i < P; ++i) {
            int n = min(row0 + (tid >> 3) + 32 * i, rows - 1);          // rows beyond the matrix: clamped (never stored)
            if (split) n = (n % split) * (rows / split) + n / split;
            ptr[i] = reinterpret_cast<const float*>(base) + (long long)n * ld + (tid & 7) * 4;
        }
    }
    __device__ __forceinline__ void load(int k0) {
#pragma unroll
        for (int i = 0; i < P; ++i) r[i] = *reinterpret_cast<const float4*>(ptr[i] + k0);
    }
    __device__ __forceinline__ void store(_Float16* tile, int tid) const {
#pragma unroll
        for (int i = 0; i < P; ++i)
            *reinterpret_cast<f16x4*>(tile + ((tid >> 3) + 32 * i) * 40 + (tid & 7) * 4) = (f16x4){(_Float16)r[i].x, (_Float16)r[i].y, (_Float16)r[i].z, (_Float16)r[i].w};
    }
};
template <int BR> struct BgStage<BR, true> {
    static constexpr int P = BR / 64;
    const _Float16* ptr[P];
    hv_u32x4 r[P];
    __device__ __forceinline__ void init(const void* base, int row0, int rows, int ld, int tid, int split) {
#pragma unroll
        for (int i = 0; i < P; ++i) {
            int n = min(row0 + (tid >> 2) + 64 * i, rows - 1);
            if (split) n = (n % split) * (rows / split) + n / split;
            ptr[i] = reinterpret_cast<const _Float16*>(base) + (long long)n * ld + (tid & 3) * 8;
        }
    }
    __device__ __forceinline__ void load(int k0) {
#pragma unroll
        for (int i = 0; i < P; ++i) r[i] = *reinterpret_cast<const hv_u32x4*>(ptr[i] + k0);
    }
    __device__ __forceinline__ void store(_Float16* tile, int tid) const {
#pragma unroll
        for (int i = 0; i < P; ++i) *reinterpret_cast<hv_u32x4*>(tile + ((tid >> 2) + 64 * i) * 40 + (tid & 3) * 8) = r[i];
    }
};

template <int BM, int BN, bool AH, bool BH>
__global__ __launch_bounds__(256, 2) void bgemm_nt_kernel(const BgemmK p) {
    constexpr int LD = 40;                      // halfs per LDS row: 32 + 8 (80 B)
    constexpr int MT = BM / 2 / 16, NT = BN / 2 / 16;
    __shared__ __attribute__((aligned(16))) _Float16 As[2][BM * LD];
    __shared__ __attribute__((aligned(16))) _Float16 Bs[2][BN * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    // tile -> (batch, m block, n block); with `swizzle` the 8 XCDs (round-robin over the linear workgroup id) each take whole batches
    const int per = p.tiles_m * p.tiles_n;
    int id = blockIdx.x, b, t;
    if (p.swizzle) {
        const int xcd = id & 7, slot = id >> 3;
        b = (slot / per) * 8 + xcd;
        t = slot % per;
    } else {
        b = id / per;
        t = id % per;
    }
    const int m_base = (t / p.tiles_n) * BM, n_base = (t % p.tiles_n) * BN;
    BgStage<BM, AH> sa;
    BgStage<BN, BH> sb;
    sa.init(reinterpret_cast<const char*>(p.A) + b * p.sA * (AH ? 2 : 4), m_base, p.M, p.lda, tid, 0);
    sb.init(reinterpret_cast<const char*>(p.B) + b * p.sB * (BH ? 2 : 4), n_base, p.N, p.ldb, tid, p.b_split);
    auto gload = [&](int k0) __attribute__((always_inline)) { sa.load(k0); sb.load(k0); };
    auto lstore = [&](int buf) __attribute__((always_inline)) { sa.store(As[buf], tid); sb.store(Bs[buf], tid); };
    // The MFMA's first operand carries the rows of B (n), the second the rows of A (m): a lane's four accumulators are then four CONSECUTIVE n of one
    // m -- one 16-byte store into row-major C.
    f32x4 acc[NT][MT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fo = (lane & 15) * LD + (lane >> 4) * 8;      // this lane's fragment piece inside a 16-row block
    const int nk = p.K / 32;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * 32);            // next k-step's operands fly behind this step's MFMAs
        f16x8 fa[MT], fb[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) fa[m] = *reinterpret_cast<const f16x8*>(&As[buf][(wm * (BM / 2) + m * 16) * LD + fo]);
#pragma unroll
        for (int n = 0; n < NT; ++n) fb[n] = *reinterpret_cast<const f16x8*>(&Bs[buf][(wn * (BN / 2) + n * 16) * LD + fo]);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[n], fa[m], acc[n][m], 0, 0, 0);
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }
    float* C = p.C + (p.c_f16 ? 0 : b * p.sC);
    _Float16* Ch = reinterpret_cast<_Float16*>(p.C) + (p.c_f16 ? b * p.sC : 0);
    const float* cs = p.colscale ? p.colscale + b * p.sS : nullptr;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row = m_base + wm * (BM / 2) + m * 16 + (lane & 15);
        if (row >= p.M) continue;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int col = n_base + wn * (BN / 2) + n * 16 + (lane >> 4) * 4;
            if (col >= p.N) continue;                       // N % 4 == 0: a lane's four columns are all inside or all outside
            float4 v = make_float4(acc[n][m][0] * p.alpha, acc[n][m][1] * p.alpha, acc[n][m][2] * p.alpha, acc[n][m][3] * p.alpha);
            if (cs) { const float4 s4 = *reinterpret_cast<const float4*>(cs + col); v.x *= s4.x; v.y *= s4.y; v.z *= s4.z; v.w *= s4.w; }
            if (p.c_f16) *reinterpret_cast<f16x4*>(Ch + (long long)row * p.ldc + col) = (f16x4){(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
            else *reinterpret_cast<float4*>(C + (long long)row * p.ldc + col) = v;
        }
    }
}

static int bgemm_impl(const void* A, int a_f16, int lda, long long strideA, const void* B, int b_f16, int ldb, long long strideB, float* C, int c_f16, int ldc,
                      long long strideC, int M, int N, int K, int batch, float alpha, const float* colscale, long long strideS, int b_split, void* stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || batch <= 0 || b_split < 0 || (b_split && N % b_split)) return HV_ERR_ARG;
    const int va = a_f16 ? 7 : 3, vb = b_f16 ? 7 : 3;        // 16-byte items: 8 halfs / 4 floats
    if ((K & 31) || (N & 3) || (lda & va) || (ldb & vb) || (ldc & 3) || lda < K || ldb < K || ldc < N) return HV_ERR_UNSUPPORTED;
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)colscale) & 15) return HV_ERR_UNSUPPORTED;
    if ((strideA & va) || (strideB & vb) || ((strideC | strideS) & 3)) return HV_ERR_UNSUPPORTED;
    if (a_f16 && !b_f16) return HV_ERR_UNSUPPORTED;          // (no caller: fp32 x fp32, fp32 x fp16 and fp16 x fp16 are instantiated)
    BgemmK k;
    k.A = A; k.B = B; k.C = C; k.colscale = colscale;
    k.sA = strideA; k.sB = strideB; k.sC = strideC; k.sS = strideS;
    k.lda = lda; k.ldb = ldb; k.ldc = ldc; k.M = M; k.N = N; k.K = K; k.batch = batch; k.alpha = alpha; k.b_split = b_split; k.c_f16 = c_f16 ? 1 : 0;
    const int BN = N % 128 == 0 ? 128 : 64;      // N = 576 (the 3x3 patch gradient): nine 64-column tiles instead of a half-empty fifth 128-column one
    k.tiles_m = hv_cdiv(M, 128); k.tiles_n = hv_cdiv(N, BN);
    const long long tiles = (long long)k.tiles_m * k.tiles_n * batch;
    if (tiles >= (1ll << 31)) return HV_ERR_UNSUPPORTED;
    static const int xcd = getenv("HV_XCD") ? atoi(getenv("HV_XCD")) : 1;
    k.swizzle = (xcd && batch % 8 == 0) ? 1 : 0;
    const dim3 grid((unsigned)tiles);
    hipStream_t s = (hipStream_t)stream;
#define HV_BG(BN_)                                                                                                       \
    do {                                                                                                                 \
        if (a_f16) hipLaunchKernelGGL((bgemm_nt_kernel<128, BN_, true, true>), grid, dim3(256), 0, s, k);                \
        else if (b_f16) hipLaunchKernelGGL((bgemm_nt_kernel<128, BN_, false, true>), grid, dim3(256), 0, s, k);          \
        else hipLaunchKernelGGL((bgemm_nt_kernel<128, BN_, false, false>), grid, dim3(256), 0, s, k);                    \
    } while (0)
    if (BN == 128) HV_BG(128);
    else HV_BG(64);
#undef HV_BG
    HV_LAUNCH_CHECK();
    return HV_OK;
}
extern "C" int hv_bgemm_nt(const void* A, int a_f16, int lda, long long strideA, const void* B, int b_f16, int ldb, long long strideB, float* C, int ldc,
                           long long strideC, int M, int N, int K, int batch, float alpha, const float* colscale, long long strideS, int b_split, void* stream) {
    return bgemm_impl(A, a_f16, lda, strideA, B, b_f16, ldb, strideB, C, 0, ldc, strideC, M, N, K, batch, alpha, colscale, strideS, b_split, stream);
}
extern "C" int hv_bgemm_nt_h(const void* A, int a_f16, int lda, long long strideA, const void* B, int b_f16, int ldb, long long strideB, void* C_h, int ldc,
                             long long strideC, int M, int N, int K, int batch, float alpha, const float* colscale, long long strideS, int b_split, void* stream) {
    return bgemm_impl(A, a_f16, lda, strideA, B, b_f16, ldb, strideB, reinterpret_cast<float*>(C_h), 1, ldc, strideC, M, N, K, batch, alpha, colscale, strideS,
                      b_split, stream);
}

// fold (col2im) of 4x4 stride-2 pad-1 patches: src[b][p][tap][c] (p over the (H/2) x (W/2) patch grid, tap = r*4 + s) ->
//   dst[b][y][x][c] (+)= alpha * sum over the taps (r, s) with (y + 1 - r), (x + 1 - s) even and the patch position inside the grid
// = F.conv_transpose2d(A, raw patches, stride 2, padding 1) after the contraction over the patches, and equally the adjoint of hv_ca_raw_patches.
__global__ __launch_bounds__(256) void ca_fold_kernel(const float* __restrict__ src, void* __restrict__ dst, int dsth, int H, int W, int C, int dst_ld, float alpha,
                                                      int accumulate, long long n) {
    const int h = H >> 1, w = W >> 1;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long long r = i / C;
        const int x = (int)(r % W);
        r /= W;
        const int y = (int)(r % H);
        const long long b = r / H;
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int rr = ((y + 1) & 1) + 2 * a, py = (y + 1 - rr) >> 1;         // filter rows of this output parity
            if ((unsigned)py >= (unsigned)h) continue;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int ss = ((x + 1) & 1) + 2 * q, px = (x + 1 - ss) >> 1;
                if ((unsigned)px >= (unsigned)w) continue;
                s += src[(((b * h + py) * w + px) * 16 + rr * 4 + ss) * C + c];
            }
        }
        const long long o = ((b * H + y) * W + x) * dst_ld + c;
        hv_st1(dst, o, accumulate ? hv_ld1(dst, o, dsth) + alpha * s : alpha * s, dsth);      // dsth: the destination map is stored as fp16
    }
}

// The same on 8-channel pieces (16 bytes of an fp16 source / destination per access), integer index arithmetic, the four taps' loads issued before the
// first add: the element-wise kernel above pays three 64-bit divisions and four 4-byte loads per output element (24 us for 67 MB at bs 16).  Same
// summation order per element: with an fp32 source the same bits.
template <bool SH, bool DH>
__global__ __launch_bounds__(256) void ca_fold_vec_kernel(const void* __restrict__ src, void* __restrict__ dst, int H, int W, int C, int dst_ld, float alpha,
                                                          int accumulate, int n) {
    const int h = H >> 1, w = W >> 1, C8 = C >> 3;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int c = (i % C8) * 8;
        int r = i / C8;
        const int x = r % W;
        r /= W;
        const int y = r % H, b = r / H;
        float v[4][8];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int rr = ((y + 1) & 1) + 2 * a, py = (y + 1 - rr) >> 1, ss = ((x + 1) & 1) + 2 * q, px = (x + 1 - ss) >> 1;
                const bool ok = (unsigned)py < (unsigned)h && (unsigned)px < (unsigned)w;
                const long long o = ((((long long)b * h + py) * w + px) * 16 + rr * 4 + ss) * C + c;
                if (SH) {
                    hv_u32x4 u = {0u, 0u, 0u, 0u};
                    if (ok) u = *reinterpret_cast<const hv_u32x4*>(reinterpret_cast<const _Float16*>(src) + o);
                    const f16x8 h8 = __builtin_bit_cast(f16x8, u);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[a * 2 + q][e] = (float)h8[e];
                } else {
                    float4 u0 = make_float4(0.f, 0.f, 0.f, 0.f), u1 = u0;
                    if (ok) { u0 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(src) + o); u1 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(src) + o + 4); }
                    v[a * 2 + q][0] = u0.x; v[a * 2 + q][1] = u0.y; v[a * 2 + q][2] = u0.z; v[a * 2 + q][3] = u0.w;
                    v[a * 2 + q][4] = u1.x; v[a * 2 + q][5] = u1.y; v[a * 2 + q][6] = u1.z; v[a * 2 + q][7] = u1.w;
                }
            }
        const long long od = (((long long)b * H + y) * W + x) * dst_ld + c;
        float o8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) s += v[t][e];      // absent taps are +0: adding them leaves the partial sum's bits (the sum starts at +0 as before)
            o8[e] = alpha * s;
        }
        if (DH) {
            _Float16* d = reinterpret_cast<_Float16*>(dst) + od;
            f16x8 h8;
            if (accumulate) {
                const f16x8 old = __builtin_bit_cast(f16x8, *reinterpret_cast<const hv_u32x4*>(d));
#pragma unroll
                for (int e = 0; e < 8; ++e) h8[e] = (_Float16)((float)old[e] + o8[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) h8[e] = (_Float16)o8[e];
            }
            *reinterpret_cast<hv_u32x4*>(d) = __builtin_bit_cast(hv_u32x4, h8);
        } else {
            float* d = reinterpret_cast<float*>(dst) + od;
            float4 a0 = make_float4(o8[0], o8[1], o8[2], o8[3]), a1 = make_float4(o8[4], o8[5], o8[6], o8[7]);
            if (accumulate) {
                const float4 p0 = *reinterpret_cast<const float4*>(d), p1 = *reinterpret_cast<const float4*>(d + 4);
                a0 = make_float4(p0.x + a0.x, p0.y + a0.y, p0.z + a0.z, p0.w + a0.w);
                a1 = make_float4(p1.x + a1.x, p1.y + a1.y, p1.z + a1.z, p1.w + a1.w);
            }
            *reinterpret_cast<float4*>(d) = a0;
            *reinterpret_cast<float4*>(d + 4) = a1;
        }
    }
}
static int ca_fold_impl(const void* src, int src_f16, void* dst, int dst_f16, int B, int H, int W, int C, int dst_ld, float alpha, int accumulate, void* stream) {
    if (!src || !dst || B <= 0 || H <= 0 || W <= 0 || C <= 0 || ((H | W) & 1) || dst_ld < C) return HV_ERR_ARG;
    const long long n8 = (long long)B * H * W * (C / 8);
    static const int vec = getenv("HV_CA_FOLD_VEC") ? atoi(getenv("HV_CA_FOLD_VEC")) : 1;      // A/B knob
    const bool vec_ok = !(C & 7) && !(dst_ld & 7) && !((uintptr_t)src & 15) && !((uintptr_t)dst & 15) && n8 < (1ll << 31);
    if ((vec || src_f16) && vec_ok) {
        long long blocks = (n8 + 255) / 256;
        if (blocks > 65536) blocks = 65536;
        const dim3 grid((unsigned)blocks);
        hipStream_t s = (hipStream_t)stream;
        if (src_f16) {
            if (dst_f16) hipLaunchKernelGGL((ca_fold_vec_kernel<true, true>), grid, dim3(256), 0, s, src, dst, H, W, C, dst_ld, alpha, accumulate, (int)n8);
            else hipLaunchKernelGGL((ca_fold_vec_kernel<true, false>), grid, dim3(256), 0, s, src, dst, H, W, C, dst_ld, alpha, accumulate, (int)n8);
        } else {
            if (dst_f16) hipLaunchKernelGGL((ca_fold_vec_kernel<false, true>), grid, dim3(256), 0, s, src, dst, H, W, C, dst_ld, alpha, accumulate, (int)n8);
            else hipLaunchKernelGGL((ca_fold_vec_kernel<false, false>), grid, dim3(256), 0, s, src, dst, H, W, C, dst_ld, alpha, accumulate, (int)n8);
        }
        HV_LAUNCH_CHECK();
        return HV_OK;
    }
    if (src_f16) return HV_ERR_UNSUPPORTED;
    return 1;      // the element-wise kernel below
}
extern "C" int hv_ca_fold_h(const void* src_h, void* dst, int dst_f16, int B, int H, int W, int C, int dst_ld, float alpha, int accumulate, void* stream) {
    return ca_fold_impl(src_h, 1, dst, dst_f16, B, H, W, C, dst_ld, alpha, accumulate, stream);
}
extern "C" int hv_ca_fold(const float* src, void* dst, int dst_f16, int B, int H, int W, int C, int dst_ld, float alpha, int accumulate, void* stream) {
    if (!src || !dst || B <= 0 || H <= 0 || W <= 0 || C <= 0 || ((H | W) & 1) || dst_ld < C) return HV_ERR_ARG;
    {
        const int rc = ca_fold_impl(src, 0, dst, dst_f16, B, H, W, C, dst_ld, alpha, accumulate, stream);
        if (rc != 1) return rc;
    }
    const long long n = (long long)B * H * W * C;
    long long blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(ca_fold_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, dst_f16, H, W, C, dst_ld, alpha, accumulate, n);
    HV_LAUNCH_CHECK();
    return HV_OK;
}

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

#include <type_traits>

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


// ---------------------------------------------------------------------------------------------------------------------------------------------
// The fp16 x fp16 products (the attention block's A V, dA and d raw at 1024^3 per sample) on an LDS-DMA ring (round 5).  bgemm_nt_kernel above moves
// 16 KB of operands global -> registers -> LDS per 64 MFMAs of a 128 x 128 tile: at full MFMA rate that is 64 B per CU cycle, twice what a CU takes in from
// L2, and it ran at 0.60-0.64 PFLOP/s.  Here a workgroup of eight waves owns a 256 x 256 tile of C (one round of 256 workgroups at 1024^2 x 16 samples;
// 32 KB of operands per 256 MFMAs = 32 B per CU cycle), a stage is 32 deep in k (one MFMA k-step), both operand tiles arrive by LDS-DMA (buffer_load ... lds,
// 1 KB per wave instruction, no staging registers) into a ring of four 32-KB stages issued three stages ahead (the operands stream from the Infinity Cache /
// HBM; with two 64-deep stages and one stage of lead: 48.2 us against 40.7), one barrier per stage of 32 MFMAs per wave.  The DMA writes lane-linear
// 1-KB pieces = 16 rows x 64 B, so rows cannot be padded: slot s of row r holds its 16-byte k piece s ^ 2 ((r >> 2) & 1) (the permutation goes on the SOURCE
// address, as for the patch rows of conv_g4_kernel): the four lane groups of a fragment's ds_read_b128 then hit every bank once.  A wave computes 128 (m) x
// 64 (n): per k-step 8 A + 4 B fragment reads for 32 MFMAs; the fragments are double-buffered over half k-steps (16 MFMAs).  Measured (tools/bench_bgemm.py,
// rotating operands): 1024^3 x 16 61.1 -> 42.9 us = 0.80 PFLOP/s, 4096 x 4096 x 1024 x 16 975 -> 746 us; same bits (same k order).
typedef __attribute__((address_space(3))) void* bg_lds_ptr;
__device__ __forceinline__ void bg_dma16(__amdgpu_buffer_rsrc_t r, bg_lds_ptr dst, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, voff, soff, 0, 0);
#endif
}

__global__ __launch_bounds__(512, 2) void bgemm_dma_kernel(const BgemmK p) {
    constexpr int BM = 256, BN = 256;
    constexpr int STAGE = 32 * 1024, RING = 4;                                // bytes per stage: [operand 2][16 pieces][1 KB]; stages in LDS
    extern __shared__ __attribute__((aligned(16))) char smem[];               // [RING][STAGE]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                                  // wave tile: rows wm * 128 .., columns wn * 64 ..
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
    const _Float16* Ab = reinterpret_cast<const _Float16*>(p.A) + b * p.sA;
    const _Float16* Bb = reinterpret_cast<const _Float16*>(p.B) + b * p.sB;
    const __amdgpu_buffer_rsrc_t asrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(Ab), 0, (unsigned)((size_t)p.M * p.lda * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t bsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(Bb), 0, (unsigned)((size_t)p.N * p.ldb * 2), 0x00020000);
    // ---- this lane's part of the wave's 4 DMA pieces per stage (stage-independent): piece idx = wave * 4 + i -> (operand idx >> 4, 16-row block idx & 15);
    // lane -> (row lane >> 2, slot lane & 3)
    unsigned doff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = wave * 4 + i, op = idx >> 4, j = idx & 15;
        const int r = lane >> 2, c8 = (lane & 3) ^ (((r >> 2) & 1) << 1);
        int row = (op ? n_base : m_base) + j * 16 + r;
        const int rows = op ? p.N : p.M, ld = op ? p.ldb : p.lda;
        const bool ok = row < rows;
        if (op && p.b_split) row = (row % p.b_split) * (p.N / p.b_split) + row / p.b_split;
        doff[i] = ok ? (unsigned)((row * ld + c8 * 8) * 2) : 0x80000000u;
    }
    auto issue = [&](int s) __attribute__((always_inline)) {
        char* dst = smem + (s & (RING - 1)) * STAGE + wave * 4 * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) bg_dma16(wave >= 4 ? bsrc : asrc, (bg_lds_ptr)(dst + i * 1024), doff[i], s * 64);
    };
    // fragment piece of this lane inside a 16-row block: row lane & 15, k piece lane >> 4 at its permuted slot
    const int fo = (lane & 15) * 64 + ((((lane >> 4) ^ ((((lane & 15) >> 2) & 1) << 1))) << 4);
    f32x4 acc[4][8];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f16x8 fb[2][4], fa[2][4];
    auto ldB = [&](const char* base, int q) __attribute__((always_inline)) {
#pragma unroll
        for (int n = 0; n < 4; ++n) fb[q][n] = *reinterpret_cast<const f16x8*>(base + (16 + wn * 4 + n) * 1024 + fo);
    };
    auto ldA = [&](const char* base, int h, int q) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 4; ++m) fa[q][m] = *reinterpret_cast<const f16x8*>(base + (wm * 8 + h * 4 + m) * 1024 + fo);
    };
    auto mfmas = [&](int qb, int qa, int h) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[n][h * 4 + m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[qb][n], fa[qa][m], acc[n][h * 4 + m], 0, 0, 0);
    };
    auto mix = [&](int reads) __attribute__((always_inline)) {      // the next half-step's fragment reads dealt among this half-step's 16 MFMAs
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < reads) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
    };
    // Stage s (32 deep in k: one MFMA k-step, 32 MFMAs per wave in two halves) computes on buffer s % 4 while the DMA of stages s + 1 .. s + 3 is in flight
    // (issued three stages ahead: the operands stream from the Infinity Cache / HBM, not from L2).  The barrier sits in front of the second half: the wave holds
    // all its fragments of stage s and has waited for its OWN pieces of stage s + 1 (counted vmcnt: the 8 newer instructions stay in flight).
    const int NS = p.K >> 5;
    issue(0);
    if (NS > 1) issue(1);
    if (NS > 2) issue(2);
    if (NS > 2) __builtin_amdgcn_s_waitcnt(0x0F70 | 8); else if (NS > 1) __builtin_amdgcn_s_waitcnt(0x0F70 | 4); else __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();
    ldB(smem, 0);
    ldA(smem, 0, 0);
    auto stage = [&](int s, auto PAR) __attribute__((always_inline)) {      // PAR: s & 1 at compile time (the B fragments' register set)
        constexpr int q = decltype(PAR)::value;
        const char* cur = smem + (s & (RING - 1)) * STAGE;
        const char* nxt = smem + ((s + 1) & (RING - 1)) * STAGE;
        // the DMA of stage s + 3 (buffer (s - 1) % 4: every wave left it at the barrier of stage s - 1).  (Measured and not kept: waves 4-7 issuing theirs behind
        // the first half's MFMAs, and the four instructions dealt among those MFMAs by the scheduler: 42.6 / 42.8 against 42.9 us.  PMC: the 262 144 DMA
        // instructions of a launch hold their waves for ~150 cycles each = a quarter of all wave cycles wherever they sit; MFMA pipes busy 38 %.)
        if (s + 3 < NS) issue(s + 3);
        __builtin_amdgcn_sched_barrier(0);
        ldA(cur, 1, 1);
        mfmas(q, 0, 0);
        mix(4);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 3 < NS) __builtin_amdgcn_s_waitcnt(0x0070 | 8);            // lgkmcnt(0); stages s + 2, s + 3 may stay in flight
        else if (s + 2 < NS) __builtin_amdgcn_s_waitcnt(0x0070 | 4);
        else __builtin_amdgcn_s_waitcnt(0x0070);
        __builtin_amdgcn_s_barrier();
        if (s + 1 < NS) { ldB(nxt, q ^ 1); ldA(nxt, 0, 0); }
        mfmas(q, 1, 1);
        mix(8);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int s = 0; s < NS; s += 2) {
        stage(s, std::integral_constant<int, 0>());
        if (s + 1 < NS) stage(s + 1, std::integral_constant<int, 1>());
    }
    float* C = p.C + (p.c_f16 ? 0 : b * p.sC);
    _Float16* Ch = reinterpret_cast<_Float16*>(p.C) + (p.c_f16 ? b * p.sC : 0);
    const float* cs = p.colscale ? p.colscale + b * p.sS : nullptr;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int row = m_base + wm * 128 + m * 16 + (lane & 15);
        if (row >= p.M) continue;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int col = n_base + wn * 64 + n * 16 + (lane >> 4) * 4;
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
    static const int xcd = getenv("HV_XCD") ? atoi(getenv("HV_XCD")) : 1;
    hipStream_t s = (hipStream_t)stream;
    {   // fp16 x fp16 with whole 64-deep stages and enough 256 x 256 tiles to go round: the LDS-DMA form
        static const int dma = getenv("HV_BGEMM_DMA") ? atoi(getenv("HV_BGEMM_DMA")) : 1;      // A/B knob
        const long long t256 = (long long)hv_cdiv(M, 256) * hv_cdiv(N, 256) * batch;
        if (dma && a_f16 && b_f16 && !(K & 63) && M >= 256 && N >= 256 && t256 >= 128 && t256 < (1ll << 31) && (size_t)M * lda * 2 < (1ull << 31) &&
            (size_t)N * ldb * 2 < (1ull << 31)) {
            k.tiles_m = hv_cdiv(M, 256); k.tiles_n = hv_cdiv(N, 256);
            k.swizzle = (xcd && batch % 8 == 0) ? 1 : 0;
            static bool raised = false;
            if (!raised) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bgemm_dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
                if (e != hipSuccess) return -1000 - (int)e;
                raised = true;
            }
            hipLaunchKernelGGL(bgemm_dma_kernel, dim3((unsigned)t256), dim3(512), 128 * 1024, s, k);
            HV_LAUNCH_CHECK();
            return HV_OK;
        }
    }
    const int BN = N % 128 == 0 ? 128 : 64;      // N = 576 (the 3x3 patch gradient): nine 64-column tiles instead of a half-empty fifth 128-column one
    k.tiles_m = hv_cdiv(M, 128); k.tiles_n = hv_cdiv(N, BN);
    const long long tiles = (long long)k.tiles_m * k.tiles_n * batch;
    if (tiles >= (1ll << 31)) return HV_ERR_UNSUPPORTED;
    k.swizzle = (xcd && batch % 8 == 0) ? 1 : 0;
    const dim3 grid((unsigned)tiles);
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

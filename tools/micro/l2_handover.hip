// Does an XCD's L2 still hold what the PREVIOUS kernel wrote there?  (queue item 1: cross-kernel L2 residency for the 64 x 64 trunk.)
//
// Kernel W: every workgroup reads its XCD (HW_REG_XCC_ID), takes a slot from that XCD's counter and writes one CHUNK of `buf` -- chunk (xcd, slot) --
// so that each XCD's writes total 1 MB (well inside its 4 MB L2).  Kernel R (next launch, same stream): every workgroup again reads its XCD, takes a
// slot and reads chunk ((xcd + shift) & 7, slot): shift 0 = the chunk THIS XCD wrote, shift 4 = a chunk another XCD wrote.  Per-workgroup read time by
// s_memrealtime (100 MHz), summed bytes / time printed per shift.  If the L2 keeps a predecessor's lines across the kernel boundary the shift-0 reads run
// at the L2 rate (66-73 GB/s per CU in MI355X_MICROARCH.md), the shift-4 reads at the Infinity-Cache rate (~33).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/l2_handover.hip -o tools/micro/l2_handover && tools/micro/l2_handover
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int CHUNK = 64 * 1024;              // bytes per workgroup
constexpr int SLOTS = 16;                     // chunks per XCD: 1 MB

__device__ __forceinline__ int xcc_id() {
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7;
}

__global__ __launch_bounds__(256) void kw(uint4* buf, int* counters, int* who) {
    __shared__ int slot, xcd;
    if (threadIdx.x == 0) { xcd = xcc_id(); slot = atomicAdd(counters + xcd, 1); }
    __syncthreads();
    if (slot >= SLOTS) return;
    uint4* p = buf + ((size_t)(xcd * SLOTS + slot) * CHUNK) / 16;
    for (int i = threadIdx.x; i < CHUNK / 16; i += 256) p[i] = make_uint4(i, xcd, slot, 7);
    if (threadIdx.x == 0) who[xcd * SLOTS + slot] = xcd;
}

__global__ __launch_bounds__(256) void kr(const uint4* buf, int* counters, int shift, unsigned long long* ticks, unsigned* sink) {
    __shared__ int slot, xcd;
    if (threadIdx.x == 0) { xcd = xcc_id(); slot = atomicAdd(counters + xcd, 1); }
    __syncthreads();
    if (slot >= SLOTS) return;
    const uint4* p = buf + ((size_t)(((xcd + shift) & 7) * SLOTS + slot) * CHUNK) / 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    uint4 v[CHUNK / 16 / 256];                            // all 16 loads of the thread in flight: the interval is ~one round trip + the data rate
#pragma unroll
    for (int k = 0; k < CHUNK / 16 / 256; ++k) v[k] = p[threadIdx.x + 256 * k];
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    unsigned acc = 0;
#pragma unroll
    for (int k = 0; k < CHUNK / 16 / 256; ++k) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    // control: the same chunk once more (another thread's pieces, so that nothing comes from this CU's L1): now it IS in this XCD's L2
    __syncthreads();
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int k = 0; k < CHUNK / 16 / 256; ++k) v[k] = p[((threadIdx.x + 128) & 255) + 256 * ((k + 5) & 15)];
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int k = 0; k < CHUNK / 16 / 256; ++k) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) { ticks[xcd * SLOTS + slot] = t1 - t0; ticks[8 * SLOTS + xcd * SLOTS + slot] = t3 - t2; }
}

int main() {
    const int NWG = 256;                      // one workgroup per CU: 32 per XCD (round-robin dispatch), the first 16 of each take the slots; few atomics beside the timed reads
    uint4* buf; int *cw, *cr, *who; unsigned long long* ticks; unsigned* sink;
    hipMalloc(&buf, (size_t)8 * SLOTS * CHUNK); hipMalloc(&cw, 32); hipMalloc(&cr, 32); hipMalloc(&who, 8 * SLOTS * 4);
    hipMalloc(&ticks, 2 * 8 * SLOTS * 8); hipMalloc(&sink, (size_t)NWG * 256 * 4);
    unsigned long long h[2 * 8 * SLOTS];
    for (int shift : {0, 4, 0, 4, 1, 0}) {
        double best = 0, best2 = 0;
        for (int rep = 0; rep < 5; ++rep) {
            hipMemset(cw, 0, 32); hipMemset(cr, 0, 32); hipMemset(ticks, 0, 2 * 8 * SLOTS * 8);
            hipDeviceSynchronize();
            hipLaunchKernelGGL(kw, dim3(NWG), dim3(256), 0, 0, buf, cw, who);
            hipLaunchKernelGGL(kr, dim3(NWG), dim3(256), 0, 0, buf, cr, shift, ticks, sink);
            hipDeviceSynchronize();
            hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
            double sum = 0, sum2 = 0; int n = 0;
            for (int i = 0; i < 8 * SLOTS; ++i) if (h[i]) { sum += (double)h[i]; sum2 += (double)h[8 * SLOTS + i]; ++n; }
            const double us = sum / n / 100.0;             // mean per-workgroup read time (100 MHz counter)
            const double gbs = CHUNK / us / 1e3;
            if (gbs > best) { best = gbs; best2 = CHUNK / (sum2 / n / 100.0) / 1e3; }
        }
        printf("shift %d: first read %.1f GB/s per workgroup, second read of the same chunk %.1f GB/s (64 KB chunk, best of 5)\n", shift, best, best2);
    }
    return 0;
}

// Does `buffer_load_dwordx4 ... lds` write zeros for lanes whose offset is beyond the descriptor's range?  (conv_g4's patch staging relies on it.)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const unsigned* src, unsigned bytes, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* l = (unsigned*)smem;
    for (int i = threadIdx.x; i < 1024; i += 64) l[i] = 0xdeadbeefu;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, bytes, 0x00020000);
    const int lane = threadIdx.x;
    // even lanes in range, odd lanes out of range
    const unsigned off = (lane & 1) ? 0x80000000u : lane * 16;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)smem, 16, off, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = l[i];
}
int main() {
    unsigned h[256], *d, *o;
    for (int i = 0; i < 256; ++i) h[i] = 1000 + i;
    hipMalloc(&d, 1024); hipMalloc(&o, 1024);
    hipMemcpy(d, h, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, 1024u, o);
    hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 8; ++l) printf("lane %d: %u %u %u %u\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    return 0;
}

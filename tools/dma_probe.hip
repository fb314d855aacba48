// Probe (round 2): does the instruction offset of global_load_lds_dwordx4 (SADDR form) apply to the global address, to the
// LDS address, or to both?  Answer on gfx950: both (LDS bytes [m0 + offset, ...), global bytes [base + offset, ...)).
//   hipcc --offload-arch=gfx950 -O2 -o dma_probe tools/dma_probe.hip && ./dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void probe(const uint32_t* src, uint32_t* out)
{
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = 0xdeadbeef;
    __syncthreads();
    const uint32_t lane16 = threadIdx.x * 16;
    const uint64_t base = (uint64_t)(uintptr_t)src;
    uint32_t lo, hi, dst;
    // (s_nop 1: the vector registers were written just before; s_nop 4: scalar registers written by vector instructions are
    // not readable as a memory instruction's address for five wait states)
    asm volatile("s_nop 1\n\tv_readfirstlane_b32 %0, %3\n\tv_readfirstlane_b32 %1, %4\n\tv_readfirstlane_b32 %2, %5\n\ts_nop 4"
                 : "=s"(lo), "=s"(hi), "=s"(dst) : "v"((uint32_t)base), "v"((uint32_t)(base >> 32)), "v"((uint32_t)(uintptr_t)lds + 4096u));
    const uint64_t b = ((uint64_t)hi << 32) | lo;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024" ::"v"(lane16), "s"(b), "s"(dst));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 64) out[i] = lds[i];
}
int main()
{
    uint32_t *src, *out;
    if (hipMalloc(&src, 1 << 20) != hipSuccess || hipMalloc(&out, 16384) != hipSuccess) return 1;
    std::vector<uint32_t> h(1 << 18);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)i;     // word index
    (void)hipMemcpy(src, h.data(), 1 << 20, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 16384, 0, src, out);
    printf("sync: %s\n", hipGetErrorString(hipDeviceSynchronize()));
    std::vector<uint32_t> o(4096);
    (void)hipMemcpy(o.data(), out, 16384, hipMemcpyDeviceToHost);
    int first = -1, last = -1;
    for (int i = 0; i < 4096; ++i) if (o[i] != 0xdeadbeef) { if (first < 0) first = i; last = i; }
    printf("LDS words written: [%d, %d]  (m0 = word 1024; with the offset: word 1280)\n", first, last);
    if (first >= 0) printf("first value: %u  (global word 0 = no offset on the global side, 256 = offset applied)\n", o[first]);
    return 0;
}

// Is a CU's instruction throughput limited by instruction FETCH when the code is long and straight-line?
// Two kernels execute the same number of dependent-free VALU instructions per wave: one as a short loop (the body
// stays in the wave's instruction buffer / one I-cache line), one as a long unrolled body (16 K instructions = 64+ KB,
// streams through the I-cache, which two CUs share).   hipcc -O3 --offload-arch=gfx950 ifetch_probe.hip -o ifetch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define OPS8(a, b) a = a * 3u + b; b = b * 5u + a; a ^= b >> 3; b += a << 2; a = a * 7u + b; b ^= a >> 5; a += b << 1; b = b * 9u + a;
#define OPS64(a, b) OPS8(a, b) OPS8(a, b) OPS8(a, b) OPS8(a, b) OPS8(a, b) OPS8(a, b) OPS8(a, b) OPS8(a, b)
#define OPS512(a, b) OPS64(a, b) OPS64(a, b) OPS64(a, b) OPS64(a, b) OPS64(a, b) OPS64(a, b) OPS64(a, b) OPS64(a, b)
#define OPS4096(a, b) OPS512(a, b) OPS512(a, b) OPS512(a, b) OPS512(a, b) OPS512(a, b) OPS512(a, b) OPS512(a, b) OPS512(a, b)
__global__ void __launch_bounds__(64) k_short(uint32_t *out, int iters)
{
    uint32_t a = threadIdx.x, b = blockIdx.x;
    for (int i = 0; i < iters * 512; i++) { OPS8(a, b) }
    if (a == 0x12345678u) out[0] = b;
}
__global__ void __launch_bounds__(64) k_long(uint32_t *out, int iters)
{
    uint32_t a = threadIdx.x, b = blockIdx.x;
    for (int i = 0; i < iters; i++) { OPS4096(a, b) }
    if (a == 0x12345678u) out[0] = b;
}
int main()
{
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int n_cu = pr.multiProcessorCount;
    uint32_t *out; hipMalloc(&out, 4);
    const int iters = 200;
    const double instr_per_wave = (double)iters * 4096 * 16.0 / 8.0;      // OPS8 = 8 statements ~ 16 VALU instructions
    for (int which = 0; which < 2; which++) for (int wpc : {4, 8, 16, 32}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        if (which) k_long<<<n_cu * wpc, 64>>>(out, 2); else k_short<<<n_cu * wpc, 64>>>(out, 2);
        hipEventRecord(e0);
        if (which) k_long<<<n_cu * wpc, 64>>>(out, iters); else k_short<<<n_cu * wpc, 64>>>(out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s body, %2d waves/CU: %7.3f ms  ~%.2f VALU instr/cycle/CU at 2.1 GHz (if 16 instr per OPS8)\n", which ? "4096-statement" : "   8-statement", wpc, ms,
               instr_per_wave * wpc / (ms * 1e-3 * 2.1e9));
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    return 0;
}

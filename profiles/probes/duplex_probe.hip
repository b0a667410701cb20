// Is the host link full duplex for us?  hipMemcpyAsync in both directions at once gave the rate of one direction (profiles/r02/pcie_probe.txt);
// here the copies are also made by kernels that read / write page-locked host memory directly, alone and against each other.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/duplex_probe profiles/probes/duplex_probe.hip && /tmp/duplex_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_copy(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const size_t MB = 1 << 20, n_in = 256 * MB, n_out = 512 * MB;
    const int blocks = argc > 1 ? atoi(argv[1]) : 64;            // workgroups of a copy kernel
    void *h_in, *h_out, *d_in, *d_out;
    CHK(hipHostMalloc(&h_in, n_in, hipHostMallocDefault)); CHK(hipHostMalloc(&h_out, n_out, hipHostMallocDefault));
    CHK(hipMalloc(&d_in, n_in)); CHK(hipMalloc(&d_out, n_out));
    CHK(hipMemset(d_out, 1, n_out)); memset(h_in, 2, n_in); memset(h_out, 0, n_out);
    hipStream_t s1, s2; CHK(hipStreamCreate(&s1)); CHK(hipStreamCreate(&s2));
    auto run = [&](const char *what, int h2d, int d2h) {          // 0 = not at all, 1 = hipMemcpyAsync, 2 = kernel
        for (int rep = 0; rep < 2; rep++) {                       // the first round warms up
            CHK(hipDeviceSynchronize());
            const double t = now();
            for (int k = 0; k < 4; k++) {
                if (h2d == 1) CHK(hipMemcpyAsync(d_in, h_in, n_in, hipMemcpyHostToDevice, s1));
                if (h2d == 2) k_copy<<<blocks, 256, 0, s1>>>((uint4 *)d_in, (const uint4 *)h_in, n_in / 16);
                if (d2h == 1) CHK(hipMemcpyAsync(h_out, d_out, n_out, hipMemcpyDeviceToHost, s2));
                if (d2h == 2) k_copy<<<blocks, 256, 0, s2>>>((uint4 *)h_out, (const uint4 *)d_out, n_out / 16);
            }
            CHK(hipDeviceSynchronize());
            const double dt = now() - t, bytes = 4.0 * ((h2d ? n_in : 0) + (d2h ? n_out : 0));
            if (rep) printf("%-58s %6.1f GB/s in total (%.1f ms)\n", what, bytes / dt / 1e9, dt * 1e3);
        }
    };
    printf("copy kernels: %d workgroups of 256\n", blocks);
    run("H2D hipMemcpyAsync alone", 1, 0);
    run("D2H hipMemcpyAsync alone", 0, 1);
    run("H2D kernel (reads host memory) alone", 2, 0);
    run("D2H kernel (writes host memory) alone", 0, 2);
    run("H2D hipMemcpyAsync + D2H hipMemcpyAsync at once", 1, 1);
    run("H2D hipMemcpyAsync + D2H kernel at once", 1, 2);
    run("H2D kernel + D2H hipMemcpyAsync at once", 2, 1);
    run("H2D kernel + D2H kernel at once", 2, 2);
    return 0;
}

// Device -> page-locked host: one copy of the contact columns of a 100 k-atom pass (18.8 MB) against five (4, 4, 4, 2, 1 B x 1.25 M)
// and against a kernel writing straight into the mapped host buffer.   hipcc --offload-arch=gfx950 -O3 d2h_rate.hip -o d2h && ./d2h
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int main() {
    const size_t P = 1250796;
    const size_t sz[5] = {4 * P, 4 * P, 4 * P, 2 * P, P};
    size_t total = 0; for (size_t s : sz) total += (s + 255) & ~(size_t)255;
    char *d, *h;
    CK(hipMalloc(&d, total)); CK(hipHostMalloc(&h, total, hipHostMallocDefault));
    CK(hipMemset(d, 1, total));
    hipStream_t st; CK(hipStreamCreate(&st));
    auto now = [] { return std::chrono::steady_clock::now(); };
    for (int mode = 0; mode < 4; ++mode) {
        double best = 1e9;
        for (int rep = 0; rep < 30; ++rep) {
            auto t0 = now();
            if (mode == 0) CK(hipMemcpyAsync(h, d, total, hipMemcpyDeviceToHost, st));
            else if (mode == 1) { size_t off = 0; for (size_t s : sz) { CK(hipMemcpyAsync(h + off, d + off, s, hipMemcpyDeviceToHost, st)); off += (s + 255) & ~(size_t)255; } }
            else if (mode == 2) hipLaunchKernelGGL(k_copy, dim3(1024), dim3(256), 0, st, (const uint4*)d, (uint4*)h, total / 16);
            else { CK(hipMemcpyAsync(h, d, total / 2, hipMemcpyDeviceToHost, st)); hipLaunchKernelGGL(k_copy, dim3(512), dim3(256), 0, st, (const uint4*)(d + total / 2), (uint4*)(h + total / 2), total / 32); }
            CK(hipStreamSynchronize(st));
            const double ms = std::chrono::duration<double, std::milli>(now() - t0).count();
            if (rep >= 5 && ms < best) best = ms;
        }
        const char* names[4] = {"one hipMemcpyAsync", "five hipMemcpyAsync", "copy kernel into mapped host memory", "half by memcpy, then half by kernel"};
        printf("%-40s %.3f ms  %.1f GB/s\n", names[mode], best, total / best * 1e-6);
    }
    return 0;
}

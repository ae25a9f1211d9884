#include <hip/hip_runtime.h>
__global__ __launch_bounds__(1024) void k(unsigned long long* out) {
    __shared__ unsigned long long a[8192];
    __shared__ unsigned int b[8192];
    for (int i = threadIdx.x; i < 8192; i += 1024) { a[i] = i; b[i] = i; }
    __syncthreads();
    out[threadIdx.x] = a[(threadIdx.x * 7) & 8191] + b[(threadIdx.x * 13) & 8191];
}
int main() { unsigned long long* d; hipMalloc(&d, 8192); hipLaunchKernelGGL(k, dim3(1), dim3(1024), 0, 0, d); hipError_t e = hipDeviceSynchronize(); printf("%s\n", hipGetErrorString(e)); hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); printf("sharedMemPerBlock %zu maxSharedMemoryPerMultiProcessor %zu\n", p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor); return 0; }

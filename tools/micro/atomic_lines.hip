// How do returning device-scope atomics of many blocks serialise: by word, by cache line, by channel?   (GPU box)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/atomic_lines.hip -o /tmp/atomic_lines && /tmp/atomic_lines
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
__global__ void k_ticket(u64* ctr, int stride_words, int nslots, u64* sink, int work) {
    // some work first so that all blocks are resident and arrive together
    float x = threadIdx.x;
    for (int i = 0; i < work; ++i) x = x * 1.0001f + 0.5f;
    if (x == 12345.f) sink[1] = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        const u64 old = atomicAdd(ctr + (size_t)(blockIdx.x % nslots) * stride_words, 1ull);
        if (old == 0xFFFFFFFFFFFFull) sink[0] = old;
    }
}
int main() {
    u64 *ctr, *sink;
    hipMalloc(&ctr, 1 << 22); hipMalloc(&sink, 64);
    hipMemset(ctr, 0, 1 << 22);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks[] = {256, 768, 1024, 2048};
    struct Cfg { const char* name; int nslots, stride; };
    const Cfg cfgs[] = {{"none (0 atomics: slot stride, 1 block does it)", 0, 0}, {"1 word", 1, 1}, {"8 words in one 64-B line", 8, 1}, {"8 words 128 B apart", 8, 16},
                        {"8 words 4 KiB apart", 8, 512}, {"16 words 128 B apart", 16, 16}, {"64 words 128 B apart", 64, 16}, {"64 words, one line each 64 B", 64, 8}};
    for (int nb : blocks)
        for (const Cfg& c : cfgs) {
            float best = 1e9f;
            for (int rep = 0; rep < 20; ++rep) {
                hipEventRecord(e0);
                if (c.nslots == 0) hipLaunchKernelGGL(k_ticket, dim3(nb), dim3(256), 0, 0, ctr, 524288 / nb, nb, sink, 2000);   // every block its own line
                else hipLaunchKernelGGL(k_ticket, dim3(nb), dim3(256), 0, 0, ctr, c.stride, c.nslots, sink, 2000);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 3 && ms < best) best = ms;
            }
            printf("blocks %4d  %-48s %7.2f us\n", nb, c.name, best * 1e3f);
        }
    return 0;
}

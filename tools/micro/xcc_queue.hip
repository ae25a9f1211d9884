// Can a kernel on one stream consume, while it runs, what a kernel on another stream produces?   (GPU box)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/xcc_queue.hip -o /tmp/xcc_queue && /tmp/xcc_queue
// 1. which XCD a block runs on (HW_REG_XCC_ID) against blockIdx % 8, for a kernel alone and for two kernels in flight;
// 2. producer blocks write 64-bit payload words with relaxed agent-scope stores (write-through, sc1), wait for them
//    (s_waitcnt vmcnt(0)) and publish a descriptor word {epoch, base, n} the same way; consumer waves of a kernel that was
//    launched FIRST on another stream spin on the descriptor with relaxed agent-scope loads and then read the payload with
//    such loads: every word must be the one the producer wrote (checked over many epochs, no fences anywhere);
// 3. the rate of such loads / stores against ordinary ones.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u; }   // HW_REG_XCC_ID[3:0]

__global__ void k_where(int* out, int spin) {
    float x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
    if (threadIdx.x == 0) out[blockIdx.x] = (int)xcc_id() | (x == 123.f ? 256 : 0);
}

#define CHUNK 256
__global__ __launch_bounds__(512) void k_produce(u64* payload, u64* desc, u64* head, unsigned epoch, int chunks_per_block, int work) {
    float x = threadIdx.x;
    for (int i = 0; i < work * (1 + (int)(blockIdx.x % 7)); ++i) x = x * 1.0001f + 0.5f;      // blocks end at different times
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int c = w; c < chunks_per_block; c += 8) {
        u64 slot = 0;
        if (lane == 0) slot = atomicAdd(head, 1ull);
        slot = __shfl(slot, 0);
        const u64 base = slot * CHUNK;
        for (int k = lane; k < CHUNK; k += 64)
            __hip_atomic_store(payload + base + k, ((u64)epoch << 40) | ((base + k) * 2654435761ull & 0xFFFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(desc + slot, ((u64)epoch << 40) | (x == 123.f ? 1ull : 0ull) | (u64)(blockIdx.x + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// consumer waves take tickets; the total number of chunks is known (nchunks)
__global__ __launch_bounds__(256) void k_consume(const u64* payload, const u64* desc, u64* tail, unsigned epoch, u64 nchunks, u64* bad, u64* spins, u64* seen) {
    const int lane = threadIdx.x & 63;
    u64 nb = 0, nspin = 0, nseen = 0;
    for (;;) {
        u64 t = 0;
        if (lane == 0) t = atomicAdd(tail, 1ull);
        t = __shfl(t, 0);
        if (t >= nchunks) break;
        u64 d;
        for (;;) {
            d = __hip_atomic_load(desc + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(d >> 40) == epoch) break;
            ++nspin;
            __builtin_amdgcn_s_sleep(8);
        }
        const u64 base = t * CHUNK;
        for (int k = lane; k < CHUNK; k += 64) {
            const u64 v = __hip_atomic_load(payload + base + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != (((u64)epoch << 40) | ((base + k) * 2654435761ull & 0xFFFFFFFFFFull))) ++nb;
        }
        ++nseen;
    }
    if (nb) atomicAdd(bad, nb);
    if (lane == 0) { atomicAdd(spins, nspin); atomicAdd(seen, nseen); }
}

__global__ void k_stream(const u64* src, u64* dst, size_t n, int mode) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        u64 v;
        if (mode & 1) v = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else v = src[i];
        if (mode & 2) __hip_atomic_store(dst + i, v + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else dst[i] = v + 1;
    }
}

int main() {
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    // ---- 1
    int *d_w1, *d_w2;
    CK(hipMalloc(&d_w1, 4096 * 4)); CK(hipMalloc(&d_w2, 4096 * 4));
    std::vector<int> h1(4096), h2(4096);
    hipLaunchKernelGGL(k_where, dim3(768), dim3(512), 0, s1, d_w1, 100);
    CK(hipStreamSynchronize(s1));
    CK(hipMemcpy(h1.data(), d_w1, 768 * 4, hipMemcpyDeviceToHost));
    int agree = 0;
    for (int b = 0; b < 768; ++b) agree += (h1[b] & 15) == (b & 7);
    printf("one kernel, 768 blocks: XCC_ID == blockIdx %% 8 for %d blocks (first ids: %d %d %d %d %d %d %d %d %d)\n", agree, h1[0], h1[1], h1[2], h1[3], h1[4], h1[5], h1[6], h1[7], h1[8]);
    hipLaunchKernelGGL(k_where, dim3(768), dim3(512), 0, s1, d_w1, 20000);
    hipLaunchKernelGGL(k_where, dim3(512), dim3(256), 0, s2, d_w2, 20000);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h1.data(), d_w1, 768 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), d_w2, 512 * 4, hipMemcpyDeviceToHost));
    int a1 = 0, a2 = 0;
    for (int b = 0; b < 768; ++b) a1 += (h1[b] & 15) == (b & 7);
    for (int b = 0; b < 512; ++b) a2 += (h2[b] & 15) == (b & 7);
    printf("two kernels in flight: %d / 768 and %d / 512 blocks on XCD blockIdx %% 8\n", a1, a2);
    // ---- 2
    const int PB = 768, CPB = 8;
    const u64 nchunks = (u64)PB * CPB;
    u64 *payload, *desc, *ctr;
    CK(hipMalloc(&payload, nchunks * CHUNK * 8)); CK(hipMalloc(&desc, nchunks * 8)); CK(hipMalloc(&ctr, 4096));
    CK(hipMemset(payload, 0, nchunks * CHUNK * 8)); CK(hipMemset(desc, 0, nchunks * 8)); CK(hipMemset(ctr, 0, 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        float tot = 0;
        for (unsigned epoch = 1; epoch <= 200; ++epoch) {
            CK(hipMemsetAsync(ctr, 0, 16 * 8 * 2, s1));     // head = ctr[0], tail = ctr[16]; the check counters ctr[32..34] accumulate
            CK(hipStreamSynchronize(s1));
            CK(hipEventRecord(e0, s1));
            if (mode == 0) {      // consumer first, on its own stream: it spins while the producer runs
                hipLaunchKernelGGL(k_consume, dim3(512), dim3(256), 0, s2, payload, desc, ctr + 16, epoch, nchunks, ctr + 32, ctr + 33, ctr + 34);
                hipLaunchKernelGGL(k_produce, dim3(PB), dim3(512), 0, s1, payload, desc, ctr, epoch, CPB, 3000);
            } else {              // one after the other on one stream
                hipLaunchKernelGGL(k_produce, dim3(PB), dim3(512), 0, s1, payload, desc, ctr, epoch, CPB, 3000);
                hipLaunchKernelGGL(k_consume, dim3(512), dim3(256), 0, s1, payload, desc, ctr + 16, epoch, nchunks, ctr + 32, ctr + 33, ctr + 34);
            }
            CK(hipStreamSynchronize(s2));
            CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (epoch > 20) tot += ms;
        }
        u64 h[3];
        CK(hipMemcpy(h, ctr + 32, 24, hipMemcpyDeviceToHost));
        printf("%s: %.1f us per round (768 producer blocks, %llu chunks of %d words); wrong words %llu, spins %llu, chunks seen %llu (of %llu in 200 rounds)\n",
               mode == 0 ? "consumer kernel in flight beside the producer" : "producer, then consumer", tot / 180 * 1e3, (unsigned long long)nchunks, CHUNK,
               (unsigned long long)h[0], (unsigned long long)h[1], (unsigned long long)h[2], (unsigned long long)(nchunks * 200));
        CK(hipMemset(ctr + 32, 0, 24));
    }
    // ---- 3
    const size_t N = 1 << 22;      // 32 MiB
    u64 *a, *b;
    CK(hipMalloc(&a, N * 8)); CK(hipMalloc(&b, N * 8)); CK(hipMemset(a, 1, N * 8));
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 10; ++rep) {
            CK(hipEventRecord(e0, s1));
            hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, s1, a, b, N, mode);
            CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 2 && ms < best) best = ms;
        }
        printf("32 MiB read + 32 MiB written, %s loads, %s stores: %.1f us (%.0f GB/s)\n", mode & 1 ? "agent-scope" : "plain", mode & 2 ? "agent-scope" : "plain", best * 1e3, 2.0 * N * 8 / best / 1e6);
    }
    return 0;
}

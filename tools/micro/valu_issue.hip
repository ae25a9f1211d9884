// What is the VALU issue peak of one MI355X SIMD — a wave64 instruction every FOUR cycles (SIMD-16, what bench.py assumed through
// round 5) or every TWO (SIMD-32, /opt/skills/guides/MI355X_MICROARCH.md)?   (GPU box)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_issue.hip -o /tmp/valu_issue && /tmp/valu_issue [out.json]
// Every wave runs ITER iterations of UNROLL independent instructions of one kind (8 accumulator chains, so a chain's own latency
// never binds), on every SIMD of the chip (256 CUs x 4 SIMDs), with 1 .. 8 waves per SIMD.  Time comes from s_memrealtime
// (100 MHz, min over the waves' starts .. max over their ends, so the launch is not in it) and from HIP events; the shader clock
// from s_memtime / s_memrealtime inside the same kernel (s_memtime counts shader cycles on gfx9) and — printed beside it — from
// `rocm-smi --showclocks` if that is on the PATH.  Output: wave-instructions per second per SIMD and per chip, and cycles per
// instruction per SIMD, for v_fma_f32, v_pk_fma_f32, v_cmp + v_addc (the pair k_search's distance loop is made of, in its SGPR and
// its VCC encodings), the plain f32 / u32 VOP2 / VOP3 forms, the packed f32 and the f64 forms, v_mul_lo_u32, v_readlane_b32;
// 1, 2, 4, 8 waves per SIMD each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define ITER 2048
#define UNROLL 64      // instructions per iteration (8 chains x 8)

enum { K_FMA32 = 0, K_PKFMA32, K_CMP_ADDC, K_ADD_U32, K_FMA64, K_MUL_LO, K_CMP_VCC, K_CMP_SGPR, K_ADDC_VCC, K_CMP_ADDC_VCC, K_CNDMASK, K_SUB_F32,
       K_MUL_F32, K_FMAC_F32, K_PK_ADD_F32, K_PK_MUL_F32, K_LSHL_OR, K_AND_OR, K_ADD_F64, K_MUL_F64, K_READLANE, K_ADD3, K_KINDS };
static const char* kind_name[K_KINDS] = {"v_fma_f32", "v_pk_fma_f32", "v_cmp_le_f32_e64(sgpr)+v_addc_co_u32_e64", "v_add_u32", "v_fma_f64", "v_mul_lo_u32",
                                         "v_cmp_le_f32_e32(vcc)", "v_cmp_le_f32_e64(sgpr)", "v_addc_co_u32_e32(vcc)", "v_cmp_le_f32_e32+v_addc_co_u32_e32(vcc)",
                                         "v_cndmask_b32_e32(vcc)", "v_sub_f32", "v_mul_f32", "v_fmac_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_lshl_or_b32",
                                         "v_and_or_b32", "v_add_f64", "v_mul_f64", "v_readlane_b32", "v_add3_u32"};
// VALU instructions one "step" of the kind issues (the cmp + addc pairs are two)
static const int kind_insts[K_KINDS] = {1, 1, 2, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
// kinds that count towards the "plain full-rate" peak bench.py uses
static const bool kind_plain[K_KINDS] = {true, false, false, true, false, false, false, false, false, false, true, true, true, true, false, false, true, true, false, false, false, true};

typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(512) void k_issue(u64* __restrict__ t_out, float* __restrict__ sink, float seed) {
    const u64 rt0 = __builtin_amdgcn_s_memrealtime();
    const u64 ct0 = __builtin_amdgcn_s_memtime();
    float a[8];
    v2f p[8];
    double d[8];
    uint32_t u[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = seed + k + threadIdx.x; p[k] = v2f{seed + k, seed - k}; d[k] = seed + k; u[k] = (uint32_t)(threadIdx.x * 8 + k); }
    const float m = 1.0000001f, c = 1e-9f;
    const v2f pm = v2f{m, m}, pc = v2f{c, c};
    const double dm = 1.0000001, dc = 1e-12;
    uint32_t one = 1u + (uint32_t)(seed == 77.f);
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < UNROLL / 8; ++r) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (KIND == K_FMA32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
                if (KIND == K_PKFMA32) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(pm), "v"(pc));
                if (KIND == K_CMP_ADDC) {
                    u64 msk;
                    asm volatile("v_cmp_le_f32_e64 %1, %2, %3\n\tv_addc_co_u32_e64 %0, %1, %0, %0, %1" : "+v"(u[k]), "=&s"(msk) : "v"(a[k]), "v"(m));
                }
                if (KIND == K_ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k]) : "v"(one));
                if (KIND == K_FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k]) : "v"(dm), "v"(dc));
                if (KIND == K_MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[k]) : "v"(2654435761u));
                if (KIND == K_CMP_VCC) asm volatile("v_cmp_le_f32_e32 vcc, %0, %1" :: "v"(a[k]), "v"(m) : "vcc");
                if (KIND == K_CMP_SGPR) { u64 msk; asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(msk) : "v"(a[k]), "v"(m)); }
                if (KIND == K_ADDC_VCC) asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(u[k]) :: "vcc");
                if (KIND == K_CMP_ADDC_VCC) asm volatile("v_cmp_le_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(u[k]) : "v"(a[k]), "v"(m) : "vcc");
                if (KIND == K_CNDMASK) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(u[k]) : "v"(one) : "vcc");
                if (KIND == K_SUB_F32) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
                if (KIND == K_MUL_F32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
                if (KIND == K_FMAC_F32) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
                if (KIND == K_PK_ADD_F32) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pc));
                if (KIND == K_PK_MUL_F32) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pm));
                if (KIND == K_LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(u[k]) : "v"(one));
                if (KIND == K_AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(u[k]) : "v"(one));
                if (KIND == K_ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[k]) : "v"(dc));
                if (KIND == K_MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k]) : "v"(dm));
                if (KIND == K_READLANE) { uint32_t sv; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sv) : "v"(u[k])); }
                if (KIND == K_ADD3) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(u[k]) : "v"(one));
            }
        }
    }
    const u64 ct1 = __builtin_amdgcn_s_memtime();
    const u64 rt1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a[k] + p[k].x + p[k].y + (float)d[k] + (float)u[k];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        t_out[4 * w + 0] = rt0; t_out[4 * w + 1] = rt1; t_out[4 * w + 2] = ct0; t_out[4 * w + 3] = ct1;
    }
}

struct Row { int kind, waves_per_simd; double us_rt, us_ev, mhz, inst_per_s_simd, cyc_per_inst; };

template <int KIND>
static int run_kind(int num_cu, u64* d_t, float* d_sink, std::vector<u64>& h_t, std::vector<Row>& rows) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wps = 1; wps <= 8; wps = (wps < 2 ? 2 : wps * 2)) {
        // one block per CU with 4 * wps waves: the dispatcher spreads a block's waves over the CU's four SIMDs
        const int threads = 64 * 4 * wps;
        const int blocks = num_cu;
        // (more than 8 waves per CU: 256-thread blocks, wps of them per CU — every block one wave per SIMD)
        int bt = threads, nb = blocks;
        if (threads > 512) { bt = 256; nb = num_cu * wps; }
        double best_rt = 1e30, best_ev = 1e30, mhz = 0;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((k_issue<KIND>), dim3(nb), dim3(bt), 0, 0, d_t, d_sink, 1.0f);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const size_t nw = (size_t)nb * bt / 64;
            CK(hipMemcpy(h_t.data(), d_t, nw * 4 * sizeof(u64), hipMemcpyDeviceToHost));
            u64 lo = ~0ull, hi = 0;
            double clk = 0;
            for (size_t w = 0; w < nw; ++w) {
                lo = std::min(lo, h_t[4 * w]); hi = std::max(hi, h_t[4 * w + 1]);
                clk += (double)(h_t[4 * w + 3] - h_t[4 * w + 2]) / std::max<double>(1.0, (double)(h_t[4 * w + 1] - h_t[4 * w])) * 100.0;   // MHz
            }
            const double us = (double)(hi - lo) / 100.0;
            if (us < best_rt) { best_rt = us; mhz = clk / nw; }
            best_ev = std::min(best_ev, (double)ms * 1000.0);
        }
        const double insts_per_wave = (double)ITER * UNROLL * kind_insts[KIND];
        const double per_simd = insts_per_wave * wps / (best_rt * 1e-6);
        Row r{KIND, wps, best_rt, best_ev, mhz, per_simd, mhz * 1e6 / per_simd};
        rows.push_back(r);
        printf("%-28s waves/SIMD %d: %8.1f us (events %8.1f)  clock %6.0f MHz  %.4e wave-inst/s/SIMD  %.3f cycles/inst  chip %.4e /s\n",
               kind_name[KIND], wps, best_rt, best_ev, mhz, per_simd, r.cyc_per_inst, per_simd * num_cu * 4);
    }
    return 0;
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int num_cu = prop.multiProcessorCount;
    printf("%s, %d CUs, clockRate %d kHz\n", prop.gcnArchName, num_cu, prop.clockRate);
    u64* d_t; float* d_sink;
    const size_t maxw = (size_t)num_cu * 32;
    CK(hipMalloc(&d_t, maxw * 4 * sizeof(u64)));
    CK(hipMalloc(&d_sink, 64));
    std::vector<u64> h_t(maxw * 4);
    std::vector<Row> rows;
#define RUN(K) if (run_kind<K>(num_cu, d_t, d_sink, h_t, rows)) return 1
    RUN(K_FMA32); RUN(K_PKFMA32); RUN(K_CMP_ADDC); RUN(K_ADD_U32); RUN(K_FMA64); RUN(K_MUL_LO); RUN(K_CMP_VCC); RUN(K_CMP_SGPR); RUN(K_ADDC_VCC);
    RUN(K_CMP_ADDC_VCC); RUN(K_CNDMASK); RUN(K_SUB_F32); RUN(K_MUL_F32); RUN(K_FMAC_F32); RUN(K_PK_ADD_F32); RUN(K_PK_MUL_F32); RUN(K_LSHL_OR);
    RUN(K_AND_OR); RUN(K_ADD_F64); RUN(K_MUL_F64); RUN(K_READLANE); RUN(K_ADD3);
    std::string smi;
    if (FILE* f = popen("rocm-smi --showclocks 2>/dev/null | grep -i sclk | head -2", "r")) {
        char buf[256];
        while (fgets(buf, sizeof buf, f)) smi += buf;
        pclose(f);
    }
    printf("rocm-smi: %s\n", smi.c_str());
    if (argc > 1) {
        FILE* f = fopen(argv[1], "w");
        if (!f) return 1;
        // the figure bench.py uses: the best sustained rate of the plain one-pass VALU kinds (f32 fma / u32 add / cmp + addc)
        double peak = 0, peak_mhz = 0;
        for (const Row& r : rows)
            if (kind_plain[r.kind] && r.inst_per_s_simd > peak) { peak = r.inst_per_s_simd; peak_mhz = r.mhz; }
        fprintf(f, "{\"device\": \"%s\", \"num_cu\": %d, \"simds\": %d, \"iter\": %d, \"unroll\": %d,\n", prop.gcnArchName, num_cu, num_cu * 4, ITER, UNROLL);
        fprintf(f, " \"valu_issue_peak_per_s_per_simd\": %.6e, \"valu_issue_peak_per_s_chip\": %.6e, \"shader_clock_mhz_measured\": %.1f,\n", peak, peak * num_cu * 4, peak_mhz);
        fprintf(f, " \"cycles_per_wave64_instruction\": %.4f,\n", peak_mhz * 1e6 / peak);
        for (size_t i = 0; i < smi.size(); ++i) if (smi[i] == '"' || smi[i] == '\\' || (unsigned char)smi[i] < 32) smi[i] = ' ';
        fprintf(f, " \"rocm_smi_sclk\": \"%s\",\n \"rows\": [\n", smi.c_str());
        for (size_t i = 0; i < rows.size(); ++i) {
            const Row& r = rows[i];
            fprintf(f, "  {\"inst\": \"%s\", \"waves_per_simd\": %d, \"us_memrealtime\": %.2f, \"us_events\": %.2f, \"clock_mhz\": %.1f, \"wave_inst_per_s_per_simd\": %.6e, \"cycles_per_inst\": %.4f}%s\n",
                    kind_name[r.kind], r.waves_per_simd, r.us_rt, r.us_ev, r.mhz, r.inst_per_s_simd, r.cyc_per_inst, i + 1 < rows.size() ? "," : "");
        }
        fprintf(f, " ]}\n");
        fclose(f);
    }
    return 0;
}

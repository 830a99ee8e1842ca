// VALU issue ceiling of one MI355X SIMD, measured: independent wave64 VALU instructions at 1, 2, 4 and 8 waves per
// SIMD.  Diagnostic program, not part of libenzymm_amd.so:
//
//     hipcc --offload-arch=gfx950 -O3 -o profiles/valu_probe profiles/valu_probe.hip && profiles/valu_probe
//
// Every wave runs `iters` rounds of 32 instructions on 8 independent accumulators (no instruction depends on the one
// before it), reads the shader clock (s_memtime) before and after, and the host times the launch with HIP events.
// Printed per (instruction, waves per SIMD): shader-clock cycles per wave64 instruction per SIMD (= W x cycles one
// wave took / its instruction count), and wave-instructions per second over the chip from the event time.
// One JSON object on stdout; profiles/summarize_pmc.py takes its VALU issue ceiling from it.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(e)                                                                       \
    do {                                                                               \
        hipError_t e_ = (e);                                                           \
        if (e_ != hipSuccess) {                                                        \
            std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_));              \
            std::exit(1);                                                              \
        }                                                                              \
    } while (0)

enum Op { FMA_F32 = 0, ADD_U32 = 1, CMP_CNDMASK = 2, FMA_F64 = 3, MIX_SEARCH = 4 };

template <int kOp>
__global__ void k_probe(uint32_t iters, unsigned long long *__restrict__ ticks, float *__restrict__ sink) {
    extern __shared__ unsigned char pad[];  // sizes the number of workgroups a CU takes
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    const float x = 1.0000001f, y = 1e-9f;
    const double dx = 1.0000001, dy = 1e-9;
    uint32_t u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (kOp == FMA_F32) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
            } else if (kOp == ADD_U32) {
                asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                             "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(u0 | 1u));
            } else if (kOp == CMP_CNDMASK) {  // compare into VCC, select: the pair the search's "write lane" compiles to
                asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
                             "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %9, vcc"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "vcc");
            } else if (kOp == FMA_F64) {
                asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dx), "v"(dy));
            } else {  // the search's blend: subtract, fma, compare, and, shift, select, readlane-free integer work
                asm volatile("v_sub_f32 %0, %0, %8\n v_fma_f32 %1, %0, %0, %1\n v_cmp_ge_f32 vcc, %1, %9\n v_cndmask_b32 %2, %2, %9, vcc\n"
                             "v_and_b32 %4, %4, %5\n v_lshrrev_b32 %5, 3, %6\n v_add_u32 %6, %6, %7\n v_xor_b32 %7, %7, %4"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(y), "v"(x) : "vcc");
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) ticks[wave] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) +
                                                  (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7);
}

template <int kOp>
static void run(const char *name, int cus, bool last) {
    const uint32_t iters = 20000;
    const double per_wave = (double)iters * 32.0;
    std::printf("  \"%s\": {", name);
    const int levels[4] = {1, 2, 4, 8};
    for (int li = 0; li < 4; ++li) {
        const int w = levels[li];
        // W waves per SIMD = 4 W waves per CU: one workgroup of 256 W threads (W <= 4), or two of 1024 (W = 8); the
        // dynamic LDS request lets a CU take exactly that many workgroups
        const int block = w <= 4 ? 256 * w : 1024, per_cu = w <= 4 ? 1 : 2;
        const size_t lds = per_cu == 1 ? 100 * 1024 : 70 * 1024;
        const int grid = cus * per_cu;
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_probe<kOp>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        unsigned long long *d_ticks;
        float *d_sink;
        const size_t waves = (size_t)grid * block / 64;
        CHECK(hipMalloc(&d_ticks, waves * sizeof(unsigned long long)));
        CHECK(hipMalloc(&d_sink, (size_t)grid * block * sizeof(float)));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_probe<kOp>, dim3(grid), dim3(block), lds, 0, 2000u, d_ticks, d_sink);  // warm
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_probe<kOp>, dim3(grid), dim3(block), lds, 0, iters, d_ticks, d_sink);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> ticks(waves);
        CHECK(hipMemcpy(ticks.data(), d_ticks, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double mean = 0;
        for (auto t : ticks) mean += (double)t;
        mean /= (double)waves;
        // s_memtime ticks at 100 MHz on this part (a constant clock, not the shader clock): cycles are derived from
        // the event time and the clock the runtime reports instead, and the tick rate is printed for the record
        const double insts = per_wave * (double)waves;
        std::printf("%s\"%d\": {\"ms\": %.4f, \"wave_instructions_per_s\": %.4e, \"ns_per_instruction_per_simd\": %.4f, "
                    "\"memtime_ticks_per_wave\": %.1f}",
                    li ? ", " : "", w, ms, insts / (ms * 1e-3), (ms * 1e6) / (per_wave * w), mean);
        CHECK(hipFree(d_ticks)); CHECK(hipFree(d_sink));
        CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    }
    std::printf("}%s\n", last ? "" : ",");
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::printf("{\n  \"device\": \"%s\", \"cus\": %d, \"simds\": %d, \"clock_khz\": %d,\n", prop.gcnArchName, cus, cus * 4, prop.clockRate);
    std::printf("  \"note\": \"ns_per_instruction_per_simd = launch time / (instructions per wave x waves per SIMD): the time one SIMD "
                "needs per wave64 instruction with W waves issuing; x clock (GHz) = cycles\",\n");
    run<FMA_F32>("v_fma_f32", cus, false);
    run<ADD_U32>("v_add_u32", cus, false);
    run<CMP_CNDMASK>("v_cmp+v_cndmask", cus, false);
    run<FMA_F64>("v_fma_f64", cus, false);
    run<MIX_SEARCH>("search_mix", cus, true);
    std::printf("}\n");
    return 0;
}

// valu_rate.hip -- issue cost of the VALU instructions the SdfFuse kernels are built from, on the GPU it runs on.
// For each opcode: every SIMD holds W waves that each issue a long run of independent instances (16 accumulators,
// inline asm so nothing is folded); reports cycles per wave-instruction per SIMD from the shader clock (s_memtime)
// and from the wall clock.  Build: hipcc --offload-arch=gfx950 -O2 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define KERNEL_SCALAR(NAME, ASM)                                                                             \
    __global__ __launch_bounds__(256) void NAME(float* out, unsigned long long* cyc, int iters, float a, float b) \
    {                                                                                                        \
        float acc[16];                                                                                       \
        for (int i = 0; i < 16; ++i) acc[i] = a + (float)(threadIdx.x + i);                                   \
        float x = a, y = b;                                                                                  \
        const unsigned long long t0 = __builtin_readcyclecounter();                                          \
        for (int it = 0; it < iters; ++it) {                                                                 \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(acc[i]) : "v"(x), "v"(y));  \
        }                                                                                                    \
        const unsigned long long t1 = __builtin_readcyclecounter();                                          \
        float s = 0;                                                                                         \
        for (int i = 0; i < 16; ++i) s += acc[i];                                                            \
        if (s == 123.456f) out[0] = s;                                                                       \
        if ((threadIdx.x & 63) == 0) atomicMax(cyc, t1 - t0);                                                \
    }

#define KERNEL_PACKED(NAME, ASM)                                                                             \
    __global__ __launch_bounds__(256) void NAME(float* out, unsigned long long* cyc, int iters, float a, float b) \
    {                                                                                                        \
        v2f acc[16];                                                                                         \
        for (int i = 0; i < 16; ++i) { acc[i].x = a + (float)(threadIdx.x + i); acc[i].y = b + (float)i; }    \
        v2f x, y;                                                                                            \
        x.x = a; x.y = b; y.x = b; y.y = a;                                                                  \
        const unsigned long long t0 = __builtin_readcyclecounter();                                          \
        for (int it = 0; it < iters; ++it) {                                                                 \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(acc[i]) : "v"(x), "v"(y));  \
        }                                                                                                    \
        const unsigned long long t1 = __builtin_readcyclecounter();                                          \
        float s = 0;                                                                                         \
        for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;                                               \
        if (s == 123.456f) out[0] = s;                                                                       \
        if ((threadIdx.x & 63) == 0) atomicMax(cyc, t1 - t0);                                                \
    }

KERNEL_SCALAR(k_fma, "v_fma_f32 %0, %1, %2, %0")
KERNEL_SCALAR(k_mul, "v_mul_f32 %0, %1, %0")
KERNEL_SCALAR(k_add, "v_add_f32 %0, %1, %0")
KERNEL_SCALAR(k_mov, "v_mov_b32 %0, %1")
KERNEL_SCALAR(k_rcp, "v_rcp_f32 %0, %0")
KERNEL_SCALAR(k_rsq, "v_rsq_f32 %0, %0")
KERNEL_SCALAR(k_sqrt, "v_sqrt_f32 %0, %0")
KERNEL_SCALAR(k_exp, "v_exp_f32 %0, %0")
KERNEL_SCALAR(k_floor, "v_floor_f32 %0, %0")
KERNEL_SCALAR(k_cvt, "v_cvt_i32_f32 %0, %0")
KERNEL_SCALAR(k_max3, "v_max3_f32 %0, %1, %2, %0")
KERNEL_SCALAR(k_med3, "v_med3_f32 %0, %1, %2, %0")
KERNEL_SCALAR(k_divscale, "v_div_scale_f32 %0, vcc, %1, %2, %0")
KERNEL_SCALAR(k_divfmas, "v_div_fmas_f32 %0, %1, %2, %0")
KERNEL_SCALAR(k_divfixup, "v_div_fixup_f32 %0, %1, %2, %0")
KERNEL_SCALAR(k_cndmask, "v_cndmask_b32 %0, %1, %0, vcc")
KERNEL_SCALAR(k_cmp, "v_cmp_lt_f32 vcc, %1, %0")
KERNEL_SCALAR(k_mul_u24, "v_mul_u32_u24 %0, %1, %0")
KERNEL_SCALAR(k_mul_lo, "v_mul_lo_u32 %0, %1, %0")
KERNEL_SCALAR(k_fmac_dpp, "v_fmac_f32 %0, %1, %2")
KERNEL_SCALAR(k_cnd_sgpr, "v_cndmask_b32 %0, %1, %0, s[20:21]")
KERNEL_SCALAR(k_cnd_vcc_init, "v_cndmask_b32_e32 %0, %1, %0, vcc")
KERNEL_SCALAR(k_cmp_cnd, "v_cmp_lt_f32 vcc, %1, %0\n\tv_cndmask_b32 %0, %1, %0, vcc")
KERNEL_SCALAR(k_cmp_sgpr_cnd, "v_cmp_lt_f32 s[20:21], %1, %0\n\tv_cndmask_b32 %0, %1, %0, s[20:21]")
KERNEL_SCALAR(k_min, "v_min_f32 %0, %1, %0")
KERNEL_SCALAR(k_max, "v_max_f32 %0, %1, %0")
KERNEL_SCALAR(k_sub, "v_sub_f32 %0, %1, %0")
KERNEL_SCALAR(k_and, "v_and_b32 %0, %1, %0")
KERNEL_SCALAR(k_addu, "v_add_u32 %0, %1, %0")
KERNEL_SCALAR(k_lshladd, "v_lshl_add_u32 %0, %1, 2, %0")
KERNEL_SCALAR(k_mad24, "v_mad_u32_u24 %0, %1, %2, %0")
KERNEL_SCALAR(k_cmpclass, "v_cmp_class_f32 vcc, %0, %1")
KERNEL_SCALAR(k_fma_neg, "v_fma_f32 %0, -%1, %2, %0")
KERNEL_SCALAR(k_fma_sgpr, "v_fma_f32 %0, s20, %2, %0")
KERNEL_SCALAR(k_mul_sgpr_e32, "v_mul_f32_e32 %0, s20, %0")
KERNEL_SCALAR(k_add_sgpr_e32, "v_add_f32_e32 %0, s20, %0")
KERNEL_SCALAR(k_fmac_sgpr_e32, "v_fmac_f32_e32 %0, s20, %1")
KERNEL_SCALAR(k_fma_sgpr_add, "v_fma_f32 %0, %1, %0, s20")
KERNEL_SCALAR(k_cmp_sgpr_src, "v_cmp_lt_f32 vcc, s20, %0")
KERNEL_SCALAR(k_cmp_e64_sgpr_dst, "v_cmp_lt_f32 s[22:23], %1, %0")
KERNEL_SCALAR(k_med3_sgpr, "v_med3_f32 %0, %0, s20, s20")
KERNEL_SCALAR(k_max_sgpr, "v_max_f32_e32 %0, s20, %0")
KERNEL_SCALAR(k_mul_lit, "v_mul_f32 %0, 0x3f8ccccd, %0")
KERNEL_SCALAR(k_salu_mix, "v_fma_f32 %0, %1, %2, %0\n\ts_and_b64 s[20:21], s[20:21], exec")
KERNEL_PACKED(k_pk_fma, "v_pk_fma_f32 %0, %1, %2, %0")
KERNEL_PACKED(k_pk_mul, "v_pk_mul_f32 %0, %1, %0")
KERNEL_PACKED(k_pk_add, "v_pk_add_f32 %0, %1, %0")
KERNEL_PACKED(k_pk_mov, "v_pk_mov_b32 %0, %1, %2")

typedef void (*kern_t)(float*, unsigned long long*, int, float, float);

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2048;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clockRate %d kHz\n", prop.name, cus, prop.clockRate);
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 64);
    hipMalloc(&cyc, 8);
    struct { const char* name; kern_t k; } tests[] = {
        {"v_fma_f32", k_fma}, {"v_mul_f32", k_mul}, {"v_add_f32", k_add}, {"v_mov_b32", k_mov}, {"v_fmac_f32", k_fmac_dpp},
        {"v_pk_fma_f32", k_pk_fma}, {"v_pk_mul_f32", k_pk_mul}, {"v_pk_add_f32", k_pk_add}, {"v_pk_mov_b32", k_pk_mov},
        {"v_rcp_f32", k_rcp}, {"v_rsq_f32", k_rsq}, {"v_sqrt_f32", k_sqrt}, {"v_exp_f32", k_exp},
        {"v_floor_f32", k_floor}, {"v_cvt_i32_f32", k_cvt}, {"v_max3_f32", k_max3}, {"v_med3_f32", k_med3},
        {"v_div_scale_f32", k_divscale}, {"v_div_fmas_f32", k_divfmas}, {"v_div_fixup_f32", k_divfixup},
        {"v_cndmask_b32", k_cndmask}, {"v_cndmask sgpr mask", k_cnd_sgpr}, {"v_cndmask_e32 vcc", k_cnd_vcc_init},
        {"v_cmp+v_cndmask vcc", k_cmp_cnd}, {"v_cmp+v_cndmask sgpr", k_cmp_sgpr_cnd}, {"v_min_f32", k_min}, {"v_max_f32", k_max},
        {"v_sub_f32", k_sub}, {"v_and_b32", k_and}, {"v_add_u32", k_addu}, {"v_lshl_add_u32", k_lshladd}, {"v_mad_u32_u24", k_mad24},
        {"v_cmp_class_f32", k_cmpclass}, {"v_fma_f32 neg", k_fma_neg}, {"v_fma_f32 sgpr", k_fma_sgpr}, {"v_cmp_lt vcc, sgpr src", k_cmp_sgpr_src}, {"v_cmp_lt sgpr dst", k_cmp_e64_sgpr_dst},
        {"v_med3 sgpr srcs", k_med3_sgpr}, {"v_max_f32 sgpr src", k_max_sgpr}, {"v_mul_f32_e32 sgpr", k_mul_sgpr_e32}, {"v_add_f32_e32 sgpr", k_add_sgpr_e32},
        {"v_fmac_f32_e32 sgpr", k_fmac_sgpr_e32}, {"v_fma_f32 sgpr addend", k_fma_sgpr_add}, {"v_mul_f32 literal", k_mul_lit},
        {"v_fma + s_and_b64", k_salu_mix}, {"v_cmp_lt_f32", k_cmp}, {"v_mul_u32_u24", k_mul_u24}, {"v_mul_lo_u32", k_mul_lo},
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("%-18s %5s %14s %14s %10s\n", "opcode", "w/SIMD", "cyc/inst(shader)", "cyc/inst(wall@2.4)", "ms");
    for (auto& t : tests) {
        for (int wps : {2, 6}) { // waves per SIMD: blocks of 256 threads = 1 wave per SIMD each; wps blocks per CU
            const int blocks = cus * wps;
            hipMemset(cyc, 0, 8);
            t.k<<<blocks, 256>>>(out, cyc, 16, 1.0001f, 0.9999f); // warm
            hipMemset(cyc, 0, 8);
            hipEventRecord(e0);
            t.k<<<blocks, 256>>>(out, cyc, iters, 1.0001f, 0.9999f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c = 0;
            hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double inst_per_simd = (double)iters * 16 * wps;
            printf("%-18s %5d %14.2f %14.2f %10.4f\n", t.name, wps, (double)c / inst_per_simd, ms * 1e-3 * 2.4e9 / inst_per_simd, ms);
        }
    }
    return 0;
}

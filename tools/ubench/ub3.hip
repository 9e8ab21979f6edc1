// ub3 -- round-3 micro-benchmarks: VALU issue cost per wave on gfx950 at 1, 2 and 3 waves per SIMD
// (diagnostic tool, never shipped):  hipcc --offload-arch=gfx950 -O3 -o ub3 ub3.hip && ./ub3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define IT 512
typedef float f2 __attribute__((ext_vector_type(2)));

#define R8(X) X X X X X X X X
// 8 independent registers, 8 instructions per group, 8 groups per iteration = 64 instructions
#define BODY_FMA                                                     \
    "v_fma_f32 %0, %8, %9, %0\n\tv_fma_f32 %1, %8, %9, %1\n\t"       \
    "v_fma_f32 %2, %8, %9, %2\n\tv_fma_f32 %3, %8, %9, %3\n\t"       \
    "v_fma_f32 %4, %8, %9, %4\n\tv_fma_f32 %5, %8, %9, %5\n\t"       \
    "v_fma_f32 %6, %8, %9, %6\n\tv_fma_f32 %7, %8, %9, %7\n\t"
#define BODY_DPPF                                                                                   \
    "v_fmac_f32_dpp %0, %0, %9 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"               \
    "v_fmac_f32_dpp %1, %1, %9 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"               \
    "v_fmac_f32_dpp %2, %2, %9 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"               \
    "v_fmac_f32_dpp %3, %3, %9 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"               \
    "v_fmac_f32_dpp %4, %4, %9 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"               \
    "v_fmac_f32_dpp %5, %5, %9 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"               \
    "v_fmac_f32_dpp %6, %6, %9 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"               \
    "v_fmac_f32_dpp %7, %7, %9 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define BODY_DPPA                                                                                   \
    "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                \
    "v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                \
    "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                \
    "v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                \
    "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                \
    "v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                \
    "v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                \
    "v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define BODY_MOV                                                     \
    "v_mov_b32 %0, %8\n\tv_mov_b32 %1, %9\n\tv_mov_b32 %2, %8\n\tv_mov_b32 %3, %9\n\t" \
    "v_mov_b32 %4, %8\n\tv_mov_b32 %5, %9\n\tv_mov_b32 %6, %8\n\tv_mov_b32 %7, %9\n\t"

#define KERNEL_F(NAME, BODY)                                                                          \
    __global__ void NAME(float* o, long* dt) {                                                        \
        float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;               \
        float x = 0.999f, y = 0.5f;                                                                   \
        __syncthreads();                                                                              \
        long t0 = clock64();                                                                          \
        for (int i = 0; i < IT; ++i)                                                                  \
            asm volatile(R8(BODY)                                                                     \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(x), "v"(y));                                                           \
        long t1 = clock64();                                                                          \
        if ((threadIdx.x & 63) == 0) dt[threadIdx.x >> 6] = t1 - t0;                                  \
        o[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                       \
    }
KERNEL_F(k_fma, BODY_FMA)
KERNEL_F(k_dppf, BODY_DPPF)
KERNEL_F(k_dppa, BODY_DPPA)
KERNEL_F(k_mov, BODY_MOV)

// packed fma: 8 independent register pairs
#define BODY_PK                                                                \
    "v_pk_fma_f32 %0, %8, %9, %0\n\tv_pk_fma_f32 %1, %8, %9, %1\n\t"           \
    "v_pk_fma_f32 %2, %8, %9, %2\n\tv_pk_fma_f32 %3, %8, %9, %3\n\t"           \
    "v_pk_fma_f32 %4, %8, %9, %4\n\tv_pk_fma_f32 %5, %8, %9, %5\n\t"           \
    "v_pk_fma_f32 %6, %8, %9, %6\n\tv_pk_fma_f32 %7, %8, %9, %7\n\t"
// the decode kernel's form: second operand a scalar broadcast to both halves (op_sel_hi:[1,0,1])
#define BODY_PKB                                                                                        \
    "v_pk_fma_f32 %0, %8, %9, %0 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %1, %8, %9, %1 op_sel_hi:[1,0,1]\n\t" \
    "v_pk_fma_f32 %2, %8, %9, %2 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %3, %8, %9, %3 op_sel_hi:[1,0,1]\n\t" \
    "v_pk_fma_f32 %4, %8, %9, %4 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %5, %8, %9, %5 op_sel_hi:[1,0,1]\n\t" \
    "v_pk_fma_f32 %6, %8, %9, %6 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %7, %8, %9, %7 op_sel_hi:[1,0,1]\n\t"
#define KERNEL_P(NAME, BODY)                                                                          \
    __global__ void NAME(float* o, long* dt) {                                                        \
        f2 a0 = {(float)threadIdx.x, 1}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0; \
        f2 x = {0.999f, 0.998f}, y = {0.5f, 0.25f};                                                   \
        __syncthreads();                                                                              \
        long t0 = clock64();                                                                          \
        for (int i = 0; i < IT; ++i)                                                                  \
            asm volatile(R8(BODY)                                                                     \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(x), "v"(y));                                                           \
        long t1 = clock64();                                                                          \
        if ((threadIdx.x & 63) == 0) dt[threadIdx.x >> 6] = t1 - t0;                                  \
        f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                 \
        o[threadIdx.x] = s.x + s.y;                                                                   \
    }
KERNEL_P(k_pk, BODY_PK)
KERNEL_P(k_pkb, BODY_PKB)

// a dependent v_fma chain in waves 0-3 (one per SIMD) beside independent pk_fma streams in the other waves
__global__ void k_mix(float* o, long* dt, int chain_prio) {
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if (wave < 4) {
        if (chain_prio) __builtin_amdgcn_s_setprio(3);
        float a = threadIdx.x, x = 0.999f, y = 0.5f;
        long t0 = clock64();
        for (int i = 0; i < IT; ++i)
            asm volatile(R8("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t"
                            "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t")
                         : "+v"(a)
                         : "v"(x), "v"(y));
        long t1 = clock64();
        if ((threadIdx.x & 63) == 0) dt[wave] = t1 - t0;
        o[threadIdx.x] = a;
    } else {
        f2 a0 = {(float)threadIdx.x, 1}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
        f2 x = {0.999f, 0.998f}, y = {0.5f, 0.25f};
        long t0 = clock64();
        for (int i = 0; i < IT; ++i)
            asm volatile(R8(BODY_PK)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(x), "v"(y));
        long t1 = clock64();
        if ((threadIdx.x & 63) == 0) dt[wave] = t1 - t0;
        f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
        o[threadIdx.x] = s.x + s.y;
    }
}

// waves 0-3: mode 0 idle (exit), 1 independent v_fma, 2 dependent v_fma chain, 3 independent pk_fma (separate copy of the code),
// 4 dependent LDS-read chain; the other waves: independent pk_fma streams
__global__ void k_mix2(float* o, long* dt, int mode) {
    __shared__ int s[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) s[i] = (i * 7 + 1) % 2048;
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if (wave < 4) {
        float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, x = 0.999f, y = 0.5f;
        long t0 = clock64();
        if (mode == 1) {
            for (int i = 0; i < IT; ++i)
                asm volatile(R8(BODY_FMA)
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : "v"(x), "v"(y));
        } else if (mode == 2) {
            for (int i = 0; i < IT; ++i)
                asm volatile(R8("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t"
                                "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t")
                             : "+v"(a0)
                             : "v"(x), "v"(y));
        } else if (mode == 3) {
            f2 b0 = {a0, 1}, b1 = b0, b2 = b0, b3 = b0, b4 = b0, b5 = b0, b6 = b0, b7 = b0;
            f2 xx = {0.999f, 0.998f}, yy = {0.5f, 0.25f};
            for (int i = 0; i < IT; ++i)
                asm volatile(R8(BODY_PK)
                             : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7)
                             : "v"(xx), "v"(yy));
            a0 = b0.x + b1.y + b2.x + b3.x + b4.x + b5.x + b6.x + b7.x;
        } else if (mode == 4) {
            int p = threadIdx.x;
            for (int i = 0; i < IT * 4; ++i) p = s[p];
            a0 = p;
        }
        long t1 = clock64();
        if ((threadIdx.x & 63) == 0) dt[wave] = t1 - t0;
        o[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    } else {
        f2 a0 = {(float)threadIdx.x, 1}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
        f2 x = {0.999f, 0.998f}, y = {0.5f, 0.25f};
        long t0 = clock64();
        for (int i = 0; i < IT; ++i)
            asm volatile(R8(BODY_PK)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(x), "v"(y));
        long t1 = clock64();
        if ((threadIdx.x & 63) == 0) dt[wave] = t1 - t0;
        f2 sm = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
        o[threadIdx.x] = sm.x + sm.y;
    }
}

int main() {
    float* o;
    long *dt, h[16];
    hipMalloc(&o, 4096);
    hipMalloc(&dt, 16 * sizeof(long));
#define RUN(K, NAME)                                                                          \
    for (int nw = 4; nw <= 12; nw += 4) {                                                     \
        hipLaunchKernelGGL(K, dim3(1), dim3(64 * nw), 0, 0, o, dt);                           \
        hipLaunchKernelGGL(K, dim3(1), dim3(64 * nw), 0, 0, o, dt);                           \
        hipDeviceSynchronize();                                                               \
        hipMemcpy(h, dt, sizeof h, hipMemcpyDeviceToHost);                                    \
        printf("%-44s %2d waves (%d per SIMD): cyc per instruction, waves 0.. :", NAME, nw, nw / 4);  \
        for (int w = 0; w < nw; ++w) printf(" %.2f", (double)h[w] / (IT * 64.0));             \
        printf("\n");                                                                         \
    }
    RUN(k_fma, "independent v_fma_f32")
    RUN(k_pk, "independent v_pk_fma_f32")
    RUN(k_pkb, "independent v_pk_fma_f32 op_sel_hi:[1,0,1]")
    RUN(k_dppf, "independent v_fmac_f32_dpp row_shl:1")
    RUN(k_dppa, "independent v_add_f32_dpp row_shr:1")
    RUN(k_mov, "independent v_mov_b32")
    for (int prio = 0; prio < 2; ++prio)
        for (int nw = 4; nw <= 12; nw += 4) {
            hipLaunchKernelGGL(k_mix, dim3(1), dim3(64 * nw), 0, 0, o, dt, prio);
            hipLaunchKernelGGL(k_mix, dim3(1), dim3(64 * nw), 0, 0, o, dt, prio);
            hipDeviceSynchronize();
            hipMemcpy(h, dt, sizeof h, hipMemcpyDeviceToHost);
            printf("dependent v_fma chain (prio %d) beside %d pk_fma wave(s) per SIMD: chain %6.2f cyc per fma", prio ? 3 : 0,
                   nw / 4 - 1, (double)h[0] / (IT * 64.0));
            if (nw > 4) printf(", pk stream %6.2f cyc per pk_fma per wave", (double)h[4] / (IT * 64.0));
            printf("\n");
        }
    const char* names[5] = {"idle", "independent v_fma", "dependent v_fma chain", "independent pk_fma (own code copy)", "dependent ds_read chain"};
    for (int mode = 0; mode < 5; ++mode)
        for (int nw = 8; nw <= 12; nw += 4) {
            hipLaunchKernelGGL(k_mix2, dim3(1), dim3(64 * nw), 0, 0, o, dt, mode);
            hipLaunchKernelGGL(k_mix2, dim3(1), dim3(64 * nw), 0, 0, o, dt, mode);
            hipDeviceSynchronize();
            hipMemcpy(h, dt, sizeof h, hipMemcpyDeviceToHost);
            printf("waves 0-3 %-36s (%6.2f cyc per op) beside %d pk_fma wave(s) per SIMD: %6.2f cyc per pk_fma per wave\n",
                   names[mode], (double)h[0] / (IT * 64.0) / (mode == 4 ? 1.0 / 16 : 1.0), nw / 4 - 1, (double)h[4] / (IT * 64.0));
        }
    return 0;
}

// ub2 -- round-2 micro-benchmarks of the gfx950 latencies the decode kernel's sample-to-sample chain is made of
// (diagnostic tool, never shipped):  hipcc --offload-arch=gfx950 -O3 -o ub2 ub2.hip && ./ub2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define IT 2048
typedef float f2 __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ void fin(float* o, float v, long dt, int slot) {
    if (threadIdx.x == 0) {
        o[0] = v;
        ((long*)o)[1 + slot] = dt;
    }
}
// ---- barrier only ----
__global__ void k_bar(float* o) {
    long t0 = clock64();
    for (int i = 0; i < IT; ++i) __builtin_amdgcn_s_barrier();
    fin(o, 0, clock64() - t0, 0);
}
// ---- write -> barrier -> read (one hop per iteration; WAR safe by double buffering) ----
__global__ void k_hop(float* o) {
    __shared__ float s[2][1024];
    float v = threadIdx.x;
    long t0 = clock64();
    for (int i = 0; i < IT; ++i) {
        s[i & 1][threadIdx.x] = v;
        __syncthreads();
        v = s[i & 1][(threadIdx.x + 64) % blockDim.x] + 1.0f;
    }
    fin(o, v, clock64() - t0, 0);
}
// ---- one hop wave 0 -> wave 1 -> wave 0 by polling an LDS word (two hops per iteration) ----
__global__ void k_poll(float* o) {
    __shared__ volatile int f0, f1;
    if (threadIdx.x == 0) f0 = f1 = -1;
    __syncthreads();
    const int w = threadIdx.x >> 6;
    long t0 = clock64();
    if (w == 0) {
        for (int i = 0; i < IT; ++i) {
            if ((threadIdx.x & 63) == 0) f0 = i;
            while (f1 != i) {}
        }
    } else if (w == 1) {
        for (int i = 0; i < IT; ++i) {
            while (f0 != i) {}
            if ((threadIdx.x & 63) == 0) f1 = i;
        }
    }
    fin(o, 0, clock64() - t0, 0);
}
// ---- dependent LDS reads ----
__global__ void k_lds32(float* o) {
    __shared__ int s[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) s[i] = (i * 7 + 1) % 2048;
    __syncthreads();
    int p = threadIdx.x;
    long t0 = clock64();
    for (int i = 0; i < IT; ++i) p = s[p];
    fin(o, p, clock64() - t0, 0);
}
__global__ void k_lds64(float* o) {
    __shared__ int2 s[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) s[i] = make_int2((i * 7 + 1) % 2048, i);
    __syncthreads();
    int p = threadIdx.x;
    long t0 = clock64();
    for (int i = 0; i < IT; ++i) {
        int2 q = s[p];
        p = q.x + (q.y & 0);
    }
    fin(o, p, clock64() - t0, 0);
}
__global__ void k_lds128(float* o) {
    __shared__ int4 s[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) s[i] = make_int4((i * 7 + 1) % 2048, i, 0, 0);
    __syncthreads();
    int p = threadIdx.x;
    long t0 = clock64();
    for (int i = 0; i < IT; ++i) {
        int4 q = s[p];
        p = q.x + (q.w & 0);
    }
    fin(o, p, clock64() - t0, 0);
}
__global__ void k_bperm(float* o) {
    int p = threadIdx.x;
    long t0 = clock64();
    for (int i = 0; i < IT; ++i) p = __builtin_amdgcn_ds_bpermute(((p + 1) & 63) << 2, p);
    fin(o, p, clock64() - t0, 0);
}
// ---- dependent DPP adds / readlane / plain fma ----
__global__ void k_dpp(float* o) {
    float v = threadIdx.x;
    long t0 = clock64();
    for (int i = 0; i < IT; ++i) v = v + dpp_f<0x111>(v);
    fin(o, v, clock64() - t0, 0);
}
__global__ void k_readlane(float* o) {
    float v = threadIdx.x;
    long t0 = clock64();
    for (int i = 0; i < IT; ++i) v = v + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
    fin(o, v, clock64() - t0, 0);
}
__global__ void k_fma(float* o, float a, float b) {
    float v = threadIdx.x;
    long t0 = clock64();
    for (int i = 0; i < IT; ++i) v = fmaf(v, a, b);
    fin(o, v, clock64() - t0, 0);
}
// dependent fma chain in wave 0 while the other waves stream independent pk_fma; PRIO: priority of wave 0
template <int PRIO>
__global__ void k_fma_contended(float* o, float a, float b) {
    const int w = threadIdx.x >> 6;
    if (w < 4) {
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
        float v = threadIdx.x;
        long t0 = clock64();
        for (int i = 0; i < IT; ++i) v = fmaf(v, a, b);
        long dt = clock64() - t0;
        if (threadIdx.x == 0) {
            o[0] = v;
            ((long*)o)[1] = dt;
        }
    } else {
        f2 x0 = {(float)threadIdx.x, 1}, x1 = {2, 3}, x2 = {4, 5}, x3 = {6, 7};
        f2 A = {a, a}, B = {b, b};
        for (int i = 0; i < IT * 2; ++i) {
            x0 = __builtin_elementwise_fma(x0, A, B);
            x1 = __builtin_elementwise_fma(x1, A, B);
            x2 = __builtin_elementwise_fma(x2, A, B);
            x3 = __builtin_elementwise_fma(x3, A, B);
        }
        if (threadIdx.x == 64 * 5) o[2] = x0.x + x1.y + x2.x + x3.y;
    }
}
// ---- table activation chain (the kernel's lut_scaled) vs hardware exp + rcp ----
__global__ void k_lut(float* o, float a) {
    __shared__ float2 T[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) T[i] = make_float2(tanhf(i / 512.0f), tanhf((i + 1) / 512.0f) - tanhf(i / 512.0f));
    __syncthreads();
    float x = 0.001f * threadIdx.x;
    long t0 = clock64();
    for (int i = 0; i < IT; ++i) {
        const float u = fminf(fabsf(x) * 512.0f, 4095.99976f);
        const float f = __builtin_amdgcn_fractf(u);
        const float2 td = T[(unsigned)u];
        x = copysignf(fmaf(f, td.y, td.x), x) + a;
    }
    fin(o, x, clock64() - t0, 0);
}
__global__ void k_hwsig(float* o, float a) {
    float x = 0.001f * threadIdx.x;
    long t0 = clock64();
    for (int i = 0; i < IT; ++i) x = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.442695f)) + a;
    fin(o, x, clock64() - t0, 0);
}
// ---- L2-resident row gather: dependent dwordx3 loads from a 3.5 MB table ----
__global__ void k_gather(const float* tab, float* o) {
    unsigned row = threadIdx.x * 7 + blockIdx.x;
    float acc = 0;
    long t0 = clock64();
    for (int i = 0; i < 512; ++i) {
        const float* p = tab + (size_t)(row % 768) * 1152 + 3 * (threadIdx.x % 384);
        const float x = p[0], y = p[1], z = p[2];
        acc += x + y + z;
        row = row * 5 + 1 + (unsigned)(acc != 12345.0f);
    }
    long dt = clock64() - t0;
    if (threadIdx.x == 0) {
        o[blockIdx.x * 8] = acc;
        ((long*)o)[1 + blockIdx.x * 4] = dt;
    }
}

int main() {
    float* o;
    hipMalloc(&o, 1 << 20);
    long h[2048];
    float* tab;
    hipMalloc(&tab, 768 * 1152 * 4);
    hipMemset(tab, 0, 768 * 1152 * 4);
    auto rd = [&]() {
        hipDeviceSynchronize();
        hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost);
    };
    for (int rep = 0; rep < 2; ++rep) {
        for (int nt = 256; nt <= 1024; nt += 256) {
            hipLaunchKernelGGL(k_bar, dim3(1), dim3(nt), 0, 0, o);
            rd();
            printf("s_barrier only, %2d waves:            %.1f cyc\n", nt / 64, (double)h[1] / IT);
            hipLaunchKernelGGL(k_hop, dim3(1), dim3(nt), 0, 0, o);
            rd();
            printf("write->barrier->read hop, %2d waves:  %.1f cyc\n", nt / 64, (double)h[1] / IT);
        }
        hipLaunchKernelGGL(k_poll, dim3(1), dim3(128), 0, 0, o);
        rd();
        printf("LDS flag poll, one hop (2 waves):      %.1f cyc\n", (double)h[1] / IT / 2);
        hipLaunchKernelGGL(k_lds32, dim3(1), dim3(64), 0, 0, o);
        rd();
        printf("dependent ds_read_b32:   %.1f cyc\n", (double)h[1] / IT);
        hipLaunchKernelGGL(k_lds64, dim3(1), dim3(64), 0, 0, o);
        rd();
        printf("dependent ds_read_b64:   %.1f cyc\n", (double)h[1] / IT);
        hipLaunchKernelGGL(k_lds128, dim3(1), dim3(64), 0, 0, o);
        rd();
        printf("dependent ds_read_b128:  %.1f cyc\n", (double)h[1] / IT);
        hipLaunchKernelGGL(k_bperm, dim3(1), dim3(64), 0, 0, o);
        rd();
        printf("dependent ds_bpermute:   %.1f cyc\n", (double)h[1] / IT);
        hipLaunchKernelGGL(k_dpp, dim3(1), dim3(64), 0, 0, o);
        rd();
        printf("dependent v_add_dpp:     %.1f cyc\n", (double)h[1] / IT);
        hipLaunchKernelGGL(k_readlane, dim3(1), dim3(64), 0, 0, o);
        rd();
        printf("dependent readlane+add:  %.1f cyc\n", (double)h[1] / IT);
        hipLaunchKernelGGL(k_fma, dim3(1), dim3(64), 0, 0, o, 1.0001f, 0.5f);
        rd();
        printf("dependent fma alone:     %.1f cyc\n", (double)h[1] / IT);
        hipLaunchKernelGGL(k_fma_contended<0>, dim3(1), dim3(768), 0, 0, o, 1.0001f, 0.5f);
        rd();
        printf("dependent fma, 2 pk_fma streams on the SIMD, prio 0: %.1f cyc\n", (double)h[1] / IT);
        hipLaunchKernelGGL(k_fma_contended<3>, dim3(1), dim3(768), 0, 0, o, 1.0001f, 0.5f);
        rd();
        printf("dependent fma, 2 pk_fma streams on the SIMD, prio 3: %.1f cyc\n", (double)h[1] / IT);
        hipLaunchKernelGGL(k_lut, dim3(1), dim3(64), 0, 0, o, 0.01f);
        rd();
        printf("table activation chain:  %.1f cyc\n", (double)h[1] / IT);
        hipLaunchKernelGGL(k_hwsig, dim3(1), dim3(64), 0, 0, o, 0.01f);
        rd();
        printf("hw exp+rcp sigmoid chain: %.1f cyc\n", (double)h[1] / IT);
        hipLaunchKernelGGL(k_gather, dim3(1), dim3(384), 0, 0, tab, o);
        rd();
        printf("dependent dwordx3 row gather (L2), 1 WG:   %.1f cyc\n", (double)h[1] / 512);
        hipLaunchKernelGGL(k_gather, dim3(256), dim3(384), 0, 0, tab, o);
        rd();
        printf("dependent dwordx3 row gather (L2), 256 WG: %.1f cyc\n", (double)h[1] / 512);
    }
    return 0;
}

// ub6: the decode kernel's gate-phase gather in isolation -- 384 lanes (6 waves) fetch THE SAME three table rows per step
// (row numbers known only when the previous step's values are in: a dependent chain like the sample loop's), one step =
// 3 x 4 608 bytes out of a 3.5 MB table that sits in the L2.  Variants of the layout / the load shape:
//   0  [row][unit][3] interleaved, one dwordx3 per row and lane (the shipped form)
//   1  [row][3][unit] planar, three dword loads per row and lane
//   2  [row][unit][4] padded to 16 bytes, one dwordx4 per row and lane (4.7 MB table)
//   3  interleaved, dwordx2 + dword
//   4  like 0 with two rows a step, 5 like 0 with one row a step (the per-row cost)
// hipcc --offload-arch=gfx950 -O3 -o tools/ubench/ub6 tools/ubench/ub6.hip
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int U = 384, ROWS = 768, IT = 2048;
template <int V>
__global__ __launch_bounds__(384) void k(const float* __restrict__ tab, float* o) {
    const unsigned l = threadIdx.x;
    unsigned r = blockIdx.x * 13u + 5u;
    float acc = 0.0f;
    const long t0 = clock64();
    for (int i = 0; i < IT; ++i) {
        const unsigned ra = r % ROWS, rb = (r * 7u + 3u) % ROWS, rc = (r * 11u + 1u) % ROWS;
        float s = 0.0f;
        auto row = [&](unsigned q) {
            if (V == 0 || V == 4 || V == 5) {
                const float* p = tab + (size_t)q * (3 * U) + 3 * l;
                struct F3 { float x, y, z; };
                const F3 v = *reinterpret_cast<const F3*>(p);
                s += (v.x + v.y) + v.z;
            } else if (V == 1) {
                const float* p = tab + (size_t)q * (3 * U) + l;
                s += (p[0] + p[U]) + p[2 * U];
            } else if (V == 2) {
                const float4 v = *reinterpret_cast<const float4*>(tab + (size_t)q * (4 * U) + 4 * l);
                s += (v.x + v.y) + v.z;
            } else {
                const float* p = tab + (size_t)q * (3 * U) + 3 * l;
                float x, y, z;
                asm volatile("global_load_dword %0, %1, off" : "=v"(x) : "v"(p));
                asm volatile("global_load_dword %0, %1, off offset:4" : "=v"(y) : "v"(p));
                asm volatile("global_load_dword %0, %1, off offset:8\n\ts_waitcnt vmcnt(0)" : "=v"(z) : "v"(p));
                s += (x + y) + z;
            }
        };
        row(ra);
        if (V != 5) row(rb);
        if (V != 5 && V != 4) row(rc);
        acc += s;
        // the next rows depend on what came back (uniform: every lane adds the same zero-valued table)
        r = r * 5u + 1u + (unsigned)(__builtin_amdgcn_readfirstlane(__float_as_int(acc)) != 12345);
        __syncthreads();
    }
    const long dt = clock64() - t0;
    if (l == 0) {
        o[blockIdx.x * 8] = acc;
        ((long*)o)[1 + blockIdx.x * 4] = dt;
    }
}
int main() {
    float *o, *tab;
    hipMalloc(&o, 1 << 20);
    hipMalloc(&tab, (size_t)ROWS * 4 * U * 4);
    hipMemset(tab, 0, (size_t)ROWS * 4 * U * 4);
    long h[2048];
    const char* name[6] = {"interleaved, dwordx3", "planar, 3 x dword", "padded, dwordx4", "interleaved, 3 x dword (asm)", "dwordx3, two rows", "dwordx3, one row"};
    for (int nb : {1, 256}) {
        for (int v = 0; v < 6; ++v) {
            for (int rep = 0; rep < 2; ++rep) {
                switch (v) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(nb), dim3(384), 0, 0, tab, o); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(nb), dim3(384), 0, 0, tab, o); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(nb), dim3(384), 0, 0, tab, o); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(nb), dim3(384), 0, 0, tab, o); break;
                    case 4: hipLaunchKernelGGL(k<4>, dim3(nb), dim3(384), 0, 0, tab, o); break;
                    default: hipLaunchKernelGGL(k<5>, dim3(nb), dim3(384), 0, 0, tab, o); break;
                }
                hipDeviceSynchronize();
            }
            hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost);
            printf("%3d workgroups, %-30s %.0f cycles per step (incl. one barrier)\n", nb, name[v], (double)h[1] / IT);
        }
    }
    return 0;
}

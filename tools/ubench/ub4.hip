// ub4 -- round-3 micro-benchmark: how fast can every CU stream the SAME L2-resident 2.67 MB (the predictor's weight set)?
// (diagnostic tool, never shipped):  hipcc --offload-arch=gfx950 -O3 -o ub4 ub4.hip && ./ub4
// One workgroup per CU sweeps the buffer REP times; variants differ in threads per workgroup, 16-byte loads in flight per
// thread, and the load flavour (plain, nontemporal, LDS-DMA).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr size_t BYTES = 2669640 / 16 * 16;  // ~2.67 MB like the weight set
constexpr int REP = 40;

template <int NT, int CD, int FLAVOUR>
__global__ __launch_bounds__(NT) void k_stream(const float4* __restrict__ w, size_t n16, float* out) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int tid = threadIdx.x;
    // every workgroup walks the buffer in the same order (what the predictor does); slice = whole buffer
    for (int rep = 0; rep < REP; ++rep) {
        for (size_t base = 0; base + (size_t)NT * CD <= n16; base += (size_t)NT * CD) {
            float4 v[CD];
#pragma unroll
            for (int j = 0; j < CD; ++j) {
                const float4* p = w + base + (size_t)j * NT + tid;
                if (FLAVOUR == 1) {
                    typedef float v4 __attribute__((ext_vector_type(4)));
                    const v4 t = __builtin_nontemporal_load(reinterpret_cast<const v4*>(p));
                    v[j] = make_float4(t.x, t.y, t.z, t.w);
                } else {
                    v[j] = *p;
                }
            }
#pragma unroll
            for (int j = 0; j < CD; ++j) {
                acc.x = fmaf(v[j].x, 1.0001f, acc.x);
                acc.y = fmaf(v[j].y, 1.0001f, acc.y);
                acc.z = fmaf(v[j].z, 1.0001f, acc.z);
                acc.w = fmaf(v[j].w, 1.0001f, acc.w);
            }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x * NT + tid] = acc.x;
}

// staggered start: workgroup b begins its sweep at a different offset (same bytes, different order)
template <int NT, int CD>
__global__ __launch_bounds__(NT) void k_stream_stagger(const float4* __restrict__ w, size_t n16, float* out) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int tid = threadIdx.x;
    const size_t chunk = (size_t)NT * CD, nchunk = n16 / chunk;
    const size_t start = ((size_t)blockIdx.x * 37) % nchunk;
    for (int rep = 0; rep < REP; ++rep) {
        for (size_t c = 0; c < nchunk; ++c) {
            const size_t base = ((c + start) % nchunk) * chunk;
            float4 v[CD];
#pragma unroll
            for (int j = 0; j < CD; ++j) v[j] = w[base + (size_t)j * NT + tid];
#pragma unroll
            for (int j = 0; j < CD; ++j) {
                acc.x = fmaf(v[j].x, 1.0001f, acc.x);
                acc.y = fmaf(v[j].y, 1.0001f, acc.y);
                acc.z = fmaf(v[j].z, 1.0001f, acc.z);
                acc.w = fmaf(v[j].w, 1.0001f, acc.w);
            }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x * NT + tid] = acc.x;
}

template <class K>
static void run(const char* name, K kern, int nt, int grid, const float4* w, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(nt), 0, 0, w, BYTES / 16, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(nt), 0, 0, w, BYTES / 16, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_cu = (double)BYTES * REP / (ms * 1e-3) / 1e9;
    printf("%-58s grid %3d: %7.3f ms  %6.1f GB/s per workgroup  %6.2f TB/s chip\n", name, grid, ms, per_cu, per_cu * grid / 1e3);
}

int main() {
    float4* w;
    float* out;
    hipMalloc(&w, BYTES);
    hipMalloc(&out, 1 << 22);
    hipMemset(w, 0, BYTES);
    for (int grid : {256, 128, 32, 2}) {
        run("576 thr, 16 x 16 B in flight, plain", k_stream<576, 16, 0>, 576, grid, w, out);
        run("576 thr, 8 x 16 B in flight, plain", k_stream<576, 8, 0>, 576, grid, w, out);
        run("576 thr, 32 x 16 B in flight, plain", k_stream<576, 32, 0>, 576, grid, w, out);
        run("1024 thr, 16 x 16 B in flight, plain", k_stream<1024, 16, 0>, 1024, grid, w, out);
        run("256 thr, 32 x 16 B in flight, plain", k_stream<256, 32, 0>, 256, grid, w, out);
        run("576 thr, 16 x 16 B in flight, nontemporal", k_stream<576, 16, 1>, 576, grid, w, out);
        run("576 thr, 16 x 16 B in flight, staggered start per workgroup", k_stream_stagger<576, 16>, 576, grid, w, out);
    }
    return 0;
}

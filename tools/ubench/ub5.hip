// ub5 -- round-3 micro-benchmark: what does one hop of the predictor's exchange cost, same XCD against different XCDs, for
// each store / load cache policy?  (diagnostic tool, never shipped):  hipcc --offload-arch=gfx950 -O3 -o ub5 ub5.hip && ./ub5
// Two workgroups of 256 threads play ping-pong with NV tagged 8-byte granules {epoch, value} per direction: side A stores
// NV granules, side B polls all of them (one lane per granule), barrier, answers with its own NV granules, A polls.
// Reported: cycles per round trip (two hops) at A, the XCC_ID of both sides; every spin is bounded (a policy that never
// becomes visible reports "never seen").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

constexpr int NV = 192, ITER = 400, NTHR = 256;

template <int SF>
__device__ __forceinline__ void st(unsigned long long* p, unsigned long long v) {
    if constexpr (SF == 0) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if constexpr (SF == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    if constexpr (SF == 2) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if constexpr (SF == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if constexpr (SF == 4) asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
}
template <int LF>
__device__ __forceinline__ unsigned long long ld(const unsigned long long* p) {
    unsigned long long v;
    if constexpr (LF == 0) asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (LF == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (LF == 2) asm volatile("global_load_dwordx2 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (LF == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int SF, int LF>
__global__ __launch_bounds__(NTHR) void k_pingpong(unsigned long long* g, int ba, int bb, long long* out) {
    const int tid = threadIdx.x;
    const bool A = blockIdx.x == ba, B = blockIdx.x == bb;
    if (!A && !B) return;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* mine = g + (A ? 0 : 1024);   // the granules this side writes
    unsigned long long* theirs = g + (A ? 1024 : 0);
    __shared__ int failed;
    if (tid == 0) failed = 0;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 1; it <= ITER; ++it) {
        if (A) {
            if (tid < NV) st<SF>(&mine[tid], ((unsigned long long)it << 32) | tid);
        }
        // poll the partner's granules of this iteration
        bool bad = false;
        if (tid < NV) {
            int spins = 0;
            while (true) {
                const unsigned long long v = ld<LF>(&theirs[tid]);
                if ((unsigned)(v >> 32) == (unsigned)it) break;
                if (++spins > 200000) {
                    bad = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (__syncthreads_or(bad)) {
            if (tid == 0) failed = 1;
            break;
        }
        if (B) {
            if (tid < NV) st<SF>(&mine[tid], ((unsigned long long)it << 32) | tid);
        }
    }
    __syncthreads();
    const long long t1 = __builtin_readcyclecounter();
    if (tid == 0) {
        out[A ? 0 : 2] = failed ? -1 : (t1 - t0) / ITER;
        out[A ? 1 : 3] = xcc & 0xf;
    }
    // a failed side leaves its partner spinning to its own bound: both end
}

template <int SF, int LF>
static void run(const char* sname, const char* lname, unsigned long long* g, long long* out_d, int ba, int bb) {
    long long h[4] = {0, 0, 0, 0};
    hipMemset(g, 0, 2048 * 8);
    hipMemset(out_d, 0, sizeof h);
    hipLaunchKernelGGL((k_pingpong<SF, LF>), dim3(64), dim3(NTHR), 0, 0, g, ba, bb, out_d);
    hipDeviceSynchronize();
    hipMemcpy(h, out_d, sizeof h, hipMemcpyDeviceToHost);
    if (h[0] < 0 || h[2] < 0)
        printf("blocks %2d,%2d (XCC %lld,%lld)  store %-8s load %-8s : never seen\n", ba, bb, h[1], h[3], sname, lname);
    else
        printf("blocks %2d,%2d (XCC %lld,%lld)  store %-8s load %-8s : %6lld cycles per round trip (2 hops of %d granules)\n", ba, bb,
               h[1], h[3], sname, lname, h[0], NV);
}

int main() {
    unsigned long long* g;
    long long* out;
    hipMalloc(&g, 2048 * 8);
    hipMalloc(&out, 64);
    for (int pair = 0; pair < 2; ++pair) {
        const int ba = 0, bb = pair == 0 ? 8 : 1;
        run<2, 0>("sc1", "sc1", g, out, ba, bb);      // the shipped exchange (agent-scope atomics)
        run<3, 1>("sc0 sc1", "sc0 sc1", g, out, ba, bb);
        run<0, 0>("plain", "sc1", g, out, ba, bb);
        run<0, 2>("plain", "nt", g, out, ba, bb);
        run<1, 0>("sc0", "sc1", g, out, ba, bb);
        run<4, 2>("nt", "nt", g, out, ba, bb);
        run<4, 0>("nt", "sc1", g, out, ba, bb);
        run<0, 3>("plain", "sc0", g, out, ba, bb);
        run<2, 2>("sc1", "nt", g, out, ba, bb);
    }
    return 0;
}

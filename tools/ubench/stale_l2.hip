// Microbenchmark: can a kernel read a stale per-XCD L2 line across a kernel boundary?
//
// Stream A runs a chain of small kernels.  In step k, block b owns segment (b + k*shift) % G of
// a buffer: every thread checks that its word holds k-1 and writes k.  With shift != 0 a segment
// is handled by a different workgroup index -- hence (round-robin dispatch) a different XCD --
// each step, and comes back to an XCD that read it a few steps earlier.  If that XCD still holds
// the old clean line, the check fails.  Optionally a second stream runs an unrelated long kernel
// at the same time.  Variants: plain, with in-kernel agent-scope acquire/release fences.
//
//   ./stale_l2 [steps=2000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <bool FENCE>
__global__ __launch_bounds__(256) void step(uint32_t *x, uint32_t k, uint32_t shift, uint32_t G, unsigned long long *err, uint32_t *first_bad)
{
    if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const uint32_t seg = (blockIdx.x + k * shift) % G;
    const uint32_t i = seg * 256u + threadIdx.x;
    const uint32_t v = x[i];
    if (v != k - 1u) {
        if (atomicAdd(err, 1ull) == 0ull) { first_bad[0] = k; first_bad[1] = v; first_bad[2] = i; }
    }
    x[i] = k;
    if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
}

__global__ __launch_bounds__(256) void noise(const float4 *t, uint32_t n, int iters, float *out)
{
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % n;
    float acc = 0.f;
    for (int i = 0; i < iters; i++) { const float4 v = t[idx]; acc += v.x; idx = (idx * 1664525u + 1013904223u + __float_as_uint(v.w)) % n; }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

int main(int argc, char **argv)
{
    const uint32_t steps = argc > 1 ? (uint32_t)atoi(argv[1]) : 2000u;
    hipStream_t sa, sb;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    const uint32_t nt = 1u << 22;
    float4 *t; float *o; hipMalloc(&t, (size_t)nt * 16); hipMemset(t, 0, (size_t)nt * 16); hipMalloc(&o, 2048 * 256 * 4);
    unsigned long long *err; uint32_t *bad; hipMalloc(&err, 8); hipMalloc(&bad, 16);
    for (uint32_t G : {64u, 1024u}) {
        uint32_t *x; hipMalloc(&x, (size_t)G * 256 * 4);
        for (int fence = 0; fence < 2; fence++)
            for (int conc = 0; conc < 2; conc++)
                for (uint32_t shift : {0u, 1u, 3u}) {
                    hipMemset(x, 0, (size_t)G * 256 * 4); hipMemset(err, 0, 8); hipMemset(bad, 0, 16);
                    hipDeviceSynchronize();
                    for (uint32_t k = 1; k <= steps; k++) {
                        if (conc && (k % 50u) == 1u) hipLaunchKernelGGL(noise, dim3(1024), dim3(256), 0, sb, t, nt, 400, o);
                        if (fence) hipLaunchKernelGGL(step<true>, dim3(G), dim3(256), 0, sa, x, k, shift, G, err, bad);
                        else hipLaunchKernelGGL(step<false>, dim3(G), dim3(256), 0, sa, x, k, shift, G, err, bad);
                    }
                    hipDeviceSynchronize();
                    unsigned long long e = 0; uint32_t b[4] = {0, 0, 0, 0};
                    hipMemcpy(&e, err, 8, hipMemcpyDeviceToHost); hipMemcpy(b, bad, 16, hipMemcpyDeviceToHost);
                    printf("{\"G\": %u, \"fence\": %d, \"concurrent\": %d, \"shift\": %u, \"steps\": %u, \"stale_reads\": %llu, \"first\": [%u, %u, %u]}\n",
                           G, fence, conc, shift, steps, e, b[0], b[1], b[2]);
                    fflush(stdout);
                }
        hipFree(x);
    }
    return 0;
}

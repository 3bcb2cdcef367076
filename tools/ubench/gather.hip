// Microbenchmark: rate of per-lane random record gathers (N x dwordx4 per record) from a table of given size.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int NV>
__global__ __launch_bounds__(64) void k(const float4 *__restrict__ tab, uint32_t nrec, int iters, float *out)
{
    uint32_t s = (blockIdx.x * 64 + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    uint32_t idx = s % nrec;
    for (int i = 0; i < iters; i++) {
        const float4 *p = tab + (size_t)idx * NV;
        float4 v[NV];
#pragma unroll
        for (int j = 0; j < NV; j++) v[j] = p[j];
#pragma unroll
        for (int j = 0; j < NV; j++) acc += v[j].x + v[j].w;
        // dependent chase: next index depends on loaded data (like a BVH walk)
        idx = (__float_as_uint(v[NV - 1].w) ^ (s += 0x9e3779b9u)) % nrec;
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}
int main(int argc, char **argv)
{
    int iters = 2000;
    for (size_t mb : {1, 16, 128, 1024}) {
        for (int nv : {1, 2, 4}) {
            uint32_t nrec = (uint32_t)(mb * 1024 * 1024 / (16 * nv));
            std::vector<float4> h((size_t)nrec * nv);
            for (size_t i = 0; i < h.size(); i++) { uint32_t r = (uint32_t)(i * 2246822519u + 3266489917u); h[i] = float4{1.f, 2.f, 3.f, 0.f}; ((uint32_t *)&h[i])[3] = r; }
            float4 *d; float *o;
            hipMalloc(&d, h.size() * 16); hipMemcpy(d, h.data(), h.size() * 16, hipMemcpyHostToDevice);
            for (int wpc : {8, 20}) {
                int blocks = 256 * wpc;
                hipMalloc(&o, blocks * 64 * 4);
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                for (int rep = 0; rep < 2; rep++) {
                    hipEventRecord(e0);
                    if (nv == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, d, nrec, iters, o);
                    if (nv == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, d, nrec, iters, o);
                    if (nv == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64), 0, 0, d, nrec, iters, o);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                }
                float ms; hipEventElapsedTime(&ms, e0, e1);
                double recs = (double)blocks * 64 * iters;
                printf("table %5zu MB rec %2d B waves/CU %2d : %7.2f G rec/s  %6.2f TB/s  %.3f lane-rec/clk/CU\n", mb, nv * 16, wpc,
                       recs / ms / 1e6, recs * nv * 16 / ms / 1e9, recs / (ms * 1e-3) / 256 / 2.4e9);
                hipFree(o);
            }
            hipFree(d);
        }
    }
    return 0;
}

// GPU box: does a line WRITTEN by one kernel come back from the Infinity Cache when the next kernel reads it?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mall_probe tools/probes/mall_probe.hip && /tmp/mall_probe
// For X = 16 MB .. 1 GB: kernel W stores X bytes, kernel R loads them (a sum per lane, one word per wave): R's rate after W,
// R's rate after R, and R's rate after W with Y MB of other traffic in between (a copy inside another buffer).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void wr(uint4 *p, size_t n, uint32_t v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(v, (uint32_t)i, v, v);
}
__global__ void rd(const uint4 *p, size_t n, unsigned long long *out) {
    unsigned long long s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s;      // (a word per wave: 8192 atomics on ONE word would take 100 us)
}
__global__ void cp(const uint4 *a, uint4 *b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main() {
    const size_t MAXB = (size_t)1 << 30;
    uint4 *buf, *other; unsigned long long *out;
    CHK(hipMalloc(&buf, MAXB)); CHK(hipMalloc(&other, 2 * MAXB)); CHK(hipMalloc(&out, 8 * 65536));
    CHK(hipMemset(other, 1, 2 * MAXB));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int G = 256 * 8, T = 256;
    auto timed_rd = [&](size_t n) { float ms; hipEventRecord(e0, 0); hipLaunchKernelGGL(rd, dim3(G), dim3(T), 0, 0, buf, n, out); hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); return ms; };
    printf("%8s %12s %12s %12s | R after W with other traffic in between (MB: TB/s)\n", "X MB", "W TB/s", "R after W", "R after R");
    for (size_t mb : {16, 32, 64, 96, 128, 192, 256, 384, 512, 1024}) {
        const size_t n = (mb << 20) / 16;
        double w = 0, rw = 0, rr = 0;
        const int reps = 5;
        for (int r = 0; r < reps + 1; ++r) {
            float ms;
            hipLaunchKernelGGL(cp, dim3(G), dim3(T), 0, 0, other, other + (MAXB / 16), MAXB / 16);          // flush: 2 GB of other traffic
            hipEventRecord(e0, 0); hipLaunchKernelGGL(wr, dim3(G), dim3(T), 0, 0, buf, n, (uint32_t)r); hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            const float a = timed_rd(n), b = timed_rd(n);
            if (r) { w += ms; rw += a; rr += b; }
        }
        const double bytes = (double)(mb << 20) * reps;
        printf("%8zu %12.2f %12.2f %12.2f |", mb, bytes / w / 1e9, bytes / rw / 1e9, bytes / rr / 1e9);
        for (size_t y : {32, 64, 128, 192}) {
            double t = 0;
            for (int r = 0; r < reps; ++r) {
                hipLaunchKernelGGL(cp, dim3(G), dim3(T), 0, 0, other, other + (MAXB / 16), MAXB / 16);
                hipLaunchKernelGGL(wr, dim3(G), dim3(T), 0, 0, buf, n, (uint32_t)r);
                hipLaunchKernelGGL(cp, dim3(G), dim3(T), 0, 0, other, other + (MAXB / 16), (y << 20) / 32);   // y MB of traffic: half read, half written
                t += timed_rd(n);
            }
            printf("  %zu: %.2f", y, bytes / t / 1e9);
        }
        printf("\n"); fflush(stdout);
    }
    return 0;
}

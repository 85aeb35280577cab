// which XCD (and CU) does block b of a 256-block grid of 1024-thread, 160-KB-LDS workgroups run on?   hipcc --offload-arch=gfx950 -o xcc_probe xcc_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(1024) void probe(unsigned *out) {
    extern __shared__ unsigned char lds[];
    if (threadIdx.x == 0) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        lds[0] = (unsigned char)xcc;
        out[2 * blockIdx.x] = xcc;
        out[2 * blockIdx.x + 1] = hwid;
    }
}
int main() {
    unsigned *d;
    const int grids[3] = {256, 512, 1024};
    hipMalloc(&d, 2 * 1024 * sizeof(unsigned));
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int gi = 0; gi < 3; ++gi) {
        const int g = grids[gi];
        hipLaunchKernelGGL(probe, dim3(g), dim3(1024), gi == 0 ? 160 * 1024 : 64 * 1024, 0, d);
        std::vector<unsigned> h(2 * g);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        int match = 0;
        printf("grid %d:", g);
        for (int b = 0; b < g; ++b) { if ((h[2 * b] & 0xf) == (unsigned)(b % 8)) ++match; if (b < 24) printf(" %u", h[2 * b] & 0xf); }
        printf(" ...  xcc == b %% 8 for %d of %d blocks\n", match, g);
    }
    return 0;
}

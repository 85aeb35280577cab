// GPU box: can an ALU/LDS-bound workgroup and a memory-streaming workgroup share a CU and overlap?  (VERDICT r4 item 4b asked
// whether ALU-bound part1 can run beside the memory-bound list passes; the CU-mask experiment of round 5 answered it for a
// SPATIAL split of the chip -- no.  This asks it for two half-size workgroups on the SAME CU.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/smt_probe tools/probes/smt_probe.hip && /tmp/smt_probe
// A ("alu"):    512 threads, 81 KB of LDS, 256 blocks: per record two 3-word strand rolls, a borrow-chain compare, one 64-bit multiply,
//               a returning LDS atomic addressed by the hash, an 8-byte LDS read, later a scattered 8-byte LDS write and one coalesced
//               8-byte global store -- the instruction mix of part1_kernel's fused loop, a barrier every 16 records per lane.
// B ("stream"): 512 threads, 72 KB of LDS, 256 blocks: a wave streams rows of 64 records (next round asked for a round ahead), ranks
//               them by 7 key bits with a returning LDS atomic, scatters them into an LDS stage, and the block copies the stage out in
//               whole lines -- part2f_kernel's shape at half its size.
// Runs: A alone, B alone (one and two blocks per CU), A and B on two streams (A launched first: two A's do not fit one CU, one A and
// one B do).  Per kernel: first start to last end by the 100-MHz clock, so the concurrent run shows what each role took.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ unsigned where_am_i() {
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    return ((xcc & 0xf) << 8) | ((hwid >> 8) & 0xff);          // XCD, then se / sh / cu
}
struct Stamp { unsigned long long t0, t1; unsigned where, pad; };

template <int TH>
__global__ __launch_bounds__(TH, 4) void alu_kernel(uint64_t *__restrict__ out, uint32_t iters, Stamp *st) {
    extern __shared__ __align__(16) unsigned char s_raw[];
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_raw);                 // 1024 + 64
    uint64_t *s_stage = reinterpret_cast<uint64_t *>(s_raw + 8192);        // 8192 records
    const int t = threadIdx.x;
    unsigned long long t0 = 0;
    if (t == 0) t0 = wall_clock64();
    for (int i = t; i < 2048; i += TH) s_cnt[i] = 0;
    for (int i = t; i < 8192; i += TH) s_stage[i] = (uint64_t)i * 0x9E3779B97F4A7C15ull;
    __syncthreads();
    uint32_t f0 = (uint32_t)t * 2654435761u + blockIdx.x, f1 = f0 * 40503u + 1u, f2 = f1 ^ 0x5bd1e995u;
    uint32_t r0 = ~f2, r1 = ~f1, r2 = ~f0;
    uint64_t acc = 0;
    uint64_t *dst = out + (uint64_t)blockIdx.x * iters * (TH * 16);
    for (uint32_t it = 0; it < iters; ++it) {
        uint64_t rec[16];
        uint32_t br[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t cj = (f0 >> 7 ^ f1 >> 13) & 3u;
            f2 = __builtin_amdgcn_alignbit(f2, f1, 30); f1 = __builtin_amdgcn_alignbit(f1, f0, 30); f0 = (f0 << 2) | cj; f2 &= 0x3ffu;
            r0 = __builtin_amdgcn_alignbit(r1, r0, 2); r1 = __builtin_amdgcn_alignbit(r2, r1, 2); r2 = (r2 >> 2) | ((cj ^ 3u) << 8);
            uint32_t m0, m1, m2, d;
            asm("v_sub_co_u32_e32 %3, vcc, %4, %7\n\tv_subb_co_u32_e32 %3, vcc, %5, %8, vcc\n\tv_subb_co_u32_e32 %3, vcc, %6, %9, vcc\n\t"
                "v_cndmask_b32_e32 %0, %7, %4, vcc\n\tv_cndmask_b32_e32 %1, %8, %5, vcc\n\tv_cndmask_b32_e32 %2, %9, %6, vcc"
                : "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(d) : "v"(r0), "v"(r1), "v"(r2), "v"(f0), "v"(f1), "v"(f2) : "vcc");
            uint64_t x = ((uint64_t)m1 << 32) | m0;
            x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 29;
            const uint32_t hi = (m2 ^ (uint32_t)(x >> 30)) & 0x3ffu;
            const uint32_t b = hi;
            rec[j] = x;
            br[j] = (b << 16) | (atomicAdd(&s_cnt[b], 1u) & 0xffffu);
            const uint64_t prev = s_stage[(t + j * TH) & 8191];                       // the copy-out side: one staged record per hashed base
            dst[(uint64_t)it * (TH * 16) + j * TH + t] = prev ^ acc;
            acc += prev;
        }
        lds_barrier();
        for (int i = t; i < 1024; i += TH) s_cnt[i] = 0;
        lds_barrier();
#pragma unroll
        for (int j = 0; j < 16; ++j) s_stage[((br[j] >> 16) * 8 + (br[j] & 7u)) & 8191] = rec[j];      // scattered by hash
        lds_barrier();
    }
    out[(uint64_t)blockIdx.x * TH + t] ^= acc;
    if (t == 0) { st[blockIdx.x].t0 = t0; st[blockIdx.x].t1 = wall_clock64(); st[blockIdx.x].where = where_am_i(); }
}

template <int ROWS, int TH>
__global__ __launch_bounds__(TH, 4) void stream_kernel(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, uint32_t rounds, Stamp *st) {
    extern __shared__ __align__(16) unsigned char s_raw[];
    constexpr int TILE = TH * ROWS;
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_raw);                 // 128 + pad
    uint64_t *s_stage = reinterpret_cast<uint64_t *>(s_raw + 1024);        // TILE
    const int t = threadIdx.x;
    const uint32_t wave = t >> 6, lane = t & 63;
    unsigned long long t0 = 0;
    if (t == 0) t0 = wall_clock64();
    if (t < 128) s_cnt[t] = 0;
    __syncthreads();
    const uint64_t *src = in + ((uint64_t)blockIdx.x * (TH / 64) + wave) * rounds * (64 * ROWS);
    uint64_t *dst = out + (uint64_t)blockIdx.x * rounds * TILE;
    uint64_t rec[ROWS], nxt[ROWS];
#pragma unroll
    for (int j = 0; j < ROWS; ++j) rec[j] = src[j * 64 + lane];
#pragma unroll
    for (int j = 0; j < ROWS; ++j) asm volatile("" : "+v"(rec[j]));
    for (uint32_t round = 0; round < rounds; ++round) {
        if (round + 1 < rounds) {
#pragma unroll
            for (int j = 0; j < ROWS; ++j) nxt[j] = src[(uint64_t)(round + 1) * (64 * ROWS) + j * 64 + lane];
        }
        uint32_t br[ROWS];
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const uint32_t b2 = (uint32_t)(rec[j] >> 50) & 127u;
            br[j] = (b2 << 16) | (atomicAdd(&s_cnt[b2], 1u) & 0xffffu);
        }
        lds_barrier();
        if (t < 128) s_cnt[t] = 0;
        lds_barrier();
#pragma unroll
        for (int j = 0; j < ROWS; ++j) s_stage[((br[j] >> 16) * (TILE / 128) + (br[j] & 0xffffu)) % TILE] = rec[j];
#pragma unroll
        for (int j = 0; j < ROWS; ++j) asm volatile("" : "+v"(nxt[j]));                  // the next round's records have arrived: before the stores
        lds_barrier();
#pragma unroll 4
        for (int i = t; i < TILE; i += TH) dst[(uint64_t)round * TILE + i] = s_stage[i];
#pragma unroll
        for (int j = 0; j < ROWS; ++j) rec[j] = nxt[j];
    }
    if (t == 0) { st[blockIdx.x].t0 = t0; st[blockIdx.x].t1 = wall_clock64(); st[blockIdx.x].where = where_am_i(); }
}
__global__ void fill(uint64_t *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint64_t x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; p[i] = x ^ (x >> 31); }
}

static double span_ms(const std::vector<Stamp> &s) {
    unsigned long long a = ~0ull, b = 0;
    for (auto &x : s) { if (x.t0 < a) a = x.t0; if (x.t1 > b) b = x.t1; }
    return (double)(b - a) / 1e5;            // 100 MHz
}
int main(int argc, char **argv) {
    constexpr int ROWS = 12, NB = 256;
    const uint32_t recs_per_cu_b = 12 * 512 * 340;                           // B: records per CU (4.3 GB in, 4.3 GB out over the chip)
    const uint32_t recs_per_cu_a = 16 * 512 * 256;                           // A: records per CU (4.3 GB out over the chip = half the 47-Mb piece)
    const size_t nrec = (size_t)NB * recs_per_cu_b;
    uint64_t *in, *outb, *outa;
    Stamp *sa, *sb;
    CHK(hipMalloc(&in, nrec * 8 + (64 << 20))); CHK(hipMalloc(&outb, nrec * 8 + (64 << 20))); CHK(hipMalloc(&outa, (size_t)NB * recs_per_cu_a * 8 + (64 << 20)));
    CHK(hipMalloc(&sa, 1024 * sizeof(Stamp))); CHK(hipMalloc(&sb, 1024 * sizeof(Stamp)));
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, in, nrec);
#define ATTR(f) CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(f), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
    ATTR(alu_kernel<1024>); ATTR(alu_kernel<768>); ATTR(alu_kernel<512>);
    ATTR((stream_kernel<ROWS, 1024>)); ATTR((stream_kernel<ROWS, 512>)); ATTR((stream_kernel<ROWS, 256>));
    hipStream_t s0, s1;
    CHK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CHK(hipDeviceSynchronize());
    auto get = [&](Stamp *d, int n) { std::vector<Stamp> h(n); (void)hipMemcpy(h.data(), d, n * sizeof(Stamp), hipMemcpyDeviceToHost); return h; };
    auto wall = [&](auto &&fn) { (void)hipDeviceSynchronize(); auto a = std::chrono::steady_clock::now(); fn(); (void)hipDeviceSynchronize(); return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    // same work whatever the workgroup size: a CU's block does recs_per_cu records
    auto launch_a = [&](int th, size_t lds, hipStream_t st) {
        const uint32_t iters = recs_per_cu_a / (16 * th);
        if (th == 1024) hipLaunchKernelGGL(alu_kernel<1024>, dim3(NB), dim3(1024), lds, st, outa, iters, sa);
        else if (th == 768) hipLaunchKernelGGL(alu_kernel<768>, dim3(NB), dim3(768), lds, st, outa, iters, sa);
        else hipLaunchKernelGGL(alu_kernel<512>, dim3(NB), dim3(512), lds, st, outa, iters, sa);
    };
    auto launch_b = [&](int th, size_t lds, hipStream_t st) {
        const uint32_t rounds = recs_per_cu_b / (ROWS * th);
        if (th == 1024) hipLaunchKernelGGL((stream_kernel<ROWS, 1024>), dim3(NB), dim3(1024), lds, st, in, outb, rounds, sb);
        else if (th == 512) hipLaunchKernelGGL((stream_kernel<ROWS, 512>), dim3(NB), dim3(512), lds, st, in, outb, rounds, sb);
        else hipLaunchKernelGGL((stream_kernel<ROWS, 256>), dim3(NB), dim3(256), lds, st, in, outb, rounds, sb);
    };
    for (int rep = 0; rep < 3; ++rep) {
        printf("--- repetition %d (A: %.2f GB written, B: %.2f GB read + as much written)\n", rep, NB * (double)recs_per_cu_a * 8 / 1e9, nrec * 8 / 1e9);
        double ta[3], tb[3];
        const int tha[3] = {1024, 768, 512}, thb[3] = {1024, 512, 256};
        for (int i = 0; i < 3; ++i) { wall([&] { launch_a(tha[i], 100 * 1024, s0); }); ta[i] = span_ms(get(sa, NB)); }
        for (int i = 0; i < 3; ++i) { wall([&] { launch_b(thb[i], 100 * 1024, s1); }); tb[i] = span_ms(get(sb, NB)); }
        printf("alone, one block per CU: A 16 / 12 / 8 waves %.3f %.3f %.3f ms | B 16 / 8 / 4 waves %.3f %.3f %.3f ms | A16 then B16: %.3f ms\n", ta[0], ta[1], ta[2], tb[0], tb[1], tb[2], ta[0] + tb[0]);
        struct Pair { int a, b; size_t la, lb; } pairs[2] = {{512, 512, 88 * 1024, 68 * 1024}, {768, 256, 124 * 1024, 35 * 1024}};
        for (auto &pr : pairs) {
            const double w = wall([&] { launch_a(pr.a, pr.la, s0); launch_b(pr.b, pr.lb, s1); });
            const auto ha = get(sa, NB), hb = get(sb, NB);
            std::map<unsigned, int> ca, cb;
            for (auto &x : ha) ca[x.where]++;
            for (auto &x : hb) cb[x.where]++;
            int paired = 0;
            for (auto &kv : ca) if (cb.count(kv.first)) ++paired;
            unsigned long long a0 = ~0ull, a1 = 0, b0 = ~0ull, b1 = 0;
            for (auto &x : ha) { if (x.t0 < a0) a0 = x.t0; if (x.t1 > a1) a1 = x.t1; }
            for (auto &x : hb) { if (x.t0 < b0) b0 = x.t0; if (x.t1 > b1) b1 = x.t1; }
            printf("A %d + B %d waves on a CU: wall %.3f, A %.3f, B %.3f, both %.3f ms (CUs with both %d)\n", pr.a / 64, pr.b / 64, w, (double)(a1 - a0) / 1e5, (double)(b1 - b0) / 1e5,
                   (double)((a1 > b1 ? a1 : b1) - (a0 < b0 ? a0 : b0)) / 1e5, paired);
        }
        fflush(stdout);
    }
    return 0;
}

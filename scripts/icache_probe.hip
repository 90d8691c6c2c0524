// Does code that a launch executes ONCE pay for its instruction fetch?  A block of straight-line, dependency-free VALU
// instructions of 4 .. 128 KB is run twice per launch (same addresses); shader cycles of pass 1 and pass 2, for the first launch
// and for a later one (the same kernel back to back, and with another kernel in between).
// Build: hipcc --offload-arch=gfx950 -O3 -w scripts/icache_probe.hip -o scripts/icache_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KB>
__global__ void __launch_bounds__(256) k_straight(long long* out) {
    long long t[3];
    unsigned a = threadIdx.x, b = 3;
    for (int pass = 0; pass < 2; ++pass) {
        t[pass] = __builtin_readcyclecounter();
        // KB * 1024 / 8 instruction pairs of 8 bytes (two 4-byte VALU adds on alternating registers)
        asm volatile(".rept %2\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %1, %1, %0\n\t.endr" : "+v"(a), "+v"(b) : "n"(KB * 128));
        asm volatile("s_nop 0" ::: "memory");
    }
    t[2] = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) {
        long long* o = out + ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
        o[0] = t[1] - t[0];
        o[1] = t[2] - t[1];
    }
    if (a + b == 0x7fffffff) out[0] = 0;
}
__global__ void k_other(double* x) { x[threadIdx.x] += 1.0; }

template <int KB>
static void run(int grid, long long* d, double* dx) {
    std::vector<long long> h((size_t)grid * 8);
    auto stats = [&](const char* what) {
        (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double p1 = 0, p2 = 0;
        for (int i = 0; i < grid * 4; ++i) { p1 += h[2 * i]; p2 += h[2 * i + 1]; }
        const double n = KB * 256.0;      // instructions per pass
        printf("  %3d KB grid %3d %-34s pass 1 %8.0f cycles (%.2f / instr)   pass 2 %8.0f (%.2f / instr)\n", KB, grid, what,
               p1 / (grid * 4), p1 / (grid * 4) / n, p2 / (grid * 4), p2 / (grid * 4) / n);
    };
    k_straight<KB><<<grid, 256>>>(d);
    (void)hipDeviceSynchronize();
    stats("first launch");
    for (int i = 0; i < 5; ++i) k_straight<KB><<<grid, 256>>>(d);
    (void)hipDeviceSynchronize();
    stats("6th launch back to back");
    for (int i = 0; i < 3; ++i) { k_other<<<1, 64>>>(dx); k_straight<KB><<<grid, 256>>>(d); }
    (void)hipDeviceSynchronize();
    stats("after a small other kernel");
}

int main() {
    long long* d; double* dx;
    (void)hipMalloc(&d, 256 * 8 * 8);
    (void)hipMalloc(&dx, 64 * 8);
    (void)hipMemset(dx, 0, 64 * 8);
    for (int grid : {1, 256}) {
        run<4>(grid, d, dx);
        run<16>(grid, d, dx);
        run<32>(grid, d, dx);
        run<48>(grid, d, dx);
        run<64>(grid, d, dx);
        run<96>(grid, d, dx);
        run<128>(grid, d, dx);
    }
    return 0;
}

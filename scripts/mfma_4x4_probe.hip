// Probe of the lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950: for every (lane of A, lane of B) pair set a
// single 1.0 in each operand and record which D lane becomes non-zero.
// build: hipcc --offload-arch=gfx950 -O2 scripts/mfma_4x4_probe.hip -o scripts/mfma_4x4_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int* out, int cbsz, int abid) {
    const int la = blockIdx.x / 64, lb = blockIdx.x % 64, lane = threadIdx.x;
    double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0, d;
    if (cbsz == 0) d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    else d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 2, 0, 0);
    if (d != 0.0) atomicOr(&out[blockIdx.x * 2 + (lane >> 5)], 1 << (lane & 31));
}
int main() {
    for (int mode = 0; mode < 2; ++mode) {
        int* d; hipMalloc(&d, 4096 * 2 * sizeof(int)); hipMemset(d, 0, 4096 * 2 * sizeof(int));
        probe<<<4096, 64>>>(d, mode, 0);
        std::vector<int> h(8192); hipMemcpy(h.data(), d, 8192 * sizeof(int), hipMemcpyDeviceToHost);
        printf("mode %d (cbsz=%d): (laneA, laneB) -> D lanes\n", mode, mode ? 2 : 0);
        for (int la = 0; la < 64; ++la) {
            printf("A%2d:", la);
            for (int lb = 0; lb < 64; ++lb) {
                unsigned long long m = ((unsigned long long)(unsigned)h[(la * 64 + lb) * 2 + 1] << 32) | (unsigned)h[(la * 64 + lb) * 2];
                if (!m) continue;
                printf(" B%d->", lb);
                for (int l = 0; l < 64; ++l) if (m >> l & 1) printf("%d,", l);
            }
            printf("\n");
        }
        hipFree(d);
    }
    return 0;
}

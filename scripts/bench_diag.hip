// Developer micro-benchmark: cycles of one 32x32 diagonal factorisation (gpsat::diag_factor), one wave per
// workgroup, with 1 or 8 workgroups per CU.  hipcc --offload-arch=gfx950 -O3 -I gpsat_amd/csrc -I include ...
#include "../gpsat_amd/csrc/gpsat_kernels.hip"
#include <cstdio>
#include <vector>
using namespace gpsat;
using namespace gpsat::w4;

__global__ void __launch_bounds__(64) k_diag(const float* in, float* out, unsigned long long* cyc, int reps) {
    const int lane = threadIdx.x;
    f32x16 W;
    for (int r = 0; r < 16; ++r) W[r] = in[(r >> 2) * 256 + lane * 4 + (r & 3)];
    f32x16 S1, S2; double ls; int bad;
    const int Ad = 0, piv = 1100;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < reps; ++i) {
        diag_factor(W, Ad, piv, lane, S1, S2, ls, bad);
        W[0] += S1[0] * 1e-30f;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) cyc[0] = (t1 - t0) / reps;
    for (int r = 0; r < 16; ++r) out[(size_t)blockIdx.x * 1024 + (r >> 2) * 256 + lane * 4 + (r & 3)] = S1[r];
    if (lane == 0 && blockIdx.x == 0) { out[0] += (float)ls * 0.f + bad; }
}

int main() {
    std::vector<float> h(1024);
    // SPD block in acc layout: A = I*40 + small symmetric
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
        int g = l & 31, hh = l >> 5; int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
        float v = (row == g) ? 40.f : 1.0f / (1 + abs(row - g));
        h[(r >> 2) * 256 + l * 4 + (r & 3)] = v;
    }
    float *din, *dout; unsigned long long* dc;
    hipMalloc(&din, 4096); hipMalloc(&dout, 4096 * 4096); hipMalloc(&dc, 8);
    hipMemcpy(din, h.data(), 4096, hipMemcpyHostToDevice);
    for (int grid : {1, 256, 2048}) {
        hipLaunchKernelGGL(k_diag, dim3(grid), dim3(64), 8192, 0, din, dout, dc, 50);
        hipDeviceSynchronize();
        unsigned long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        printf("grid %d: %llu cycles per diag_factor\n", grid, c);
    }
    return 0;
}

// Bare fp32 MFMA loops on random operands held in registers: v_mfma_f32_32x32x2_f32 against v_mfma_f32_16x16x4_f32 at equal FLOP
// per wave (64 x 32 output tile per wave = 32 accumulator registers either way), every CU busy, 1 or 2 waves per SIMD, >= 10 ms
// per launch (the chip lowers its clock under matrix load: MI355X_MICROARCH.md, DVFS give-back -- the shapes are ranked by wall
// time on random data, not by cycles).  build: hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_probe.hip -o tools/_exp/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void loop_kernel(const float* __restrict__ in, float* __restrict__ out, int iters, long long* clk) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float a[16], b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = in[(t * 16 + i) & 0xfffff]; b[i] = in[(t * 16 + i + 7777) & 0xfffff]; }
    const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (SHAPE == 32) {
        f32x16 acc[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
            // 16 k-pairs x 2 row blocks = 32 MFMAs = 131072 FLOP x 64 lanes ... (64 x 32 tile, 32 k)
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(s + m) & 15], b[s], acc[m], 0, 0, 0);
        }
        float v = 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) v += acc[m][e];
        out[t] = v;
    } else {
        f32x4 acc[8];
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[m][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
            // 8 k-quads x (4 x 2) tiles = 64 MFMAs of 2048 FLOP: the same 64 x 32 tile over 32 k
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(s + (m >> 1)) & 15], b[(s + 8 * (m & 1)) & 15], acc[m], 0, 0, 0);
        }
        float v = 0.f;
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) v += acc[m][e];
        out[t] = v;
    }
    const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200000;
    float *in, *out;
    long long* clk;
    const int maxblocks = 512;
    hipMalloc(&in, (1 << 20) * 4);
    hipMalloc(&out, maxblocks * 256 * 4);
    hipMalloc(&clk, maxblocks * 16);
    std::vector<float> h(1 << 20);
    srand(1);
    for (auto& x : h) x = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<long long> hc(maxblocks * 2);
    printf("{");
    bool first = true;
    for (int rep = 0; rep < 2; ++rep)
        for (int blocks : {256, 512})
            for (int shape : {32, 16}) {
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                if (shape == 32) hipLaunchKernelGGL(loop_kernel<32>, dim3(blocks), dim3(256), 0, 0, in, out, iters, clk);
                else hipLaunchKernelGGL(loop_kernel<16>, dim3(blocks), dim3(256), 0, 0, in, out, iters, clk);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                hipMemcpy(hc.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
                double cyc = 0, real = 0;
                for (int i = 0; i < blocks; ++i) { cyc += hc[2 * i]; real += hc[2 * i + 1]; }
                const double flop = (double)blocks * 4 * iters * 32 * 4096.0;     // 32 MFMAs of 4096 FLOP (or 64 of 2048) per wave and iteration
                if (rep == 1) {
                    printf("%s\n \"%s, %d waves per SIMD\": {\"ms\": %.2f, \"TFLOPs\": %.1f, \"clock_GHz\": %.3f, \"cycles_per_32x32x2_equivalent\": %.1f}", first ? "" : ",",
                           shape == 32 ? "32x32x2" : "16x16x4", blocks / 256, ms, flop / ms * 1e-9, cyc / real * 0.1, cyc / blocks / iters / 32.0);
                    first = false;
                }
            }
    printf("\n}\n");
    return 0;
}

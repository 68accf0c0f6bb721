// Dev probe: sustained rate of the two fp16 MFMA shapes from registers only (no memory traffic), every CU busy.
// waves_per_simd = 1 or 2 (256- or 512-thread workgroups, one per CU).  Reports nothing itself: time it from the host.
#include <hip/hip_runtime.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int SHAPE>
__global__ __launch_bounds__(512) void mfma_probe_kernel(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    h8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (lane + e)); b[e] = (_Float16)(0.002f * (lane - e)); }
    float sum = 0.f;
    if (SHAPE == 16) {                                            // 16 independent 16x16x32 accumulators (64 acc VGPRs)
        f4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) sum += acc[i][0] + acc[i][3];
    } else {                                                      // 4 independent 32x32x16 accumulators (64 acc VGPRs), same flops per iteration
        f16v acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) sum += acc[i][0] + acc[i][15];
    }
    if (sum == 123.456f) out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}
extern "C" int mfma_probe_launch(float* out, int shape, int iters, int threads, int wgs, void* stream) {
    if (shape == 16) hipLaunchKernelGGL(mfma_probe_kernel<16>, dim3(wgs), dim3(threads), 0, static_cast<hipStream_t>(stream), out, iters);
    else hipLaunchKernelGGL(mfma_probe_kernel<32>, dim3(wgs), dim3(threads), 0, static_cast<hipStream_t>(stream), out, iters);
    return (int)hipGetLastError();
}

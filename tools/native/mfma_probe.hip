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

// The k-loop of the 256x256 GEMM tile without its global traffic: per k32 slice a wave reads 8 A + 4 B fragments from LDS
// (ds_read_b128, the GEMM's row pitch and swizzle) and issues 32 MFMAs (wave tile 128 x 64); optionally every k-step also
// writes the 64 KB stage a workgroup receives by LDS-DMA in the real kernel (plain ds_write_b128 of register data here).
__global__ __launch_bounds__(512) void mfma_lds_probe_kernel(float* out, int iters, int with_writes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];          // 2 x (256 + 256) rows x 128 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
    const int wr = wave >> 2, wc = wave & 3;
    for (int i = tid; i < 2 * 512 * 128 / 16; i += 512) reinterpret_cast<float4*>(smem)[i] = make_float4(1e-3f * i, 0.f, 1.f, 2.f);
    __syncthreads();
    f4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    const float4 wv = make_float4(1.f, 2.f, 3.f, 4.f);
    for (int it = 0; it < iters; ++it) {
        const unsigned char* la = smem + (it & 1) * (512 * 128);
        const unsigned char* lb = la + 256 * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            h8 af[8], bf[4];
#pragma unroll
            for (int i = 0; i < 8; ++i) { const int r = wr * 128 + i * 16 + l15; af[i] = *reinterpret_cast<const h8*>(la + r * 128 + (((ks * 4 + lg) ^ (r & 7)) << 4)); }
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int r = wc * 64 + j * 16 + l15; bf[j] = *reinterpret_cast<const h8*>(lb + r * 128 + (((ks * 4 + lg) ^ (r & 7)) << 4)); }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
        if (with_writes) {                                       // the other buffer receives the next stage: 64 KB per workgroup
            unsigned char* dst = smem + ((it + 1) & 1) * (512 * 128);
#pragma unroll
            for (int q = 0; q < 8; ++q) *reinterpret_cast<float4*>(dst + (q * 512 + tid) * 16) = wv;
        }
        __syncthreads();
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) sum += acc[i][j][0];
    if (sum == 123.456f) out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}
extern "C" int mfma_lds_probe_launch(float* out, int iters, int with_writes, int wgs, void* stream) {
    static bool set = false;
    if (!set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_lds_probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 131072); set = true; }
    hipLaunchKernelGGL(mfma_lds_probe_kernel, dim3(wgs), dim3(512), 131072, static_cast<hipStream_t>(stream), out, iters, with_writes);
    return (int)hipGetLastError();
}
extern "C" int mfma_probe_launch(float* out, int shape, int iters, int threads, int wgs, void* stream) {
    if (shape == 16) hipLaunchKernelGGL(mfma_probe_kernel<16>, dim3(wgs), dim3(threads), 0, static_cast<hipStream_t>(stream), out, iters);
    else hipLaunchKernelGGL(mfma_probe_kernel<32>, dim3(wgs), dim3(threads), 0, static_cast<hipStream_t>(stream), out, iters);
    return (int)hipGetLastError();
}

// Dev tool: one wave samples the core clock (s_memtime cycles per 100 MHz s_memrealtime tick) while other streams run kernels.
#include <hip/hip_runtime.h>
__global__ void clock_probe_kernel(long long* out, int samples, int spin) {
    if (threadIdx.x != 0) return;
    for (int s = 0; s < samples; ++s) {
        const long long t0 = (long long)wall_clock64(), c0 = (long long)clock64();
        for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(2);
        const long long t1 = (long long)wall_clock64(), c1 = (long long)clock64();
        out[2 * s] = t1 - t0;
        out[2 * s + 1] = c1 - c0;
    }
}
extern "C" int clock_probe_launch(long long* out, int samples, int spin, void* stream) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), out, samples, spin);
    return (int)hipGetLastError();
}

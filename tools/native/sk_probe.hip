// Dev probe: what does a stream-K fix-up cost?  512 workgroups of 256 threads; odd local workgroups dump a 64 KB partial
// accumulator (64 floats per thread) and raise a flag, their partners wait for the flag, read the partial and reduce it.
// partner distance 8 = same XCD (workgroup id mod 8), 1 = the neighbouring XCD.  mode 0: no exchange (baseline: everybody
// just writes its 64 KB), 1: agent-scope release / acquire fences, 2: s_waitcnt vmcnt(0) + relaxed flag, acquire fence on the
// reader only.  Spins are bounded; err[0] counts give-ups, err[1] counts wrong sums.
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(256) void sk_probe_kernel(float* part, int* flag, float* out, int* err, int mode, int dist, int epoch,
                                                       int spin_work) {
    const int wg = blockIdx.x, tid = threadIdx.x;
    for (int i = 0; i < spin_work; ++i) __builtin_amdgcn_s_sleep(32);         // stands for the k-loop
    const int local = wg / dist, role = local & 1;                            // odd: contributor, even: owner
    float4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = make_float4((float)(wg + u), 1.f, 2.f, (float)epoch);
    float4* mine = reinterpret_cast<float4*>(part) + (size_t)wg * 16 * 256;
    if (mode == 0 || role == 1) {
#pragma unroll
        for (int u = 0; u < 16; ++u) mine[u * 256 + tid] = v[u];
        if (mode == 0) return;
        if (mode == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        else __builtin_amdgcn_s_waitcnt(0x0070);                              // vmcnt(0) (expcnt 7, lgkmcnt 0)
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flag + wg, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int partner = wg + dist;                                            // the contributor
    if (tid == 0) {
        int n = 0;
        while (__hip_atomic_load(flag + partner, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
            __builtin_amdgcn_s_sleep(2);
            if (++n > (1 << 22)) { atomicAdd(err, 1); break; }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const float4* theirs = reinterpret_cast<const float4*>(part) + (size_t)partner * 16 * 256;
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) { const float4 t = theirs[u * 256 + tid]; s += t.x + t.y + t.z + t.w; v[u].x += t.x; }
    float want = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) want += (float)(partner + u) + 3.f + (float)epoch;
    if (s != want) atomicAdd(err + 1, 1);
#pragma unroll
    for (int u = 0; u < 16; ++u) reinterpret_cast<float4*>(out)[((size_t)wg * 16 + u) * 256 + tid] = v[u];
}
extern "C" int sk_probe_launch(float* part, int* flag, float* out, int* err, int mode, int dist, int epoch, int spin_work, int wgs,
                               void* stream) {
    hipLaunchKernelGGL(sk_probe_kernel, dim3(wgs), dim3(256), 0, static_cast<hipStream_t>(stream), part, flag, out, err, mode, dist,
                       epoch, spin_work);
    return (int)hipGetLastError();
}

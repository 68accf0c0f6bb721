// Stand-alone check + timing of cc_wgrad_tn_f16 (dev tool):  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include
//   -I centerclip_amd/csrc tools/native/wgrad_probe.hip -o tools/native/wgrad_probe && tools/native/wgrad_probe
#include "../../centerclip_amd/csrc/wgrad.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }

static int run(int M, int N1, int N2, bool check) {
    unsigned seed = 12345u + M + N1 * 3 + N2 * 7;
    std::vector<_Float16> dy((size_t)M * N1), x((size_t)M * N2);
    for (auto& v : dy) v = (_Float16)(frand(seed) * 2.0f);
    for (auto& v : x) v = (_Float16)(frand(seed) * 2.0f);
    _Float16 *ddy, *dx; float *ddw, *dscale; void* ws;
    const size_t wsb = cc_wgrad_tn_workspace_bytes(M, N1, N2);
    hipMalloc(&ddy, dy.size() * 2); hipMalloc(&dx, x.size() * 2); hipMalloc(&ddw, (size_t)N1 * N2 * 4); hipMalloc(&dscale, 4);
    hipMalloc(&ws, wsb); hipMemset(ws, 0, wsb);
    hipMemcpy(ddy, dy.data(), dy.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dx, x.data(), x.size() * 2, hipMemcpyHostToDevice);
    const float scale = 4.0f;
    hipMemcpy(dscale, &scale, 4, hipMemcpyHostToDevice);
    int rc = cc_wgrad_tn_f16(ddy, dx, ddw, M, N1, N2, dscale, nullptr, 0, nullptr, ws, wsb, nullptr);
    hipError_t e = hipDeviceSynchronize();
    if (rc || e != hipSuccess) { printf("M=%d N1=%d N2=%d: rc %d hip %d\n", M, N1, N2, rc, (int)e); return 1; }
    int bad = 0;
    if (check) {
        std::vector<float> dw((size_t)N1 * N2), dw2((size_t)N1 * N2);
        hipMemcpy(dw.data(), ddw, dw.size() * 4, hipMemcpyDeviceToHost);
        cc_wgrad_tn_f16(ddy, dx, ddw, M, N1, N2, dscale, nullptr, 0, nullptr, ws, wsb, nullptr);          // second call: tickets were left at zero
        hipDeviceSynchronize();
        hipMemcpy(dw2.data(), ddw, dw2.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0;
        // every element of a sample of rows, transpose-detecting (N1 != N2 in the shapes below, random data)
        for (int n1 = 0; n1 < N1; n1 += (N1 > 256 ? 37 : 1))
            for (int n2 = 0; n2 < N2; ++n2) {
                double ref = 0;
                for (int m = 0; m < M; ++m) ref += (double)(float)dy[(size_t)m * N1 + n1] * (double)(float)x[(size_t)m * N2 + n2];
                ref /= scale;
                const double d = fabs(ref - dw[(size_t)n1 * N2 + n2]);
                worst = fmax(worst, d);
                if (d > 2e-3 * sqrt((double)M) * 0.1 + 1e-3) { if (bad < 5) printf("  bad [%d][%d] got %f want %f\n", n1, n2, dw[(size_t)n1 * N2 + n2], ref); ++bad; }
                if (dw[(size_t)n1 * N2 + n2] != dw2[(size_t)n1 * N2 + n2]) { if (bad < 5) printf("  run-to-run difference [%d][%d]\n", n1, n2); ++bad; }
            }
        printf("M=%d N1=%d N2=%d S=%d: worst |err| %.3g, %d bad\n", M, N1, N2, wgrad_slices(M, N1, N2), worst, bad);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) cc_wgrad_tn_f16(ddy, dx, ddw, M, N1, N2, dscale, nullptr, 0, nullptr, ws, wsb, nullptr);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) cc_wgrad_tn_f16(ddy, dx, ddw, M, N1, N2, dscale, nullptr, 0, nullptr, ws, wsb, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("M=%d N1=%d N2=%d: %.1f us  %.0f TFLOP/s\n", M, N1, N2, ms * 50, 2.0 * M * N1 * N2 / (ms / 20 * 1e-3) / 1e12);
    hipFree(ddy); hipFree(dx); hipFree(ddw); hipFree(dscale); hipFree(ws);
    return bad;
}

int main() {
    int bad = 0;
    bad += run(200, 128, 256, true);
    bad += run(1232, 512, 1536, true);
    bad += run(1232, 2048, 512, true);
    bad += run(9600, 768, 768, true);
    bad += run(9600, 2304, 768, false);
    bad += run(9600, 3072, 768, false);
    bad += run(9600, 768, 3072, false);
    printf(bad ? "FAILED\n" : "all ok\n");
    return bad != 0;
}

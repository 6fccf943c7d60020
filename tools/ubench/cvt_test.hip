// Semantics of v_cvt_pk_u8_f32 on gfx950 (the instruction frame_ops.hip / dibr.hip / jpeg.hip use for float -> uint8):
// round-half-to-even, saturate to [0, 255], insert into the selected byte.  Prints the table DESIGN.md section 3.3 cites.
//     hipcc --offload-arch=gfx950 -O2 tools/ubench/cvt_test.hip -o /tmp/cvt_test && /tmp/cvt_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

__global__ void k(const float* in, unsigned* out, int n) {
    int i = threadIdx.x;
    if (i < n) {
        unsigned r = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0xAABBCC00u);      // byte 0 of an existing word
        r = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 2, r);                          // and byte 2
        out[i] = r;
    }
}

int main() {
    const float v[] = {0.f, 0.49f, 0.5f, 0.51f, 1.5f, 2.5f, 3.5f, 126.5f, 127.5f, 254.5f, 255.49f, 255.5f, 256.f, 300.f, 1e9f,
                       -0.4f, -0.5f, -5.f, -1e9f, NAN, INFINITY};
    const int n = sizeof(v) / sizeof(v[0]);
    float* d_in; unsigned* d_out; unsigned h[64];
    hipMalloc(&d_in, sizeof(v)); hipMalloc(&d_out, n * 4);
    hipMemcpy(d_in, v, sizeof(v), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_in, d_out, n);
    hipMemcpy(h, d_out, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        unsigned b0 = h[i] & 255u, b2 = (h[i] >> 16) & 255u;
        float x = v[i];
        unsigned want = std::isnan(x) ? 0u : (x <= 0.f ? 0u : (x >= 255.f ? 255u : (unsigned)std::nearbyint(x)));   // RNE + saturate
        bool ok = b0 == want && b2 == want && (h[i] & 0xFF00FF00u) == 0xAA00CC00u;
        printf("%14g -> byte %3u (expected %3u), word %08x %s\n", x, b0, want, h[i], ok ? "" : "MISMATCH");
        bad += !ok;
    }
    printf(bad ? "FAILED\n" : "v_cvt_pk_u8_f32: round-half-even + saturate + byte insert, as assumed\n");
    return bad != 0;
}

// Known-size streaming kernels to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md: FETCH_SIZE
// reports half the bytes of a wide coalesced read; WRITE_SIZE is uncalibrated).  tools/pmc_traffic.py divides the bytes these
// kernels are KNOWN to move by what the counters report and applies the factors to the engine's kernels.
//     hipcc --offload-arch=gfx950 -O2 tools/ubench/copy_calib.hip -o /tmp/copy_calib
//     rocprofv3 --pmc FETCH_SIZE -d out/fetch -o fetch --output-format csv -- /tmp/copy_calib   (and WRITE_SIZE likewise)
#include <hip/hip_runtime.h>
#include <cstdio>

// 512 MiB in, 512 MiB out, 16 B per lane: larger than the 256 MiB Infinity Cache, so the counters see HBM traffic
__global__ void d2s_calib_copy16(const float4* __restrict__ in, float4* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = in[i];
}
// 8-byte stores (the width of a bf16 x4 epilogue store) and 16-byte loads
__global__ void d2s_calib_copy16to8(const float4* __restrict__ in, float2* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float4 v = in[i];
        out[i] = make_float2(v.x + v.z, v.y + v.w);
    }
}

int main() {
    const long n = 32L << 20;                                   // float4 elements: 512 MiB
    float4 *a, *b;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16);
    hipMemset(a, 1, n * 16);
    for (int it = 0; it < 3; ++it) {
        hipLaunchKernelGGL(d2s_calib_copy16, dim3(4096), dim3(256), 0, 0, a, b, n);
        hipLaunchKernelGGL(d2s_calib_copy16to8, dim3(4096), dim3(256), 0, 0, a, (float2*)b, n);
    }
    hipDeviceSynchronize();
    printf("d2s_calib_copy16: read %ld B, wrote %ld B per launch; d2s_calib_copy16to8: read %ld B, wrote %ld B\n", n * 16, n * 16, n * 16, n * 8);
    return 0;
}

// What ds_read_b64_tr_b16 (gfx950) returns: LDS holds the 16-bit values 0, 1, 2, ...; every lane gives the address `base + lane * stride` and prints its four results.
// Measured (gfx950, ROCm 7.2): the instruction works on groups of 16 lanes.  With D[l][e] = the e-th 16-bit value at lane l's address (l = 0..15, e = 0..3), lane i
// receives O[i][j] = D[(i >> 2) + 4 j][i & 3].  So 16 lanes that load a 4 (k) x 16 (m) block of a k-major tile -- lane l: row k0 + (l >> 2), m-chunk (l & 3) of 8 bytes --
// get it back transposed: lane i holds T[k0 .. k0+3][m = i], four consecutive k for its own m.  Two such reads (k0 = 8 q, 8 q + 4) make the 8-element A / B fragment of
// v_mfma_f32_16x16x32_bf16 from an operand stored k-major: the way to drop the explicit transposes of kf_linear_backward (not built yet).
//   hipcc --offload-arch=gfx950 -O2 -o scratch/dbg/ds_read_tr_probe scratch/dbg/ds_read_tr_probe.hip && scratch/dbg/ds_read_tr_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(int stride_bytes, short* out) {
    __shared__ short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)((__attribute__((address_space(3))) char*)lds + threadIdx.x * stride_bytes));
    out[threadIdx.x * 4 + 0] = r.x, out[threadIdx.x * 4 + 1] = r.y, out[threadIdx.x * 4 + 2] = r.z, out[threadIdx.x * 4 + 3] = r.w;
}
int main() {
    short* d;
    hipMalloc(&d, 64 * 4 * 2);
    for (int stride : {8, 32, 64}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, stride, d);
        short h[256];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("stride %d bytes (lane i reads at 16-bit index %d * i):\n", stride, stride / 2);
        for (int l = 0; l < 64; l++) printf("  lane %2d: %4d %4d %4d %4d%s", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3], (l & 3) == 3 ? "\n" : "");
    }
    return 0;
}

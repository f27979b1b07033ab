// tools/dpp_probe.hip -- prints which lane a DPP control reads from, and which lanes a bank_mask
// enables, on the machine it runs on (used once to pin the direction of row_shl/row_shr/row_ror).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(int *out)
{
    int lane = threadIdx.x;
    out[0 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x104, 0xF, 0xF, false);  // row_shl:4
    out[1 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x114, 0xF, 0xF, false);  // row_shr:4
    out[2 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x124, 0xF, 0xF, false);  // row_ror:4
    out[3 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x12C, 0xF, 0xF, false);  // row_ror:12
    out[4 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x128, 0xF, 0x5, false);  // row_ror:8 bank_mask 0x5
    out[5 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x128, 0xF, 0x3, false);  // row_ror:8 bank_mask 0x3
    out[6 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x128, 0x5, 0xF, false);  // row_ror:8 row_mask 0x5
}
int main()
{
    int *d, h[7 * 64];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[] = {"row_shl:4", "row_shr:4", "row_ror:4", "row_ror:12", "ror8 bank 0x5", "ror8 bank 0x3", "ror8 rowmask 0x5"};
    for (int t = 0; t < 7; ++t) {
        printf("%-18s:", names[t]);
        for (int l = 0; l < 32; ++l) printf(" %d", h[t * 64 + l]);
        printf("\n");
    }
    return 0;
}

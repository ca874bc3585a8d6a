// does gfx950 execute DPP wave_shr:1 / wave_shl:1 (whole-wavefront shifts)?  compared against __shfl_up / __shfl_down
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const double *in, double *up, double *dn, double *rup, double *rdn) {
    const int t = threadIdx.x;
    double v = in[t];
    int lo = __double2loint(v), hi = __double2hiint(v);
    int lo1 = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
    int hi1 = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    up[t] = __hiloint2double(hi1, lo1);
    int lo2 = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
    int hi2 = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    dn[t] = __hiloint2double(hi2, lo2);
    rup[t] = __shfl_up(v, 1, 64);
    rdn[t] = __shfl_down(v, 1, 64);
}
int main() {
    double h[128], *d[5];
    for (int i = 0; i < 128; i++) h[i] = 1.0 + i * 0.5;
    for (int q = 0; q < 5; q++) hipMalloc(&d[q], sizeof(h));
    hipMemcpy(d[0], h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, d[0], d[1], d[2], d[3], d[4]);
    double r[4][128];
    for (int q = 0; q < 4; q++) hipMemcpy(r[q], d[q + 1], sizeof(h), hipMemcpyDeviceToHost);
    int bad_up = 0, bad_dn = 0;
    for (int i = 0; i < 128; i++) {
        if ((i & 63) != 0 && r[0][i] != r[2][i]) bad_up++;
        if ((i & 63) != 63 && r[1][i] != r[3][i]) bad_dn++;
    }
    printf("DPP wave_shr mismatches %d, wave_shl mismatches %d; lane0 up %g (own %g), lane63 dn %g (own %g)\n", bad_up, bad_dn, r[0][0], h[0], r[1][63], h[63]);
    printf("sample up: %g %g %g %g | %g %g\n", r[0][1], r[0][15], r[0][16], r[0][17], r[0][32], r[0][33]);
    return 0;
}

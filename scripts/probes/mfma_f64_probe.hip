// Probe: v_mfma_f64_16x16x4_f64 operand/result lane maps on gfx950 (exact integer data).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void probe(const double* A, const double* B, double* D) {   // A 16x4 row-major, B 4x16 row-major
    int l = threadIdx.x;
    double a = A[(l & 15) * 4 + (l >> 4)];       // A[i=l&15][k=l>>4]
    double b = B[(l >> 4) * 16 + (l & 15)];      // B[k=l>>4][j=l&15]
    double4_t c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}
int main() {
    double hA[64], hB[64], hD[256], ref[256];
    unsigned s = 12345;
    for (int i = 0; i < 64; ++i) { s = s * 1664525u + 1013904223u; hA[i] = (double)((s >> 16) % 97) - 40; s = s * 1664525u + 1013904223u; hB[i] = (double)((s >> 16) % 89) - 30; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double t = 0; for (int k = 0; k < 4; ++k) t += hA[i * 4 + k] * hB[k * 16 + j]; ref[i * 16 + j] = t; }
    double *dA, *dB, *dD; (void)hipMalloc(&dA, 512); (void)hipMalloc(&dB, 512); (void)hipMalloc(&dD, 2048);
    (void)hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD); (void)hipMemcpy(hD, dD, 2048, hipMemcpyDeviceToHost);
    int okA = 1, okB = 1;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        int rowA = (l >> 4) + 4 * r, rowB = 4 * (l >> 4) + r, col = l & 15;
        if (hD[l * 4 + r] != ref[rowA * 16 + col]) okA = 0;
        if (hD[l * 4 + r] != ref[rowB * 16 + col]) okB = 0;
    }
    printf("map row=(l>>4)+4r, col=l&15 : %s\nmap row=4(l>>4)+r, col=l&15 : %s\n", okA ? "MATCH" : "no", okB ? "MATCH" : "no");
    // brute force: where does each (l,r) value appear in ref?
    for (int l = 0; l < 64; l += 13) for (int r = 0; r < 4; ++r) {
        printf("lane %2d reg %d = %8.0f ->", l, r, hD[l * 4 + r]);
        for (int i = 0; i < 256; ++i) if (ref[i] == hD[l * 4 + r]) printf(" (row %d,col %d)", i / 16, i % 16);
        printf("\n");
    }
    return 0;
}

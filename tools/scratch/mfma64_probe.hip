// probe the operand / result layout of v_mfma_f64_16x16x4_f64 (one wave)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(const double *A, const double *B, double *D) {
    // A [16][4] row-major, B [4][16] row-major; hypothesis: lane l holds A[l & 15][l >> 4], B[l >> 4][l & 15]
    const int l = threadIdx.x;
    d4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = acc[r];
}
int main() {
    double hA[64], hB[64], hD[256], ref[256];
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) hA[i * 4 + k] = 1 + i + 0.01 * k;
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) hB[k * 16 + j] = (k == 0 ? 1.0 : 0.0) * (100 * (j + 1)) + (k == 1 ? 1e-3 * j : 0);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
    double *dA, *dB, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    // find for lane l, reg r the (i, j) whose ref matches
    int okA = 1;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        int fi = -1, fj = -1;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (fabs(ref[i * 16 + j] - hD[l * 4 + r]) < 1e-9) { fi = i; fj = j; }
        if (l < 20 || l % 16 == 0) printf("lane %2d reg %d -> D[%d][%d]\n", l, r, fi, fj);
        if (fi != 4 * (l >> 4) + r || fj != (l & 15)) okA = 0;
    }
    printf("hypothesis i = 4 * (l >> 4) + r, j = l & 15: %s\n", okA ? "CONFIRMED" : "NO");
    return 0;
}

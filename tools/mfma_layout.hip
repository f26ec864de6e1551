// mfma_layout.hip -- prints the lane/register layout of v_mfma_f64_16x16x4_f64 operands (A: 16x4, B: 4x16, D: 16x16).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(double* out, int mode) {
  const int l = threadIdx.x;
  // hypothesis: A[i][k] with i = l % 16, k = l / 16;  B[k][j] with j = l % 16, k = l / 16
  const int i = l % 16, k = l / 16;
  double a, b;
  if (mode == 0) { a = (k == 0) ? (double)(i + 1) : 0.0; b = (k == 0) ? 100.0 * (i + 1) : 0.0; }     // D[i][j] = (i+1) * 100 (j+1)
  else { a = (double)(i + 1) + 0.01 * (k + 1); b = (i == 3) ? 1.0 : 0.0; }                               // D[i][3] = sum_k A[i][k]
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
  double* d; hipMalloc(&d, 64 * 4 * 8);
  double h[256];
  for (int mode = 0; mode < 2; ++mode) {
    probe<<<1, 64>>>(d, mode); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; l += 1) {
      if (l % 16 < 2 || l % 16 == 3 || l % 16 == 15) printf("lane %2d: %10.2f %10.2f %10.2f %10.2f\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
  }
  return 0;
}

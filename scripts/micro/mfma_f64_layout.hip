// Which lane holds what in v_mfma_f64_16x16x4_f64 (gfx950)?  Inputs in the layout A: lane l holds A[l % 16][l / 16],
// B: lane l holds B[l / 16][l % 16]; every output register is matched by VALUE against the host product and the
// (lane, register) -> (row, column) map is printed.
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/mfma_f64_layout.hip -o /tmp/mfma_layout && /tmp/mfma_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* raw) {
  const int l = threadIdx.x;
  const double a = A[(l % 16) * 4 + l / 16];
  const double b = B[(l / 16) * 16 + l % 16];
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) raw[l * 4 + r] = acc[r];
}
int main() {
  double hA[64], hB[64], raw[256], ref[256];
  srand(1);
  for (int i = 0; i < 64; ++i) { hA[i] = rand() / (double)RAND_MAX - 0.5; hB[i] = rand() / (double)RAND_MAX - 0.5; }
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double s = 0;
      for (int q = 0; q < 4; ++q) s += hA[i * 4 + q] * hB[q * 16 + j];
      ref[i * 16 + j] = s;
    }
  double *dA, *dB, *dD;
  (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dD, sizeof raw);
  (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  (void)hipMemcpy(raw, dD, sizeof raw, hipMemcpyDeviceToHost);
  int unmatched = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      int hit = -1;
      for (int e = 0; e < 256; ++e) if (fabs(raw[l * 4 + r] - ref[e]) < 1e-14) hit = e;
      if (hit < 0) ++unmatched;
      if (l < 20 || l % 16 == 0) printf("lane %2d reg %d -> D[%2d][%2d]\n", l, r, hit / 16, hit % 16);
    }
  printf("unmatched registers: %d (0 = the INPUT layout assumption holds)\n", unmatched);
  return 0;
}

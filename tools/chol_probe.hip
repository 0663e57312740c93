// Phase timing of k_cholesky (shader-clock timestamps from thread 0): build with
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/chol_probe.hip -o tools/chol_probe
// Not part of the library.
#define M355_CHOL_PROBE 1
#include "../distributions_amd/csrc/m355_mvn.hip"
#include <vector>
namespace m355 { int fail(int code, const char*, ...) { return code; } }
int main(int argc, char** argv) {
  const int d = argc > 1 ? atoi(argv[1]) : 64;
  std::vector<double> A(d * d), C(d * d);
  srand(6);
  for (auto& v : A) v = rand() / (double)RAND_MAX - 0.5;
  for (int i = 0; i < d; ++i) for (int j = 0; j < d; ++j) {
    double s = i == j ? 1.0 : 0.0;
    for (int k = 0; k < d; ++k) s += A[i * d + k] * A[j * d + k] / d;
    C[i * d + j] = s;
  }
  double *dc, *dw;
  size_t wsb = m355_mvn_workspace_bytes(d, 1);
  hipMalloc(&dc, d * d * 8); hipMalloc(&dw, wsb);
  hipMemcpy(dc, C.data(), d * d * 8, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; ++rep) {
    run_cholesky(dc, 1, d, dw, dw + m355::chol_stride(d), 0);
    hipDeviceSynchronize();
  }
  long long t[64];
  hipMemcpyFromSymbol(t, HIP_SYMBOL(m355::g_chol_ticks), sizeof(t));
  const int NB = (d + 15) / 16;
  printf("d=%d load %lld\n", d, t[1] - t[0]);
  for (int J = 0; J < NB; ++J)
    printf(" panel %d: A %lld  B %lld  C %lld  (total %lld)\n", J, t[3 + 4 * J] - t[2 + 4 * J], t[4 + 4 * J] - t[3 + 4 * J],
           t[5 + 4 * J] - t[4 + 4 * J], t[5 + 4 * J] - t[2 + 4 * J]);
  printf(" store+logdet %lld  inverse %lld  frags %lld  total %lld cycles\n", t[41] - t[40], t[42] - t[41], t[43] - t[42], t[43] - t[0]);
  if (d == 64) {
    const long long n = 100000;
    double *dx, *dmu, *dout;
    hipMalloc(&dx, n * d * 8); hipMalloc(&dmu, d * 8); hipMalloc(&dout, n * 8);
    hipMemset(dx, 0, n * d * 8); hipMemset(dmu, 0, d * 8);
    m355::MfmaArgs m{dx, dmu, dw + m355::chol_stride(d), dw, dout, n, d, 1, 0.0, 1.0};
    for (int rep = 0; rep < 3; ++rep) { run_mfma<false>(m, 0); hipDeviceSynchronize(); }
    hipMemcpyFromSymbol(t, HIP_SYMBOL(m355::g_chol_ticks), sizeof(t));
    printf("solve: prologue %lld", t[51] - t[50]);
    for (int k = 52; k + 2 < 64 && t[k + 2] > t[k]; k += 3)
      printf(" | tile: stage %lld chain %lld", t[k + 1] - t[k], t[k + 2] - t[k + 1]);
    printf("\n");
  }
  return 0;
}

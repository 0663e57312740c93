// Phase timing of k_mvn_shared's factorisation (shader-clock timestamps from thread 0 of workgroup 0): build with
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off [-DM355_SHARED_VARIANT=v] tools/shared_probe.hip -o tools/shared_probe
// Not part of the library.
#define M355_CHOL_PROBE 1
#include "../distributions_amd/csrc/m355_mvn.hip"
#include <vector>
namespace m355 { int fail(int code, const char*, ...) { return code; } }
int main(int argc, char** argv) {
  const int d = argc > 1 ? atoi(argv[1]) : 64;
  const long long n = argc > 2 ? atoll(argv[2]) : 100000;
  std::vector<double> A(d * d), C(d * d);
  srand(6);
  for (auto& v : A) v = rand() / (double)RAND_MAX - 0.5;
  for (int i = 0; i < d; ++i) for (int j = 0; j < d; ++j) {
    double s = i == j ? 1.0 : 0.0;
    for (int k = 0; k < d; ++k) s += A[i * d + k] * A[j * d + k] / d;
    C[i * d + j] = s;
  }
  double *dc, *dx, *dmu, *dout;
  hipMalloc(&dc, d * d * 8); hipMalloc(&dx, n * d * 8); hipMalloc(&dmu, d * 8); hipMalloc(&dout, n * 8);
  hipMemcpy(dc, C.data(), d * d * 8, hipMemcpyHostToDevice);
  hipMemset(dx, 0, n * d * 8); hipMemset(dmu, 0, d * 8);
  m355::MfmaArgs m{dx, dmu, nullptr, nullptr, dout, n, d, 1, 0.0, 1.0};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0, 0);
    run_shared<false>(dc, m, 0);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  long long t[64];
  hipMemcpyFromSymbol(t, HIP_SYMBOL(m355::g_chol_ticks), sizeof(t));
  const int NB = (d + 15) / 16;
  printf("variant %d d=%d n=%lld kernel %.1f us | load %lld", M355_SHARED_VARIANT, d, n, best * 1e3, t[1] - t[0]);
  for (int J = 0; J < NB; ++J) printf(" | panel %d: transpose %lld sweep %lld mfma %lld", J, t[2 + 3 * J] - (J ? t[4 + 3 * (J - 1)] : t[1]), t[3 + 3 * J] - t[2 + 3 * J], t[4 + 3 * J] - t[3 + 3 * J]);
  printf(" | logdet+inverse %lld publish %lld barrier %lld | factor total %lld cycles (100 MHz ticks if clock64 is the fixed counter)\n", t[21] - t[20], t[22] - t[21], t[23] - t[22], t[22] - t[0]);
  printf("wavefront 1 tiles (cycles from the barrier):");
  for (int i = 0; i < 9; ++i) printf(" | #%d start %lld mfma-done %lld out %lld staged-next %lld", i, t[24 + 4 * i] - t[23], t[25 + 4 * i] - t[23], t[26 + 4 * i] - t[23], t[27 + 4 * i] - t[23]);
  printf("\n");
  return 0;
}

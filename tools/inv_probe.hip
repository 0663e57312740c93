// Timing of k_rvs_inv<poisson> (1e9 draws, int64) against the grid size: tools/inv_probe <mu> <blocks>.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/inv_probe.hip -o tools/inv_probe
#include "../distributions_amd/csrc/m355_rvs.hip"
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
  using DD = m355::Dist<M355_POISSON>;
  const long long n = 1000000000ll;
  const double mu = argc > 1 ? atof(argv[1]) : 100.0;
  const int blocks = argc > 2 ? atoi(argv[2]) : m355::kInvMaxBlocks;
  const double pp[4] = {mu, 0, 0, 0};
  const DD::K k = DD::derive(pp);
  double lo = 0.0;
  int len = 0;
  if (!DD::dtab_range(k, lo, len)) return 2;
  void* out;
  CHECK(hipMalloc(&out, n * 8));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 4; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((m355::k_rvs_inv<M355_POISSON>), dim3(blocks), dim3(m355::kInvBlock), 0, 0, out, (int)M355_I64, n, k, lo, len, 20260313ull, 7u, 0ull);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep == 3) printf("mu %g window %d blocks %d: %.3f ms  (%.3g draws/s, %.2f TB/s)\n", mu, len, blocks, ms, n / (ms * 1e-3), 8.0 * n / (ms * 1e-3) / 1e12);
  }
  return 0;
}

// Timing variants of k_rvs_normal_zig (M355_ZIG_VARIANT, see m355_rvs.hip): 0 the library's kernel, 1 fast path only
// (non-temporal 16-byte stores, no queue), 2 the same with plain stores, 3 full 16-byte stores plus the queue.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DM355_ZIG_VARIANT=N tools/zig_probe.hip -o tools/zig_probe_vN
// other layer counts / block sizes: python tools/gen_ziggurat.py 2048 > /tmp/zigtab2048.h and
//   -DM355_ZIGTAB_HEADER='"/tmp/zigtab2048.h"' -DM355_ZIG_BLOCK=1024
#include "../distributions_amd/csrc/m355_rvs.hip"
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
  const long long n = 1000000000ll, npairs = n / 2;
  const int blocks = argc > 1 ? atoi(argv[1]) : m355::kMaxBlocks / 2;
  double* out;
  CHECK(hipMalloc(&out, n * 8));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 4; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(m355::k_rvs_normal_zig<false>, dim3(blocks), dim3(m355::kZigBlock), 0, 0, out, npairs, m355::DevOperand{nullptr, 0.0, 1, 1, M355_F64},
                       m355::DevOperand{nullptr, 1.0, 1, 1, M355_F64}, 20260313ull, 7u, 0ull);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) printf("variant %d blocks %d: %.3f ms  (%.3g draws/s, %.2f TB/s)\n", M355_ZIG_VARIANT, blocks, ms, n / (ms * 1e-3), 8.0 * n / (ms * 1e-3) / 1e12);
  }
  return 0;
}

// Diagnostic: what is the LDS-DMA bf16-split GEMM's (pf_gemm3f.h) time made of?  Built once per -DG3F_ABLATE mask by
// tools/gemm_dma_ablate.sh; results of an ablated build are wrong by construction -- only times are read.
//   gemm_dma_ablate MODE M N K BN [ZEROS] [PAIR]      MODE 0 FWD (x W^T + b, ELU), 1 BWD_X;  BN = kernel variant: 0 (128 columns), 1 (64), -1 auto;
//   PAIR 1: two problems of this shape in one launch (as the PPO update issues the actor's and the critic's layer)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../pointfoot_amd/csrc/pf_gemm3f.h"
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
  const int mode = atoi(argv[1]), M = atoi(argv[2]), N = atoi(argv[3]), K = atoi(argv[4]), bn = atoi(argv[5]);
  float *A, *Cm, *bias, *H; unsigned short* Bp;
  const int ldbp = (K + 7) / 8 * 8;
  OK(hipMalloc(&A, (size_t)M * K * 4)); OK(hipMalloc(&Cm, (size_t)M * N * 4));
  OK(hipMalloc(&bias, (size_t)N * 4)); OK(hipMalloc(&H, (size_t)M * N * 4)); OK(hipMalloc(&Bp, (size_t)3 * N * ldbp * 2));
  std::vector<float> h((size_t)M * (K > N ? K : N));
  const bool zeros = argc > 6 && atoi(argv[6]) != 0;
  for (auto& v : h) v = zeros ? 0.0f : (float)rand() / RAND_MAX - 0.5f;
  OK(hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice));
  OK(hipMemset(bias, 0, N * 4)); OK(hipMemcpy(H, h.data(), (size_t)M * N * 4, hipMemcpyHostToDevice));
  OK(hipMemset(Bp, zeros ? 0 : 0x3c, (size_t)3 * N * ldbp * 2));
  GemmArgs g{};
  g.A = A; g.C = Cm; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldc = N; g.bias = bias; g.H = H; g.ldh = N;
  g.Bp = Bp; g.ldbp = ldbp; g.bplane = (long long)N * ldbp;
  const bool pair = argc > 7 && atoi(argv[7]) != 0;
#if G3F_EMIT_PLANES
  { unsigned short* Cp; OK(hipMalloc(&Cp, (size_t)3 * M * N * 2)); g.Cp = Cp; g.cplane = (long long)M * N; }
#endif
  GemmArgs g2 = g;
  if (pair) { float* C2; OK(hipMalloc(&C2, (size_t)M * N * 4)); g2.C = C2; }
#if G3F_EMIT_PLANES
  if (pair) { unsigned short* Cp2; OK(hipMalloc(&Cp2, (size_t)3 * M * N * 2)); g2.Cp = Cp2; }
#endif
  hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  auto launch = [&]() {
    if (pair) { if (mode == 0) pf_gemm3f_launch2<GEMM_EPI_BIAS_ELU>(g, g2, 0, bn); else pf_gemm3f_launch2<GEMM_EPI_MUL_DELU>(g, g2, 0, bn); }
    else if (mode == 0) pf_gemm3f_launch<GEMM_EPI_BIAS_ELU>(g, 0, bn);
    else pf_gemm3f_launch<GEMM_EPI_MUL_DELU>(g, 0, bn);
  };
  for (int i = 0; i < 5; ++i) launch();
  OK(hipDeviceSynchronize());
  const int iters = 50;
  OK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) launch();
  OK(hipEventRecord(e1, 0)); OK(hipEventSynchronize(e1));
  float ms; OK(hipEventElapsedTime(&ms, e0, e1));
  printf("ablate %2d%s mode %d M=%d N=%d K=%d bn %d%s%s: %.1f us\n", G3F_ABLATE, G3F_EMIT_PLANES ? " +planes" : "", mode, M, N, K, bn, zeros ? " zeros" : "", pair ? " pair" : "", ms * 1000.0f / iters);
  return 0;
}

// Diagnostic: what is the bf16-split GEMM's main loop made of?  Built once per -DG3_ABLATE mask by tools/gemm_ablate.sh; every
// build runs the same shape and prints its time.  Results of an ablated build are wrong by construction -- only times are read.
//   gemm_ablate MODE M N K TILE BP [ZEROS]      MODE 0 FWD (x W^T + b, ELU), 1 BWD_X, 2 BWD_W (M = out, N = in, K = batch rows);  TILE 0 128x128, 1 128x64, 3 64x64;  BP 1: B as bf16 planes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../pointfoot_amd/csrc/pf_gemm3.h"
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
  const int mode = atoi(argv[1]), M = atoi(argv[2]), N = atoi(argv[3]), K = atoi(argv[4]), tile = atoi(argv[5]), bp = atoi(argv[6]);
  float *A, *B, *Cm, *bias, *H; unsigned short* Bp;
  const int ldbp = (K + 7) / 8 * 8;
  OK(hipMalloc(&A, (size_t)M * K * 4)); OK(hipMalloc(&B, (size_t)N * K * 4)); OK(hipMalloc(&Cm, (size_t)M * N * 4));
  OK(hipMalloc(&bias, (size_t)N * 4)); OK(hipMalloc(&H, (size_t)M * N * 4)); OK(hipMalloc(&Bp, (size_t)3 * N * ldbp * 2));
  std::vector<float> h((size_t)(M > N ? M : N) * K);
  const bool zeros = argc > 7 && atoi(argv[7]) != 0;      // zero operands: same instruction stream, less switching power
  for (auto& v : h) v = zeros ? 0.0f : (float)rand() / RAND_MAX - 0.5f;
  OK(hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice));
  OK(hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice));
  OK(hipMemset(bias, 0, N * 4)); OK(hipMemcpy(H, h.data(), (size_t)std::min((size_t)M * N, h.size()) * 4, hipMemcpyHostToDevice));
  OK(hipMemset(Bp, zeros ? 0 : 0x3c, (size_t)3 * N * ldbp * 2));
  GemmArgs g{};
  g.A = A; g.C = Cm; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldc = N; g.bias = bias; g.H = H; g.ldh = N;
  if (mode == 0) { g.B = B; g.ldb = K; } else { g.B = B; g.ldb = N; }      // FWD: B [N][K]; BWD_X: B [K][N]
  float* slabs = nullptr; int S = 0;
  if (mode == 2) {      // weight gradient: C[M,N] = A[K][M]^T B[K][N], split-K slabs (tile 0: 128x128; 64 splits = 512 workgroups, what the update's pair launch of layer 1 puts on the chip)
    S = 64; const int per = ((K + S - 1) / S + 31) / 32 * 32; S = (K + per - 1) / per;
    OK(hipMalloc(&slabs, (size_t)S * ((size_t)M * N + M) * 4));
    g.lda = M; g.ldb = N; g.C = slabs; g.k_per_split = per; g.Bp = nullptr;
  }
  if (bp) { g.Bp = Bp; g.ldbp = ldbp; g.bplane = (long long)N * ldbp; }
  hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  auto launch = [&]() {
    if (mode == 2) pf_gemm3_launch<false, false, GEMM_EPI_NONE, true>(g, S, 0, tile);
    else if (mode == 0) pf_gemm3_launch<true, true, GEMM_EPI_BIAS_ELU>(g, 0, 0, tile);
    else if (bp) pf_gemm3_launch<true, true, GEMM_EPI_MUL_DELU>(g, 0, 0, tile);
    else pf_gemm3_launch<true, false, GEMM_EPI_MUL_DELU>(g, 0, 0, tile);
  };
  for (int i = 0; i < 5; ++i) launch();
  OK(hipDeviceSynchronize());
  const int iters = 50;
  OK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) launch();
  OK(hipEventRecord(e1, 0)); OK(hipEventSynchronize(e1));
  float ms; OK(hipEventElapsedTime(&ms, e0, e1));
  printf("ablate %2d mode %d M=%d N=%d K=%d tile %d bp %d%s: %.1f us\n", G3_ABLATE, mode, M, N, K, tile, bp, zeros ? " zeros" : "", ms * 1000.0f / iters);
  return 0;
}

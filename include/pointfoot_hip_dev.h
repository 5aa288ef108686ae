/*
 * pointfoot_hip_dev.h -- development / test hooks exported by libpointfoot_hip.so IN ADDITION to the drop-in boundary of
 * pointfoot_hip.h.  Nothing here replaces a reference interface; a maintainer binding the library to the reference
 * needs none of it.  Users: tests/test_gpu_ppo.py (GEMM family test), tools/gemm_bench.py, tools/step_stamps.py.
 *
 * Environment switches read by the library (diagnostics only; the defaults are the product path):
 *   PF_PPO_GEMM=1|2        dense layers on the fp32 MFMA kernels (1) or on the bf16-split kernels (2), overriding PpoConfig.gemm_mode
 *   PF_PPO_PAIRED=0        actor / critic layers as separate launches on two streams instead of one paired launch
 *   PF_PPO_FUSED_HEADS=0   unfused head / loss / head-backward kernels
 *   PF_PPO_DUAL_STREAM=0   (with PAIRED=0) single stream
 * They exist so that tests/test_gpu_ppo.py can hold every code path to the same oracle.
 */
#ifndef POINTFOOT_HIP_DEV_H
#define POINTFOOT_HIP_DEV_H
#include "pointfoot_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Run one GEMM of the PPO kernel family (modes 0-2: fp32 MFMA, pf_gemm.h; 10-12: the same arrangements on the bf16-split
 * kernels, pf_gemm3.h; 20 / 21: forward / data gradient with B pre-split into bf16 planes in `ws`).  mode 0: C[M,N] = A[M,K] B[N,K]^T + bias, ELU;  1: C[M,N] = A[M,K] B[K,N] * ELU'(H);
 * 2: C[M,N] = A[K,M]^T B[K,N] as split-K slabs into `ws`, reduced into C, column sums of A in C[M*N .. M*N+M).
 * tile: -1 auto, 0: 128x128, 1: 128x64, 2: 64x64. */
int pf_gemm_test(int mode, int M, int N, int K, const float* A, const float* B, float* C, const float* bias, const float* H,
                 float* ws, int splits, int tile, void* stream);

/* Diagnostic build (-DPF_STAMPS, tools/step_stamps.py) only: per-env phase cycle counters, 16 unsigned per env. */
int pf_debug_set_stamps(PfEnv* env, unsigned* d_stamps);

/* Calibration of the rocprofv3 FETCH_SIZE counter on the step kernel's own read pattern (dword per lane, 32 lanes per
 * record row): reads rows x row_floats floats once; tools/calib_fetch.py compares the counter with the known byte count. */
int pf_debug_calib_read(const float* d_src, int rows, int row_floats, float* d_sink, void* stream);

/* The same for the height scan's narrow gathers (three int16 samples per scan point, 121 points per robot, one point per lane):
 * `robots` 32-lane groups sample the 12 x 12 cells at base[2 r], base[2 r + 1] of an int16 field with `cols` columns;
 * tools/calib_gather.py counts the 64-byte sectors / 128-byte lines they touch on the host and compares. */
int pf_debug_calib_gather(const int16_t* d_field, int cols, const int* d_base, int robots, float* d_sink, void* stream);

#ifdef __cplusplus
}
#endif
#endif

// pf_gemm.h — fp32 MFMA GEMM for the actor-critic MLPs (gfx950: v_mfma_f32_32x32x2_f32, exact f32,
// 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak).
//
//   C[M,N] = op(A) * op(B)  (+ epilogue), three operand arrangements of an nn.Linear layer y = x W^T + b:
//     FWD  : C = X[M,K]  * W[N,K]^T      A k-contiguous, B k-contiguous   -> bias + ELU epilogue
//     BWD_X: C = G[M,K]  * W[K,N]        A k-contiguous, B n-contiguous   -> * ELU'(H) epilogue
//     BWD_W: C = G[Kr,M]^T * X[Kr,N]     A m-contiguous, B n-contiguous, reduction over rows Kr split across
//                                         blockIdx.z into slabs (deterministic two-stage reduce, no atomics)
// Tiling: 256 threads = 4 wavefronts (2x2); block tile BM x BN with BM,BN in {64,128}, BK = 32; each wave owns
// (BM/2)x(BN/2) = TMxTN tiles of 32x32.  Both operands are staged in LDS row-major over the tile's m (or n) index with the
// 32 k values of a slice permuted so that the values one lane feeds to four consecutive MFMA steps are adjacent:
// position(k) = ((k >> 1) & 1) * 16 + 2 * (k >> 2) + (k & 1)  (lane half h = lane >> 5 supplies, at MFMA step s,
// k = 4 (s >> 1) + 2 h + (s & 1); any one-to-one assignment works as long as A and B share it).  A fragment read is then one
// ds_read_b128 per four MFMAs per 32-row operand tile (row stride 36 floats: the 16 lanes of a b128 phase hit 64 distinct
// banks), and a thread that holds four consecutive k of one row stores them as two ds_write_b64.  k-contiguous operands
// are fetched as float4 along k; m-contiguous operands as four row-strided dwords (each wave load is 256 contiguous
// bytes).  The next k-slice is prefetched into registers while the current one feeds the MFMAs.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { GEMM_EPI_NONE = 0, GEMM_EPI_BIAS = 1, GEMM_EPI_BIAS_ELU = 2, GEMM_EPI_MUL_DELU = 3 };

struct GemmArgs {
  const float* A; const float* B; float* C;
  int M, N, K;          // C is M x N, reduction length K
  int lda, ldb, ldc;    // row strides (floats) of the arrays as stored
  const float* bias;    // [N]            (EPI_BIAS*)
  const float* H;       // [M, ldh]       (EPI_MUL_DELU: C *= (H > 0 ? 1 : H + 1))
  int ldh;
  int k_per_split;      // BWD_W: rows of the reduction handled per split (multiple of 32); C slab z at C + z*M*ldc
  int nsplit;           // number of splits (set by pf_gemm_launch)
  // pf_gemm3.h only: B handed over already split into three bf16 planes, [3][N][ldbp] with k contiguous (weights are split
  // once per optimiser step, not once per block): plane p at Bp + p * bplane (elements); rows padded with zeros to ldbp % 8 == 0
  const unsigned short* Bp;
  int ldbp;
  long long bplane;
  // diagnostic builds of pf_gemm3f.h only (-DG3F_EMIT_PLANES, tools/gemm_dma_ablate.hip): the write-out ALSO stores the result as
  // three bf16 planes [3][M][ldc] -- what a weight-gradient GEMM fed by LDS-DMA would consume; the product never sets it
  unsigned short* Cp;
  long long cplane;
};

__device__ __forceinline__ float pf_elu(float x) { return x > 0.0f ? x : (__expf(x) - 1.0f); }

// A_KC: A stored [M][K] (k contiguous) else [K][M] (m contiguous).  B_KC: B stored [N][K] else [K][N].
//
// The grid is one-dimensional and *persistent-capable*: block b walks tiles b, b + gridDim.x, ... of the
// (M tiles) x (N tiles) x (splits) space.  Before a block writes a finished tile it has already issued the global loads
// of its next tile's first k-slice, so the C-tile write (with bias/ELU) and the next tile's load latency overlap instead
// of adding up.  Tile numbers are remapped so the eight XCDs (round-robin over workgroup ids, one L2 each) each own a
// contiguous range of tiles: blocks that share A rows / B columns hit the same L2.
constexpr int PF_GEMM_BK = 32;
constexpr int PF_GEMM_LDS_S = PF_GEMM_BK + 4;      // LDS row stride (floats): 16 B aligned rows, conflict-free b128 fragment reads
// The kernel body: block `bid` of `nblk` blocks working on problem `g` (so one launch can carry two problems, see
// pf_gemm_kernel2); As / Bs are the workgroup's LDS tiles (BM x LDS_S and BN x LDS_S floats).
template <int BM, int BN, bool A_KC, bool B_KC, int EPI, bool COLSUM, bool VEC>
__device__ __forceinline__ void pf_gemm_body(const GemmArgs& g, const int bid, const int nblk, float* As, float* Bs) {
  constexpr int BK = PF_GEMM_BK;
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int LDS_S = PF_GEMM_LDS_S;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_x = (g.N + BN - 1) / BN, tiles_y = (g.M + BM - 1) / BM;
  const int nsplit = g.nsplit > 0 ? g.nsplit : 1;
  const int ntiles = tiles_x * tiles_y * nsplit;
  const int per_xcd = (ntiles + 7) >> 3, tmax = per_xcd << 3;

  struct Tile { int m0, n0, k_begin, k_end, bx; float* C; };
  auto decode = [&](int t, Tile& T) -> bool {
    int lt = (t & 7) * per_xcd + (t >> 3);      // XCD (t & 7) owns logical tiles [xcd * per_xcd, (xcd + 1) * per_xcd)
    if (lt >= ntiles) return false;
    int bx = lt % tiles_x, rest = lt / tiles_x;
    int by = rest % tiles_y, bz = rest / tiles_y;
    T.bx = bx; T.m0 = by * BM; T.n0 = bx * BN; T.k_begin = 0; T.k_end = g.K; T.C = g.C;
    if (g.k_per_split > 0) {
      T.k_begin = bz * g.k_per_split;
      T.k_end = min(g.K, T.k_begin + g.k_per_split);
      T.C += (size_t)bz * ((size_t)g.M * g.ldc + (COLSUM ? g.M : 0));
    }
    return true;
  };

  constexpr int A_V4 = BM * BK / 4 / 256, B_V4 = BN * BK / 4 / 256;   // float4 chunks per thread
  float4 ra[A_V4], rb[B_V4];

  // Operand loads go through buffer descriptors: an out-of-range offset returns 0 in hardware, so tile edges (rows past
  // the array, k past the end of the reduction range) need no branches and no value selects -- a lane that must read
  // zeros just gets an out-of-range offset -- and the prefetch stays in flight across the MFMAs.  VEC: k-contiguous rows are 16 B aligned and
  // the reduction length is a multiple of 4 (checked by pf_gemm_launch), so a row quad is one dwordx4 load.
  constexpr unsigned OOB = 0x80000000u;                     // arrays are < 2 GiB (checked on the host)
  const unsigned a_bytes = (unsigned)(A_KC ? g.M : g.K) * (unsigned)g.lda * 4u;
  const unsigned b_bytes = (unsigned)(B_KC ? g.N : g.K) * (unsigned)g.ldb * 4u;
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B), 0, b_bytes, 0x00020000);
  auto ld1 = [&](__amdgpu_buffer_rsrc_t rs, unsigned off) -> float {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0));
  };
  // k-contiguous operand: 4 consecutive k of row `row`; zero outside [nrows) x [k_end)
  auto load_kc = [&](__amdgpu_buffer_rsrc_t rs, int ld, int row, int k, int nrows, int k_end) -> float4 {
    const unsigned off = ((unsigned)row * (unsigned)ld + (unsigned)k) * 4u;
    const bool rv = row < nrows;
    if constexpr (VEC)
      return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((rv && k < k_end) ? off : OOB), 0, 0));
    return make_float4(ld1(rs, (rv && k < k_end) ? off : OOB), ld1(rs, (rv && k + 1 < k_end) ? off + 4 : OOB),
                       ld1(rs, (rv && k + 2 < k_end) ? off + 8 : OOB), ld1(rs, (rv && k + 3 < k_end) ? off + 12 : OOB));
  };
  // m-contiguous operand: rows k..k+3 of column `col` (a wave reads 64 consecutive columns per row)
  auto load_mc = [&](__amdgpu_buffer_rsrc_t rs, int ld, int k, int col, int k_end, int ncols) -> float4 {
    const unsigned off = ((unsigned)k * (unsigned)ld + (unsigned)col) * 4u, st = (unsigned)ld * 4u;
    const bool cv = col < ncols;
    return make_float4(ld1(rs, (cv && k < k_end) ? off : OOB), ld1(rs, (cv && k + 1 < k_end) ? off + st : OOB),
                       ld1(rs, (cv && k + 2 < k_end) ? off + 2 * st : OOB), ld1(rs, (cv && k + 3 < k_end) ? off + 3 * st : OOB));
  };
  // thread -> (tile row, k quad): k-contiguous: 8 quads of one row are adjacent threads; m-contiguous: adjacent threads
  // are adjacent rows
  auto fetch = [&](const Tile& T, int k0) {
#pragma unroll
    for (int i = 0; i < A_V4; ++i) {
      int idx = tid + 256 * i;
      if (A_KC) ra[i] = load_kc(a_rs, g.lda, T.m0 + idx / (BK / 4), k0 + (idx % (BK / 4)) * 4, g.M, T.k_end);
      else ra[i] = load_mc(a_rs, g.lda, k0 + (idx / BM) * 4, T.m0 + idx % BM, T.k_end, g.M);
    }
#pragma unroll
    for (int i = 0; i < B_V4; ++i) {
      int idx = tid + 256 * i;
      if (B_KC) rb[i] = load_kc(b_rs, g.ldb, T.n0 + idx / (BK / 4), k0 + (idx % (BK / 4)) * 4, g.N, T.k_end);
      else rb[i] = load_mc(b_rs, g.ldb, k0 + (idx / BN) * 4, T.n0 + idx % BN, T.k_end, g.N);
    }
  };
  // v = values at k = 4q .. 4q+3 of one row: (k, k+1) feed lane half 0 at MFMA steps (2q, 2q+1), (k+2, k+3) lane half 1 at
  // the same steps -- register pairs go to LDS as they are (one ds_write2_b64), no shuffles between load and store
  auto put = [&](float* S, int row, int q, const float4& v) {
    *(float2*)&S[row * LDS_S + 2 * q] = make_float2(v.x, v.y);
    *(float2*)&S[row * LDS_S + 16 + 2 * q] = make_float2(v.z, v.w);
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < A_V4; ++i) {
      int idx = tid + 256 * i;
      if (A_KC) put(As, idx / (BK / 4), idx % (BK / 4), ra[i]); else put(As, idx % BM, idx / BM, ra[i]);
    }
#pragma unroll
    for (int i = 0; i < B_V4; ++i) {
      int idx = tid + 256 * i;
      if (B_KC) put(Bs, idx / (BK / 4), idx % (BK / 4), rb[i]); else put(Bs, idx % BN, idx / BN, rb[i]);
    }
  };

  Tile cur;
  int t = bid;
  while (t < tmax && !decode(t, cur)) t += nblk;
  if (t >= tmax) return;
  fetch(cur, cur.k_begin);
  const int arow = wr * (BM / 2) + (lane & 31), bcol = wc * (BN / 2) + (lane & 31), kh = lane >> 5;
  // dwordx4 write-out possible: rows of C (and of H, and the bias) 16-byte aligned, width a multiple of 4
  bool cvec = ((g.N & 3) == 0) && ((g.ldc & 3) == 0) && ((((size_t)g.C) & 15) == 0) && (!COLSUM || (g.M & 3) == 0);
  if (EPI == GEMM_EPI_BIAS || EPI == GEMM_EPI_BIAS_ELU) cvec = cvec && ((((size_t)g.bias) & 15) == 0);
  if (EPI == GEMM_EPI_MUL_DELU) cvec = cvec && ((g.ldh & 3) == 0) && ((((size_t)g.H) & 15) == 0);

  while (true) {
    // ---- v_mfma_f32_32x32x2_f32: wave tile = TM x TN tiles of 32x32
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    float csum = 0.0f;   // COLSUM: sum over the reduction rows of A[:, m0 + tid] (bias gradient), tiles with bx == 0
    for (int k0 = cur.k_begin; k0 < cur.k_end; k0 += BK) {
      __syncthreads();          // previous k-slice (or previous tile) fully consumed
      stage();
      __syncthreads();
      if (k0 + BK < cur.k_end) fetch(cur, k0 + BK);
      if (COLSUM && cur.bx == 0 && tid < BM) {
#pragma unroll
        for (int q = 0; q < BK / 4; ++q) { float4 v = *(const float4*)&As[tid * LDS_S + 4 * q]; csum += (v.x + v.y) + (v.z + v.w); }
      }
#pragma unroll
      for (int q = 0; q < BK / 8; ++q) {       // four MFMA steps per fragment read
        float4 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *(const float4*)&As[(arow + 32 * i) * LDS_S + kh * 16 + 4 * q];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *(const float4*)&Bs[(bcol + 32 * j) * LDS_S + kh * 16 + 4 * q];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
          }
      }
    }
    // next tile of this block: get its first k-slice in flight before the write-out below
    Tile nxt;
    int tn = t + nblk;
    bool more = false;
    while (tn < tmax) { if (decode(tn, nxt)) { more = true; break; } tn += nblk; }
    if (more) fetch(nxt, nxt.k_begin);

    if (COLSUM && cur.bx == 0 && tid < BM && cur.m0 + tid < g.M) cur.C[(size_t)g.M * g.ldc + cur.m0 + tid] = csum;
    bool done = false;
    if constexpr (BM == 64 && BN == 64) {
      if (cvec) {      // uniform
        // 64x64 tiles, 16-byte aligned C rows: transpose the accumulators through LDS (the operand tiles are dead) so a lane
        // writes 4 x dwordx4 of one row instead of 16 dwords in 16 rows -- the write-out is store-issue bound.
        float* Cs = As;                      // As and Bs are contiguous: (64 + 64) * 36 floats >= 64 * 64
        __syncthreads();                     // the last slice's fragment reads are done in every wave
#pragma unroll
        for (int r = 0; r < 16; ++r)         // C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
          Cs[(wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 64 + wc * 32 + (lane & 31)] = acc[0][0][r];
        __syncthreads();
        const int c4 = tid & 15, r0 = tid >> 4;
        const int col = cur.n0 + 4 * c4;
        if (col < g.N) {                     // N % 4 == 0: the whole quad is inside
          float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
          if (EPI == GEMM_EPI_BIAS || EPI == GEMM_EPI_BIAS_ELU) b4 = *reinterpret_cast<const float4*>(g.bias + col);
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const int rl = r0 + 16 * p, row = cur.m0 + rl;
            if (row >= g.M) continue;
            float4 v = *reinterpret_cast<const float4*>(&Cs[rl * 64 + 4 * c4]);
            if (EPI == GEMM_EPI_BIAS || EPI == GEMM_EPI_BIAS_ELU) { v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w; }
            if (EPI == GEMM_EPI_BIAS_ELU) { v.x = pf_elu(v.x); v.y = pf_elu(v.y); v.z = pf_elu(v.z); v.w = pf_elu(v.w); }
            if (EPI == GEMM_EPI_MUL_DELU) {
              const float4 h = *reinterpret_cast<const float4*>(g.H + (size_t)row * g.ldh + col);
              v.x *= (h.x > 0.0f ? 1.0f : h.x + 1.0f); v.y *= (h.y > 0.0f ? 1.0f : h.y + 1.0f);
              v.z *= (h.z > 0.0f ? 1.0f : h.z + 1.0f); v.w *= (h.w > 0.0f ? 1.0f : h.w + 1.0f);
            }
            *reinterpret_cast<float4*>(cur.C + (size_t)row * g.ldc + col) = v;
          }
        }
        done = true;
      }
    }
    if (!done) {
    // C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = cur.n0 + wc * (BN / 2) + 32 * j + (lane & 31);
        float bias = 0.0f;
        if (EPI == GEMM_EPI_BIAS || EPI == GEMM_EPI_BIAS_ELU) bias = col < g.N ? g.bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = cur.m0 + wr * (BM / 2) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (row >= g.M || col >= g.N) continue;
          float v = acc[i][j][r];
          if (EPI == GEMM_EPI_BIAS || EPI == GEMM_EPI_BIAS_ELU) v += bias;
          if (EPI == GEMM_EPI_BIAS_ELU) v = pf_elu(v);
          if (EPI == GEMM_EPI_MUL_DELU) { float h = g.H[(size_t)row * g.ldh + col]; v *= (h > 0.0f ? 1.0f : h + 1.0f); }
          cur.C[(size_t)row * g.ldc + col] = v;
        }
      }
    }
    if (!more) break;
    cur = nxt; t = tn;
  }
}


template <int BM, int BN, bool A_KC, bool B_KC, int EPI, bool COLSUM = false, bool VEC = true>
__global__ __launch_bounds__(256) void pf_gemm_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * PF_GEMM_LDS_S];     // A tile then B tile, contiguous
  float* As = smem; float* Bs = smem + BM * PF_GEMM_LDS_S;
  pf_gemm_body<BM, BN, A_KC, B_KC, EPI, COLSUM, VEC>(g, (int)blockIdx.x, (int)gridDim.x, As, Bs);
}
// Two independent problems of the same arrangement in one launch (the actor's and the critic's layer i): blocks
// [0, nb0) work on g0, the rest on g1.  One grid instead of two streams: the tails of one problem are filled by the other
// without cross-stream events.  nb0 is a multiple of 8 so the XCD of a block is the same function of its local id.
template <int BM, int BN, bool A_KC, bool B_KC, int EPI, bool COLSUM, bool VEC0, bool VEC1>
__global__ __launch_bounds__(256, 6) void pf_gemm_kernel2(GemmArgs g0, GemmArgs g1, int nb0) {
  __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * PF_GEMM_LDS_S];     // A tile then B tile, contiguous
  float* As = smem; float* Bs = smem + BM * PF_GEMM_LDS_S;
  if ((int)blockIdx.x < nb0) pf_gemm_body<BM, BN, A_KC, B_KC, EPI, COLSUM, VEC0>(g0, (int)blockIdx.x, nb0, As, Bs);
  else pf_gemm_body<BM, BN, A_KC, B_KC, EPI, COLSUM, VEC1>(g1, (int)blockIdx.x - nb0, (int)gridDim.x - nb0, As, Bs);
}

// ---- weight-gradient GEMM without LDS staging ("direct fragments").
// C[M,N] (split-K slab z) = sum_k A[k][m] B[k][n] with A = g[rows, out], B = x[rows, in], both row-major over the reduction
// index.  The MFMA operand layout of v_mfma_f32_32x32x2_f32 is exactly two rows of such a matrix: lane l supplies
// A[k = 2 step + (l >> 5)][m0 + (l & 31)] -- 32 consecutive floats per half-wave.  So every wave loads its fragments
// straight from global memory (L1/L2 absorb the 2x reuse between the four waves of a tile): no LDS writes, no bank
// conflicts and, above all, NO workgroup barriers in the main loop -- the four waves of a block run free.  Per slice of 32
// rows a wave issues 32 dword loads (one instruction per fragment; the row advance rides in the scalar offset, so there
// is no vector address arithmetic) one slice ahead of the 16 MFMAs that consume them.  COLSUM: the column sums of A (bias
// gradient) fall out of the A fragments a lane already holds.
__device__ __forceinline__ void pf_gemm_bww_body(const GemmArgs& g, const int bid, const int nblk, float* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, kh = lane >> 5, lc = lane & 31;
  const int tiles_x = (g.N + 63) / 64, tiles_y = (g.M + 63) / 64;
  const int nsplit = g.nsplit > 0 ? g.nsplit : 1;
  const int ntiles = tiles_x * tiles_y * nsplit;
  const int per_xcd = (ntiles + 7) >> 3, tmax = per_xcd << 3;
  constexpr unsigned OOB = 0x80000000u;
  const bool cvec = ((g.N & 3) == 0) && ((g.ldc & 3) == 0) && ((((size_t)g.C) & 15) == 0) && ((g.M & 3) == 0);
  for (int t = bid; t < tmax; t += nblk) {
    const int lt = (t & 7) * per_xcd + (t >> 3);
    if (lt >= ntiles) continue;
    const int bx = lt % tiles_x, rest = lt / tiles_x, by = rest % tiles_y, bz = rest / tiles_y;
    const int m0 = by * 64, n0 = bx * 64;
    int k_begin = 0, k_end = g.K;
    float* C = g.C;
    if (g.k_per_split > 0) {
      k_begin = bz * g.k_per_split; k_end = min(g.K, k_begin + g.k_per_split);
      C += (size_t)bz * ((size_t)g.M * g.ldc + g.M);
    }
    // per-lane byte offset of row `kh` inside a 16-row group; columns past the matrix read zeros (out-of-range offset)
    const int ca = m0 + 32 * wr + lc, cb = n0 + 32 * wc + lc;
    const unsigned va = ca < g.M ? ((unsigned)kh * (unsigned)g.lda + (unsigned)ca) * 4u : OOB;
    const unsigned vb = cb < g.N ? ((unsigned)kh * (unsigned)g.ldb + (unsigned)cb) * 4u : OOB;
    const unsigned sa2 = 2u * (unsigned)g.lda * 4u, sb2 = 2u * (unsigned)g.ldb * 4u;      // two rows per MFMA step
    // 16 rows (8 MFMA steps) per fetch, double-buffered.  The descriptor of a fetch starts at its first row and ends at
    // the split's last row, so rows past k_end read zeros in hardware: one code path for whole and ragged groups.
    float a[2][8], b[2][8];
    auto fetch = [&](int k0, auto set_c) {
      constexpr int set = decltype(set_c)::value;
      const unsigned rows = (unsigned)max(0, min(16, k_end - k0));      // 0: every lane out of range, no memory traffic
      // (wave-uniform by construction; readfirstlane makes that provable so the loads are not wrapped in waterfall loops)
      auto uni = [](const float* p) {
        const unsigned long long v = (unsigned long long)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (float*)(((unsigned long long)hi << 32) | lo);
      };
      const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(uni(g.A + (size_t)k0 * g.lda), 0, __builtin_amdgcn_readfirstlane(rows * (unsigned)g.lda * 4u), 0x00020000);
      const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(uni(g.B + (size_t)k0 * g.ldb), 0, __builtin_amdgcn_readfirstlane(rows * (unsigned)g.ldb * 4u), 0x00020000);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        a[set][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, (int)(va + s * sa2), 0, 0));
        b[set][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, (int)(vb + s * sb2), 0, 0));
      }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    float csum = 0.0f;
    // straight-line body (no branches between a fetch and its use, so the waits count exactly the newer loads in flight):
    // 32 rows per trip, the group after next always requested -- past k_end it is all zeros and costs nothing
    fetch(k_begin, std::integral_constant<int, 0>{});
    for (int k0 = k_begin; k0 < k_end; k0 += 32) {
      fetch(k0 + 16, std::integral_constant<int, 1>{});
#pragma unroll
      for (int s = 0; s < 8; ++s) { acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][s], b[0][s], acc, 0, 0, 0); csum += a[0][s]; }
      fetch(k0 + 32, std::integral_constant<int, 0>{});
#pragma unroll
      for (int s = 0; s < 8; ++s) { acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][s], b[1][s], acc, 0, 0, 0); csum += a[1][s]; }
    }
    // bias gradient: column sums of A over this split's rows (waves that own the tile's A columns, first column tile only)
    csum += __shfl_xor(csum, 32, 64);
    if (bx == 0 && wc == 0 && kh == 0 && ca < g.M) C[(size_t)g.M * g.ldc + ca] = csum;
    // write-out (C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
    if (cvec) {
      __syncthreads();                   // previous tile's LDS reads are done (persistent grids)
#pragma unroll
      for (int r = 0; r < 16; ++r) smem[(wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * 64 + wc * 32 + lc] = acc[r];
      __syncthreads();
      const int c4 = tid & 15, r0 = tid >> 4, col = n0 + 4 * c4;
      if (col < g.N) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int rl = r0 + 16 * p, row = m0 + rl;
          if (row < g.M) *reinterpret_cast<float4*>(C + (size_t)row * g.ldc + col) = *reinterpret_cast<const float4*>(&smem[rl * 64 + 4 * c4]);
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < g.M && cb < g.N) C[(size_t)row * g.ldc + cb] = acc[r];
      }
    }
  }
}
__global__ __launch_bounds__(256, 6) void pf_gemm_bww_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float smem[64 * 64];
  pf_gemm_bww_body(g, (int)blockIdx.x, (int)gridDim.x, smem);
}
__global__ __launch_bounds__(256, 6) void pf_gemm_bww_kernel2(GemmArgs g0, GemmArgs g1, int nb0) {
  __shared__ __attribute__((aligned(16))) float smem[64 * 64];
  if ((int)blockIdx.x < nb0) pf_gemm_bww_body(g0, (int)blockIdx.x, nb0, smem);
  else pf_gemm_bww_body(g1, (int)blockIdx.x - nb0, (int)gridDim.x - nb0, smem);
}
// PF_GEMM_BWW=0 falls back to the LDS-staged kernel for the weight-gradient GEMM
static inline bool pf_gemm_use_bww() { static const bool v = [] { const char* e = getenv("PF_GEMM_BWW"); return e ? atoi(e) != 0 : true; }(); return v; }

// PF_GEMM_PERSIST = resident blocks per CU for the persistent grid (0: one block per tile)
static inline int pf_gemm_persist() { static const int v = [] { const char* e = getenv("PF_GEMM_PERSIST"); return e ? atoi(e) : 0; }(); return v; }
template <int BM, int BN>
static inline int pf_gemm_blocks(GemmArgs& g, int splits) {
  g.nsplit = splits > 0 ? splits : 1;
  int ntiles = ((g.N + BN - 1) / BN) * ((g.M + BM - 1) / BM) * g.nsplit;
  int blocks = (ntiles + 7) / 8 * 8;
  int p = pf_gemm_persist();
  if (p > 0 && blocks > p * 256) blocks = p * 256;
  return blocks;
}
// float4 row loads of the k-contiguous operands need 16 B aligned rows and K % 4 == 0
template <bool A_KC, bool B_KC>
static inline bool pf_gemm_vec(const GemmArgs& g) {
  if (!A_KC && !B_KC) return true;      // no k-contiguous operand: one instance
  bool vec = (g.K & 3) == 0;
  if (A_KC) vec = vec && ((((size_t)g.A) & 15) == 0) && ((g.lda & 3) == 0);
  if (B_KC) vec = vec && ((((size_t)g.B) & 15) == 0) && ((g.ldb & 3) == 0);
  return vec;
}
template <int BM, int BN, bool A_KC, bool B_KC, int EPI, bool COLSUM = false>
static inline void pf_gemm_launch(const GemmArgs& g_in, int splits, hipStream_t s) {
  GemmArgs g = g_in;
  const int blocks = pf_gemm_blocks<BM, BN>(g, splits);
  if constexpr (BM == 64 && BN == 64 && !A_KC && !B_KC && EPI == GEMM_EPI_NONE && COLSUM) {
    if (pf_gemm_use_bww()) { hipLaunchKernelGGL(pf_gemm_bww_kernel, dim3(blocks), dim3(256), 0, s, g); return; }
  }
  if (pf_gemm_vec<A_KC, B_KC>(g)) hipLaunchKernelGGL((pf_gemm_kernel<BM, BN, A_KC, B_KC, EPI, COLSUM, true>), dim3(blocks), dim3(256), 0, s, g);
  else hipLaunchKernelGGL((pf_gemm_kernel<BM, BN, A_KC, B_KC, EPI, COLSUM, (A_KC || B_KC) ? false : true>), dim3(blocks), dim3(256), 0, s, g);
}
// two problems, one launch (64x64 tiles)
template <bool A_KC, bool B_KC, int EPI, bool COLSUM = false>
static inline void pf_gemm_launch2(const GemmArgs& g0_in, int splits0, const GemmArgs& g1_in, int splits1, hipStream_t s) {
  GemmArgs g0 = g0_in, g1 = g1_in;
  const int nb0 = pf_gemm_blocks<64, 64>(g0, splits0), nb1 = pf_gemm_blocks<64, 64>(g1, splits1);
  if constexpr (!A_KC && !B_KC && EPI == GEMM_EPI_NONE && COLSUM) {
    if (pf_gemm_use_bww()) { hipLaunchKernelGGL(pf_gemm_bww_kernel2, dim3(nb0 + nb1), dim3(256), 0, s, g0, g1, nb0); return; }
  }
  const bool v0 = pf_gemm_vec<A_KC, B_KC>(g0), v1 = pf_gemm_vec<A_KC, B_KC>(g1);
  constexpr bool HAS_KC = A_KC || B_KC;
  const dim3 grid(nb0 + nb1), block(256);
  if (v0 && v1) hipLaunchKernelGGL((pf_gemm_kernel2<64, 64, A_KC, B_KC, EPI, COLSUM, true, true>), grid, block, 0, s, g0, g1, nb0);
  else if (!v0 && v1) hipLaunchKernelGGL((pf_gemm_kernel2<64, 64, A_KC, B_KC, EPI, COLSUM, HAS_KC ? false : true, true>), grid, block, 0, s, g0, g1, nb0);
  else if (v0 && !v1) hipLaunchKernelGGL((pf_gemm_kernel2<64, 64, A_KC, B_KC, EPI, COLSUM, true, HAS_KC ? false : true>), grid, block, 0, s, g0, g1, nb0);
  else hipLaunchKernelGGL((pf_gemm_kernel2<64, 64, A_KC, B_KC, EPI, COLSUM, HAS_KC ? false : true, HAS_KC ? false : true>), grid, block, 0, s, g0, g1, nb0);
}

// Tile choice from a small cost model fitted to measurements on MI355X (tools/gemm_bench.py): a k-slice costs the tile's
// MFMA time (64 cycles per v_mfma_f32_32x32x2_f32) divided by a per-tile efficiency, every block pays a prologue/epilogue
// cost that grows with its tile (first global->LDS fill, C-tile write with bias/ELU), and the grid runs in
// ceil(blocks / 256 CUs) rounds.  Only the ranking matters (every PPO shape ranks 64x64 first; 4096^3 ranks 128x64 first,
// as measured).  Returns the modelled time in cycles; `tile_out` = 0: 128x128, 1: 128x64, 2: 64x64.
static inline float pf_tile_cost(int M, int N, int K, int splits, int t) {
  const int bm[3] = {128, 128, 64}, bn[3] = {128, 64, 64};
  const float cyc[3] = {4096.0f, 2048.0f / 0.95f, 1024.0f / 0.90f}, ovh[3] = {30000.0f, 11000.0f, 4200.0f};
  int S = splits > 0 ? splits : 1;
  long blocks = (long)((M + bm[t] - 1) / bm[t]) * ((N + bn[t] - 1) / bn[t]) * S;
  long rounds = (blocks + 255) / 256;
  int ktiles = ((K + S - 1) / S + 31) / 32;
  return (float)rounds * ((float)ktiles * cyc[t] + ovh[t]);
}
static inline int pf_pick_tile(int M, int N, int K, int splits) {
  int best = 0; float bc = 1e30f;
  for (int t = 0; t < 3; ++t) { float c = pf_tile_cost(M, N, K, splits, t); if (c < bc) { bc = c; best = t; } }
  return best;
}
// Split-K choice for the weight-gradient GEMM (output only out x in, reduction over tens of thousands of rows): measured
// on MI355X with the direct-fragment kernel (tools/gemm_sweep_bwdw.py): 512x256: 69.0 us at 512 blocks, 67.7 at 1024,
// 68.7 at 2048, 74.8 at 4096; 512x148: 59.1 / 47.3 at 768 / 50.7 at 1152; 256x128 and 512x27 (8 tiles): 22.3 / 21.6 at 512
// blocks, 24.5 / 22.8 at 1024.  So: ~1024 blocks for >= 32 tiles, ~768 for >= 16, ~512 below, capped by the slab workspace.
static inline void pf_pick_split(int M, int N, int K, int max_splits, int* tile_out, int* splits_out) {
  int tiles = ((M + 63) / 64) * ((N + 63) / 64);
  const int target = tiles >= 32 ? 1024 : (tiles >= 16 ? 768 : 512);
  int S = (target + tiles - 1) / tiles;
  if (S > max_splits) S = max_splits;
  while (S > 1 && (K + S - 1) / S < 64) --S;
  *tile_out = 2; *splits_out = S < 1 ? 1 : S;
}
template <bool A_KC, bool B_KC, int EPI, bool COLSUM = false>
static inline void pf_gemm_auto(const GemmArgs& g, int splits, hipStream_t s);
// the pair in one launch when both problems want 64x64 tiles (every PPO shape does), else one launch each
template <bool A_KC, bool B_KC, int EPI, bool COLSUM = false>
static inline void pf_gemm_auto2(const GemmArgs& g0, int splits0, const GemmArgs& g1, int splits1, hipStream_t s) {
  if (pf_pick_tile(g0.M, g0.N, g0.K, splits0) == 2 && pf_pick_tile(g1.M, g1.N, g1.K, splits1) == 2)
    pf_gemm_launch2<A_KC, B_KC, EPI, COLSUM>(g0, splits0, g1, splits1, s);
  else { pf_gemm_auto<A_KC, B_KC, EPI, COLSUM>(g0, splits0, s); pf_gemm_auto<A_KC, B_KC, EPI, COLSUM>(g1, splits1, s); }
}
template <bool A_KC, bool B_KC, int EPI, bool COLSUM>
static inline void pf_gemm_auto(const GemmArgs& g, int splits, hipStream_t s) {
  switch (pf_pick_tile(g.M, g.N, g.K, splits)) {
    case 0: pf_gemm_launch<128, 128, A_KC, B_KC, EPI, COLSUM>(g, splits, s); break;
    case 1: pf_gemm_launch<128, 64, A_KC, B_KC, EPI, COLSUM>(g, splits, s); break;
    default: pf_gemm_launch<64, 64, A_KC, B_KC, EPI, COLSUM>(g, splits, s); break;
  }
}

// pf_gemm3.h — fp32 GEMM on the bf16 matrix pipes by exact operand splitting ("bf16x6"), gfx950.
//
// v_mfma_f32_32x32x2_f32 runs at 64 FLOP/clk/SIMD, v_mfma_f32_32x32x16_bf16 at 1024.  An fp32 number is the EXACT sum of
// three bf16 numbers: a = a1 + a2 + a3 with a1 = trunc_bf16(a), a2 = trunc_bf16(a - a1), a3 = a - a1 - a2 (8 + 8 + 8 = 24
// significand bits; the subtractions are exact in fp32).  Then
//     a b = a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1) + [a2 b3 + a3 b2 + a3 b3],
// every bf16 x bf16 product is exact in the MFMA's fp32 accumulation, and the bracket -- the only thing dropped -- is below
// 2^-23 |a b|: the rounding class of ONE fp32 multiply.  Six bf16 MFMAs replace one fp32 MFMA step of the same k-extent at
// 16x the rate: 2.67x the fp32 matrix peak, with fp32 accumulation and fp32-grade products.  tests/test_gpu_ppo.py holds
// this path to the same tolerances against the fp32 autograd oracle as the plain fp32 MFMA path (pf_gemm.h), and
// test_gemm_kernel_family_on_ragged_shapes compares both against float64.
//
// Same three arrangements and epilogues as pf_gemm.h (FWD / BWD_X / BWD_W with split-K slabs and fused bias column sums).
// Tiling: 256 threads = 2x2 wavefronts, block tile BM x BN in {64,128}^2, BK = 32; a wave owns (BM/2) x (BN/2) as 32x32
// MFMA tiles.  Operands are fetched as fp32 (buffer loads, out-of-range -> 0), split in registers and staged in LDS as three
// bf16 planes per operand, row-major over the tile's m (n) index, 32 k = 64 bytes per row with the 16-byte chunks XOR-swizzled
// (g3_soff: conflict-free b128 stores and reads); lane l of an MFMA reads row l & 31, k = 8 (l >> 5) .. +7 of a 16-wide
// k-step: one ds_read_b128 per plane per 32x32 operand tile per k-step.
#pragma once
#include "pf_gemm.h"

typedef __bf16 pf_bf16x8 __attribute__((ext_vector_type(8)));

// Diagnostic builds only (tools/gemm_ablate.hip): -DG3_ABLATE=mask removes one ingredient of the main loop at a time to see
// what the loop's time is made of (results are wrong by construction).  1 MFMAs, 2 operand split, 4 LDS stores, 8 in-loop
// global loads, 16 fragment reads, 32 barriers.  The library is built with 0 and contains none of it.
#ifndef G3_ABLATE
#define G3_ABLATE 0
#endif
#define G3_KEEP(x) asm volatile("" ::"v"(x))
constexpr int G3_BK = 32;
constexpr int G3_RS = 32;      // LDS row stride in bf16 elements: one 64-byte row per 32-wide k-slice, no padding
// A row holds four 16-byte chunks (8 k each); chunk c of row r lives at chunk position c ^ ((r >> 2) & 3).  A 16-lane phase
// of a ds_read_b128 takes the same chunk of 16 consecutive rows: rows r, r+4, r+8, r+12 (same bank group unswizzled) land in
// four different positions, so the phase covers all 64 banks once; a 16-lane phase of the staging ds_write_b128 covers four
// whole rows.  No padding: 36 KB for a 128 x 64 tile, so LDS never limits the number of resident workgroups.
__device__ __forceinline__ int g3_soff(int row, int chunk) { return row * G3_RS + ((chunk ^ ((row >> 2) & 3)) << 3); }

// exact split of 4 consecutive-k fp32 values of one row into three packed-bf16 quads
__device__ __forceinline__ void g3_split4(const float4& v, uint2& p0, uint2& p1, uint2& p2) {
  const float x[4] = {v.x, v.y, v.z, v.w};
  if constexpr ((G3_ABLATE & 2) != 0) {
    p0 = make_uint2(__builtin_bit_cast(uint32_t, v.x), __builtin_bit_cast(uint32_t, v.y));
    p1 = make_uint2(__builtin_bit_cast(uint32_t, v.z), __builtin_bit_cast(uint32_t, v.w));
    p2 = p0;
    return;
  }
  uint32_t h[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t u = __builtin_bit_cast(uint32_t, x[i]);
    const float r1 = x[i] - __builtin_bit_cast(float, u & 0xFFFF0000u);        // exact
    const uint32_t u1 = __builtin_bit_cast(uint32_t, r1);
    const float r2 = r1 - __builtin_bit_cast(float, u1 & 0xFFFF0000u);         // exact; <= 8 significant bits left
    h[i] = u; m[i] = u1; l[i] = __builtin_bit_cast(uint32_t, r2);
  }
  // pack the high halves: {hi16(e0), hi16(e1)} -> one dword (v_perm_b32)
  auto pk = [](uint32_t e0, uint32_t e1) { return __builtin_amdgcn_perm(e1, e0, 0x07060302u); };
  p0 = make_uint2(pk(h[0], h[1]), pk(h[2], h[3]));
  p1 = make_uint2(pk(m[0], m[1]), pk(m[2], m[3]));
  p2 = make_uint2(pk(l[0], l[1]), pk(l[2], l[3]));
}

// 8 consecutive-k fp32 values -> one 16-byte chunk per plane
__device__ __forceinline__ void g3_split8(const float4& v0, const float4& v1, uint4& p0, uint4& p1, uint4& p2) {
  uint2 a0, a1, a2, b0, b1, b2;
  g3_split4(v0, a0, a1, a2);
  g3_split4(v1, b0, b1, b2);
  p0 = make_uint4(a0.x, a0.y, b0.x, b0.y); p1 = make_uint4(a1.x, a1.y, b1.x, b1.y); p2 = make_uint4(a2.x, a2.y, b2.x, b2.y);
}

// Write-out of a block tile's accumulators (shared by the row-image and the transposed-image kernels).  C/D layout of a
// 32x32 tile: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  The caller has synchronised the workgroup:
// the operand tiles in `smem` are dead.
template <int BM, int BN, int EPI, bool COLSUM>
__device__ __forceinline__ void g3_writeout(const GemmArgs& g, float* C, f32x16 (&acc)[BM / 64][BN / 64], int m0, int n0, unsigned char* smem) {
  constexpr int TM = BM / 64, TN = BN / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, kh = lane >> 5;
  bool cvec = ((g.N & 3) == 0) && ((g.ldc & 3) == 0) && ((((size_t)C) & 15) == 0) && (!COLSUM || (g.M & 3) == 0);
  if (EPI == GEMM_EPI_BIAS || EPI == GEMM_EPI_BIAS_ELU) cvec = cvec && ((((size_t)g.bias) & 15) == 0);
  if (EPI == GEMM_EPI_MUL_DELU) cvec = cvec && ((g.ldh & 3) == 0) && ((((size_t)g.H) & 15) == 0);
  if (cvec) {
    // every wave transposes its own 32-row strips through a private LDS region and stores whole rows as dwordx4.
    // All global LOADS of the epilogue (bias, the ELU' operand of every strip) are issued before the first STORE: vmcnt retires
    // in order, so a load issued behind a store is not usable until that store has been acknowledged by memory, and a
    // "load h, multiply, store" loop pays a store round trip per pass (profiles/r3_gemm_dma_ablation.txt).
    constexpr int WN = BN / 2, CS = WN + 4;                         // strip row stride (floats): 16-byte aligned rows
    constexpr int Q = WN / 4, P = Q / 2, RP = 64 / Q;               // float4 per strip row; passes per strip; rows per pass
    float* Cs = reinterpret_cast<float*>(smem) + wave * (32 * CS);
    const int c4 = lane % Q, r0 = lane / Q, col = n0 + wc * WN + 4 * c4;
    const bool cok = col < g.N;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), hv[TM][P];
    if (EPI == GEMM_EPI_BIAS || EPI == GEMM_EPI_BIAS_ELU) { if (cok) b4 = *reinterpret_cast<const float4*>(g.bias + col); }
    if (EPI == GEMM_EPI_MUL_DELU) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const int row = m0 + wr * (BM / 2) + 32 * i + r0 + RP * p;
          hv[i][p] = (row < g.M && cok) ? *reinterpret_cast<const float4*>(g.H + (size_t)row * g.ldh + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) Cs[((r & 3) + 8 * (r >> 2) + 4 * kh) * CS + 32 * j + (lane & 31)] = acc[i][j][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int rl = r0 + RP * p, row = m0 + wr * (BM / 2) + 32 * i + rl;
        float4 v = *reinterpret_cast<const float4*>(&Cs[rl * CS + 4 * c4]);
        if (EPI == GEMM_EPI_BIAS || EPI == GEMM_EPI_BIAS_ELU) { v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w; }
        if (EPI == GEMM_EPI_BIAS_ELU) { v.x = pf_elu(v.x); v.y = pf_elu(v.y); v.z = pf_elu(v.z); v.w = pf_elu(v.w); }
        if (EPI == GEMM_EPI_MUL_DELU) {
          const float4 h = hv[i][p];
          v.x *= (h.x > 0.0f ? 1.0f : h.x + 1.0f); v.y *= (h.y > 0.0f ? 1.0f : h.y + 1.0f);
          v.z *= (h.z > 0.0f ? 1.0f : h.z + 1.0f); v.w *= (h.w > 0.0f ? 1.0f : h.w + 1.0f);
        }
        if (row < g.M && cok) *reinterpret_cast<float4*>(C + (size_t)row * g.ldc + col) = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wc * (BN / 2) + 32 * j + (lane & 31);
        float bias = 0.0f;
        if (EPI == GEMM_EPI_BIAS || EPI == GEMM_EPI_BIAS_ELU) bias = col < g.N ? g.bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wr * (BM / 2) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kh;
          if (row >= g.M || col >= g.N) continue;
          float v = acc[i][j][r];
          if (EPI == GEMM_EPI_BIAS || EPI == GEMM_EPI_BIAS_ELU) v += bias;
          if (EPI == GEMM_EPI_BIAS_ELU) v = pf_elu(v);
          if (EPI == GEMM_EPI_MUL_DELU) { float h = g.H[(size_t)row * g.ldh + col]; v *= (h > 0.0f ? 1.0f : h + 1.0f); }
          C[(size_t)row * g.ldc + col] = v;
        }
      }
  }
}

template <int BM, int BN, bool A_KC, bool B_KC, int EPI, bool COLSUM, bool VEC, bool BP = false>
__device__ __forceinline__ void pf_gemm3_body(const GemmArgs& g, const int bid, const int nblk, unsigned char* smem) {
  static_assert(!COLSUM || !A_KC, "column sums are taken over an m-contiguous A (weight-gradient arrangement)");
  constexpr int BK = G3_BK, RS = G3_RS;
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int PLANE_A = BM * RS, PLANE_B = BN * RS;        // bf16 elements per plane
  uint16_t* As = reinterpret_cast<uint16_t*>(smem);          // [3][BM][RS]
  uint16_t* Bs = As + 3 * PLANE_A;                           // [3][BN][RS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_x = (g.N + BN - 1) / BN, tiles_y = (g.M + BM - 1) / BM;
  const int nsplit = g.nsplit > 0 ? g.nsplit : 1;
  const int ntiles = tiles_x * tiles_y * nsplit;
  const int per_xcd = (ntiles + 7) >> 3, tmax = per_xcd << 3;

  constexpr unsigned OOB = 0x80000000u;
  const unsigned a_bytes = (unsigned)(A_KC ? g.M : g.K) * (unsigned)g.lda * 4u;
  const unsigned b_bytes = (unsigned)(B_KC ? g.N : g.K) * (unsigned)g.ldb * 4u;
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B), 0, b_bytes, 0x00020000);
  auto ld1 = [&](__amdgpu_buffer_rsrc_t rs, unsigned off) -> float {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0));
  };
  auto load_kc = [&](__amdgpu_buffer_rsrc_t rs, int ld, int row, int k, int nrows, int k_end) -> float4 {
    const unsigned off = ((unsigned)row * (unsigned)ld + (unsigned)k) * 4u;
    const bool rv = row < nrows;
    if constexpr (VEC)
      return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((rv && k < k_end) ? off : OOB), 0, 0));
    return make_float4(ld1(rs, (rv && k < k_end) ? off : OOB), ld1(rs, (rv && k + 1 < k_end) ? off + 4 : OOB),
                       ld1(rs, (rv && k + 2 < k_end) ? off + 8 : OOB), ld1(rs, (rv && k + 3 < k_end) ? off + 12 : OOB));
  };
  auto load_mc = [&](__amdgpu_buffer_rsrc_t rs, int ld, int k, int col, int k_end, int ncols) -> float4 {
    const unsigned off = ((unsigned)k * (unsigned)ld + (unsigned)col) * 4u, st = (unsigned)ld * 4u;
    const bool cv = col < ncols;
    return make_float4(ld1(rs, (cv && k < k_end) ? off : OOB), ld1(rs, (cv && k + 1 < k_end) ? off + st : OOB),
                       ld1(rs, (cv && k + 2 < k_end) ? off + 2 * st : OOB), ld1(rs, (cv && k + 3 < k_end) ? off + 3 * st : OOB));
  };
  constexpr int A_V4 = BM * BK / 4 / 256, B_V4 = BP ? 1 : BN * BK / 4 / 256;
  float4 ra[A_V4], rb[B_V4];
  // BP: B arrives as bf16 planes (k contiguous): a row of a k-slice is 4 chunks of 8 bf16 per plane, staged by plain copies.
  // (Loading the B fragments straight from the global planes into registers -- 16 bytes of 32 different rows per wave load --
  // was measured and is slower: 53 vs 43 us on 24576 x 256 x 512; the texture path serialises the 32 lines.)
  constexpr int B_P8 = BP ? BN * 12 / 256 : 1;
  uint4 rbp[B_P8];
  const unsigned bp_bytes = BP ? (unsigned)(3ll * g.bplane * 2ll) : 0u;
  const __amdgpu_buffer_rsrc_t bp_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(BP ? g.Bp : nullptr), 0, bp_bytes, 0x00020000);

  for (int t = bid; t < tmax; t += nblk) {
    const int lt = (t & 7) * per_xcd + (t >> 3);      // XCD (t & 7) owns a contiguous range of logical tiles
    if (lt >= ntiles) continue;
    const int bx = lt % tiles_x, rest = lt / tiles_x, by = rest % tiles_y, bz = rest / tiles_y;
    const int m0 = by * BM, n0 = bx * BN;
    int k_begin = 0, k_end = g.K;
    float* C = g.C;
    if (g.k_per_split > 0) {
      k_begin = bz * g.k_per_split; k_end = min(g.K, k_begin + g.k_per_split);
      C += (size_t)bz * ((size_t)g.M * g.ldc + (COLSUM ? g.M : 0));
    }
    auto fetch = [&](int k0) {
#pragma unroll
      for (int i = 0; i < A_V4; ++i) {
        const int idx = tid + 256 * i;
        // k-contiguous operand: float4 2t and 2t + 1 are the two halves of 16-byte chunk (T & 3) of row T >> 2, T = tid + 256 t
        if (A_KC) ra[i] = load_kc(a_rs, g.lda, m0 + ((tid + 256 * (i >> 1)) >> 2), k0 + ((tid + 256 * (i >> 1)) & 3) * 8 + 4 * (i & 1), g.M, k_end);
        else ra[i] = load_mc(a_rs, g.lda, k0 + (idx / BM) * 4, m0 + idx % BM, k_end, g.M);
      }
      if constexpr (BP) {
#pragma unroll
        for (int i = 0; i < B_P8; ++i) {
          const int idx = tid + 256 * i, c = idx & 3, r = (idx >> 2) % BN, p = idx / (4 * BN);
          const int row = n0 + r, k = k0 + 8 * c;
          const unsigned off = (unsigned)(((long long)p * g.bplane + (long long)row * g.ldbp + k) * 2ll);
          rbp[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(bp_rs, (int)((row < g.N && k < g.ldbp && k < k_end) ? off : OOB), 0, 0));
        }
      } else {
#pragma unroll
        for (int i = 0; i < B_V4; ++i) {
          const int idx = tid + 256 * i;
          if (B_KC) rb[i] = load_kc(b_rs, g.ldb, n0 + ((tid + 256 * (i >> 1)) >> 2), k0 + ((tid + 256 * (i >> 1)) & 3) * 8 + 4 * (i & 1), g.N, k_end);
          else rb[i] = load_mc(b_rs, g.ldb, k0 + (idx / BN) * 4, n0 + idx % BN, k_end, g.N);
        }
      }
    };
    // m-contiguous operand: a thread holds 4 consecutive k (half a chunk) of one row
    auto put = [&](uint16_t* S, int plane_elems, int row, int q, const float4& v) {
      uint2 p0, p1, p2;
      g3_split4(v, p0, p1, p2);
      uint16_t* d = S + g3_soff(row, q >> 1) + 4 * (q & 1);
      if constexpr ((G3_ABLATE & 4) != 0) { G3_KEEP(p0.x); G3_KEEP(p0.y); G3_KEEP(p1.x); G3_KEEP(p1.y); G3_KEEP(p2.x); G3_KEEP(p2.y); return; }
      *reinterpret_cast<uint2*>(d) = p0;
      *reinterpret_cast<uint2*>(d + plane_elems) = p1;
      *reinterpret_cast<uint2*>(d + 2 * plane_elems) = p2;
    };
    auto put8 = [&](uint16_t* S, int plane_elems, int t, const float4& v0, const float4& v1) {
      uint4 p0, p1, p2;
      g3_split8(v0, v1, p0, p1, p2);
      const int T = tid + 256 * t;
      uint16_t* d = S + g3_soff(T >> 2, T & 3);
      if constexpr ((G3_ABLATE & 4) != 0) {
        G3_KEEP(p0.x); G3_KEEP(p0.y); G3_KEEP(p0.z); G3_KEEP(p0.w); G3_KEEP(p1.x); G3_KEEP(p1.y); G3_KEEP(p1.z); G3_KEEP(p1.w);
        G3_KEEP(p2.x); G3_KEEP(p2.y); G3_KEEP(p2.z); G3_KEEP(p2.w);
        return;
      }
      *reinterpret_cast<uint4*>(d) = p0;
      *reinterpret_cast<uint4*>(d + plane_elems) = p1;
      *reinterpret_cast<uint4*>(d + 2 * plane_elems) = p2;
    };
    float csum[A_V4];     // COLSUM (A is m-contiguous): this thread's partial column sums of A over the split's rows
#pragma unroll
    for (int i = 0; i < A_V4; ++i) csum[i] = 0.0f;
    auto stage = [&]() {
      if constexpr (A_KC) {
#pragma unroll
        for (int t = 0; t < A_V4 / 2; ++t) put8(As, PLANE_A, t, ra[2 * t], ra[2 * t + 1]);
      } else {
#pragma unroll
        for (int i = 0; i < A_V4; ++i) {
          const int idx = tid + 256 * i;
          put(As, PLANE_A, idx % BM, idx / BM, ra[i]);
          if (COLSUM) csum[i] += (ra[i].x + ra[i].y) + (ra[i].z + ra[i].w);
        }
      }
      if constexpr (BP) {
#pragma unroll
        for (int i = 0; i < B_P8; ++i) {
          const int idx = tid + 256 * i, c = idx & 3, r = (idx >> 2) % BN, p = idx / (4 * BN);
          if constexpr ((G3_ABLATE & 4) != 0) { G3_KEEP(rbp[i].x); G3_KEEP(rbp[i].y); G3_KEEP(rbp[i].z); G3_KEEP(rbp[i].w); continue; }
          *reinterpret_cast<uint4*>(Bs + p * PLANE_B + g3_soff(r, c)) = rbp[i];
        }
      } else if constexpr (B_KC) {
#pragma unroll
        for (int t = 0; t < B_V4 / 2; ++t) put8(Bs, PLANE_B, t, rb[2 * t], rb[2 * t + 1]);
      } else {
#pragma unroll
        for (int i = 0; i < B_V4; ++i) {
          const int idx = tid + 256 * i;
          put(Bs, PLANE_B, idx % BN, idx / BN, rb[i]);
        }
      }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int arow = wr * (BM / 2) + (lane & 31), bcol = wc * (BN / 2) + (lane & 31), kh = lane >> 5;
    fetch(k_begin);
    pf_bf16x8 a[TM][3], b[TN][3];
    auto frags = [&](int kk) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          a[i][p] = *reinterpret_cast<const pf_bf16x8*>(As + p * PLANE_A + g3_soff(arow + 32 * i, 2 * kk + kh));
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          b[j][p] = *reinterpret_cast<const pf_bf16x8*>(Bs + p * PLANE_B + g3_soff(bcol + 32 * j, 2 * kk + kh));
    };
    if constexpr ((G3_ABLATE & 16) != 0) frags(0);
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
      if constexpr ((G3_ABLATE & 32) == 0) __syncthreads();          // previous k-slice (or the previous tile's write-out) fully consumed
      stage();
      if constexpr ((G3_ABLATE & 32) == 0) __syncthreads();
      if constexpr ((G3_ABLATE & 8) == 0) { if (k0 + BK < k_end) fetch(k0 + BK); }
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) {
        if constexpr ((G3_ABLATE & 16) == 0) frags(kk);
        if constexpr ((G3_ABLATE & 1) != 0) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) G3_KEEP(a[i][p]);
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) G3_KEEP(b[j][p]);
          continue;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            // smallest terms first
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
          }
      }
    }
    __syncthreads();            // every wave is done with the operand tiles: LDS becomes the write-out staging area
    if constexpr (COLSUM) {
      // threads tid and tid + BM (+ ...) hold partial sums of the same column: combine through LDS in a fixed order
      float* cs = reinterpret_cast<float*>(smem);
      constexpr int PER = 256 / BM;            // threads per column in one chunk pass (m-contiguous: idx % BM = column)
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < A_V4; ++i) s += csum[i];
      cs[tid] = s;
      __syncthreads();
      if (bx == 0 && tid < BM && m0 + tid < g.M) {
        float tot = 0.0f;
#pragma unroll
        for (int u = 0; u < PER; ++u) tot += cs[tid + u * BM];
        C[(size_t)g.M * g.ldc + m0 + tid] = tot;
      }
      __syncthreads();
    }
    g3_writeout<BM, BN, EPI, COLSUM>(g, C, acc, m0, n0, smem);
  }
}

// ---- weight-gradient arrangement with a TRANSPOSED LDS image:  C[M,N] (split-K slab) = sum_k A[k][m] B[k][n], both operands
// row-major over the reduction index k (A = g[rows, out], B = x[rows, in]).  The tile is staged exactly as it lies in memory --
// [k][m] rows, so a thread's load is one dwordx4 of 4 consecutive m (a wave covers whole 512-byte row segments), split, and
// written as one 8-byte store per plane -- and the MFMA fragments, which need 8 consecutive k for one m, come out of
// ds_read_b64_tr_b16: a 16-lane group reads a 4-row x 16-column block and every lane receives one column (k-contiguous).
// Row stride BM + 32 elements: the four rows of a block fall into four different 16-bank groups (conflict-free reads).
typedef short pf_s16x4 __attribute__((ext_vector_type(4)));
// Diagnostic build only (-DG3_TN_STAMPS, tools/gemm_tn_stamps.py): lane 0 of wavefront 0 of every workgroup accumulates s_memtime deltas per
// phase of the weight-gradient loop -- [0] first barrier of a slice, [1] staging (wait for the operand loads, split, plane stores),
// [2] second barrier, [3] issue of the next slice's loads, [4] fragment reads + MFMAs, [5] prologue, [6] column sums + write-out, [7] slices
#if defined(G3_TN_STAMPS)
__device__ unsigned long long g3_tn_stamp_acc[8];
#define G3_STAMP(id) do { if (tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_[id] += t_ - tl_; tl_ = t_; } } while (0)
#else
#define G3_STAMP(id) ((void)0)
#endif
template <int BM, int BN>
__device__ __forceinline__ void pf_gemm3_tn_body(const GemmArgs& g, const int bid, const int nblk, unsigned char* smem) {
  // dwordx4 loads need 16-byte aligned rows (a 27-wide observation matrix has none): decided per operand, uniform per launch
  const bool VEC_A = ((g.lda & 3) == 0) && ((((size_t)g.A) & 15) == 0), VEC_B = ((g.ldb & 3) == 0) && ((((size_t)g.B) & 15) == 0);
  constexpr int BK = G3_BK;
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int RSA = BM + 32, RSB = BN + 32;                 // row strides (bf16 elements)
  constexpr int PLANE_A = BK * RSA, PLANE_B = BK * RSB;
  uint16_t* As = reinterpret_cast<uint16_t*>(smem);            // [3][BK][RSA]
  uint16_t* Bs = As + 3 * PLANE_A;                             // [3][BK][RSB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_x = (g.N + BN - 1) / BN, tiles_y = (g.M + BM - 1) / BM;
  const int nsplit = g.nsplit > 0 ? g.nsplit : 1;
  const int ntiles = tiles_x * tiles_y * nsplit;
  const int per_xcd = (ntiles + 7) >> 3, tmax = per_xcd << 3;
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, (unsigned)g.K * (unsigned)g.lda * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B), 0, (unsigned)g.K * (unsigned)g.ldb * 4u, 0x00020000);
  // 4 consecutive columns of row k; zero outside [k_end) x [ncols)
  auto load4 = [&](__amdgpu_buffer_rsrc_t rs, int ld, int k, int col, int k_end, int ncols, bool vec) -> float4 {
    const unsigned off = ((unsigned)k * (unsigned)ld + (unsigned)col) * 4u;
    const bool kv = k < k_end;
    if (vec)
      return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((kv && col < ncols) ? off : OOB), 0, 0));
    auto ld1 = [&](unsigned o, bool ok) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(ok ? o : OOB), 0, 0)); };
    return make_float4(ld1(off, kv && col < ncols), ld1(off + 4, kv && col + 1 < ncols), ld1(off + 8, kv && col + 2 < ncols), ld1(off + 12, kv && col + 3 < ncols));
  };
  constexpr int A_V4 = BM * BK / 4 / 256, B_V4 = BN * BK / 4 / 256;
  constexpr int CA = BM / 4, CB = BN / 4;                      // float4 chunks per tile row
  float4 ra[A_V4], rb[B_V4];
  // fragment addresses of this lane (T10): group g = lane >> 4 takes columns 16 (g & 1) .. +15 and k rows 8 (g >> 1) .. +7 of a
  // 16-deep k-step; inside the group lane 4 q + p supplies the address of row q, columns 4 p .. 4 p + 3
  const int grp = lane >> 4, lq = (lane >> 2) & 3, lp = lane & 3;
  const int frow = 8 * (grp >> 1) + lq, fcol = 16 * (grp & 1) + 4 * lp;

#if defined(G3_TN_STAMPS)
  unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl_ = __builtin_amdgcn_s_memtime();
#endif
  for (int t = bid; t < tmax; t += nblk) {
    const int lt = (t & 7) * per_xcd + (t >> 3);
    if (lt >= ntiles) continue;
    const int bx = lt % tiles_x, rest = lt / tiles_x, by = rest % tiles_y, bz = rest / tiles_y;
    const int m0 = by * BM, n0 = bx * BN;
    int k_begin = 0, k_end = g.K;
    float* C = g.C;
    if (g.k_per_split > 0) {
      k_begin = bz * g.k_per_split; k_end = min(g.K, k_begin + g.k_per_split);
      C += (size_t)bz * ((size_t)g.M * g.ldc + g.M);
    }
    auto fetch = [&](int k0) {
#pragma unroll
      for (int i = 0; i < A_V4; ++i) { const int idx = tid + 256 * i; ra[i] = load4(a_rs, g.lda, k0 + idx / CA, m0 + (idx % CA) * 4, k_end, g.M, VEC_A); }
#pragma unroll
      for (int i = 0; i < B_V4; ++i) { const int idx = tid + 256 * i; rb[i] = load4(b_rs, g.ldb, k0 + idx / CB, n0 + (idx % CB) * 4, k_end, g.N, VEC_B); }
    };
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);      // column sums of A (bias gradient): this thread's 4 columns, its rows
    auto stage = [&]() {
#pragma unroll
      for (int i = 0; i < A_V4; ++i) {
        const int idx = tid + 256 * i;
        uint2 p0, p1, p2;
        g3_split4(ra[i], p0, p1, p2);
        uint16_t* d = As + (idx / CA) * RSA + (idx % CA) * 4;
        if constexpr ((G3_ABLATE & 4) != 0) { G3_KEEP(p0.x); G3_KEEP(p0.y); G3_KEEP(p1.x); G3_KEEP(p1.y); G3_KEEP(p2.x); G3_KEEP(p2.y); }
        else { *reinterpret_cast<uint2*>(d) = p0; *reinterpret_cast<uint2*>(d + PLANE_A) = p1; *reinterpret_cast<uint2*>(d + 2 * PLANE_A) = p2; }
        csum.x += ra[i].x; csum.y += ra[i].y; csum.z += ra[i].z; csum.w += ra[i].w;
      }
#pragma unroll
      for (int i = 0; i < B_V4; ++i) {
        const int idx = tid + 256 * i;
        uint2 p0, p1, p2;
        g3_split4(rb[i], p0, p1, p2);
        uint16_t* d = Bs + (idx / CB) * RSB + (idx % CB) * 4;
        if constexpr ((G3_ABLATE & 4) != 0) { G3_KEEP(p0.x); G3_KEEP(p0.y); G3_KEEP(p1.x); G3_KEEP(p1.y); G3_KEEP(p2.x); G3_KEEP(p2.y); }
        else { *reinterpret_cast<uint2*>(d) = p0; *reinterpret_cast<uint2*>(d + PLANE_B) = p1; *reinterpret_cast<uint2*>(d + 2 * PLANE_B) = p2; }
      }
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    auto frag = [&](const uint16_t* S, int rs, int plane, int kk, int col0) -> pf_bf16x8 {
      const uint16_t* a0 = S + plane + (16 * kk + frow) * rs + col0 + fcol;
      const pf_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pf_s16x4 __attribute__((address_space(3)))*)(a0));
      const pf_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pf_s16x4 __attribute__((address_space(3)))*)(a0 + 4 * rs));
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      return __builtin_bit_cast(pf_bf16x8, v);
    };
    fetch(k_begin);
    G3_STAMP(5);
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
      __syncthreads();
      G3_STAMP(0);
      stage();
#if defined(G3_TN_STAMPS)
      __builtin_amdgcn_s_waitcnt(0xC07F);      // (lgkmcnt 0: the plane stores have left the wave)
#endif
      G3_STAMP(1);
      __syncthreads();
      G3_STAMP(2);
      if (k0 + BK < k_end) fetch(k0 + BK);
      G3_STAMP(3);
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) {
        pf_bf16x8 a[TM][3], b[TN][3];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int p = 0; p < 3; ++p) a[i][p] = frag(As, RSA, p * PLANE_A, kk, wr * (BM / 2) + 32 * i);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int p = 0; p < 3; ++p) b[j][p] = frag(Bs, RSB, p * PLANE_B, kk, wc * (BN / 2) + 32 * j);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if constexpr ((G3_ABLATE & 1) != 0) {
#pragma unroll
              for (int p = 0; p < 3; ++p) { G3_KEEP(a[i][p]); G3_KEEP(b[j][p]); }
              continue;
            }
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
          }
      }
#if defined(G3_TN_STAMPS)
      if (tid == 0) st_[7] += 1;
#endif
      G3_STAMP(4);
    }
    __syncthreads();
    {
      // column sums: thread tid holds columns 4 (tid % CA) .. +3 for the rows tid / CA, + 256 / CA, ...: combine the 256 / CA
      // row groups through LDS in a fixed order
      float4* cs = reinterpret_cast<float4*>(smem);
      cs[tid] = csum;
      __syncthreads();
      if (bx == 0 && tid < CA) {
        float4 tot = cs[tid];
#pragma unroll
        for (int u = 1; u < 256 / CA; ++u) { const float4 v = cs[tid + u * CA]; tot.x += v.x; tot.y += v.y; tot.z += v.z; tot.w += v.w; }
        const int col = m0 + 4 * tid;
        float* o = C + (size_t)g.M * g.ldc;
        if (col < g.M) o[col] = tot.x;
        if (col + 1 < g.M) o[col + 1] = tot.y;
        if (col + 2 < g.M) o[col + 2] = tot.z;
        if (col + 3 < g.M) o[col + 3] = tot.w;
      }
      __syncthreads();
    }
    g3_writeout<BM, BN, GEMM_EPI_NONE, true>(g, C, acc, m0, n0, smem);
    G3_STAMP(6);
  }
#if defined(G3_TN_STAMPS)
  if (tid == 0) for (int u = 0; u < 8; ++u) atomicAdd(&g3_tn_stamp_acc[u], st_[u]);
#endif
}
template <int BM, int BN>
constexpr size_t pf_gemm3_tn_lds() {
  size_t op = (size_t)3 * G3_BK * ((BM + 32) + (BN + 32)) * 2, st = (size_t)4 * 32 * (BN / 2 + 4) * 4;
  return op > st ? op : st;
}
template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void pf_gemm3_tn_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[pf_gemm3_tn_lds<BM, BN>()];
  pf_gemm3_tn_body<BM, BN>(g, (int)blockIdx.x, (int)gridDim.x, smem);
}
// the actor's and the critic's weight gradient of one layer in one launch: the pair fills the chip with half the splits each
// (half the slab traffic, half the slab-reduction work)
template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void pf_gemm3_tn_kernel2(GemmArgs g0, GemmArgs g1, int nb0) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[pf_gemm3_tn_lds<BM, BN>()];
  if ((int)blockIdx.x < nb0) pf_gemm3_tn_body<BM, BN>(g0, (int)blockIdx.x, nb0, smem);
  else pf_gemm3_tn_body<BM, BN>(g1, (int)blockIdx.x - nb0, (int)gridDim.x - nb0, smem);
}

template <int BM, int BN, bool BP = false>
constexpr size_t pf_gemm3_lds() {
  // operand planes (3 x (BM + BN) x RS bf16); the write-out strips
  // (4 waves x 32 x (BN/2 + 4) floats) and the column-sum scratch (256 floats) reuse the same bytes
  size_t op = (size_t)3 * (BM + BN) * G3_RS * 2, st = (size_t)4 * 32 * (BN / 2 + 4) * 4;
  return op > st ? op : st;
}

template <int BM, int BN, bool A_KC, bool B_KC, int EPI, bool COLSUM, bool VEC, bool BP = false>
__global__ __launch_bounds__(256, (BM * BN <= 128 * 64 ? 4 : 2)) void pf_gemm3_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[pf_gemm3_lds<BM, BN, BP>()];
  pf_gemm3_body<BM, BN, A_KC, B_KC, EPI, COLSUM, VEC, BP>(g, (int)blockIdx.x, (int)gridDim.x, smem);
}
// two problems of the same arrangement in one launch (the actor's and the critic's layer i)
template <int BM, int BN, bool A_KC, bool B_KC, int EPI, bool COLSUM, bool VEC0, bool VEC1, bool BP = false>
__global__ __launch_bounds__(256, (BM * BN <= 128 * 64 ? 4 : 2)) void pf_gemm3_kernel2(GemmArgs g0, GemmArgs g1, int nb0) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[pf_gemm3_lds<BM, BN, BP>()];
  if ((int)blockIdx.x < nb0) pf_gemm3_body<BM, BN, A_KC, B_KC, EPI, COLSUM, VEC0, BP>(g0, (int)blockIdx.x, nb0, smem);
  else pf_gemm3_body<BM, BN, A_KC, B_KC, EPI, COLSUM, VEC1, BP>(g1, (int)blockIdx.x - nb0, (int)gridDim.x - nb0, smem);
}

// Tile choice from measurements on MI355X (tools/gemm_bench.py, profiles/r2_gemm_shapes.txt): 128 x 128 only pays when
// there are >= 1024 of them (4096^3: 189 TFLOP/s); the PPO shapes (M = 24 576, N = 256 / 512) run best on 128 x 64 -- twice the
// blocks, so one block's staging hides behind another's MFMAs -- and narrow or short problems on 64 x 64.
// (0: 128x128, 1: 128x64, 2: 64x128, 3: 64x64)
static inline int pf_gemm3_pick_tile(int M, int N, int splits) {
  const long S = splits > 0 ? splits : 1;
  auto tiles = [&](int bm, int bn) { return (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn) * S; };
  if (tiles(128, 128) >= 1024) return 0;
  if (N >= 256 && tiles(128, 64) >= 512) return 1;
  return 3;
}
template <int BM, int BN>
static inline int pf_gemm3_blocks(GemmArgs& g, int splits) {
  g.nsplit = splits > 0 ? splits : 1;
  int ntiles = ((g.N + BN - 1) / BN) * ((g.M + BM - 1) / BM) * g.nsplit;
  return (ntiles + 7) / 8 * 8;
}
// (with a pre-split B only A decides whether dwordx4 loads are possible)
template <bool A_KC>
static inline bool pf_gemm3_vec_a(const GemmArgs& g) {
  return !A_KC || (((g.K & 3) == 0) && ((((size_t)g.A) & 15) == 0) && ((g.lda & 3) == 0));
}
template <int BM, int BN, bool A_KC, bool B_KC, int EPI, bool COLSUM>
static inline void pf_gemm3_launch_t(const GemmArgs& g_in, int splits, hipStream_t s) {
  GemmArgs g = g_in;
  const int blocks = pf_gemm3_blocks<BM, BN>(g, splits);
  if constexpr (!A_KC && !B_KC && COLSUM && EPI == GEMM_EPI_NONE) {      // weight gradient: transposed-image kernel
    hipLaunchKernelGGL((pf_gemm3_tn_kernel<BM, BN>), dim3(blocks), dim3(256), 0, s, g);
    return;
  }
  if constexpr (B_KC && !COLSUM) {
    if (g.Bp) {
      if (pf_gemm3_vec_a<A_KC>(g)) hipLaunchKernelGGL((pf_gemm3_kernel<BM, BN, A_KC, true, EPI, false, true, true>), dim3(blocks), dim3(256), 0, s, g);
      else hipLaunchKernelGGL((pf_gemm3_kernel<BM, BN, A_KC, true, EPI, false, false, true>), dim3(blocks), dim3(256), 0, s, g);
      return;
    }
  }
  if (pf_gemm_vec<A_KC, B_KC>(g)) hipLaunchKernelGGL((pf_gemm3_kernel<BM, BN, A_KC, B_KC, EPI, COLSUM, true>), dim3(blocks), dim3(256), 0, s, g);
  else hipLaunchKernelGGL((pf_gemm3_kernel<BM, BN, A_KC, B_KC, EPI, COLSUM, (A_KC || B_KC) ? false : true>), dim3(blocks), dim3(256), 0, s, g);
}
template <bool A_KC, bool B_KC, int EPI, bool COLSUM = false>
static inline void pf_gemm3_launch(const GemmArgs& g, int splits, hipStream_t s, int tile = -1) {
  if (tile < 0) tile = pf_gemm3_pick_tile(g.M, g.N, splits);
  switch (tile) {
    case 0: pf_gemm3_launch_t<128, 128, A_KC, B_KC, EPI, COLSUM>(g, splits, s); break;
    case 1: pf_gemm3_launch_t<128, 64, A_KC, B_KC, EPI, COLSUM>(g, splits, s); break;
    case 2: pf_gemm3_launch_t<64, 128, A_KC, B_KC, EPI, COLSUM>(g, splits, s); break;
    default: pf_gemm3_launch_t<64, 64, A_KC, B_KC, EPI, COLSUM>(g, splits, s); break;
  }
}
// the pair in one launch when both problems pick the same tile, else one launch each
template <int BM, int BN, bool A_KC, bool B_KC, int EPI, bool COLSUM>
static inline void pf_gemm3_launch2_t(const GemmArgs& g0_in, int splits0, const GemmArgs& g1_in, int splits1, hipStream_t s) {
  GemmArgs g0 = g0_in, g1 = g1_in;
  const int nb0 = pf_gemm3_blocks<BM, BN>(g0, splits0), nb1 = pf_gemm3_blocks<BM, BN>(g1, splits1);
  const dim3 grid(nb0 + nb1), block(256);
  if constexpr (B_KC && !COLSUM) {
    if (g0.Bp && g1.Bp) {
      const bool a0 = pf_gemm3_vec_a<A_KC>(g0), a1 = pf_gemm3_vec_a<A_KC>(g1);
      if (a0 && a1) hipLaunchKernelGGL((pf_gemm3_kernel2<BM, BN, A_KC, true, EPI, false, true, true, true>), grid, block, 0, s, g0, g1, nb0);
      else if (!a0 && a1) hipLaunchKernelGGL((pf_gemm3_kernel2<BM, BN, A_KC, true, EPI, false, false, true, true>), grid, block, 0, s, g0, g1, nb0);
      else if (a0 && !a1) hipLaunchKernelGGL((pf_gemm3_kernel2<BM, BN, A_KC, true, EPI, false, true, false, true>), grid, block, 0, s, g0, g1, nb0);
      else hipLaunchKernelGGL((pf_gemm3_kernel2<BM, BN, A_KC, true, EPI, false, false, false, true>), grid, block, 0, s, g0, g1, nb0);
      return;
    }
  }
  const bool v0 = pf_gemm_vec<A_KC, B_KC>(g0), v1 = pf_gemm_vec<A_KC, B_KC>(g1);
  constexpr bool NV = (A_KC || B_KC) ? false : true;      // the "not vectorisable" instance (no k-contiguous operand: always vec)
  if (v0 && v1) hipLaunchKernelGGL((pf_gemm3_kernel2<BM, BN, A_KC, B_KC, EPI, COLSUM, true, true>), grid, block, 0, s, g0, g1, nb0);
  else if (!v0 && v1) hipLaunchKernelGGL((pf_gemm3_kernel2<BM, BN, A_KC, B_KC, EPI, COLSUM, NV, true>), grid, block, 0, s, g0, g1, nb0);
  else if (v0 && !v1) hipLaunchKernelGGL((pf_gemm3_kernel2<BM, BN, A_KC, B_KC, EPI, COLSUM, true, NV>), grid, block, 0, s, g0, g1, nb0);
  else hipLaunchKernelGGL((pf_gemm3_kernel2<BM, BN, A_KC, B_KC, EPI, COLSUM, NV, NV>), grid, block, 0, s, g0, g1, nb0);
}
template <bool A_KC, bool B_KC, int EPI, bool COLSUM = false>
static inline void pf_gemm3_launch2(const GemmArgs& g0, int splits0, const GemmArgs& g1, int splits1, hipStream_t s) {
  // one tile shape for the pair: judged on the two problems together (their blocks share the grid)
  const int t0 = pf_gemm3_pick_tile(g0.M > g1.M ? g0.M : g1.M, g0.N > g1.N ? g0.N : g1.N, splits0 > splits1 ? splits0 : splits1);
  switch (t0) {
    case 0: pf_gemm3_launch2_t<128, 128, A_KC, B_KC, EPI, COLSUM>(g0, splits0, g1, splits1, s); break;
    case 1: pf_gemm3_launch2_t<128, 64, A_KC, B_KC, EPI, COLSUM>(g0, splits0, g1, splits1, s); break;
    case 2: pf_gemm3_launch2_t<64, 128, A_KC, B_KC, EPI, COLSUM>(g0, splits0, g1, splits1, s); break;
    default: pf_gemm3_launch2_t<64, 64, A_KC, B_KC, EPI, COLSUM>(g0, splits0, g1, splits1, s); break;
  }
}

// Split-K for the weight gradient (output out x in, reduction over the minibatch rows): tile by shape, then enough splits
// for ~2 resident blocks per CU.  tile: 0 128x128, 1 128x64, 2 64x128, 3 64x64.
static inline void pf_gemm3_pick_split(int M, int N, int K, int max_splits, int* tile_out, int* splits_out) {
  int t = N <= 64 ? (M <= 64 ? 3 : 1) : (M <= 128 ? 3 : 0);
  const int bm = (t == 0 || t == 1) ? 128 : 64, bn = (t == 0 || t == 2) ? 128 : 64;
  const int tiles = ((M + bm - 1) / bm) * ((N + bn - 1) / bn);
  int S = (512 + tiles - 1) / tiles;
  if (S > max_splits) S = max_splits;
  while (S > 1 && (K + S - 1) / S < 128) --S;
  *tile_out = t; *splits_out = S < 1 ? 1 : S;
}

// ---- weights -> bf16 planes, once per optimiser step.  dst[p][r][c] (row stride ldp, zero padded) = p-th split term of
// src[r][c] (transposed: of src[c][r]); one launch covers every layer of both networks.
struct SplitJob { const float* src; unsigned short* dst; int rows, cols, ld_src, ldp, transposed; long long plane; int blocks; };
struct SplitJobs { SplitJob j[4 * 4]; int n; };
__global__ __launch_bounds__(256) void k_split_weights(SplitJobs js) {
  int b = (int)blockIdx.x;
  for (int i = 0; i < js.n; ++i) {
    const SplitJob& J = js.j[i];
    if (b < J.blocks) {
      const long long e = (long long)b * 256 + threadIdx.x;
      if (e >= (long long)J.rows * J.ldp) return;
      const int r = (int)(e / J.ldp), c = (int)(e - (long long)r * J.ldp);
      float v = 0.0f;
      if (c < J.cols) v = J.transposed ? J.src[(size_t)c * J.ld_src + r] : J.src[(size_t)r * J.ld_src + c];
      const unsigned u = __builtin_bit_cast(unsigned, v);
      const float r1 = v - __builtin_bit_cast(float, u & 0xFFFF0000u);
      const unsigned u1 = __builtin_bit_cast(unsigned, r1);
      const float r2 = r1 - __builtin_bit_cast(float, u1 & 0xFFFF0000u);
      J.dst[e] = (unsigned short)(u >> 16);
      J.dst[J.plane + e] = (unsigned short)(u1 >> 16);
      J.dst[2 * J.plane + e] = (unsigned short)(__builtin_bit_cast(unsigned, r2) >> 16);
      return;
    }
    b -= J.blocks;
  }
}

// weight-gradient pair (same tile shape for both problems): tile 0 128x128, 1 128x64, 2 64x128, 3 64x64
static inline void pf_gemm3_tn_launch2(const GemmArgs& g0_in, int S0, const GemmArgs& g1_in, int S1, int tile, hipStream_t s) {
  GemmArgs g0 = g0_in, g1 = g1_in;
  auto go = [&](auto bm, auto bn) {
    constexpr int BM = decltype(bm)::value, BN = decltype(bn)::value;
    const int nb0 = pf_gemm3_blocks<BM, BN>(g0, S0), nb1 = pf_gemm3_blocks<BM, BN>(g1, S1);
    hipLaunchKernelGGL((pf_gemm3_tn_kernel2<BM, BN>), dim3(nb0 + nb1), dim3(256), 0, s, g0, g1, nb0);
  };
  using I128 = std::integral_constant<int, 128>; using I64 = std::integral_constant<int, 64>;
  switch (tile) {
    case 0: go(I128{}, I128{}); break;
    case 1: go(I128{}, I64{}); break;
    case 2: go(I64{}, I128{}); break;
    default: go(I64{}, I64{}); break;
  }
}
// split count for a problem that shares its launch with `share` problems of the same shape
static inline void pf_gemm3_pick_split_shared(int M, int N, int K, int max_splits, int share, int* tile_out, int* splits_out) {
  int t, S;
  pf_gemm3_pick_split(M, N, K, max_splits, &t, &S);
  S = (S + share - 1) / share;
  *tile_out = t; *splits_out = S < 1 ? 1 : S;
}

// pf_gemm3f.h — the "bf16x6" dense layer (pf_gemm3.h) with LDS-DMA staging: C[M,N] = A[M,K] * W[N,K]^T (+ epilogue) for the
// two k-contiguous arrangements of an nn.Linear layer (forward x W^T + b, ELU; data gradient g W * ELU'(h) with the
// pre-transposed weight planes), gfx950.
//
// What changed against pf_gemm3_body (profiles/r2_gemm3_ablation.txt: loads + split + ds_write + barriers cost as much as
// the MFMAs and did not overlap them):
//  * nothing is staged through registers.  The fp32 activation tile (128 rows x 32 k) and the three bf16 weight planes
//    (BN rows x 32 k each) go global -> LDS by `buffer_load_dwordx4 ... lds` (1 KB per wave instruction, out-of-range -> 0),
//    two LDS stages, so the DMA of k-slice t + 1 runs under the MFMAs of slice t with ONE barrier per slice;
//  * the workgroup's four wavefronts are stacked 4 x 1: a wave owns 32 rows x all BN columns.  The exact three-way split of
//    the activations (a = a1 + a2 + a3, bf16 each) is done on the 8 values a lane reads for its A fragment -- every
//    activation element is split exactly once per block, by the one wave that owns its row, 44 VALU instructions per
//    6 * BN/32 MFMAs, issued in the MFMA shadow -- and there is no ds_write in the loop at all;
//  * LDS images are lane-linear (an LDS-DMA writes base + 16 * lane): the bank swizzles sit on the per-lane SOURCE address
//    and on the fragment read (cdna_hip_programming.md 5.4 rule 21).  A rows are 128 B: chunk c of row r lives at position
//    c ^ ((r >> 1) & 7), so the 16 lanes of a ds_read_b128 phase (16 different rows, same chunk) cover all 64 banks;
//    B rows are 64 B with pf_gemm3.h's g3_soff swizzle.
// HBM formats and the numerics are unchanged (same products, same accumulation order within a k-step).
#pragma once
#include "pf_gemm3.h"
#include <type_traits>

// An LDS-DMA issued through the builtin is counted by hipcc, which then drains it (s_waitcnt vmcnt(0)) ahead of the first
// ds_read that follows -- the MFMAs of a slice would wait for the NEXT slice's DMA.  The loads therefore go out as inline
// asm (hipcc does not count them; the loop waits for them itself, once per slice), with the buffer descriptor built by hand.
// M0 (the LDS destination base of the DMA) is compiler-reserved: saved and restored inside the statement.
typedef unsigned g3f_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ g3f_u4 g3f_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  g3f_u4 r = {(unsigned)a, (unsigned)(a >> 32) & 0xFFFFu, bytes, 0x00020000u};      // raw buffer (stride 0), out-of-range reads return 0
  return r;
}
__device__ __forceinline__ void g3f_dma16(const g3f_u4& rsrc, unsigned voff, unsigned lds_byte) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_byte), "s"(rsrc) : "memory");
}
// Diagnostic builds only (tools/gemm_dma_ablate.hip): -DG3F_ABLATE=mask removes one ingredient at a time (results are wrong by
// construction).  1 MFMAs, 2 activation split, 4 in-loop DMA, 8 B fragment reads, 16 A fragment reads, 32 write-out, 64 the write-out's global
// stores only, 128 the write-out's LDS transpose only.
#ifndef G3F_ABLATE
#define G3F_ABLATE 0
#endif
// Diagnostic builds only: 1 = the write-out also emits the result as three bf16 planes (GemmArgs::Cp) -- the producer side of
// "activations as planes for the weight gradients", measured in profiles/r4_plane_copy_ablation.txt.  The library builds with 0.
#ifndef G3F_EMIT_PLANES
#define G3F_EMIT_PLANES 0
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define G3F_OPAQUE(...) asm volatile("" : __VA_ARGS__)      // register constraints of the ablation sinks only parse for the device
#else
#define G3F_OPAQUE(...)
#endif
constexpr int G3F_BM = 128;
// Stage geometry for a k-slice of BK (32 or 16): A rows are BK fp32 (128 / 64 B), B rows BK bf16 per plane (64 / 32 B).  A 1 KB
// DMA piece is 8 / 16 rows of A or 16 / 32 rows of one B plane.  16-byte chunk c of row r sits at chunk position c ^ swz(r),
// swz(r) = (r / rows-per-256-B) mod chunks-per-row: the 16 lanes of a ds_read_b128 phase (16 different rows, same chunk)
// then cover all 64 banks.
template <int BN, int BK> struct G3fGeom {
  static constexpr int A_ROWB = BK * 4, A_CH = BK / 4, A_RPP = 1024 / A_ROWB, A_PW = 32 / A_RPP;      // bytes / chunks per row, rows per piece, pieces per wave
  static constexpr int B_ROWB = BK * 2, B_CH = BK / 8, B_RPP = 1024 / B_ROWB, B_PPP = BN / B_RPP, NPB = 3 * B_PPP;      // ... pieces per plane, B pieces
  static constexpr int A_BYTES = G3F_BM * A_ROWB, PLANE_B = BN * B_ROWB, STAGE = A_BYTES + 3 * PLANE_B;
  static constexpr int SW = BN < 64 ? BN : 64;                 // write-out strip width (columns): 4 waves x 32 x (SW + 4) floats
  static constexpr int STRIPS = 4 * 32 * (SW + 4) * 4;
  static constexpr int LDS = 2 * STAGE > STRIPS ? 2 * STAGE : STRIPS;
  template <int NSTG> static constexpr int lds() { return NSTG * STAGE > STRIPS ? NSTG * STAGE : STRIPS; }
  static constexpr int B_PW = (NPB + 3) / 4, NP = A_PW + B_PW;      // B pieces per wave (BN = 32: six pieces, the last wave has none)
  __device__ static __forceinline__ int a_swz(int r) { return (r / (256 / A_ROWB)) & (A_CH - 1); }
  __device__ static __forceinline__ int b_swz(int r) { return (r / (256 / B_ROWB)) & (B_CH - 1); }
};

// wait until at most N of this wave's vector-memory operations (here: DMA pieces) are outstanding
template <int N> __device__ __forceinline__ void g3f_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NSTG LDS stages: the DMA of k-slice t + NSTG - 1 is issued under the MFMAs of slice t.  Two stages (two workgroups per CU) for the
// update's launches; four for the rollout's 4096-row launches, where there is ONE workgroup per CU and nothing else hides the
// ~1 us a slice takes to land.
template <int BN, int BK, int EPI, int NSTG = 2>
__device__ __forceinline__ void pf_gemm3f_body(const GemmArgs& g, const int bid, const int nblk, unsigned char* smem) {
  using Ge = G3fGeom<BN, BK>;
  constexpr int BM = G3F_BM, TN = BN / 32;
  constexpr int A_BYTES = Ge::A_BYTES, PLANE_B = Ge::PLANE_B, STAGE = Ge::STAGE, NP = Ge::NP, A_PW = Ge::A_PW, B_PW = Ge::B_PW;
  constexpr unsigned OOB = 0x80000000u;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_x = (g.N + BN - 1) / BN, tiles_y = (g.M + BM - 1) / BM;
  const int ntiles = tiles_x * tiles_y;
  const int per_xcd = (ntiles + 7) >> 3, tmax = per_xcd << 3;
  const g3f_u4 a_rs = g3f_rsrc(g.A, (unsigned)g.M * (unsigned)g.lda * 4u), b_rs = g3f_rsrc(g.Bp, (unsigned)(3ll * g.bplane * 2ll));
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;      // LDS byte address of the stages

  for (int t = bid; t < tmax; t += nblk) {
    const int lt = (t & 7) * per_xcd + (t >> 3);      // XCD (t & 7) owns a contiguous range of logical tiles (shared A rows hit one L2)
    if (lt >= ntiles) continue;
    const int bx = lt % tiles_x, by = lt / tiles_x;
    const int m0 = by * BM, n0 = bx * BN;
    // per-lane source offsets of this wave's DMA pieces (k0 = 0); the A pieces of a wave are its own 32 rows
    unsigned a_off[A_PW], b_off[B_PW];
    int a_k[A_PW], b_k[B_PW];
#pragma unroll
    for (int i = 0; i < A_PW; ++i) {
      const int rl = 32 * wave + Ge::A_RPP * i + lane / Ge::A_CH, c = (lane % Ge::A_CH) ^ Ge::a_swz(rl), row = m0 + rl;
      a_k[i] = 4 * c;
      a_off[i] = row < g.M ? ((unsigned)row * (unsigned)g.lda + 4u * c) * 4u : OOB;
    }
#pragma unroll
    for (int i = 0; i < B_PW; ++i) {
      const int q = wave * B_PW + i, p = q / Ge::B_PPP, rl = Ge::B_RPP * (q % Ge::B_PPP) + lane / Ge::B_CH, c = (lane % Ge::B_CH) ^ Ge::b_swz(rl), row = n0 + rl;
      b_k[i] = 8 * c;
      b_off[i] = row < g.N ? (unsigned)(((long long)p * g.bplane + (long long)row * g.ldbp + 8 * c) * 2ll) : OOB;
    }
    // DMA piece i of this wave for the slice at k0 into stage `buf`: first its A rows, then its share of the B planes
    auto piece = [&](int i, int k0, int buf, bool full) {
      const unsigned st = lds0 + buf * STAGE;
      if (i < A_PW) {
        unsigned off = a_off[i] + 4u * (unsigned)k0;
        if (!full) off = (k0 + a_k[i] < g.K) ? off : OOB;
        g3f_dma16(a_rs, off, st + (A_PW * wave + i) * 1024);
      } else {
        if (Ge::NPB % 4 != 0 && wave * B_PW + i - A_PW >= Ge::NPB) return;      // (uniform: `wave` is scalar)
        unsigned off = b_off[i - A_PW] + 2u * (unsigned)k0;
        if (!full) off = (k0 + b_k[i - A_PW] < g.ldbp) ? off : OOB;
        g3f_dma16(b_rs, off, st + A_BYTES + (wave * B_PW + i - A_PW) * 1024);
      }
    };
    auto is_full = [&](int k0) { return k0 + BK <= g.K && k0 + BK <= g.ldbp; };      // uniform: only a ragged last slice checks k per lane

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    const int rl = 32 * wave + (lane & 31), kh = lane >> 5, aswz = Ge::a_swz(rl);
    const int nt = (g.K + BK - 1) / BK;
    // one k-slice of MFMAs out of stage it & 1; with NEXT, the DMA pieces of slice it + 1 go out between the MFMA groups
    // (a burst of them at the head of the slice keeps every wave of the CU in the load queue while the matrix pipes idle)
    auto slice = [&](int it, auto next_tag) {
      constexpr bool NEXT = decltype(next_tag)::value && (G3F_ABLATE & 4) == 0;
      const int k0n = (it + NSTG - 1) * BK, bufn = (it + NSTG - 1) % NSTG;
      const bool fulln = is_full(k0n);
      const unsigned char* st = smem + ((G3F_ABLATE & 4) ? 0 : (it % NSTG)) * STAGE;
      const unsigned char* Bs = st + A_BYTES;
      constexpr int G = (BK / 16) * TN;      // MFMA groups of a slice
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) {
        const int c0 = 4 * kk + 2 * kh;
        float4 v0, v1;
        if constexpr ((G3F_ABLATE & 16) == 0) {
          v0 = *reinterpret_cast<const float4*>(st + rl * Ge::A_ROWB + ((c0 ^ aswz) << 4));
          v1 = *reinterpret_cast<const float4*>(st + rl * Ge::A_ROWB + (((c0 + 1) ^ aswz) << 4));
        } else { v0 = make_float4(1.0f + lane, 2.0f, 3.0f + kk, 4.0f); v1 = v0; G3F_OPAQUE("+v"(v0.x), "+v"(v0.y), "+v"(v1.x), "+v"(v1.w)); }
        uint4 q0, q1, q2;
        if constexpr ((G3F_ABLATE & 2) == 0) g3_split8(v0, v1, q0, q1, q2);
        else { q0 = __builtin_bit_cast(uint4, v0); q1 = __builtin_bit_cast(uint4, v1); q2 = q0; }
        const pf_bf16x8 a0 = __builtin_bit_cast(pf_bf16x8, q0), a1 = __builtin_bit_cast(pf_bf16x8, q1), a2 = __builtin_bit_cast(pf_bf16x8, q2);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = 32 * j + (lane & 31);
          const unsigned char* bp = Bs + n * Ge::B_ROWB + (((2 * kk + kh) ^ Ge::b_swz(n)) << 4);
          pf_bf16x8 b0, b1, b2;
          if constexpr ((G3F_ABLATE & 8) == 0) {
            b0 = *reinterpret_cast<const pf_bf16x8*>(bp);
            b1 = *reinterpret_cast<const pf_bf16x8*>(bp + PLANE_B);
            b2 = *reinterpret_cast<const pf_bf16x8*>(bp + 2 * PLANE_B);
          } else { b0 = a0; b1 = a1; b2 = a2; G3F_OPAQUE("+v"(b0), "+v"(b1), "+v"(b2)); }      // opaque: no CSE across j
          if constexpr ((G3F_ABLATE & 1) != 0) { G3_KEEP(a0); G3_KEEP(a1); G3_KEEP(a2); G3_KEEP(b0); G3_KEEP(b1); G3_KEEP(b2); }
          else {
            // smallest terms first
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[j], 0, 0, 0);
          }
          if constexpr (NEXT) {
            const int grp = kk * TN + j;
#pragma unroll
            for (int i = 0; i < NP; ++i)
              if (i * G / NP == grp) piece(i, k0n, bufn, fulln);
          }
        }
      }
    };
    static_assert((NSTG - 2) * NP <= 63, "vmcnt is a 6-bit counter");
    static_assert(NSTG == 2 || Ge::NPB % 4 == 0, "counted waits need the same number of pieces in every wave");
#pragma unroll
    for (int sl = 0; sl < NSTG - 1; ++sl)
      if (sl < nt) {
#pragma unroll
        for (int i = 0; i < NP; ++i) piece(i, sl * BK, sl, is_full(sl * BK));
      }
    int it = 0;
    for (; it + NSTG - 1 < nt; ++it) {
      g3f_wait_vm<(NSTG - 2) * NP>();                       // this wave's pieces of slice `it` have landed (NSTG - 2 newer slices may be in flight)
      __syncthreads();                                      // ... everybody's have; and everybody is done reading the stage that is refilled now
      slice(it, std::true_type{});
    }
    for (; it < nt; ++it) {                                 // the last NSTG - 1 slices: nothing left to issue, fewer and fewer in flight
      const int newer = nt - 1 - it;
      if (NSTG > 3 && newer >= 2) g3f_wait_vm<(NSTG > 3 ? 2 : 0) * NP>();
      else if (NSTG > 2 && newer == 1) g3f_wait_vm<(NSTG > 2 ? 1 : 0) * NP>();
      else g3f_wait_vm<0>();
      __syncthreads();
      slice(it, std::false_type{});
    }
    __syncthreads();            // every wave is done with the operand stages: LDS becomes the write-out staging area
    if constexpr ((G3F_ABLATE & 32) != 0) {
#pragma unroll
      for (int j = 0; j < TN; ++j) G3F_OPAQUE(: "v"(acc[j]));      // all accumulators stay live (no dead MFMA chains)
      continue;
    }
    // Write-out: a wave transposes its own 32-row strip, SW columns at a time, through a private LDS region and stores whole
    // rows as dwordx4.  C/D layout of a 32x32 tile: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
    // Every global LOAD of the epilogue (bias, the ELU' operand) is issued before the first STORE that could delay it: vmcnt
    // retires in order, so a load issued after a store cannot be consumed before that store has been acknowledged by memory
    // -- a loop of "load h, multiply, store" pays one full store round trip per pass.  The ELU' operand of column group s + 1
    // is fetched before the stores of group s go out (its registers are the accumulators group s has just vacated).
    {
      constexpr int SW = Ge::SW, NS = BN / SW, JS = SW / 32;          // strip width; column groups; 32x32 tiles per group
      constexpr int CS = SW + 4, Q = SW / 4, P = Q / 2, RP = 64 / Q;  // float4 per strip row; passes; rows per pass
      float* Cs = reinterpret_cast<float*>(smem) + wave * (32 * CS);
      const int c4 = lane % Q, r0 = lane / Q;
      const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, (unsigned)g.M * (unsigned)g.ldc * 4u, 0x00020000);
      const __amdgpu_buffer_rsrc_t h_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(EPI == GEMM_EPI_MUL_DELU ? g.H : g.C), 0, (unsigned)g.M * (unsigned)(EPI == GEMM_EPI_MUL_DELU ? g.ldh : g.ldc) * 4u, 0x00020000);
      float4 b4[NS], hv[2][P];
      auto load_h = [&](int sgrp, float4 (&dst)[P]) {
        const int col = n0 + SW * sgrp + 4 * c4;
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const int row = m0 + 32 * wave + r0 + RP * p;
          dst[p] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(h_rs, (int)((row < g.M && col < g.N) ? ((unsigned)row * (unsigned)g.ldh + (unsigned)col) * 4u : OOB), 0, 0));
        }
      };
      if (EPI == GEMM_EPI_BIAS || EPI == GEMM_EPI_BIAS_ELU) {
#pragma unroll
        for (int sg = 0; sg < NS; ++sg) { const int col = n0 + SW * sg + 4 * c4; b4[sg] = col < g.N ? *reinterpret_cast<const float4*>(g.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f); }
      }
      if (EPI == GEMM_EPI_MUL_DELU) load_h(0, hv[0]);
#pragma unroll
      for (int sg = 0; sg < NS; ++sg) {
        if (sg > 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }      // the previous group's strip has been read
#pragma unroll
        for (int jj = 0; jj < JS; ++jj)
#pragma unroll
          for (int r = 0; r < 16; ++r) { if constexpr ((G3F_ABLATE & 128) == 0) Cs[((r & 3) + 8 * (r >> 2) + 4 * kh) * CS + 32 * jj + (lane & 31)] = acc[JS * sg + jj][r]; }
        if (EPI == GEMM_EPI_MUL_DELU && sg + 1 < NS) load_h(sg + 1, hv[(sg + 1) & 1]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int col = n0 + SW * sg + 4 * c4;
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const int r = r0 + RP * p, row = m0 + 32 * wave + r;
          float4 v;
          if constexpr ((G3F_ABLATE & 128) == 0) v = *reinterpret_cast<const float4*>(&Cs[r * CS + 4 * c4]);
          else v = make_float4(acc[sg][0], acc[sg][1 + p % 8], acc[sg][5 + p % 8], acc[sg][10]);
          if (EPI == GEMM_EPI_BIAS || EPI == GEMM_EPI_BIAS_ELU) { v.x += b4[sg].x; v.y += b4[sg].y; v.z += b4[sg].z; v.w += b4[sg].w; }
          if (EPI == GEMM_EPI_BIAS_ELU) { v.x = pf_elu(v.x); v.y = pf_elu(v.y); v.z = pf_elu(v.z); v.w = pf_elu(v.w); }
          if (EPI == GEMM_EPI_MUL_DELU) {
            const float4 h = hv[sg & 1][p];
            v.x *= (h.x > 0.0f ? 1.0f : h.x + 1.0f); v.y *= (h.y > 0.0f ? 1.0f : h.y + 1.0f);
            v.z *= (h.z > 0.0f ? 1.0f : h.z + 1.0f); v.w *= (h.w > 0.0f ? 1.0f : h.w + 1.0f);
          }
          if constexpr ((G3F_ABLATE & 64) != 0) { G3_KEEP(v.x); G3_KEEP(v.y); G3_KEEP(v.z); G3_KEEP(v.w); continue; }
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(g3f_u4, v), c_rs, (int)((row < g.M && col < g.N) ? ((unsigned)row * (unsigned)g.ldc + (unsigned)col) * 4u : OOB), 0, 0);
          if constexpr (G3F_EMIT_PLANES != 0) {
            if (g.Cp && row < g.M && col < g.N) {      // exact three-way split of the four values, one 8-byte store per plane
              uint2 q0, q1, q2;
              g3_split4(v, q0, q1, q2);
              unsigned short* d = g.Cp + (size_t)row * g.ldc + col;
              *reinterpret_cast<uint2*>(d) = q0; *reinterpret_cast<uint2*>(d + g.cplane) = q1; *reinterpret_cast<uint2*>(d + 2 * g.cplane) = q2;
            }
          }
        }
      }
    }
    __syncthreads();            // (a block that walks on to another tile re-fills the stages)
  }
}

template <int BN, int BK, int EPI, int NSTG = 2>
__global__ __launch_bounds__(256, NSTG > 2 ? 1 : 2) void pf_gemm3f_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[G3fGeom<BN, BK>::template lds<NSTG>()];
  pf_gemm3f_body<BN, BK, EPI, NSTG>(g, (int)blockIdx.x, (int)gridDim.x, smem);
}
// layer i of the actor and of the critic in one launch
template <int BN, int BK, int EPI, int NSTG = 2>
__global__ __launch_bounds__(256, NSTG > 2 ? 1 : 2) void pf_gemm3f_kernel2(GemmArgs g0, GemmArgs g1, int nb0) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[G3fGeom<BN, BK>::template lds<NSTG>()];
  if ((int)blockIdx.x < nb0) pf_gemm3f_body<BN, BK, EPI, NSTG>(g0, (int)blockIdx.x, nb0, smem);
  else pf_gemm3f_body<BN, BK, EPI, NSTG>(g1, (int)blockIdx.x - nb0, (int)gridDim.x - nb0, smem);
}

// The DMA path moves whole 16-byte chunks and the write-out whole float4: rows of A, C (and H) 16-byte aligned, K and N
// multiples of 4, B handed over as planes.  Anything else stays on pf_gemm3_body.
template <int EPI>
static inline bool pf_gemm3f_ok(const GemmArgs& g) {
  auto al = [](const void* p) { return (((size_t)p) & 15) == 0; };
  bool ok = g.Bp && al(g.Bp) && al(g.A) && al(g.C) && (g.lda & 3) == 0 && (g.ldc & 3) == 0 && (g.K & 3) == 0 && (g.N & 3) == 0 && (g.ldbp & 7) == 0 &&
            ((g.bplane * 2) & 15) == 0 && (long long)g.M * g.lda * 4ll < (1ll << 31) && (long long)g.M * g.ldc * 4ll < (1ll << 31) && 3ll * g.bplane * 2ll < (1ll << 31);
  if (EPI == GEMM_EPI_BIAS || EPI == GEMM_EPI_BIAS_ELU) ok = ok && al(g.bias);
  if (EPI == GEMM_EPI_MUL_DELU) ok = ok && al(g.H) && (g.ldh & 3) == 0 && (long long)g.M * g.ldh * 4ll < (1ll << 31);
  return ok;
}
// Variant 0: 128 columns per block; 1: 64 columns -- taken when 128 would leave the chip short of blocks (`tiles128` = the
// 128 x 128 tiles of everything in the launch: the actor's and the critic's layer share one) or N has a 64-wide remainder.
// (Measured and dropped, profiles/r3_gemm_dma_ablation.txt: BK = 16 stages for four workgroups per CU; 192-row tiles on
// v_mfma_f32_16x16x32_bf16.)
// (Also measured and dropped: 64-deep k-slices for the rollout's 4096-row launches -- half as many DMA / barrier round trips per
// block, but 21.5 vs 18.1 us on 4096 x 256 x 512: the first stage takes twice as long to land and one workgroup per CU hides nothing.)
static inline int pf_gemm3f_pick(int N, long tiles128) { return ((N % 128 == 0 && tiles128 >= 512) || N > 512) ? 0 : 1; }
static inline long pf_gemm3f_tiles128(const GemmArgs& g) { return (long)((g.M + 127) / 128) * ((g.N + 127) / 128); }
// Variant 2: 32 columns per block, for launches whose 64-column tiling leaves HALF the CUs without a workgroup (the rollout's
// 4096 x 128 x 256 layer pair: 128 workgroups; 10.5 -> 10.2 us in the bench, 9.5 -> 9.0 alone).  Not for the 256-workgroup layer
// (4096 x 256 x 512: 19.9 -> 21.4 us with two thin workgroups per CU).  These launches are a chain of small latencies inside one
// wavefront per SIMD -- barrier, fragment reads, split, MFMAs, write-out: removing any one ingredient saves 1-3 of the 17 us,
// removing the in-loop DMA 0.9, an empty launch takes 4.7 (profiles/r4_rollout_gemm_ablation.txt) -- and neither a deeper pipeline
// (four LDS stages, NSTG = 4: 21.5 us) nor a second workgroup per CU shortens the chain.  PF_GEMM3F_THIN=0: never.
static inline bool pf_gemm3f_thin(int blocks64) {
  if (blocks64 > 128) return false;
  const char* e = getenv("PF_GEMM3F_THIN");      // (read per launch: tests switch it)
  return !(e && atoi(e) == 0);
}
template <int BN>
static inline int pf_gemm3f_blocks(const GemmArgs& g) { return (((g.N + BN - 1) / BN) * ((g.M + G3F_BM - 1) / G3F_BM) + 7) / 8 * 8; }
template <int EPI>
static inline void pf_gemm3f_launch(const GemmArgs& g, hipStream_t s, int variant = -1) {
  if (variant < 0) variant = pf_gemm3f_pick(g.N, pf_gemm3f_tiles128(g));
  if (variant == 0) hipLaunchKernelGGL((pf_gemm3f_kernel<128, 32, EPI>), dim3(pf_gemm3f_blocks<128>(g)), dim3(256), 0, s, g);
  else if (variant == 2 || (variant == 1 && pf_gemm3f_thin(pf_gemm3f_blocks<64>(g)))) hipLaunchKernelGGL((pf_gemm3f_kernel<32, 32, EPI>), dim3(pf_gemm3f_blocks<32>(g)), dim3(256), 0, s, g);
  else hipLaunchKernelGGL((pf_gemm3f_kernel<64, 32, EPI>), dim3(pf_gemm3f_blocks<64>(g)), dim3(256), 0, s, g);
}
template <int EPI>
static inline void pf_gemm3f_launch2(const GemmArgs& g0, const GemmArgs& g1, hipStream_t s, int variant = -1) {
  if (variant < 0) variant = pf_gemm3f_pick(g0.N < g1.N ? g0.N : g1.N, pf_gemm3f_tiles128(g0) + pf_gemm3f_tiles128(g1));
  if (variant == 1) {
    const int nb0 = pf_gemm3f_blocks<64>(g0), nb1 = pf_gemm3f_blocks<64>(g1);
    if (pf_gemm3f_thin(nb0 + nb1)) variant = 2;
    else { hipLaunchKernelGGL((pf_gemm3f_kernel2<64, 32, EPI>), dim3(nb0 + nb1), dim3(256), 0, s, g0, g1, nb0); return; }
  }
  if (variant == 2) {
    const int nb0 = pf_gemm3f_blocks<32>(g0), nb1 = pf_gemm3f_blocks<32>(g1);
    hipLaunchKernelGGL((pf_gemm3f_kernel2<32, 32, EPI>), dim3(nb0 + nb1), dim3(256), 0, s, g0, g1, nb0);
    return;
  }
  const int nb0 = pf_gemm3f_blocks<128>(g0), nb1 = pf_gemm3f_blocks<128>(g1);
  hipLaunchKernelGGL((pf_gemm3f_kernel2<128, 32, EPI>), dim3(nb0 + nb1), dim3(256), 0, s, g0, g1, nb0);
}


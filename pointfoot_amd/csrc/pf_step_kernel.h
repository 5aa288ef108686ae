// pf_step_kernel.h — the fused env-step program: PD torques, `decimation` articulated-body substeps,
// and the whole post_physics_step (base-frame quantities, height scan, feet state machine, heading /
// push callback, termination, reward stack, masked reset with curriculum, observations + noise).
//
// Replaces PointFoot.step / post_physics_step (reference legged_gym/envs/pointfoot/point_foot.py:170-230)
// and the Isaac Gym simulate call inside it (:183).  One wavefront per robot; the env's state record,
// the body frames, the 12x12 mass matrix and the constraint rows live in LDS (struct Shared).
// Written in the phase vocabulary of pf_simt.h.  Line references (:NNN) are to point_foot.py.
#pragma once
#include "../../include/pointfoot_hip.h"
#include "pf_simt.h"

#define PF_REC_MAX 176
#define PF_MAXC 6          /* contacts kept per robot (first active in sphere order: feet first): PointFoot.  The quadruped instance keeps
                              PF_MAXC_QUAD = 10 (4 feet x floor + riser face, and two more for a shank or the base): measured on the curriculum field, 0.68 % of its (env, substep)
                              pairs had more than 6 active candidates, PointFoot's 3.4e-5 (profiles/r3_contact_cap.txt) */
#define PF_MAXC_QUAD 10
#define PF_MAX_CHAINS 8
#define PF_MAX_CHAIN_DEPTH 8
#include <cmath>
#include <type_traits>
#include "pf_lump.h"
#ifndef PF_DUAL_INERTIA
#define PF_DUAL_INERTIA 0  /* DUAL instances: 1 = the partner wavefront builds the body inertias and their composites, 0 = the dynamics side */
#endif
#ifndef PF_DUAL_SHARED_FK
#define PF_DUAL_SHARED_FK 1  /* DUAL instances: 1 = the contact wavefront waits for the dynamics side's body frames of the substep, 0 = it walks the chains itself (own copy, same bits) */
#endif
#ifndef PF_DUAL_CREG
#define PF_DUAL_CREG 4     /* DUAL instances: contact rows / joint-limit rows the PGS sweep keeps in registers (128 VGPRs) */
#endif
#ifndef PF_DUAL_LREG
#define PF_DUAL_LREG 2
#endif
#define PF_KSPH 24         /* collision spheres staged per robot (PointFoot instance; the quadruped instance uses 32) */
#define PF_KSPH_QUAD 32
#define PF_KPAIR_QUAD 56      /* self-collision capsule pairs of the quadruped instance built with it (4 legs x {thigh, shank, foot}: 54) */

// ------------------------------------------------------------------------------------ small math
PF_DEV float pf_clip(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
// 1 / x for the solver's pivots, normalisations and friction ratios: v_rcp_f32 (1 ulp) + one Newton step instead of the IEEE
// division sequence (v_div_scale x2, v_rcp, 4 fma, v_div_fmas, v_div_fixup: 48 of them were 12 % of a substep's VALU work).
// Within an ulp of the quotient; the host build of the step program divides.
PF_DEV float pf_rcp(float x) {
#if defined(PF_HOST_EMU)
  return 1.0f / x;
#else
  const float r = __builtin_amdgcn_rcpf(x);
  return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
#endif
}
// sin and cos of a joint angle (|x| well inside [-pi, pi]: joint limits): Cody-Waite reduction to [-pi/4, pi/4] and the Cephes
// single-precision minimax polynomials, both values from one reduction -- within 1-2 ulp of sinf / cosf, without the library
// routines' large-argument machinery (three joints per chain, five kinematics passes per step).  The host build calls libm.
PF_DEV void pf_sincos(float x, float* sn, float* cs) {
#if defined(PF_HOST_EMU)
  *sn = sinf(x); *cs = cosf(x);
#else
  const float ax = fabsf(x);
  int j = (int)(ax * 1.27323954473516f);          // 4 / pi
  j = (j + 1) & ~1;                                // nearest even multiple of pi / 4 at or above
  const float y = (float)j;
  float z = __builtin_fmaf(y, -0.78515625f, ax);
  z = __builtin_fmaf(y, -2.4187564849853515625e-4f, z);
  z = __builtin_fmaf(y, -3.77489497744594108e-8f, z);
  const float zz = z * z;
  float ps = __builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, zz, 8.3321608736e-3f), zz, -1.6666654611e-1f);
  ps = __builtin_fmaf(ps * zz, z, z);
  float pc = __builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, zz, -1.388731625493765e-3f), zz, 4.166664568298827e-2f);
  pc = __builtin_fmaf(pc * zz, zz, __builtin_fmaf(-0.5f, zz, 1.0f));
  const int q = (j >> 1) & 3;                      // quadrant
  float s_ = (q & 1) ? pc : ps, c_ = (q & 1) ? ps : pc;
  if (q == 1 || q == 2) c_ = -c_;
  if (q >= 2) s_ = -s_;
  *sn = x < 0.0f ? -s_ : s_;
  *cs = c_;
#endif
}
// sqrt(x) for the same places (x >= 0, not denormal there): v_sqrt_f32 + one Newton step instead of the scaled IEEE sequence
PF_DEV float pf_sqrt(float x) {
#if defined(PF_HOST_EMU)
  return sqrtf(x);
#else
  const float s = __builtin_amdgcn_sqrtf(x);
  // s + (x - s^2) / (2 s), the residual in one fma; x == 0 gives s == 0 and rcp(0) * 0 -> NaN, so guard it
  const float h = 0.5f * __builtin_amdgcn_rcpf(s);
  return s > 0.0f ? __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s) : s;
#endif
}
PF_DEV void pf_cross(const float* a, const float* b, float* o) {
  float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
PF_DEV float pf_dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
PF_DEV void pf_mat_vec(const float* A, const float* v, float* o) {
  float x = A[0] * v[0] + A[1] * v[1] + A[2] * v[2];
  float y = A[3] * v[0] + A[4] * v[1] + A[5] * v[2];
  float z = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
PF_DEV void pf_mat_mul(const float* A, const float* B, float* C) {
  float t[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) t[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
#pragma unroll
  for (int i = 0; i < 9; ++i) C[i] = t[i];
}
// sym 3x3 stored xx,yy,zz,xy,xz,yz
PF_DEV void pf_sym_vec(const float* I, const float* v, float* o) {
  float x = I[0] * v[0] + I[3] * v[1] + I[4] * v[2];
  float y = I[3] * v[0] + I[1] * v[1] + I[5] * v[2];
  float z = I[4] * v[0] + I[5] * v[1] + I[2] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
// spatial inertia about O: I10 = {m, hx,hy,hz, Ixx,Iyy,Izz,Ixy,Ixz,Iyz}; v6/o6 = {ang(3), lin(3)}
PF_DEV void pf_spI_mul(const float* I10, const float* v6, float* o6) {
  float a[3], hxl[3], hxa[3];
  pf_sym_vec(I10 + 4, v6, a);
  pf_cross(I10 + 1, v6 + 3, hxl);
  pf_cross(I10 + 1, v6, hxa);
  o6[0] = a[0] + hxl[0]; o6[1] = a[1] + hxl[1]; o6[2] = a[2] + hxl[2];
  o6[3] = I10[0] * v6[3] - hxa[0]; o6[4] = I10[0] * v6[4] - hxa[1]; o6[5] = I10[0] * v6[5] - hxa[2];
}
// isaacgym.torch_utils.quat_rotate (+1) / quat_rotate_inverse (-1), torch op order
PF_DEV void pf_quat_rotate_dir(const float* q, const float* v, float sign, float* out) {
#if !defined(PF_HOST_EMU)
#pragma clang fp contract(off)
#endif
  float w = q[3];
  float s = 2.0f * (w * w) - 1.0f;
  float cx = q[1] * v[2] - q[2] * v[1], cy = q[2] * v[0] - q[0] * v[2], cz = q[0] * v[1] - q[1] * v[0];
  float d = q[0] * v[0] + q[1] * v[1] + q[2] * v[2];
  float o0 = (v[0] * s + sign * (cx * w * 2.0f)) + q[0] * d * 2.0f;
  float o1 = (v[1] * s + sign * (cy * w * 2.0f)) + q[1] * d * 2.0f;
  float o2 = (v[2] * s + sign * (cz * w * 2.0f)) + q[2] * d * 2.0f;
  out[0] = o0; out[1] = o1; out[2] = o2;
}
PF_DEV void pf_quat_apply(const float* a, const float* b, float* out) {
#if !defined(PF_HOST_EMU)
#pragma clang fp contract(off)
#endif
  float tx = (a[1] * b[2] - a[2] * b[1]) * 2.0f, ty = (a[2] * b[0] - a[0] * b[2]) * 2.0f, tz = (a[0] * b[1] - a[1] * b[0]) * 2.0f;
  float ux = a[1] * tz - a[2] * ty, uy = a[2] * tx - a[0] * tz, uz = a[0] * ty - a[1] * tx;
  out[0] = (b[0] + a[3] * tx) + ux; out[1] = (b[1] + a[3] * ty) + uy; out[2] = (b[2] + a[3] * tz) + uz;
}
// legged_gym/utils/math.py:45-48 with torch.remainder semantics
PF_DEV float pf_wrap_to_pi(float a) {
#if !defined(PF_HOST_EMU)
#pragma clang fp contract(off)
#endif
  const float two_pi = 6.28318530717958647692f, pi = 3.14159265358979323846f;
  float mod = fmodf(a, two_pi);
  if (mod != 0.0f && mod < 0.0f) mod += two_pi;
  if (mod > pi) mod -= two_pi;
  return mod;
}

// Philox4x32-10 (Salmon et al. 2011); counter = (global env, step, stream, slot/4), key = seed
// the whole block of four slots 4 b .. 4 b + 3 (pf_philox_slot keeps one of them): a block costs ten rounds of two 32 x 32 -> 64
// multiplies, so long runs of consecutive slots are drawn block-wise, one block per lane (PfKernel::noise_prepare)
PF_DEV void pf_philox_block(uint32_t k0, uint32_t k1, uint32_t env, uint32_t step, uint32_t stream, uint32_t block, float* out4) {
  uint32_t c0 = env, c1 = step, c2 = stream, c3 = block;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out4[0] = (float)(c0 >> 8) * (1.0f / 16777216.0f); out4[1] = (float)(c1 >> 8) * (1.0f / 16777216.0f);
  out4[2] = (float)(c2 >> 8) * (1.0f / 16777216.0f); out4[3] = (float)(c3 >> 8) * (1.0f / 16777216.0f);
}
PF_DEV float pf_philox_slot(uint32_t k0, uint32_t k1, uint32_t env, uint32_t step, uint32_t stream, uint32_t slot) {
  uint32_t c0 = env, c1 = step, c2 = stream, c3 = slot >> 2;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  uint32_t s = slot & 3u;
  uint32_t x = s == 0 ? c0 : (s == 1 ? c1 : (s == 2 ? c2 : c3));
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}

// the four raw words (pf_ppo.hip philox4: the action sampler's counter layout is (env, sample counter, 1, action index))
PF_DEV void pf_philox_raw(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* o) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3; k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// Diagnostic build only (-DPF_STAMPS, tools/step_stamps.py): lane 0 accumulates s_memtime deltas per phase into
// args.stamps[env][PF_NSTAMP]; the real kernel contains no stamp.
#define PF_NSTAMP 16
#if defined(PF_STAMPS) && !defined(PF_HOST_EMU)
#define PF_STAMP(sh, id) do { if (threadIdx.x == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); \
                              (sh).stamp_acc[id] += (unsigned)(t_ - (sh).stamp_last); (sh).stamp_last = t_; } } while (0)
#else
#define PF_STAMP(sh, id) ((void)0)
#endif

struct PfStepArgs {
  const void* packed;     // the instance's ModelS image (PfKernel::pack_model), device copy: staged into LDS by one flat copy
  const PfConfig* cfg;    // device copy
  PfLayout L;
  PfBuffers b;
  PfInject inj;
  const float* actions;
  long long step_counter;
  int mode;               // 0 = step, 1 = reset_idx(all), 2 = compute_observations only, 3 = the height scan only (fills d_measured_heights)
  unsigned* stamps;       // diagnostic build only, else NULL
  float* stats_next;      // next step's statistics slot, cleared by this launch (NULL: the host clears slots itself)
  const uint8_t* reset_mask;   // mode 1 only: envs to reset (NULL: all) -- reset_idx(env_ids), point_foot.py:249
};

// KPR: self-collision capsule pairs the instance is built for (0: built without self-collision)
// NBD: bodies of the model the dynamics run on -- the robot with its fixed-joint bodies (the feet) lumped into their parents
// (pf_lump.h); NB stays the number of bodies the record reports on (contact_forces, feet)
// DUAL_: two wavefronts per robot group (GPU only; the host build runs the two roles one after the other): wave 0 runs the
// dynamics of a substep (kinematics with velocities, inertias, mass matrix, factorisation), wave 1 -- at the same time -- its
// contact side (kinematics without velocities, collision queries, candidate compaction, contact rows); they meet before the
// constraint solves.  The launch is one wavefront's dependent chain: the chain gets shorter, not the work.
template <int NB, int ND, int NF, int KIND = PF_KIND_POINTFOOT, int KS = PF_KSPH, int RPW = 1, int KPR = (KIND == PF_KIND_POINTFOOT ? 16 : 0), int NBD = NB, bool DUAL_ = false>
struct PfKernel {
#if defined(PF_HOST_EMU)
  static constexpr bool DUAL = false;
#else
  static constexpr bool DUAL = DUAL_;
#endif
  static constexpr int PF_W = 64 / RPW;         // lanes per robot (pf_simt.h)
  static constexpr bool LR = KIND == PF_KIND_LEGGED_ROBOT;      // LeggedRobot / Anymal semantics (legged_robot.py)
  static constexpr bool PF_OPAQUE_LANE = LR || DUAL;                  // pf_simt.h: no hoisting of lane-derived values (register pressure)
  static constexpr int REC_MAX = LR ? 704 : PF_REC_MAX;
  static constexpr int NV = 6 + ND;
  static constexpr int NVP = NV | 1;          // odd row stride: per-lane rows hit distinct LDS banks
  static constexpr int MAXC = LR ? PF_MAXC_QUAD : PF_MAXC;
  // contact rows the PGS sweep holds in registers; the rest stay in LDS (a DUAL instance has 128 VGPRs: four -- a biped rarely has more)
  static constexpr int CREG = DUAL_ ? PF_DUAL_CREG : (MAXC < 6 ? MAXC : 6);
  // self-collision capsules / capsule pairs this instance is built for (PointFoot: 2 legs x 4 links = 8 / 16; a quadruped with
  // self-collision: 4 legs x {thigh, shank, foot} = 12 / 54, anymal_c_flat_config.py:42; anymal_c_rough switches it off,
  // anymal_c_rough_config.py:77, and runs the instance built without)
  static constexpr bool SELF = KPR > 0;
  static constexpr int KC = SELF ? (LR ? PF_MAX_CAPS : 8) : 1, KP = SELF ? KPR : 1;
  static constexpr int NCW = (2 * KS + (SELF ? KP : 0) > 64) ? 2 : 1;      // 64-bit words of the contact-candidate set
  static_assert(2 * KS + (SELF ? KP : 0) <= 128, "the contact-candidate set is at most two 64-bit words");
  static constexpr int CROWS = 3 * MAXC;
  static constexpr int RFREE = CROWS + ND;    // extra right-hand side: free acceleration
  static constexpr int NROWS = RFREE + 1;
#if defined(PF_HOST_EMU)
  static constexpr bool CHOLR = false;
#else
  // <= 12 velocity components: the Cholesky factor lives in registers (packed lower triangle, every lane holds all of it
  // and factorises redundantly with static indices: no LDS round trips, no barriers), and the substitutions read it there
  // (DUAL instances run four wavefronts per SIMD: 128 VGPRs -- the packed triangle does not fit; see CHOLD)
  static constexpr bool CHOLR = NV <= 12 && !DUAL;
#endif
#if defined(PF_HOST_EMU)
  static constexpr bool CHOLL = false;
#else
  // wider systems, one robot per wavefront: lane i keeps row i of the factor in registers and the pivot row reaches the
  // other lanes through v_readlane (wave-uniform SGPR operands) -- no LDS traffic or barriers during the factorisation
  static constexpr bool CHOLL = !CHOLR && RPW == 1 && NV <= 32;
#endif
#if defined(PF_HOST_EMU)
  static constexpr bool CHOLD = false;
#else
  // DUAL instances, <= 16 velocity components: the factor is DISTRIBUTED over the lanes of every 16-lane DPP row -- lane i holds row
  // i (NV registers instead of NV (NV + 1) / 2 in every lane) and an entry of another row reaches a lane as the DPP operand of the
  // multiply-add that consumes it (row_newbcast: lane n of the row to all 16).  Same operations in the same order as CHOLR: the
  // factor and every solve come out bit for bit the same.  Every lane must be active wherever the factor is read (a DPP read
  // of a lane that EXEC has switched off returns 0): the solves run unpredicated on clamped rows and predicate their stores.
  static constexpr bool CHOLD = DUAL && NV <= 16;
#endif
  static constexpr int NLT = CHOLR ? NV * (NV + 1) / 2 : (CHOLD ? NV : 1);
  static constexpr int NDV = CHOLR ? NV : 1;
#if !defined(PF_HOST_EMU)
  // x as lane N of the 16-lane DPP row holds it (N static)
  template <int N>
  PF_DEV float row_bcast(float x) {
    static_assert(N >= 0 && N < 16, "row_newbcast takes a lane of the row");
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x150 + N, 0xF, 0xF, true));
  }
  // the static-index trampolines of the unrolled factorisation / substitutions
  template <int I, int T> PF_DEV float chold_L(const float (&Lr)[NLT]) { return row_bcast<I>(Lr[CHOLD ? T : 0]); }      // L[I][T], T < I
#endif
  static constexpr int lt(int i, int t) { return i * (i + 1) / 2 + t; }


  // Record offsets as compile-time constants (same rule as pf_compute_layout with num_actions == ND; checked against the
  // runtime PfLayout in pf_create): addresses fold into ds_read/ds_write immediates instead of living in SGPRs.
  struct LC {
    static constexpr int root = 0, dof_pos = root + 13, dof_vel = dof_pos + ND, last_actions = dof_vel + ND,
                         last_dof_vel = last_actions + ND, last_root_vel = last_dof_vel + ND, commands = last_root_vel + 6,
                         feet_air_time = commands + 4,
                         last_feet_air_time = feet_air_time + NF, cur_max_feet_height = last_feet_air_time + NF,
                         last_max_feet_height = cur_max_feet_height + NF, feet_height = last_max_feet_height + NF,
                         contact_filt = feet_height + NF, first_contact = contact_filt + NF, last_contacts = first_contact + NF,
                         ext_force = last_contacts + NF,
                         env_origin = ext_force + 4, friction = env_origin + 3, added_mass = friction + 1, added_com = added_mass + 1,
                         lstm_state = added_com + 3, episode_sums = lstm_state + (LR ? 32 * ND : 0);
    // the write-only part starts on the 64-byte line after the (fixed-size) episode-sum block: a step does not load it
    static constexpr int rw_floats = (episode_sums + PF_MAX_REWARDS + 1 + 15) & ~15;
    static constexpr int actions = rw_floats, torques = actions + ND, base_lin_vel = torques + ND, base_ang_vel = base_lin_vel + 3,
                         projected_gravity = base_ang_vel + 3, contact_forces = projected_gravity + 3,
                         feet_pos = contact_forces + 3 * NB, rec_used = feet_pos + 3 * NF;
  };
  static bool layout_matches(const PfLayout& L) {
    return L.root == LC::root && L.dof_pos == LC::dof_pos && L.dof_vel == LC::dof_vel && L.actions == LC::actions &&
           L.last_actions == LC::last_actions && L.last_dof_vel == LC::last_dof_vel && L.last_root_vel == LC::last_root_vel &&
           L.commands == LC::commands && L.torques == LC::torques && L.base_lin_vel == LC::base_lin_vel &&
           L.base_ang_vel == LC::base_ang_vel && L.projected_gravity == LC::projected_gravity && L.feet_air_time == LC::feet_air_time &&
           L.last_feet_air_time == LC::last_feet_air_time && L.cur_max_feet_height == LC::cur_max_feet_height &&
           L.last_max_feet_height == LC::last_max_feet_height && L.feet_height == LC::feet_height && L.contact_filt == LC::contact_filt &&
           L.first_contact == LC::first_contact && L.last_contacts == LC::last_contacts && L.contact_forces == LC::contact_forces &&
           L.feet_pos == LC::feet_pos && L.ext_force == LC::ext_force && L.env_origin == LC::env_origin && L.friction == LC::friction &&
           L.added_mass == LC::added_mass && L.added_com == LC::added_com && L.lstm_state == LC::lstm_state &&
           L.episode_sums == LC::episode_sums && L.rw_floats == LC::rw_floats && L.rec_used == LC::rec_used && L.rec_floats <= REC_MAX;
  }

  // The robot model, re-packed to this instance's sizes and staged in LDS once per launch: the per-body / per-sphere
  // tables are indexed by lane, so reading them from global memory would put an L2 round trip into every phase.
  struct ModelS {
    int nsph, n_term, n_pen;
    int parent[NBD], depth[NBD], subtree[NBD], dof_of_body[NBD], body_of_dof[ND], feet_body[NF];
    int feet_dyn[NF]; float feet_off[NF][3];      // carrier body of foot f and the foot frame's origin on it (pf_lump.h)
    int term_body[PF_MAX_TERM], pen_body[PF_MAX_PEN];
    float joint_pos[NBD][3], joint_rot[NBD][9], joint_axis[NBD][3];
    float mass[NBD], com[NBD][3], inertia[NBD][6];
    float q_lo[ND], q_hi[ND], vel_limit[ND];
    int sph_body[KS], sph_rep[KS];                // body that carries sphere s / body its contact force is reported on
    float sph_pos[KS][3], sph_radius[KS];
    int ncap, npair, cap_s0[KC], cap_s1[KC], pair_a[KP], pair_b[KP];
    float pair_reach[KP];      // half lengths + radii of the pair's two capsules: midpoints further apart than this + contact_offset cannot touch
    // root-to-leaf chains (one lane walks one chain in forward kinematics) and per-body 'fixed rotation is identity' flags
    int nchain, chain_len[PF_MAX_CHAINS], chain[PF_MAX_CHAINS][PF_MAX_CHAIN_DEPTH], rot_ident[NBD];
  };

  // Host side (pf_create; the host build of the step program): ModelS from the model with lumped fixed bodies and its
  // reporting tables (pf_lump.h), chains and identity flags included -- the launch only copies the image into LDS.
  static void pack_model(const PfModel& gm, const PfLump& lump, ModelS& ms) {
    std::memset(&ms, 0, sizeof(ModelS));
    for (int i = 0; i < NBD; ++i) {
      ms.parent[i] = gm.parent[i]; ms.depth[i] = gm.depth[i]; ms.subtree[i] = gm.subtree[i]; ms.dof_of_body[i] = gm.dof_of_body[i];
      ms.mass[i] = gm.mass[i];
      for (int t = 0; t < 3; ++t) { ms.joint_pos[i][t] = gm.joint_pos[i][t]; ms.joint_axis[i][t] = gm.joint_axis[i][t]; ms.com[i][t] = gm.com[i][t]; }
      for (int t = 0; t < 9; ++t) ms.joint_rot[i][t] = gm.joint_rot[i][t];
      for (int t = 0; t < 6; ++t) ms.inertia[i][t] = gm.inertia[i][t];
    }
    for (int k = 0; k < ND; ++k) { ms.body_of_dof[k] = gm.body_of_dof[k]; ms.q_lo[k] = gm.q_lo[k]; ms.q_hi[k] = gm.q_hi[k]; ms.vel_limit[k] = gm.vel_limit[k]; }
    for (int s2 = 0; s2 < KS; ++s2) {
      ms.sph_body[s2] = gm.sph_body[s2]; ms.sph_radius[s2] = gm.sph_radius[s2]; ms.sph_rep[s2] = lump.sph_rep[s2];
      for (int t = 0; t < 3; ++t) ms.sph_pos[s2][t] = gm.sph_pos[s2][t];
    }
    for (int f = 0; f < NF; ++f) {
      ms.feet_body[f] = gm.feet_body[f]; ms.feet_dyn[f] = lump.feet_dyn[f];
      for (int t = 0; t < 3; ++t) ms.feet_off[f][t] = lump.feet_off[f][t];
    }
    for (int t = 0; t < KC && t < PF_MAX_CAPS; ++t) { ms.cap_s0[t] = gm.cap_s0[t]; ms.cap_s1[t] = gm.cap_s1[t]; }
    for (int t = 0; t < KP && t < PF_MAX_PAIRS; ++t) {
      ms.pair_a[t] = gm.pair_a[t]; ms.pair_b[t] = gm.pair_b[t];
      float reach = 0.0f;
      if (t < gm.npair) {
        const int caps[2] = {gm.pair_a[t], gm.pair_b[t]};
        for (int u = 0; u < 2; ++u) {
          const float *p0 = gm.sph_pos[gm.cap_s0[caps[u]]], *p1 = gm.sph_pos[gm.cap_s1[caps[u]]];
          const double dx = (double)p1[0] - p0[0], dy = (double)p1[1] - p0[1], dz = (double)p1[2] - p0[2];
          reach += (float)(0.5 * std::sqrt(dx * dx + dy * dy + dz * dz)) + gm.sph_radius[gm.cap_s0[caps[u]]];
        }
      }
      ms.pair_reach[t] = reach * 1.0001f + 1e-4f;      // (margin against the rounding of the test below)
    }
    for (int t = 0; t < PF_MAX_TERM; ++t) ms.term_body[t] = gm.term_body[t];
    for (int t = 0; t < PF_MAX_PEN; ++t) ms.pen_body[t] = gm.pen_body[t];
    ms.nsph = gm.nsph; ms.n_term = gm.n_term; ms.n_pen = gm.n_pen; ms.ncap = gm.ncap; ms.npair = gm.npair;
    // chains: one per leaf body (subtree == 1); chain[c] lists the bodies from the base's child down to the leaf
    int nch = 0;
    for (int b = 0; b < NBD; ++b) {
      bool ident = true;
      for (int t = 0; t < 9; ++t) ident = ident && (ms.joint_rot[b][t] == ((t % 4 == 0) ? 1.0f : 0.0f));
      ms.rot_ident[b] = ident ? 1 : 0;
      if (b > 0 && ms.subtree[b] == 1) {
        const int ci = nch++, d = ms.depth[b];
        if (ci < PF_MAX_CHAINS && d <= PF_MAX_CHAIN_DEPTH) {
          ms.chain_len[ci] = d;
          int j = b;
          for (int kk = d - 1; kk >= 0; --kk) { ms.chain[ci][kk] = j; j = ms.parent[j]; }
        }
      }
    }
    ms.nchain = nch < PF_MAX_CHAINS ? nch : PF_MAX_CHAINS;
  }

  // read-only per launch, identical for every robot: one copy per workgroup, shared by all its robots
  struct SharedRO {
    ModelS mdl;
    float act[LR ? PF_ACT_PARAMS : 1];        // actuator-net parameters (LeggedRobot)
  };
  struct Shared {
#if defined(PF_STAMPS) && !defined(PF_HOST_EMU)
    unsigned long long stamp_last;
    unsigned stamp_acc[PF_NSTAMP];
#endif
    float rec[REC_MAX];
    // LSTM gate pre-activations, gate-major: gate c of (joint j, unit u) at c * GST + 8 j + u -- the 32 rows a joint's lanes
    // write (4 gates x 8 units) and the 4 values a (joint, unit) lane reads fall into distinct banks (GST = 8 ND + 8)
    // (the gates live from the start of a substep to its torques, the mass matrix / its factor from the mass-matrix phase to the
    // last solve of the same substep: one storage)
    union {
      float gates[LR ? 4 * (ND * 8 + 8) : 1];
      float M[NV][NVP];
    };
    float Rw[NBD][9], r[NBD][3], ax[NBD][3];
    // The dynamics scratch (dead once the mass matrix and the bias are built) and the contact candidates (born after the
    // factorisation, dead after the contact rows) share storage when ONE wavefront runs both (union); with a wavefront for
    // each side (DUAL) both are live at once, and the contact side has frames of its own (kb).
    struct DynS {
      float Sj[NBD][6], V[NBD][6], A[NBD][6];      // {ang, lin@O}
      float I[NBD][10], Ic[NBD][10];
      float f[NBD][6], fc[NBD][6];                // {moment@O, force}
      float F[ND][6];
    };
    struct ConS {
      float sphc[KS][3];                        // sphere centres relative to the base origin
      float cn[2 * KS][3], cd[2 * KS];          // normal / signed distance: floor candidate s at [s], face candidate at [KS + s]
      float pn[KP][3], px[KP][3], pd[KP];       // capsule pairs: normal (b -> a), contact point, distance
      int ccell[KS];                            // 4 * grid cell + face of the sphere's terrain query (diagnostic word)
    };
    union ScratchU { DynS d; ConS c; };
    struct ScratchP { DynS d; ConS c; };
    typename std::conditional<DUAL, ScratchP, ScratchU>::type u;
    struct KinB { float Rw[NBD][9], r[NBD][3], ax[NBD][3]; };
    struct None {};
    typename std::conditional<DUAL && PF_DUAL_SHARED_FK == 0, KinB, None>::type kb;      // the contact wavefront's own body frames (same arithmetic, same bits)
    int ncon_x, anchor_x[2], flag_x; float hsum_x;             // handed from one wavefront to the other (DUAL)
    float park_x[4];                                           // values of the step parked across the substep loop (DUAL: 128 VGPRs)
    float dinv[NV];
    float bias[NV], vel[NV], vold[NV];
    float J[NROWS][NVP], W[NROWS][NVP];
    float rb[NROWS], rAinv[NROWS], lam[NROWS];
    float cA[MAXC][9], cdir[MAXC][9];
    int cbody[MAXC], cbody_b[MAXC], con_cand[MAXC], lim_dof[ND], lim_side[ND];
  };
  // the height scan (post-physics only) reuses the constraint-row storage (physics only)
  static_assert(NROWS * NVP >= PF_MAX_HPTS, "height scan does not fit the aliased row storage");
  // (a DUAL instance keeps both in its contact scratch -- dead once the last substep's rows are handed over --, where the partner
  // wavefront fills them while the constraint-row storage is still in use)
  static_assert(!DUAL || sizeof(typename Shared::ConS) >= 4 * (160 + PF_MAX_HPTS), "noise + height scratch does not fit the contact scratch");
  PF_DEV float* heights_of(Shared& sh) { if constexpr (DUAL) return reinterpret_cast<float*>(&sh.u.c) + 160; else return &sh.J[0][0]; }
  // ... and the block-wise uniform draws of the observation noise reuse the other row array
  static_assert(NROWS * NVP >= 4 * (PF_MAX_OBS / 4 + 2), "noise scratch does not fit the aliased row storage");
  PF_DEV float* noise_of(Shared& sh) { if constexpr (DUAL) return reinterpret_cast<float*>(&sh.u.c); else return &sh.W[0][0]; }
  static_assert(4 * ((27 + PF_MAX_HPTS) / 4 + 2) <= 4 * 160 || !DUAL, "noise scratch of a DUAL instance: 160 floats");

  // ------------------------------------------------------------------ terrain
  // point_foot.py:1100-1112 for one point; exact float32 op order of the reference, no FMA contraction
  // index of the height sample at the low corner of the cell under (wx, wy) -- _get_heights' clipped integer coordinates (:1062-1068)
  PF_DEV int height_cell(const PfConfig& c, float wx, float wy) {
#if !defined(PF_HOST_EMU)
#pragma clang fp contract(off)
#endif
    float fx = (wx + c.border) / c.hscale;
    float fy = (wy + c.border) / c.hscale;
    int ix = (int)fx, iy = (int)fy;
    ix = ix < 0 ? 0 : (ix > c.hf_rows - 2 ? c.hf_rows - 2 : ix);
    iy = iy < 0 ? 0 : (iy > c.hf_cols - 2 ? c.hf_cols - 2 : iy);
    return ix * c.hf_cols + iy;
  }
  PF_DEV float height_raw(const PfConfig& c, const int16_t* hs, float wx, float wy) {
    const int at = height_cell(c, wx, wy);
    int h1 = hs[at], h2 = hs[at + c.hf_cols], h3 = hs[at + 1];
    int h = h1 < h2 ? h1 : h2;
    h = h < h3 ? h : h3;
    return (float)h;
  }
  // Collision surface (own contact model; the reference hands PhysX the trimesh of terrain.py:69-73).  Floor: bilinear over
  // the cell's corners, a corner more than wall_threshold above the cell's lowest corner taking the low height (it is the
  // top of a vertical face).  Faces: on each of the cell's four grid lines, where the raw profile stands more than
  // wall_threshold above the floor profile; the sphere touches the nearest point of the face segment.  Returns the floor
  // height / normal and the closest face (wd = 1e9: none).  Same arithmetic as oracle/pf_oracle.cpp Terrain::query.
  PF_DEV void terrain_query(const PfConfig& c, const int16_t* hs, int ix0, int iy0, float lx, float ly, float wz, float rad, float* h, float* n, float* wn, float* wd, int* cellface) {
    *cellface = 0;
    *wd = 1e9f; wn[0] = 0.0f; wn[1] = 0.0f; wn[2] = 1.0f;
    if (c.terrain_type == PF_TERRAIN_PLANE || hs == nullptr) { *h = 0.0f; n[0] = 0.0f; n[1] = 0.0f; n[2] = 1.0f; return; }
    // (lx, ly): metres from the corner of grid cell (ix0, iy0), the anchor of this step's local coordinates (run())
    const float fx = lx / c.hscale, fy = ly / c.hscale;
    const float ox = floorf(fx), oy = floorf(fy);
    int ix = ix0 + (int)ox, iy = iy0 + (int)oy;
    float tx = fx - ox, ty = fy - oy;
    if (ix < 0) { ix = 0; tx = 0.0f; } else if (ix > c.hf_rows - 2) { ix = c.hf_rows - 2; tx = 1.0f - 1e-3f; }
    if (iy < 0) { iy = 0; ty = 0.0f; } else if (iy > c.hf_cols - 2) { iy = c.hf_cols - 2; ty = 1.0f - 1e-3f; }
    *cellface = 4 * (ix * c.hf_cols + iy);
    const float vs = c.vscale, hsc = c.hscale, thr = c.wall_threshold;
    const float H00 = vs * (float)hs[ix * c.hf_cols + iy], H10 = vs * (float)hs[(ix + 1) * c.hf_cols + iy],
                H01 = vs * (float)hs[ix * c.hf_cols + iy + 1], H11 = vs * (float)hs[(ix + 1) * c.hf_cols + iy + 1];
    float E00 = H00, E10 = H10, E01 = H01, E11 = H11;
    const float lo = fminf(fminf(H00, H10), fminf(H01, H11)), hi = fmaxf(fmaxf(H00, H10), fmaxf(H01, H11));
    // a cell whose corners lie within wall_threshold of each other has no face and no lowered corner: the common case,
    // skipped as a whole when no lane of the wavefront stands on a step
    const bool steep = thr > 0.0f && hi - lo > thr;
    if (steep) {
      E00 = (H00 - lo > thr) ? lo : H00; E10 = (H10 - lo > thr) ? lo : H10;
      E01 = (H01 - lo > thr) ? lo : H01; E11 = (H11 - lo > thr) ? lo : H11;
    }
    *h = (1.0f - tx) * ((1.0f - ty) * E00 + ty * E01) + tx * ((1.0f - ty) * E10 + ty * E11);
    const float ihs = pf_rcp(hsc);
    float dhdx = ((1.0f - ty) * (E10 - E00) + ty * (E11 - E01)) * ihs;
    float dhdy = ((1.0f - tx) * (E01 - E00) + tx * (E11 - E10)) * ihs;
    float inv = pf_rcp(pf_sqrt(dhdx * dhdx + dhdy * dhdy + 1.0f));
    n[0] = -dhdx * inv; n[1] = -dhdy * inv; n[2] = inv;
    if (steep) {
      // 0: x-far (normal -x), 1: x-near (+x), 2: y-far (-y), 3: y-near (+y)
      const float top[4] = {(1.0f - ty) * H10 + ty * H11, (1.0f - ty) * H00 + ty * H01, (1.0f - tx) * H01 + tx * H11, (1.0f - tx) * H00 + tx * H10};
      const float bot[4] = {(1.0f - ty) * E10 + ty * E11, (1.0f - ty) * E00 + ty * E01, (1.0f - tx) * E01 + tx * E11, (1.0f - tx) * E00 + tx * E10};
      const float gap[4] = {(1.0f - tx) * hsc, tx * hsc, (1.0f - ty) * hsc, ty * hsc};
      float best = 1e9f, bh = 1.0f, bz = 0.0f; int bf = -1;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const float dz = fmaxf(wz - top[f], 0.0f);
        const float len = pf_sqrt(gap[f] * gap[f] + dz * dz);
        const float dist = len - rad;
        const bool take = (top[f] - bot[f] > thr) && dist < best;
        const float il = pf_rcp(fmaxf(len, 1e-9f));
        const bool deg = !(len > 1e-9f);
        best = take ? dist : best; bf = take ? f : bf;
        bh = take ? (deg ? 1.0f : gap[f] * il) : bh; bz = take ? (deg ? 0.0f : dz * il) : bz;
      }
      if (bf >= 0) {
        *wd = best; *cellface += bf;
        wn[0] = bf == 0 ? -bh : (bf == 1 ? bh : 0.0f);
        wn[1] = bf == 2 ? -bh : (bf == 3 ? bh : 0.0f);
        wn[2] = bz;
      }
    }
  }
  // closest points of two segments (Ericson 5.1.9): v = c1 - c2, and c2
  PF_DEV void segment_segment(const float* p1, const float* q1, const float* p2, const float* q2, float* v, float* c2) {
    const float eps = 1e-12f;
    float d1[3], d2[3], r[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) { d1[t] = q1[t] - p1[t]; d2[t] = q2[t] - p2[t]; r[t] = p1[t] - p2[t]; }
    const float a = pf_dot3(d1, d1), e = pf_dot3(d2, d2), f = pf_dot3(d2, r);
    float sp, tp;
    if (a <= eps && e <= eps) { sp = 0.0f; tp = 0.0f; }
    else if (a <= eps) { sp = 0.0f; tp = pf_clip(f / e, 0.0f, 1.0f); }
    else {
      const float cc = pf_dot3(d1, r);
      if (e <= eps) { tp = 0.0f; sp = pf_clip(-cc / a, 0.0f, 1.0f); }
      else {
        const float b = pf_dot3(d1, d2), den = a * e - b * b;
        sp = den > eps ? pf_clip((b * f - cc * e) / den, 0.0f, 1.0f) : 0.0f;
        tp = (b * sp + f) / e;
        if (tp < 0.0f) { tp = 0.0f; sp = pf_clip(-cc / a, 0.0f, 1.0f); }
        else if (tp > 1.0f) { tp = 1.0f; sp = pf_clip((b - cc) / a, 0.0f, 1.0f); }
      }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) { c2[t] = p2[t] + tp * d2[t]; v[t] = (p1[t] + sp * d1[t]) - c2[t]; }
  }

  PF_DEV float rnd(const PfConfig& c, const PfLayout& L, const PfInject& inj, int env, long long step, int slot) {
    if (inj.d_rand) return inj.d_rand[(size_t)env * L.rand_floats + slot];
    return pf_philox_slot(c.seed_lo, c.seed_hi, (uint32_t)(c.env_index_base + env), (uint32_t)step, 0u, (uint32_t)slot);
  }

  // Uniform draws for the consecutive slots [first, first + count): one Philox block (four slots) per lane into LDS, instead of
  // one block per lane PER SLOT.  noise_at(slot) then returns exactly what rnd(slot) would.  (Tests that inject the draws
  // bypass it.)
  PF_DEV void noise_prepare(Shared& sh, const PfConfig& c, const PfInject& inj, int env, long long step, int first, int count) {
    if (inj.d_rand) return;
    const int b0 = first >> 2, nb = ((first + count - 1) >> 2) - b0 + 1;
    float* sc = noise_of(sh);
    PF_LANES(b, nb) pf_philox_block(c.seed_lo, c.seed_hi, (uint32_t)(c.env_index_base + env), (uint32_t)step, 0u, (uint32_t)(b0 + b), sc + 4 * b);
    PF_SYNC();
  }
  PF_DEV float noise_at(Shared& sh, const PfLayout& L, const PfInject& inj, int env, int first, int slot) {
    if (inj.d_rand) return inj.d_rand[(size_t)env * L.rand_floats + slot];
    return noise_of(sh)[slot - 4 * (first >> 2)];
  }

  // ------------------------------------------------------------------ actuator network (anymal.py:71-78)
  // per joint: x = [a*scale + q0 - q, qd] * in_scale -> LSTM(2->8) x 2 layers (state h,c persists, zeroed on reset) ->
  // Linear(8->1) * out_scale.  No torque clip on this branch.  Lanes: (joint, gate row) pairs, then (joint, unit) pairs.
  // LSTM gate activations: v_exp_f32-based forms (absolute error ~1e-7) instead of the library expf / tanhf and an IEEE division --
  // 5 per unit, 96 units per layer, every substep.  The host build keeps libm.
  PF_DEV float sigm(float x) {
#if defined(PF_HOST_EMU)
    return 1.0f / (1.0f + expf(-x));
#else
    return pf_rcp(1.0f + __expf(-x));
#endif
  }
  PF_DEV float tanh_(float x) {
#if defined(PF_HOST_EMU)
    return tanhf(x);
#else
    const float xc = pf_clip(x, -15.0f, 15.0f);              // exp(2 x) stays finite; tanh is +-1 to float precision beyond
    return 1.0f - 2.0f * pf_rcp(__expf(2.0f * xc) + 1.0f);
#endif
  }
  // The three 32 x 8 weight blocks are staged TRANSPOSED in LDS (stage_actuator): lane (joint, gate row g) reads element t of its
  // row at base + 32 t + g -- the rows of a wavefront's lanes lie side by side -- instead of base + 8 g + t, where 32 rows share
  // four banks (an 8-way conflict on every weight read: most of the quadruped kernel's 8.8 % LDS conflict cycles, round 3).
  static constexpr int GST = ND * 8 + 8;
  PF_DEV int act_w(int base, int g, int t) { return base + 32 * t + g; }
  PF_DEV void stage_actuator(float* dst, const float* __restrict__ src) {
    PF_LANES(i, PF_ACT_PARAMS) {
      int o = i;
      const int blk = (i >= PF_ACT_W_HH0 && i < PF_ACT_W_HH0 + 256) ? PF_ACT_W_HH0 : (i >= PF_ACT_W_IH1 && i < PF_ACT_W_IH1 + 256) ? PF_ACT_W_IH1
                    : (i >= PF_ACT_W_HH1 && i < PF_ACT_W_HH1 + 256) ? PF_ACT_W_HH1 : -1;
      if (blk >= 0) { const int r = i - blk; o = act_w(blk, r >> 3, r & 7); }
      if (i >= PF_ACT_W_IH0 && i < PF_ACT_W_IH0 + 64) { const int r = i - PF_ACT_W_IH0; o = act_w(PF_ACT_W_IH0, r >> 1, r & 1); }      // 32 x 2, same rule
      dst[o] = src[i];
    }
  }
  // The actuator state inside the LDS record: joint j's 32 floats [layer][h(8) c(8)] are rotated by 8 j (rec_slot below, applied by
  // the record load / store), so the (joint, unit) lanes of the unit phase -- four joints per 32 lanes, all at the same offset of
  // their block -- fall on 32 different banks instead of 8 (a 4-way conflict on every state read and write, and a 12-way one in the
  // old per-joint output phase: ~150 of the ~190 conflict cycles per actuator call, profiles/r4_step_kernel_pmc.md).  Global
  // memory keeps the plain [joint][layer][h c] order.
  #ifndef PF_ACT_ROT
#define PF_ACT_ROT 0
#endif
#ifndef PF_ACT_DPPLIN
#define PF_ACT_DPPLIN 1
#endif
  PF_DEV int act_st(int j, int k8) { return LC::lstm_state + 32 * j + (PF_ACT_ROT ? ((k8 + 8 * j) & 31) : k8); }      // k8: 0 h0, 8 c0, 16 h1, 24 c1
  PF_DEV int rec_slot(int i) {
    if constexpr (LR) {
      const int r = i - LC::lstm_state;
      if (PF_ACT_ROT && (unsigned)r < (unsigned)(32 * ND)) return LC::lstm_state + (r & ~31) + ((r + ((r >> 5) << 3)) & 31);
    }
    return i;
  }
  PF_DEV void actuator_torques(Shared& sh, const float* P, const PfConfig& c) {
    float* S = sh.rec;
    for (int layer = 0; layer < 2; ++layer) {
      PF_LANES(e, ND * 32) {
        int j = e >> 5, g = e & 31;
        const float* h0 = S + act_st(j, 0);
        float acc;
        if (layer == 0) {
          float x0 = ((S[LC::actions + j] * c.action_scale + c.default_dof_pos[j]) - S[LC::dof_pos + j]) * P[PF_ACT_IN_SCALE];
          float x1 = S[LC::dof_vel + j] * P[PF_ACT_IN_SCALE + 1];
          acc = P[PF_ACT_B_IH0 + g] + P[PF_ACT_B_HH0 + g] + P[act_w(PF_ACT_W_IH0, g, 0)] * x0 + P[act_w(PF_ACT_W_IH0, g, 1)] * x1;
          for (int t = 0; t < 8; ++t) acc += P[act_w(PF_ACT_W_HH0, g, t)] * h0[t];
        } else {
          const float* h1 = S + act_st(j, 16);
          acc = P[PF_ACT_B_IH1 + g] + P[PF_ACT_B_HH1 + g];
          for (int t = 0; t < 8; ++t) acc += P[act_w(PF_ACT_W_IH1, g, t)] * h0[t] + P[act_w(PF_ACT_W_HH1, g, t)] * h1[t];
        }
        sh.gates[(g >> 3) * GST + 8 * j + (g & 7)] = acc;
      }
      PF_SYNC();
      PF_LANES(e, ND * 8) {
        int j = e >> 3, u = e & 7;
        const float* G = sh.gates + e;
        float* hs = S + act_st(j, layer * 16);
        float* cs = S + act_st(j, layer * 16 + 8);
        float ig = sigm(G[0]), fg = sigm(G[GST]), gg = tanh_(G[2 * GST]), og = sigm(G[3 * GST]);
        float cn = fg * cs[u] + ig * gg;
        cs[u] = cn;
        const float hn = og * tanh_(cn);
        hs[u] = hn;
#if !defined(PF_HOST_EMU) && PF_ACT_DPPLIN
        // Linear(8 -> 1) on the second layer's output while it is still in registers: the eight unit lanes of a joint are half a
        // DPP row, three v_add_f32_dpp sum them (whole groups are active in both passes of the lane loop)
        if (layer == 1) {
          float y = P[PF_ACT_W_LIN + u] * hn;
          y += pf_dpp<0xB1>(y);
          y += pf_dpp<0x4E>(y);
          y += pf_dpp<0x141>(y);
          if (u == 0) S[LC::torques + j] = (P[PF_ACT_B_LIN] + y) * P[PF_ACT_OUT_SCALE];
        }
#endif
      }
      PF_SYNC();
    }
#if defined(PF_HOST_EMU) || !PF_ACT_DPPLIN
    PF_LANES(j, ND) {
      const float* h2 = S + act_st(j, 16);
      // the device sum's shape: pairs, quads, halves
      const float p01 = P[PF_ACT_W_LIN + 0] * h2[0] + P[PF_ACT_W_LIN + 1] * h2[1], p23 = P[PF_ACT_W_LIN + 2] * h2[2] + P[PF_ACT_W_LIN + 3] * h2[3];
      const float p45 = P[PF_ACT_W_LIN + 4] * h2[4] + P[PF_ACT_W_LIN + 5] * h2[5], p67 = P[PF_ACT_W_LIN + 6] * h2[6] + P[PF_ACT_W_LIN + 7] * h2[7];
      const float y = (p01 + p23) + (p45 + p67);
      S[LC::torques + j] = (P[PF_ACT_B_LIN] + y) * P[PF_ACT_OUT_SCALE];
    }
    PF_SYNC();
#endif
  }

  // ------------------------------------------------------------------ PD torques (:523-547)
  PF_DEV void torques(Shared& sh, const PfConfig& c, const PfLayout& L, const float* qsrc, int qstride, int voff) {
#if !defined(PF_HOST_EMU)
#pragma clang fp contract(off)
#endif
    PF_LANES(k, ND) {
      // branch-free on purpose: hipcc (ROCm 7.2) miscompiled the if / else-if / else form of this body
      // (lane index left undefined on the 'T' path -> wild load of torque_limits); selects are also cheaper.
      float qp = qsrc[k * qstride], qv = qsrc[k * qstride + voff];
      float as = sh.rec[LC::actions + k] * c.action_scale;
      float lim = c.torque_limits[k];
      float tP = c.p_gains[k] * ((as + c.default_dof_pos[k]) - qp) - c.d_gains[k] * qv;
      float tV = c.p_gains[k] * (as - qv) - c.d_gains[k] * (qv - sh.rec[LC::last_dof_vel + k]) / c.sim_dt;
      float t = c.control_type == PF_CTRL_P ? tP : (c.control_type == PF_CTRL_V ? tV : as);
      sh.rec[LC::torques + k] = pf_clip(t, -lim, lim);
    }
    PF_SYNC();
  }

  // ------------------------------------------------------------------ forward kinematics (+ velocities)
  // body frames of the wavefront that runs the dynamics (sh.Rw / r / ax) or -- KB, DUAL instances only -- of the contact wavefront
  template <bool KB> PF_DEV float (&Rw_of(Shared& sh))[NBD][9] { if constexpr (KB && DUAL) return sh.kb.Rw; else return sh.Rw; }
  template <bool KB> PF_DEV float (&r_of(Shared& sh))[NBD][3] { if constexpr (KB && DUAL) return sh.kb.r; else return sh.r; }
  template <bool KB> PF_DEV float (&ax_of(Shared& sh))[NBD][3] { if constexpr (KB && DUAL) return sh.kb.ax; else return sh.ax; }
  template <bool WITH_VEL, bool KB = false>
  PF_DEV void kinematics(Shared& sh, const ModelS& m, const PfConfig& c, const PfLayout& L) {
    static_assert(!(WITH_VEL && KB), "the contact wavefront needs positions only");
    const float* S = sh.rec;
    float (&kRw)[NBD][9] = Rw_of<KB>(sh); float (&kr)[NBD][3] = r_of<KB>(sh); float (&kax)[NBD][3] = ax_of<KB>(sh);
    // base frame / velocity / bias acceleration: uniform (every lane needs them as the root of its chain)
    float R0[9], V0[6], A0[6];
    {
      float x = S[LC::root + 3], y = S[LC::root + 4], z = S[LC::root + 5], w = S[LC::root + 6];
      R0[0] = 1 - 2 * (y * y + z * z); R0[1] = 2 * (x * y - z * w); R0[2] = 2 * (x * z + y * w);
      R0[3] = 2 * (x * y + z * w); R0[4] = 1 - 2 * (x * x + z * z); R0[5] = 2 * (y * z - x * w);
      R0[6] = 2 * (x * z - y * w); R0[7] = 2 * (y * z + x * w); R0[8] = 1 - 2 * (x * x + y * y);
      if (WITH_VEL) {
        float vb[3] = {S[LC::root + 7], S[LC::root + 8], S[LC::root + 9]}, wb[3] = {S[LC::root + 10], S[LC::root + 11], S[LC::root + 12]};
        float wxv[3]; pf_cross(wb, vb, wxv);
#pragma unroll
        for (int t = 0; t < 3; ++t) { V0[t] = wb[t]; V0[3 + t] = vb[t]; A0[t] = 0.0f; A0[3 + t] = -wxv[t] - c.gravity[t]; }
      }
    }
    PF_ONE {
#pragma unroll
      for (int t = 0; t < 9; ++t) kRw[0][t] = R0[t];
#pragma unroll
      for (int t = 0; t < 3; ++t) { kr[0][t] = 0.0f; kax[0][t] = 0.0f; }
      if (WITH_VEL) {
#pragma unroll
        for (int t = 0; t < 6; ++t) { sh.u.d.V[0][t] = V0[t]; sh.u.d.A[0][t] = A0[t]; sh.u.d.Sj[0][t] = 0.0f; }
      }
    }
    // one lane per root-to-leaf chain: the parent's frame stays in registers, no barrier between tree levels.  (A level-
    // synchronous form with 16 lanes per chain -- one matrix element / vector component per lane, 3-term dot products through
    // LDS -- was built and measured: 133.6 vs 126.3 us per launch.  With two wavefronts per SIMD the kernel is bound by
    // LDS-round-trip latency, not by instruction count: the serial register walk keeps the VALU fed.)
    PF_LANES(ch, m.nchain) {
      float Rp[9], rp[3] = {0.0f, 0.0f, 0.0f}, Vp[6], Ap[6];
#pragma unroll
      for (int t = 0; t < 9; ++t) Rp[t] = R0[t];
      if (WITH_VEL) {
#pragma unroll
        for (int t = 0; t < 6; ++t) { Vp[t] = V0[t]; Ap[t] = A0[t]; }
      }
      const int len = m.chain_len[ch];
      for (int d = 0; d < len; ++d) {
        const int i = m.chain[ch][d];
        float Rj[9];
        if (m.rot_ident[i]) {
#pragma unroll
          for (int t = 0; t < 9; ++t) Rj[t] = Rp[t];
        } else pf_mat_mul(Rp, m.joint_rot[i], Rj);
        float jp[3] = {m.joint_pos[i][0], m.joint_pos[i][1], m.joint_pos[i][2]}, ro[3];
        pf_mat_vec(Rp, jp, ro);
        float ri[3] = {rp[0] + ro[0], rp[1] + ro[1], rp[2] + ro[2]};
        kr[i][0] = ri[0]; kr[i][1] = ri[1]; kr[i][2] = ri[2];
        const int k = m.dof_of_body[i];
        if (k >= 0) {
          float a[3] = {m.joint_axis[i][0], m.joint_axis[i][1], m.joint_axis[i][2]};
          float aw[3]; pf_mat_vec(Rj, a, aw);
          float th = S[LC::dof_pos + k];
          float cs, sn; pf_sincos(th, &sn, &cs);
          const float t1 = 1.0f - cs;
          float Ra[9] = {t1 * a[0] * a[0] + cs, t1 * a[0] * a[1] - sn * a[2], t1 * a[0] * a[2] + sn * a[1],
                         t1 * a[0] * a[1] + sn * a[2], t1 * a[1] * a[1] + cs, t1 * a[1] * a[2] - sn * a[0],
                         t1 * a[0] * a[2] - sn * a[1], t1 * a[1] * a[2] + sn * a[0], t1 * a[2] * a[2] + cs};
          pf_mat_mul(Rj, Ra, Rp);
          kax[i][0] = aw[0]; kax[i][1] = aw[1]; kax[i][2] = aw[2];
          if (WITH_VEL) {
            float sl[3]; pf_cross(ri, aw, sl);
            float qd = S[LC::dof_vel + k];
            float Vi[6];
#pragma unroll
            for (int t = 0; t < 3; ++t) { sh.u.d.Sj[i][t] = aw[t]; sh.u.d.Sj[i][3 + t] = sl[t]; Vi[t] = Vp[t] + qd * aw[t]; Vi[3 + t] = Vp[3 + t] + qd * sl[t]; }
            float ca[3], c1[3], c2[3];
            pf_cross(Vi, aw, ca); pf_cross(Vi, sl, c1); pf_cross(Vi + 3, aw, c2);
#pragma unroll
            for (int t = 0; t < 3; ++t) { Ap[t] += qd * ca[t]; Ap[3 + t] += qd * (c1[t] + c2[t]); }
#pragma unroll
            for (int t = 0; t < 6; ++t) Vp[t] = Vi[t];
          }
        } else {
#pragma unroll
          for (int t = 0; t < 9; ++t) Rp[t] = Rj[t];
          kax[i][0] = 0.0f; kax[i][1] = 0.0f; kax[i][2] = 0.0f;
          if (WITH_VEL) {
#pragma unroll
            for (int t = 0; t < 6; ++t) sh.u.d.Sj[i][t] = 0.0f;
          }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) kRw[i][t] = Rp[t];
        if (WITH_VEL) {
#pragma unroll
          for (int t = 0; t < 6; ++t) { sh.u.d.V[i][t] = Vp[t]; sh.u.d.A[i][t] = Ap[t]; }
        }
        rp[0] = ri[0]; rp[1] = ri[1]; rp[2] = ri[2];
      }
    }
    PF_SYNC();
  }

  // ------------------------------------------------------------------ one physics substep of sim_dt
  // ---- spatial inertia of body i about O (world axes), from its frame (KB: the contact wavefront's own frames): I10 and sh.u.d.I[i]
  template <bool KB>
  PF_DEV void body_inertia(Shared& sh, const ModelS& m, int i, float (&I10)[10]) {
    const float* S = sh.rec;
    float mass = m.mass[i];
    float cl[3] = {m.com[i][0], m.com[i][1], m.com[i][2]};
    float Il[6];
    for (int t = 0; t < 6; ++t) Il[t] = m.inertia[i][t];
    if (i == 0) {   // base mass / COM randomisation (:466-477), inertia scaled with mass
      float added = S[LC::added_mass];
      float sc = (mass + added) / mass;
      mass += added;
      for (int t = 0; t < 3; ++t) cl[t] += S[LC::added_com + t];
      for (int t = 0; t < 6; ++t) Il[t] *= sc;
    }
    const float* R = Rw_of<KB>(sh)[i];
    float cw[3]; pf_mat_vec(R, cl, cw);
    for (int t = 0; t < 3; ++t) cw[t] += r_of<KB>(sh)[i][t];
    float Ilm[9] = {Il[0], Il[3], Il[4], Il[3], Il[1], Il[5], Il[4], Il[5], Il[2]};
    float T[9]; pf_mat_mul(R, Ilm, T);
    float Rt[9] = {R[0], R[3], R[6], R[1], R[4], R[7], R[2], R[5], R[8]};
    float Iw[9]; pf_mat_mul(T, Rt, Iw);
    float c2 = pf_dot3(cw, cw);
    I10[0] = mass; I10[1] = mass * cw[0]; I10[2] = mass * cw[1]; I10[3] = mass * cw[2];
    I10[4] = Iw[0] + mass * (c2 - cw[0] * cw[0]);
    I10[5] = Iw[4] + mass * (c2 - cw[1] * cw[1]);
    I10[6] = Iw[8] + mass * (c2 - cw[2] * cw[2]);
    I10[7] = Iw[1] - mass * cw[0] * cw[1];
    I10[8] = Iw[2] - mass * cw[0] * cw[2];
    I10[9] = Iw[5] - mass * cw[1] * cw[2];
    for (int t = 0; t < 10; ++t) sh.u.d.I[i][t] = I10[t];
  }
  // composite inertias / composite bias wrenches alone (the two halves of the composite pass; same sums, same order)
  PF_DEV void composite_inertias(Shared& sh, const ModelS& m) {
    PF_LANES(e, NBD * 10) {
      int i = e / 10, t = e - 10 * i;
      int end = i + m.subtree[i];
      float acc = 0.0f;
      for (int j = i; j < end; ++j) acc += sh.u.d.I[j][t];
      sh.u.d.Ic[i][t] = acc;
    }
  }
  PF_DEV void composite_wrenches(Shared& sh, const ModelS& m) {
    PF_LANES(e, NBD * 6) {
      int i = e / 6, t = e - 6 * i;
      int end = i + m.subtree[i];
      float acc = 0.0f;
      for (int j = i; j < end; ++j) acc += sh.u.d.f[j][t];
      sh.u.d.fc[i][t] = acc;
    }
  }
  // ---- mass matrix entries (packed lower triangle) from the composite inertias, F and the joint motion subspaces
  PF_DEV void mass_entries(Shared& sh, const ModelS& m) {
    PF_LANES(e, NV * (NV + 1) / 2) {      // entry e of the packed lower triangle: row a = largest a with a (a + 1) / 2 <= e
      int a = (int)((pf_sqrt(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
      a += ((a + 1) * (a + 2) / 2 <= e) ? 1 : 0;      // (the square root may be an ulp off at the row starts)
      a -= (a * (a + 1) / 2 > e) ? 1 : 0;
      const int b = e - a * (a + 1) / 2;
      float v = 0.0f;
      if (a < 6) {
        const float* C = sh.u.d.Ic[0];
        if (a < 3) v = (a == b) ? C[0] : 0.0f;
        else if (b >= 3) {
          int x = a - 3, y = b - 3;
          v = (x == y) ? C[4 + x] : C[7 + (x + y - 1)];      // (0,1)->xy, (0,2)->xz, (1,2)->yz
        } else {
          // M_ang,lin = [h]x : row x = a-3, col y = b
          int x = a - 3, y = b;
          if (x == y) v = 0.0f;
          else if ((x + 1) % 3 == y) v = -C[1 + (3 - x - y)];   // (0,1)->-hz,(1,2)->-hx,(2,0)->-hy
          else v = C[1 + (3 - x - y)];
        }
      } else {
        int ka = a - 6;
        if (b < 6) v = b < 3 ? sh.u.d.F[ka][3 + b] : sh.u.d.F[ka][b - 3];
        else {
          int kb = b - 6, ia = m.body_of_dof[ka], ib = m.body_of_dof[kb];
          if (ib <= ia && ia < ib + m.subtree[ib]) v = pf_dot3(sh.u.d.Sj[ib], sh.u.d.F[ka]) + pf_dot3(sh.u.d.Sj[ib] + 3, sh.u.d.F[ka] + 3);
          else if (ia <= ib && ib < ia + m.subtree[ia]) v = pf_dot3(sh.u.d.Sj[ia], sh.u.d.F[kb]) + pf_dot3(sh.u.d.Sj[ia] + 3, sh.u.d.F[kb] + 3);
        }
      }
      sh.M[a][b] = v;
    }
  }
  // ---- Cholesky factor of sh.M into registers (Lr / dv; CHOLL, LDS forms: in place in sh.M / sh.dinv)
  PF_DEV void factorise(Shared& sh, float (&Lr)[NLT], float (&dv)[NDV]) {
    if constexpr (CHOLR) {
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int t = 0; t <= i; ++t) Lr[lt(i, t)] = sh.M[i][t];
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        float d = Lr[lt(j, j)];
#pragma unroll
        for (int t = 0; t < j; ++t) d -= Lr[lt(j, t)] * Lr[lt(j, t)];
        d = pf_sqrt(fmaxf(d, 1e-12f));
        const float di = pf_rcp(d);
        dv[j] = di;
#pragma unroll
        for (int i = j + 1; i < NV; ++i) {
          float sacc = Lr[lt(i, j)];
#pragma unroll
          for (int t = 0; t < j; ++t) sacc -= Lr[lt(i, t)] * Lr[lt(j, t)];
          Lr[lt(i, j)] = sacc * di;
        }
      }
    } else if constexpr (CHOLD) {
#if !defined(PF_HOST_EMU)
      const int li = PF_LANE_ID & 15;
#pragma unroll
      for (int t = 0; t < NV; ++t) Lr[t] = (li < NV && t <= li) ? sh.M[li < NV ? li : 0][t] : 0.0f;
      float mydi = 0.0f;
      chold_factor<0>(Lr, mydi, li);
      dv[0] = mydi;
#endif
    } else if constexpr (CHOLL) {
#if !defined(PF_HOST_EMU)
      const int li = PF_LANE_ID;
      float r[NV];
#pragma unroll
      for (int t = 0; t < NV; ++t) r[t] = (li < NV && t <= li) ? sh.M[li < NV ? li : 0][t] : 0.0f;
      // right-looking: once column j is scaled, every remaining entry (i, k), j < k <= i, takes its update -L[i][j] L[k][j] at
      // once -- NV - 1 - j independent multiply-adds per column (L[k][j] by v_readlane: wave-uniform) instead of one dependent
      // chain per entry; every entry still receives its updates in the order j = 0, 1, ..: the same sums, bit for bit
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r[j]), j));
        d = pf_sqrt(fmaxf(d, 1e-12f));
        const float di = pf_rcp(d);
        const float lij = r[j] * di;
        r[j] = li > j ? lij : r[j];
        if (li == j) sh.dinv[j] = di;
#pragma unroll
        for (int k = j + 1; k < NV; ++k) {
          const float lkj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r[j]), k));
          const float upd = r[k] - r[j] * lkj;
          r[k] = li >= k ? upd : r[k];
        }
      }
      if (li < NV) {
#pragma unroll
        for (int t = 0; t < NV; ++t) if (t < li) sh.M[li][t] = r[t];
      }
      PF_SYNC();
#endif
    } else {
      // in place in LDS, one column per phase
      for (int j = 0; j < NV; ++j) {
        float d = sh.M[j][j];
        for (int t = 0; t < j; ++t) d -= sh.M[j][t] * sh.M[j][t];
        d = pf_sqrt(fmaxf(d, 1e-12f));
        float di = pf_rcp(d);
        PF_LANES(ii, NV - j - 1) {
          int i = j + 1 + ii;
          float s = sh.M[i][j];
          for (int t = 0; t < j; ++t) s -= sh.M[i][t] * sh.M[j][t];
          sh.M[i][j] = s * di;
        }
        PF_ONE { sh.dinv[j] = di; }
        PF_SYNC();
      }
    }
  }
  // One substep = three parts: (1) the dynamics side, (2) the contact side, (3) constraint solves + integration.  (1) and (2) are
  // independent of each other; a DUAL instance runs them on two wavefronts at once (run()), everything else one after the other.
  PF_DEV void substep(Shared& sh, const ModelS& m, const PfConfig& c, const PfLayout& L, const PfBuffers& B, bool apply_ext, int env, int sub, int ix0, int iy0) {
    float Lr[NLT], dv[NDV];
    substep_dyn(sh, m, c, L, B, apply_ext, env, Lr, dv);
    const int ncon = substep_contacts<false>(sh, m, c, L, B, env, sub, ix0, iy0);
    substep_solve(sh, m, c, L, B, env, sub, ncon, Lr, dv);
  }
  // (1) kinematics with velocities, per-body inertias and bias wrenches, composites, mass matrix + bias, Cholesky factor (Lr / dv)
  PF_DEV void substep_dyn(Shared& sh, const ModelS& m, const PfConfig& c, const PfLayout& L, const PfBuffers& B, bool apply_ext, int env, float (&Lr)[NLT], float (&dv)[NDV], int handoff = 0) {
    float* S = sh.rec;
    kinematics<true>(sh, m, c, L);
    if constexpr (DUAL && PF_DUAL_SHARED_FK != 0) PF_XSYNC();      // the body frames of this substep are in place: the partner starts its contact side
    PF_STAMP(sh, 1);
    // ---- per-body spatial inertia about O and bias wrench
    // (DUAL: inertias and their composites are position-only -- the partner wavefront builds them, run(), and flags them)
    if constexpr (DUAL && PF_DUAL_INERTIA) PF_HANDOFF_WAIT(sh.flag_x, handoff);
    PF_LANES(i, NBD) {
      float I10[10];
      if constexpr (DUAL && PF_DUAL_INERTIA) { for (int t = 0; t < 10; ++t) I10[t] = sh.u.d.I[i][t]; }
      else body_inertia<false>(sh, m, i, I10);
      float Vi[6], Ai[6], IA[6], IV[6];
      for (int t = 0; t < 6; ++t) { Vi[t] = sh.u.d.V[i][t]; Ai[t] = sh.u.d.A[i][t]; }
      pf_spI_mul(I10, Ai, IA); pf_spI_mul(I10, Vi, IV);
      float x1[3], x2[3], x3[3];
      pf_cross(Vi, IV, x1); pf_cross(Vi + 3, IV + 3, x2); pf_cross(Vi, IV + 3, x3);
      for (int t = 0; t < 3; ++t) { sh.u.d.f[i][t] = IA[t] + x1[t] + x2[t]; sh.u.d.f[i][3 + t] = IA[3 + t] + x3[t]; }
      if (i == 0 && apply_ext)
        for (int t = 0; t < 3; ++t) sh.u.d.f[0][3 + t] -= S[LC::ext_force + t];
    }
    PF_SYNC();
    // ---- composites over DFS-contiguous subtrees: one lane per (body, component)
    if constexpr (DUAL && PF_DUAL_INERTIA) composite_wrenches(sh, m);
    else {
      PF_LANES(e, NBD * 16) {
        int i = e >> 4, t = e & 15;
        int end = i + m.subtree[i];
        float acc = 0.0f;
        if (t < 10) { for (int j = i; j < end; ++j) acc += sh.u.d.I[j][t]; sh.u.d.Ic[i][t] = acc; }
        else { for (int j = i; j < end; ++j) acc += sh.u.d.f[j][t - 10]; sh.u.d.fc[i][t - 10] = acc; }
      }
    }
    PF_SYNC();
    PF_LANES(k, ND) {
      int i = m.body_of_dof[k];
      float Ic[10], Sv[6], Fo[6];
      for (int t = 0; t < 10; ++t) Ic[t] = sh.u.d.Ic[i][t];
      for (int t = 0; t < 6; ++t) Sv[t] = sh.u.d.Sj[i][t];
      pf_spI_mul(Ic, Sv, Fo);
      for (int t = 0; t < 6; ++t) sh.u.d.F[k][t] = Fo[t];
    }
    PF_SYNC();
    PF_STAMP(sh, 2);
    // ---- mass matrix (lower triangle; generalised velocity [v_O, w, qd]) + bias + old velocity
    mass_entries(sh, m);
    PF_LANES(a, NV) {
      float bv, vo;
      if (a < 3) { bv = sh.u.d.fc[0][3 + a]; vo = S[LC::root + 7 + a]; }
      else if (a < 6) { bv = sh.u.d.fc[0][a - 3]; vo = S[LC::root + 7 + a]; }
      else {
        int k = a - 6, i = m.body_of_dof[k];
        bv = pf_dot3(sh.u.d.Sj[i], sh.u.d.fc[i]) + pf_dot3(sh.u.d.Sj[i] + 3, sh.u.d.fc[i] + 3);
        vo = S[LC::dof_vel + k];
      }
      sh.bias[a] = bv; sh.vold[a] = vo;
      float rhs = -bv + (a >= 6 ? S[LC::torques + a - 6] : 0.0f);
      if (B.d_joint_friction && a >= 6)     // regularised Coulomb joint friction (update_frictions, :103-108)
        rhs -= B.d_joint_friction[(size_t)env * ND + a - 6] * pf_clip(vo * pf_rcp(c.joint_friction_vel), -1.0f, 1.0f);
      sh.J[RFREE][a] = rhs; sh.W[RFREE][a] = rhs;
    }
    PF_SYNC();
    PF_STAMP(sh, 3);
    factorise(sh, Lr, dv);
    PF_STAMP(sh, 4);
  }
  // (2) contact candidates, compaction, contact rows (Jacobians and targets).  KB: on the contact wavefront's own body frames.
  // Returns the number of contacts kept.
  template <bool KB>
  PF_DEV int substep_contacts(Shared& sh, const ModelS& m, const PfConfig& c, const PfLayout& L, const PfBuffers& B, int env, int sub, int ix0, int iy0) {
    float* S = sh.rec;
    const float dt = c.sim_dt;
    float (&kRw)[NBD][9] = Rw_of<KB>(sh); float (&kr)[NBD][3] = r_of<KB>(sh); float (&kax)[NBD][3] = ax_of<KB>(sh);
    // ---- contact candidates, in a fixed order: floor contact of sphere s (candidate s), vertical-face contact of sphere s
    //      (nsph + s), self-collision capsule pair p (2 nsph + p); the first MAXC active ones are kept
    const int16_t* hs = B.d_height_samples;
    const int nsph = m.nsph;
    PF_LANES(s, nsph) {
      int i = m.sph_body[s];
      float xs[3]; pf_mat_vec(kRw[i], m.sph_pos[s], xs);
      for (int t = 0; t < 3; ++t) xs[t] += kr[i][t];
      float h, n[3], wn[3], wd;
      int cellface;
      float rad = m.sph_radius[s];
      terrain_query(c, hs, ix0, iy0, S[LC::root] + xs[0], S[LC::root + 1] + xs[1], S[LC::root + 2] + xs[2], rad, &h, n, wn, &wd, &cellface);
      sh.u.c.ccell[s] = cellface;
      sh.u.c.cd[s] = (S[LC::root + 2] + xs[2] - h) * n[2] - rad;
      sh.u.c.cd[KS + s] = wd;
      for (int t = 0; t < 3; ++t) { sh.u.c.sphc[s][t] = xs[t]; sh.u.c.cn[s][t] = n[t]; sh.u.c.cn[KS + s][t] = wn[t]; }
    }
    PF_SYNC();
    int npair = 0;
    if constexpr (SELF) {
      npair = c.self_collisions ? m.npair : 0;
      PF_LANES(p, npair) {
        const int ca = m.pair_a[p], cb = m.pair_b[p];
        const int a0 = m.cap_s0[ca], b0 = m.cap_s0[cb];
        // capsules whose midpoints are further apart than their reach cannot be within contact_offset: the closest-point
        // computation is skipped (by the whole wavefront when no pair of its robots is near -- the usual case)
        const float *pa0 = sh.u.c.sphc[a0], *pa1 = sh.u.c.sphc[m.cap_s1[ca]], *pb0 = sh.u.c.sphc[b0], *pb1 = sh.u.c.sphc[m.cap_s1[cb]];
        const float mx = 0.5f * ((pa0[0] + pa1[0]) - (pb0[0] + pb1[0])), my = 0.5f * ((pa0[1] + pa1[1]) - (pb0[1] + pb1[1])), mz = 0.5f * ((pa0[2] + pa1[2]) - (pb0[2] + pb1[2]));
        const float lim = m.pair_reach[p] + c.contact_offset;
        if (mx * mx + my * my + mz * mz > lim * lim) { sh.u.c.pd[p] = 1e9f; continue; }
        float v[3], c2[3];
        segment_segment(pa0, pa1, pb0, pb1, v, c2);
        const float len = pf_sqrt(pf_dot3(v, v)), rb = m.sph_radius[b0];
        const float dist = len - m.sph_radius[a0] - rb;
        const bool ok = len > 1e-9f;
        const float il = pf_rcp(fmaxf(len, 1e-9f));
        float n[3] = {ok ? v[0] * il : 0.0f, ok ? v[1] * il : 1.0f, ok ? v[2] * il : 0.0f};
        sh.u.c.pd[p] = dist;
        for (int t = 0; t < 3; ++t) { sh.u.c.pn[p][t] = n[t]; sh.u.c.px[p][t] = c2[t] + (rb + 0.5f * dist) * n[t]; }
      }
      PF_SYNC();
    }
    const int ncand = 2 * nsph + npair;
    auto cand_dist = [&](int cand) { return cand < nsph ? sh.u.c.cd[cand] : (cand < 2 * nsph ? sh.u.c.cd[KS + cand - nsph] : sh.u.c.pd[SELF ? cand - 2 * nsph : 0]); };
    int ncon = 0;
    unsigned long long cset[NCW];      // candidates 0..63 (/ 64..127) -- diagnostic output only
#pragma unroll
    for (int w = 0; w < NCW; ++w) cset[w] = 0ull;
#if !defined(PF_HOST_EMU)
    {
      // compaction by ballot: lane l tests candidates l, l + PF_W, ...; its slot is the number of active candidates before it
      const unsigned long long below = (1ull << PF_LANE_ID) - 1ull;
      const int half_shift = (PF_W == 64) ? 0 : ((int)threadIdx.x & 32);
      for (int c0 = 0; c0 < ncand; c0 += PF_W) {
        const int cand = c0 + PF_LANE_ID;
        const bool act = cand < ncand && cand_dist(cand) < c.contact_offset;
        unsigned long long grp = __ballot(act) >> half_shift;
        if (PF_W < 64) grp &= 0xFFFFFFFFull;
        const int idx = ncon + __popcll(grp & below);
        if (act && idx < MAXC) sh.con_cand[idx] = cand;
        ncon += __popcll(grp);
        if (NCW == 1 || c0 < 64) cset[0] |= grp << (c0 & 63); else cset[NCW - 1] |= grp << (c0 & 63);
      }
    }
#else
    for (int cand = 0; cand < ncand; ++cand)
      if (cand_dist(cand) < c.contact_offset) { if (ncon < MAXC) sh.con_cand[ncon] = cand; ++ncon; cset[NCW == 1 ? 0 : cand >> 6] |= 1ull << (cand & 63); }
#endif
    if (ncon > MAXC) {      // candidates beyond the cap are not contacts: drop them from the reported set
      // ... and count the event: [PF_STAT_CAP_HITS] (env, substep) pairs with more active candidates than the solver keeps,
      // [PF_STAT_CAP_DROPPED] the candidates dropped (tools/train_check.py reports the fraction; profiles/r3_contact_cap.txt)
      PF_ONE { if (B.d_reset_stats) { PF_ATOMIC_ADD(B.d_reset_stats + PF_STAT_CAP_HITS, 1.0f); PF_ATOMIC_ADD(B.d_reset_stats + PF_STAT_CAP_DROPPED, (float)(ncon - MAXC)); } }
      int left = MAXC;
#pragma unroll
      for (int w = 0; w < NCW; ++w) {
        unsigned long long kept = 0ull, rest = cset[w];
        for (; left > 0 && rest != 0ull; --left) { kept |= rest & (~rest + 1ull); rest &= rest - 1ull; }
        cset[w] = kept;
      }
      ncon = MAXC;
    }
    if (B.d_contact_set) {      // diagnostic (parity tests): which candidates were kept, and in which cell / on which face
      unsigned long long cells = 0ull;
      for (int ci = 0; ci < ncon; ++ci) {
        const int cand = sh.con_cand[ci];
        if (cand < 2 * nsph) { const int cf = sh.u.c.ccell[cand < nsph ? cand : cand - nsph]; cells += (unsigned long long)(cand + 1) * (unsigned long long)(cand < nsph ? (cf & ~3) : cf); }
      }
      PF_ONE {
        uint64_t* o = B.d_contact_set + ((size_t)env * c.decimation + sub) * PF_CSET_WORDS;
        o[0] = cset[0]; o[2] = cells; o[3] = NCW > 1 ? cset[NCW - 1] : 0ull;
      }
    }
    PF_SYNC();
    PF_STAMP(sh, 5);
    // ---- contact rows: Jacobians, targets, then W = M^-1 J^T (one lane per row; + the free-acceleration row)
    const float rest = 0.5f * c.terrain_restitution;
    PF_LANES(rr, 3 * ncon) {
      int ci = rr / 3, d = rr - 3 * ci;
      const int cand = sh.con_cand[ci];
      int i, ib = -1, irep, ibrep = -1;      // bodies of the dynamics / bodies the force is reported on
      float n[3], xc[3], dist;
      if (cand < 2 * nsph) {
        const int s = cand < nsph ? cand : cand - nsph, slot = cand < nsph ? cand : KS + s;
        i = m.sph_body[s]; irep = m.sph_rep[s];
        const float rad = m.sph_radius[s];
        dist = sh.u.c.cd[slot];
        for (int t = 0; t < 3; ++t) { n[t] = sh.u.c.cn[slot][t]; xc[t] = sh.u.c.sphc[s][t] - rad * n[t]; }
      } else {
        const int p = SELF ? cand - 2 * nsph : 0;
        i = m.sph_body[m.cap_s0[m.pair_a[p]]]; ib = m.sph_body[m.cap_s0[m.pair_b[p]]];
        irep = m.sph_rep[m.cap_s0[m.pair_a[p]]]; ibrep = m.sph_rep[m.cap_s0[m.pair_b[p]]];
        dist = sh.u.c.pd[p];
        for (int t = 0; t < 3; ++t) { n[t] = sh.u.c.pn[p][t]; xc[t] = sh.u.c.px[p][t]; }
      }
      float ref[3] = {fabsf(n[0]) < 0.9f ? 1.0f : 0.0f, fabsf(n[0]) < 0.9f ? 0.0f : 1.0f, 0.0f};
      float t1[3]; pf_cross(n, ref, t1);
      float il = pf_rcp(pf_sqrt(pf_dot3(t1, t1)));
      for (int t = 0; t < 3; ++t) t1[t] *= il;
      float t2[3]; pf_cross(n, t1, t2);
      float dd[3];
      for (int t = 0; t < 3; ++t) dd[t] = d == 0 ? n[t] : (d == 1 ? t1[t] : t2[t]);
      for (int t = 0; t < 3; ++t) sh.cdir[ci][3 * d + t] = dd[t];
      if (d == 0) { sh.cbody[ci] = irep; sh.cbody_b[ci] = ibrep; }
      float row[NV];
#pragma unroll
      for (int a = 0; a < NV; ++a) row[a] = 0.0f;
      const float gnd = ib < 0 ? 1.0f : 0.0f;       // against the terrain the base columns see the contact point
      row[0] = gnd * dd[0]; row[1] = gnd * dd[1]; row[2] = gnd * dd[2];
      float xa[3]; pf_cross(xc, dd, xa);
      row[3] = gnd * xa[0]; row[4] = gnd * xa[1]; row[5] = gnd * xa[2];
#pragma unroll
      for (int k = 0; k < ND; ++k) {
        int j = m.body_of_dof[k];
        // +1 for an ancestor of body a, -1 for an ancestor of body b (relative velocity at the shared contact point)
        float sg = (j <= i && i < j + m.subtree[j]) ? 1.0f : 0.0f;
        if (SELF) sg -= (j <= ib && ib < j + m.subtree[j]) ? 1.0f : 0.0f;
        if (sg != 0.0f) {
          float rel[3] = {xc[0] - kr[j][0], xc[1] - kr[j][1], xc[2] - kr[j][2]};
          float cr[3]; pf_cross(rel, dd, cr);
          row[6 + k] = sg * pf_dot3(kax[j], cr);
        }
      }
      float un0 = 0.0f;
#pragma unroll
      for (int a = 0; a < NV; ++a) {      // (the velocity before the substep, as the dynamics side copies it into sh.vold)
        sh.J[rr][a] = row[a]; sh.W[rr][a] = row[a]; un0 += row[a] * (a < 6 ? S[LC::root + 7 + a] : S[LC::dof_vel + (a < 6 ? 0 : a - 6)]);
      }
      float bn = 0.0f;
      if (d == 0) {
        const float idt = pf_rcp(dt);
        bn = dist >= 0.0f ? -dist * idt : fminf(0.2f * (-dist) * idt, c.max_depenetration_vel);
        if (un0 < -c.bounce_threshold) bn = fmaxf(bn, -rest * un0);
      }
      sh.rb[rr] = bn; sh.lam[rr] = 0.0f;
    }
    PF_SYNC();
    PF_STAMP(sh, 6);
    return ncon;
  }
  // (3) W = M^-1 J^T, free velocity, joint-limit rows, Delassus blocks, projected Gauss-Seidel, integration, contact forces
  PF_DEV void substep_solve(Shared& sh, const ModelS& m, const PfConfig& c, const PfLayout& L, const PfBuffers& B, int env, int sub, int ncon, const float (&Lr)[NLT], const float (&dv)[NDV]) {
    float* S = sh.rec;
    const float dt = c.sim_dt;
    solve_rows(sh, 3 * ncon, true, Lr, dv);
    PF_SYNC();
    PF_STAMP(sh, 7);
    // ---- free velocity
    PF_LANES(a, NV) sh.vel[a] = sh.vold[a] + dt * sh.W[RFREE][a];
    PF_SYNC();
    // ---- joint-limit rows (need the free velocity)
    PF_LANES(k, ND) {
      float q = S[LC::dof_pos + k], qn = q + dt * sh.vel[6 + k];
      float lo = m.q_lo[k], hi = m.q_hi[k];
      int side = 0;
      if (q < lo || qn < lo) side = 1; else if (q > hi || qn > hi) side = -1;
      sh.lim_side[k] = side;
    }
    PF_SYNC();
    int nlim = 0;
    for (int k = 0; k < ND; ++k) nlim += sh.lim_side[k] != 0 ? 1 : 0;
    if (B.d_contact_set) {      // diagnostic (parity tests): which joint-limit rows are active
      unsigned long long lset = 0ull;
      for (int k = 0; k < ND; ++k) lset |= (unsigned long long)(sh.lim_side[k] > 0 ? 1 : (sh.lim_side[k] < 0 ? 2 : 0)) << (2 * k);
      PF_ONE { B.d_contact_set[((size_t)env * c.decimation + sub) * PF_CSET_WORDS + 1] = lset; }
    }
    PF_LANES(k, ND) {
      int side = sh.lim_side[k];
      if (side == 0) continue;
      int idx = 0;
      for (int t = 0; t < k; ++t) idx += sh.lim_side[t] != 0 ? 1 : 0;
      int rr = CROWS + idx;
      sh.lim_dof[idx] = k;
      for (int a = 0; a < NV; ++a) { float v = (a == 6 + k) ? (float)side : 0.0f; sh.J[rr][a] = v; sh.W[rr][a] = v; }
      float q = S[LC::dof_pos + k];
      float viol = side > 0 ? m.q_lo[k] - q : q - m.q_hi[k];
      const float idt = pf_rcp(dt);
      sh.rb[rr] = viol > 0.0f ? fminf(0.2f * viol * idt, 10.0f * c.max_depenetration_vel) : viol * idt;
      sh.lam[rr] = 0.0f;
    }
    PF_SYNC();
    solve_rows_range(sh, CROWS, nlim, Lr, dv);
    PF_SYNC();
    PF_STAMP(sh, 8);
    // ---- per-contact 3x3 Delassus blocks
    PF_LANES(e, ncon * 9) {
      int ci = e / 9, ab = e - 9 * ci, a = ab / 3, b = ab - 3 * a;
      float s = 0.0f;
      for (int t = 0; t < NV; ++t) s += sh.J[3 * ci + a][t] * sh.W[3 * ci + b][t];
      sh.cA[ci][ab] = s;
    }
    PF_SYNC();
    PF_STAMP(sh, 9);
    // ---- projected Gauss-Seidel (uniform impulse math, lane-parallel velocity update)
    // Every lane runs the whole sweep redundantly with the velocity in registers (static indices); impulses live in LDS
    // and are stored by all lanes with the same value (each lane later re-reads what it wrote), so the dependent
    // updates need no barrier; J and W rows are LDS broadcast reads.
    const float mu = 0.5f * (S[LC::friction] + c.terrain_friction);
#if !defined(PF_HOST_EMU)
    if constexpr (NV <= 32) {
      // GPU form: lane t of each 16-lane DPP row (NV <= 16) or 32-lane group (NV <= 32) owns v[t] and column t of every J / W row
      // (loop-invariant, in registers with the impulses); a row's J.v is one multiply + a 4-step DPP butterfly whose result
      // is bit-identical in all lanes (IEEE add is commutative and every lane walks the same tree), so the uniform impulse
      // math stays consistent.  No LDS traffic inside the sweep except the broadcast reads of rb / 1/A / the Delassus block.
      const int tl = PF_LANE_ID & (NV <= 16 ? 15 : 31);
      const bool tv = tl < NV;
      auto rsum = [](float x) { if constexpr (NV <= 16) return pf_row16_sum(x); else return pf_group32_sum(x); };
      float v = tv ? sh.vel[tl] : 0.0f;
      // joint-limit rows: the first LREG live in registers like the contact rows; the 12-DOF instance keeps the rest in LDS
      // (J = +-e_k is rebuilt from lim_dof / lim_side, W and the impulse are read where they lie) -- 24 VGPRs the
      // quadruped kernel does not have (ANYmal's joints are continuous: no limit row is ever active there)
      constexpr int LREG = DUAL ? PF_DUAL_LREG : (ND <= 6 ? ND : 4);
      float jc[CREG][3], wc[CREG][3], lc[CREG][3], jl[LREG], wl[LREG], ll[LREG];
#pragma unroll
      for (int ci = 0; ci < CREG; ++ci)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const bool on = ci < ncon;
          jc[ci][d] = (on && tv) ? sh.J[3 * ci + d][tl] : 0.0f;
          wc[ci][d] = (on && tv) ? sh.W[3 * ci + d][tl] : 0.0f;
          lc[ci][d] = on ? sh.lam[3 * ci + d] : 0.0f;
        }
#pragma unroll
      for (int li = 0; li < LREG; ++li) {
        const bool on = li < nlim;
        jl[li] = (on && tv) ? sh.J[CROWS + li][tl] : 0.0f;
        wl[li] = (on && tv) ? sh.W[CROWS + li][tl] : 0.0f;
        ll[li] = on ? sh.lam[CROWS + li] : 0.0f;
      }
      // one contact's block update (normal, then the two friction directions with the cone projection); j / w: this lane's
      // components of the three rows, l: the three impulses (uniform)
      auto contact_update = [&](int ci, const float (&j)[3], const float (&w)[3], float (&l)[3]) {
        const int r0 = 3 * ci;
        const float u0 = rsum(j[0] * v), u1 = rsum(j[1] * v), u2 = rsum(j[2] * v);
        const float l0 = l[0], l1o = l[1], l2o = l[2];
        const float ln = fmaxf(0.0f, l0 + (sh.rb[r0] - u0) * sh.rAinv[r0]);
        const float d0 = ln - l0;
        const float u1c = u1 + sh.cA[ci][3] * d0;
        float l1 = l1o - u1c * sh.rAinv[r0 + 1];
        const float d1p = l1 - l1o;
        const float u2c = u2 + sh.cA[ci][6] * d0 + sh.cA[ci][7] * d1p;
        float l2 = l2o - u2c * sh.rAinv[r0 + 2];
        const float lim = mu * ln, nt = pf_sqrt(l1 * l1 + l2 * l2);
        if (nt > lim) { float sc = nt > 0.0f ? lim * pf_rcp(nt) : 0.0f; l1 *= sc; l2 *= sc; }
        const float d1 = l1 - l1o, d2 = l2 - l2o;
        v += w[0] * d0 + w[1] * d1 + w[2] * d2;
        l[0] = ln; l[1] = l1; l[2] = l2;
      };
      for (int it = 0; it < c.solver_iterations; ++it) {
#pragma unroll
        for (int ci = 0; ci < CREG; ++ci) {
          if (ci < ncon) contact_update(ci, jc[ci], wc[ci], lc[ci]);
        }
        if constexpr (CREG < MAXC) {
          for (int ci = CREG; ci < ncon; ++ci) {      // contacts 7, 8 (quadruped): rows and impulses read where they lie in LDS
            float j3[3], w3[3], l3[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) { j3[d] = tv ? sh.J[3 * ci + d][tl] : 0.0f; w3[d] = tv ? sh.W[3 * ci + d][tl] : 0.0f; l3[d] = sh.lam[3 * ci + d]; }
            contact_update(ci, j3, w3, l3);
#pragma unroll
            for (int d = 0; d < 3; ++d) sh.lam[3 * ci + d] = l3[d];      // every lane stores the same impulse: no barrier needed
          }
        }
#pragma unroll
        for (int li = 0; li < LREG; ++li) {
          if (li < nlim) {
            const int rr = CROWS + li;
            const float u = rsum(jl[li] * v);
            const float lo = ll[li];
            const float ln = fmaxf(0.0f, lo + (sh.rb[rr] - u) * sh.rAinv[rr]);
            v += wl[li] * (ln - lo);
            ll[li] = ln;
          }
        }
        if constexpr (LREG < ND) {
          for (int li = LREG; li < nlim; ++li) {      // rows kept in LDS (every lane stores the same impulse: no barrier needed)
            const int rr = CROWS + li;
            const float jv = (tv && tl == 6 + sh.lim_dof[li]) ? (float)sh.lim_side[sh.lim_dof[li]] : 0.0f;
            const float u = rsum(jv * v);
            const float lo = sh.lam[rr];
            const float ln = fmaxf(0.0f, lo + (sh.rb[rr] - u) * sh.rAinv[rr]);
            v += (tv ? sh.W[rr][tl] : 0.0f) * (ln - lo);
            sh.lam[rr] = ln;
          }
        }
      }
      PF_SYNC();
      if (PF_LANE_ID < NV) sh.vel[PF_LANE_ID] = v;
      PF_ONE {
#pragma unroll
        for (int ci = 0; ci < CREG; ++ci) if (ci < ncon) { sh.lam[3 * ci] = lc[ci][0]; sh.lam[3 * ci + 1] = lc[ci][1]; sh.lam[3 * ci + 2] = lc[ci][2]; }
#pragma unroll
        for (int li = 0; li < LREG; ++li) if (li < nlim) sh.lam[CROWS + li] = ll[li];
      }
      PF_SYNC();
    } else
#endif
    {
      float v[NV];
#pragma unroll
      for (int t = 0; t < NV; ++t) v[t] = sh.vel[t];
      for (int it = 0; it < c.solver_iterations; ++it) {
#pragma unroll 1
        for (int ci = 0; ci < ncon; ++ci) {
          const int r0 = 3 * ci;
          float u0 = 0.0f, u1 = 0.0f, u2 = 0.0f;
#pragma unroll
          for (int t = 0; t < NV; ++t) { u0 += sh.J[r0][t] * v[t]; u1 += sh.J[r0 + 1][t] * v[t]; u2 += sh.J[r0 + 2][t] * v[t]; }
          float l0 = sh.lam[r0], l1o = sh.lam[r0 + 1], l2o = sh.lam[r0 + 2];
          float ln = fmaxf(0.0f, l0 + (sh.rb[r0] - u0) * sh.rAinv[r0]);
          float d0 = ln - l0;
          float u1c = u1 + sh.cA[ci][3] * d0;
          float l1 = l1o - u1c * sh.rAinv[r0 + 1];
          float d1p = l1 - l1o;
          float u2c = u2 + sh.cA[ci][6] * d0 + sh.cA[ci][7] * d1p;
          float l2 = l2o - u2c * sh.rAinv[r0 + 2];
          float lim = mu * ln, nt = pf_sqrt(l1 * l1 + l2 * l2);
          if (nt > lim) { float sc = nt > 0.0f ? lim * pf_rcp(nt) : 0.0f; l1 *= sc; l2 *= sc; }
          float d1 = l1 - l1o, d2 = l2 - l2o;
#pragma unroll
          for (int t = 0; t < NV; ++t) v[t] += sh.W[r0][t] * d0 + sh.W[r0 + 1][t] * d1 + sh.W[r0 + 2][t] * d2;
          sh.lam[r0] = ln; sh.lam[r0 + 1] = l1; sh.lam[r0 + 2] = l2;
        }
#pragma unroll 1
        for (int li = 0; li < nlim; ++li) {
          const int rr = CROWS + li;
          float u = 0.0f;
#pragma unroll
          for (int t = 0; t < NV; ++t) u += sh.J[rr][t] * v[t];
          float lo = sh.lam[rr];
          float ln = fmaxf(0.0f, lo + (sh.rb[rr] - u) * sh.rAinv[rr]);
          float d = ln - lo;
#pragma unroll
          for (int t = 0; t < NV; ++t) v[t] += sh.W[rr][t] * d;
          sh.lam[rr] = ln;
        }
      }
      PF_SYNC();
      PF_ONE {
#pragma unroll
        for (int t = 0; t < NV; ++t) sh.vel[t] = v[t];
      }
      PF_SYNC();
    }
    PF_STAMP(sh, 10);
    // ---- integrate (semi-implicit Euler), net contact force per body
    PF_LANES(k, ND) {
      float vl = m.vel_limit[k];
      float v = fminf(fmaxf(sh.vel[6 + k], -vl), vl);
      S[LC::dof_vel + k] = v;
      S[LC::dof_pos + k] += dt * v;
    }
    PF_LANES(i, NB) {
      float Fx = 0.0f, Fy = 0.0f, Fz = 0.0f;
      for (int ci = 0; ci < ncon; ++ci) {
        const float sg = sh.cbody[ci] == i ? 1.0f : ((SELF && sh.cbody_b[ci] == i) ? -1.0f : 0.0f);   // a self-contact pushes the two links apart
        if (sg == 0.0f) continue;
        for (int d = 0; d < 3; ++d) {
          float l = sg * sh.lam[3 * ci + d] * pf_rcp(dt);
          Fx += l * sh.cdir[ci][3 * d]; Fy += l * sh.cdir[ci][3 * d + 1]; Fz += l * sh.cdir[ci][3 * d + 2];
        }
      }
      S[LC::contact_forces + 3 * i] = Fx; S[LC::contact_forces + 3 * i + 1] = Fy; S[LC::contact_forces + 3 * i + 2] = Fz;
    }
    {
      float w0 = sh.vel[3], w1 = sh.vel[4], w2 = sh.vel[5];
      float qx = S[LC::root + 3], qy = S[LC::root + 4], qz = S[LC::root + 5], qw = S[LC::root + 6];
      float hx = 0.5f * dt * w0, hy = 0.5f * dt * w1, hz = 0.5f * dt * w2;
      float nx = qx + (hx * qw + hy * qz - hz * qy);
      float ny = qy + (-hx * qz + hy * qw + hz * qx);
      float nz = qz + (hx * qy - hy * qx + hz * qw);
      float nw = qw + (-hx * qx - hy * qy - hz * qz);
      float inv = pf_rcp(pf_sqrt(nx * nx + ny * ny + nz * nz + nw * nw));
      float p0 = S[LC::root] + dt * sh.vel[0], p1 = S[LC::root + 1] + dt * sh.vel[1], p2 = S[LC::root + 2] + dt * sh.vel[2];
      float v6[6];
      for (int t = 0; t < 6; ++t) v6[t] = sh.vel[t];
      PF_SYNC();
      PF_ONE {
        S[LC::root] = p0; S[LC::root + 1] = p1; S[LC::root + 2] = p2;
        S[LC::root + 3] = nx * inv; S[LC::root + 4] = ny * inv; S[LC::root + 5] = nz * inv; S[LC::root + 6] = nw * inv;
        for (int t = 0; t < 6; ++t) S[LC::root + 7 + t] = v6[t];
      }
    }
    PF_SYNC();
    PF_STAMP(sh, 11);
  }

#if !defined(PF_HOST_EMU)
  // column J of the distributed factorisation (CHOLD), columns J + 1 .. by recursion, right-looking: the pivot leaves lane J by
  // broadcast, every row below scales its entry of column J, and every remaining entry (i, k), J < k <= i, takes its update
  // -L[i][J] L[k][J] at once -- NV - 1 - J independent multiply-adds per lane and column (the left-looking form is one dependent
  // chain per entry).  Every entry still receives its updates in the order J = 0, 1, ..: the same sums, bit for bit, as CHOLR.
  template <int J>
  PF_DEV void chold_factor(float (&Lr)[NLT], float& mydi, int li) {
    if constexpr (CHOLD && J < NV) {
      float d = row_bcast<J>(Lr[J]);
      d = pf_sqrt(fmaxf(d, 1e-12f));
      const float di = pf_rcp(d);
      const float lij = Lr[J] * di;               // L[li][J] for the rows below the pivot
      Lr[J] = li > J ? lij : Lr[J];
      mydi = li == J ? di : mydi;
      chold_update<J, J + 1>(Lr, li);
      chold_factor<J + 1>(Lr, mydi, li);
    }
  }
  // A[li][K] -= L[li][J] * L[K][J] for K = K0 .. NV - 1 (entries of this lane's row right of column J, up to its diagonal)
  template <int J, int K0>
  PF_DEV void chold_update(float (&Lr)[NLT], int li) {
    if constexpr (CHOLD && K0 < NV) {
      const float upd = Lr[K0] - Lr[J] * row_bcast<K0>(Lr[J]);
      Lr[K0] = li >= K0 ? upd : Lr[K0];
      chold_update<J, K0 + 1>(Lr, li);
    }
  }
  // forward substitution, rows I .. NV - 1:  w[I] = (w[I] - sum_{t < I} L[I][t] w[t]) / L[I][I]
  template <int I, int T>
  PF_DEV void chold_fwd_row(const float (&Lr)[NLT], const float (&w)[NV], float& sacc) {
    if constexpr (CHOLD && T < I) { sacc -= row_bcast<I>(Lr[T]) * w[T]; chold_fwd_row<I, T + 1>(Lr, w, sacc); }
  }
  template <int I>
  PF_DEV void chold_fwd(const float (&Lr)[NLT], float mydi, float (&w)[NV]) {
    if constexpr (CHOLD && I < NV) {
      float sacc = w[I];
      chold_fwd_row<I, 0>(Lr, w, sacc);
      w[I] = sacc * row_bcast<I>(mydi);
      chold_fwd<I + 1>(Lr, mydi, w);
    }
  }
  // back substitution, rows I .. 0:  w[I] = (w[I] - sum_{t > I} L[t][I] w[t]) / L[I][I]
  template <int I, int T>
  PF_DEV void chold_bwd_row(const float (&Lr)[NLT], const float (&w)[NV], float& sacc) {
    if constexpr (CHOLD && T < NV) { sacc -= row_bcast<T>(Lr[I]) * w[T]; chold_bwd_row<I, T + 1>(Lr, w, sacc); }
  }
  template <int I>
  PF_DEV void chold_bwd(const float (&Lr)[NLT], float mydi, float (&w)[NV]) {
    if constexpr (CHOLD && I >= 0) {
      float sacc = w[I];
      chold_bwd_row<I, I + 1>(Lr, w, sacc);
      w[I] = sacc * row_bcast<I>(mydi);
      chold_bwd<I - 1>(Lr, mydi, w);
    }
  }
#endif
  // W[row] <- M^-1 J[row]^T by forward/back substitution on the in-place Cholesky factor; Ainv = 1/(J.W)
  // (CHOLD: called by EVERY lane -- `store` says whether this lane's row is real)
  PF_DEV void solve_one(Shared& sh, int rr, const float (&Lr)[NLT], const float (&dv)[NDV], bool store = true) {
    float w[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) w[i] = sh.W[rr][i];
    if constexpr (CHOLD) {
#if !defined(PF_HOST_EMU)
      chold_fwd<0>(Lr, dv[0], w);
      chold_bwd<NV - 1>(Lr, dv[0], w);
#endif
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        float s = w[i];
#pragma unroll
        for (int t = 0; t < i; ++t) s -= (CHOLR ? Lr[CHOLR ? lt(i, t) : 0] : sh.M[i][t]) * w[t];
        w[i] = s * (CHOLR ? dv[CHOLR ? i : 0] : sh.dinv[i]);
      }
#pragma unroll
      for (int i = NV - 1; i >= 0; --i) {
        float s = w[i];
#pragma unroll
        for (int t = i + 1; t < NV; ++t) s -= (CHOLR ? Lr[CHOLR ? lt(t, i) : 0] : sh.M[t][i]) * w[t];
        w[i] = s * (CHOLR ? dv[CHOLR ? i : 0] : sh.dinv[i]);
      }
    }
    if (!store) return;
    float a = 0.0f;
#pragma unroll
    for (int t = 0; t < NV; ++t) { a += sh.J[rr][t] * w[t]; sh.W[rr][t] = w[t]; }
    sh.rAinv[rr] = pf_rcp(fmaxf(a, 1e-9f));
  }
  PF_DEV void solve_rows(Shared& sh, int nrows, bool with_free, const float (&Lr)[NLT], const float (&dv)[NDV]) {
    if constexpr (CHOLD) {      // at most PF_W rows (3 MAXC + 1 <= 32): one pass, every lane active
      const int x = PF_LANE_ID, n = nrows + (with_free ? 1 : 0);
      solve_one(sh, x < nrows ? x : RFREE, Lr, dv, x < n);
    } else {
      PF_LANES(x, nrows + (with_free ? 1 : 0)) { solve_one(sh, x < nrows ? x : RFREE, Lr, dv); }
    }
  }
  PF_DEV void solve_rows_range(Shared& sh, int first, int n, const float (&Lr)[NLT], const float (&dv)[NDV]) {
    if constexpr (CHOLD) {
      const int x = PF_LANE_ID;
      solve_one(sh, first + (x < n ? x : 0), Lr, dv, x < n);
    } else {
      PF_LANES(x, n) { solve_one(sh, first + x, Lr, dv); }
    }
  }

  // ------------------------------------------------------------------ observations (:320-382, :189-192)
  // noise_ready (DUAL): the partner wavefront has drawn the uniforms of slots [r_obs, r_obs + noise_count()) already
  PF_DEV int noise_count(const PfConfig& c, const PfBuffers& B) {
    return c.num_obs + ((B.d_priv && c.measure_heights && c.num_priv_obs > c.num_obs) ? c.num_height_points : 0);
  }
  PF_DEV void observations(Shared& sh, const PfConfig& c, const PfLayout& L, const PfBuffers& B, const PfInject& inj, int env, long long step, bool noise_ready = false) {
#if !defined(PF_HOST_EMU)
#pragma clang fp contract(off)
#endif
    const float* S = sh.rec;
    const float* hts = heights_of(sh);
    const int no = c.num_obs, np_ = c.num_priv_obs, na = c.num_actions;
    if constexpr (LR) {
      // obs = [lin_vel*s (3), ang_vel*s (3), gravity (3), cmd[:3]*scale (3), q-q0 (ND), qd*s (ND), actions (ND), heights (H)]
      const int nprop = 12 + 3 * ND;
      if (c.add_noise) noise_prepare(sh, c, inj, env, step, L.r_obs, no);
      PF_LANES(i, no) {
        float v;
        if (i < 3) v = S[LC::base_lin_vel + i] * c.obs_scale_lin_vel;
        else if (i < 6) v = S[LC::base_ang_vel + i - 3] * c.obs_scale_ang_vel;
        else if (i < 9) v = S[LC::projected_gravity + i - 6];
        else if (i < 12) v = S[LC::commands + i - 9] * c.commands_scale[i - 9];
        else if (i < 12 + ND) v = (S[LC::dof_pos + i - 12] - c.default_dof_pos[i - 12]) * c.obs_scale_dof_pos;
        else if (i < 12 + 2 * ND) v = S[LC::dof_vel + i - 12 - ND] * c.obs_scale_dof_vel;
        else if (i < nprop) v = S[LC::actions + i - 12 - 2 * ND];
        else v = pf_clip((S[LC::root + 2] - 0.5f) - hts[i - nprop], -1.0f, 1.0f) * c.obs_scale_height;
        if (c.add_noise) v += (2.0f * noise_at(sh, L, inj, env, L.r_obs, L.r_obs + i) - 1.0f) * c.obs_noise_vec[i];
        B.d_obs[(size_t)env * no + i] = pf_clip(v, -c.clip_obs, c.clip_obs);
      }
      return;
    }
    // proprioceptive block (:375-382), then -- cfg.terrain.measure_heights_actor (:361, num_obs = 27 + 121) -- the height
    // measurements.  The privileged buffer repeats these entries with the SAME noise draws (:340-345: the actor's noise buffer
    // is concatenated in front of the privileged extras), so the copy below covers it; critic-only heights (num_priv_obs >
    // num_obs) are the separate block with noise of their own.
    const int nprop = 6 + 2 * ND + na + 3;
    if constexpr (DUAL) { if (c.add_noise && !noise_ready) noise_prepare(sh, c, inj, env, step, L.r_obs, noise_count(c, B)); }
    PF_LANES(i, no) {
      float v;
      if (i < 3) v = S[LC::base_ang_vel + i] * c.obs_scale_ang_vel;
      else if (i < 6) v = S[LC::projected_gravity + i - 3];
      else if (i < 6 + ND) v = (S[LC::dof_pos + i - 6] - c.default_dof_pos[i - 6]) * c.obs_scale_dof_pos;
      else if (i < 6 + 2 * ND) v = S[LC::dof_vel + i - 6 - ND] * c.obs_scale_dof_vel;
      else if (i < 6 + 2 * ND + na) v = S[LC::actions + i - 6 - 2 * ND];
      else if (i < nprop) { int k = i - 6 - 2 * ND - na; v = S[LC::commands + k] * c.commands_scale[k]; }
      else v = pf_clip((S[LC::root + 2] - 0.5f) - hts[i - nprop], -1.0f, 1.0f) * c.obs_scale_height;
      if (c.add_noise) v += (2.0f * (DUAL ? noise_at(sh, L, inj, env, L.r_obs, L.r_obs + i) : rnd(c, L, inj, env, step, L.r_obs + i)) - 1.0f) * c.obs_noise_vec[i];
      v = pf_clip(v, -c.clip_obs, c.clip_obs);
      B.d_obs[(size_t)env * no + i] = v;
      if (B.d_priv) B.d_priv[(size_t)env * np_ + i] = v;
    }
    if (B.d_priv && c.measure_heights && np_ > no) {
      if (c.add_noise && !DUAL) noise_prepare(sh, c, inj, env, step, L.r_priv, c.num_height_points);
      PF_LANES(j, c.num_height_points) {    // :367-373
        float h = pf_clip((S[LC::root + 2] - 0.5f) - hts[j], -1.0f, 1.0f) * c.obs_scale_height;
        if (c.add_noise) h += (2.0f * noise_at(sh, L, inj, env, DUAL ? L.r_obs : L.r_priv, L.r_priv + j) - 1.0f) * c.priv_height_noise;
        B.d_priv[(size_t)env * np_ + no + j] = pf_clip(h, -c.clip_obs, c.clip_obs);
      }
    }
  }

  // ------------------------------------------------------------------ _resample_commands (legged_robot.py:337-351)
  PF_DEV void resample_commands(Shared& sh, const PfConfig& c, const PfLayout& L, const PfInject& inj, int env, long long step, int slot) {
#if !defined(PF_HOST_EMU)
#pragma clang fp contract(off)
#endif
    float* S = sh.rec;
    float cx = (c.cmd_hi[0] - c.cmd_lo[0]) * rnd(c, L, inj, env, step, slot) + c.cmd_lo[0];
    float cy = (c.cmd_hi[1] - c.cmd_lo[1]) * rnd(c, L, inj, env, step, slot + 1) + c.cmd_lo[1];
    int k = c.heading_command ? 3 : 2;
    float cz = (c.cmd_hi[k] - c.cmd_lo[k]) * rnd(c, L, inj, env, step, slot + 2) + c.cmd_lo[k];
    float keep = sqrtf(cx * cx + cy * cy) > 0.2f ? 1.0f : 0.0f;      // set small commands to zero
    PF_SYNC();
    PF_ONE { S[LC::commands] = cx * keep; S[LC::commands + 1] = cy * keep; S[LC::commands + k] = cz; }
    PF_SYNC();
  }

  // ------------------------------------------------------------------ reset_idx for this env (:249-299)
  PF_DEV void reset_env(Shared& sh, const ModelS& m, const PfConfig& c, const PfLayout& L, const PfBuffers& B, const PfInject& inj, int env, long long step) {
#if !defined(PF_HOST_EMU)
#pragma clang fp contract(off)
#endif
    float* S = sh.rec;
    if (c.curriculum) {     // :613-636
      float dx = S[LC::root] - S[LC::env_origin], dy = S[LC::root + 1] - S[LC::env_origin + 1];
      float dist = sqrtf(dx * dx + dy * dy);
      bool up = dist > c.terrain_length / 2.0f;
      float cn = sqrtf(S[LC::commands] * S[LC::commands] + S[LC::commands + 1] * S[LC::commands + 1]);
      bool down = (dist < cn * c.max_episode_length_s * 0.5f) && !up;
      int lvl = B.d_terrain_level[env] + (up ? 1 : 0) - (down ? 1 : 0);
      if (lvl >= c.max_terrain_level) {
        float u = rnd(c, L, inj, env, step, L.r_level);
        if (inj.d_rand) lvl = (int)u;
        else { lvl = (int)(u * (float)c.max_terrain_level); lvl = lvl < c.max_terrain_level - 1 ? lvl : c.max_terrain_level - 1; }
      } else lvl = lvl > 0 ? lvl : 0;
      const float* o = B.d_terrain_origins + ((size_t)lvl * c.terrain_num_cols + B.d_terrain_type[env]) * 3;
      float o0 = o[0], o1 = o[1], o2 = o[2];
      PF_SYNC();
      PF_ONE { B.d_terrain_level[env] = lvl; S[LC::env_origin] = o0; S[LC::env_origin + 1] = o1; S[LC::env_origin + 2] = o2; }
      PF_SYNC();
    }
    PF_LANES(k, ND) {       // :549-559
      S[LC::dof_pos + k] = c.default_dof_pos[k] * ((1.5f - 0.5f) * rnd(c, L, inj, env, step, L.r_dof + k) + 0.5f);
      S[LC::dof_vel + k] = 0.0f;
    }
    PF_LANES(i, 13) {       // :566-584
      float v = c.base_init_state[i];
      if (i < 3) v += S[LC::env_origin + i];
      if (i < 2 && c.custom_origins) v += (1.0f - (-1.0f)) * rnd(c, L, inj, env, step, L.r_xy + i) + (-1.0f);
      if (i >= 7) v = (0.5f - (-0.5f)) * rnd(c, L, inj, env, step, L.r_vel + i - 7) + (-0.5f);
      S[LC::root + i] = v;
    }
    if constexpr (LR) resample_commands(sh, c, L, inj, env, step, L.r_cmd_reset);     // legged_robot.py:168
    // :290-299 (inside a step the two `last_*` rows are overwritten again at :225-227; a host reset_idx leaves them zero)
    PF_LANES(k, c.num_actions) S[LC::last_actions + k] = 0.0f;
    PF_LANES(k, ND) S[LC::last_dof_vel + k] = 0.0f;
    PF_LANES(f, NF) {
      S[LC::feet_air_time + f] = 0.0f; S[LC::last_feet_air_time + f] = 0.0f;
      S[LC::cur_max_feet_height + f] = 0.0f; S[LC::last_max_feet_height + f] = 0.0f;
    }
    if constexpr (LR) { PF_LANES(e, 32 * ND) S[LC::lstm_state + e] = 0.0f; }     // anymal.py:56-60
    int nsum = c.num_rewards + (c.termination_scale != 0.0f ? 1 : 0);
    PF_LANES(ri, nsum) {    // :276-280
      if (B.d_reset_stats) PF_ATOMIC_ADD(B.d_reset_stats + ri, S[LC::episode_sums + ri]);
      S[LC::episode_sums + ri] = 0.0f;
    }
    PF_ONE { if (B.d_reset_stats) PF_ATOMIC_ADD(B.d_reset_stats + PF_MAX_REWARDS, 1.0f); }
    PF_SYNC();
  }

  // ------------------------------------------------------------------ the whole step for env `env`
  // Every wavefront stages the workgroup's read-only block itself (same values from every wave: no cross-wave ordering is
  // needed, so PF_SYNC stays wave-local and robots of one workgroup may diverge freely).
#if !defined(PF_HOST_EMU)
  // pf_step_policy: what k_act_heads (pf_ppo.hip) does for one env, by the env's own lanes before they read the action --
  // rsl_rl PPO.act behind the last hidden layer (both thin heads, Normal(mu, std).sample(), log-prob, RolloutStorage.add_transitions
  // for transition t) and the deferred PPO.process_env_step of transition t - 1.  One launch fewer per policy step.
  // SAME SUMS IN THE SAME ORDER as k_act_heads, so that both paths give the same bits: there, lane8 of an 8-lane group adds the
  // products of the 16-byte chunks lane8, lane8 + 8, lane8 + 16, lane8 + 24 of the hidden row in that order and the eight partial
  // sums meet in xor-1 / 2 / 4 butterflies; here lane l holds the product of chunk l and the same additions are made from shuffles.
  PF_DEV void policy_heads(Shared& sh, const PfConfig& c, const PfPolicyHeads& h, int env) {
    float* S = sh.rec;
    const int lane = PF_LANE_ID, wl = (int)threadIdx.x & 63, hb = wl & ~(PF_W - 1), b = lane & 7;
    constexpr int A = ND;      // pf_step_policy checks h.num_actions against the env's (== ND)
    const int Ka = h.ka, Kc = h.kc, nca = Ka >> 2, ncc = Kc >> 2;
    // every global load the sums need goes out first
    float4 xa = make_float4(0.0f, 0.0f, 0.0f, 0.0f), xc = xa, wc = xa;
    if (lane < nca) xa = reinterpret_cast<const float4*>(h.d_ha + (size_t)env * Ka)[lane];
    if (lane < ncc) { xc = reinterpret_cast<const float4*>(h.d_hc + (size_t)env * Kc)[lane]; wc = reinterpret_cast<const float4*>(h.d_wc)[lane]; }
    float p[A], pv = 0.0f;
#pragma unroll
    for (int o = 0; o < A; ++o) {
      p[o] = 0.0f;
      if (lane < nca) {
        const float4 w = reinterpret_cast<const float4*>(h.d_wa + (size_t)o * Ka)[lane];
        p[o] = (xa.x * w.x + xa.y * w.y) + (xa.z * w.z + xa.w * w.w);
      }
    }
    if (lane < ncc) pv = (xc.x * wc.x + xc.y * wc.y) + (xc.z * wc.z + xc.w * wc.w);
    // the previous transition's deferred process_env_step: loads now, the rest at the end
    const bool proc = h.p_rew != nullptr && lane == 0;
    float pr = 0.0f, per = 0.0f, pel = 0.0f, pval = 0.0f; bool pdone = false, pto = false;
    if (proc) {
      pr = h.p_rew[env]; pdone = h.p_dones[env] != 0;
      if (h.p_ep_ret) { per = h.p_ep_ret[env]; pel = h.p_ep_len[env]; }
      if (h.p_time_outs) { pto = h.p_time_outs[env] != 0; pval = h.p_values[env]; }
    }
    // RolloutStorage.add_transitions: the observations the policy acted on (this env's rows)
    // (eight loads in flight per lane, then the stores: the compiler cannot prove that the storage rows do not alias the observation
    // buffers, so a load-store loop would be one memory round trip per iteration)
    { const int O = h.num_obs, P = h.num_critic_obs;
      const float* so = h.d_obs + (size_t)env * O; const float* sc = h.d_critic_obs + (size_t)env * P;
      float* to = h.s_obs + (size_t)env * O; float* tc = h.s_critic_obs + (size_t)env * P;
      for (int i0 = 0; i0 < O + P; i0 += 8 * PF_W) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * PF_W + lane; t[u] = i < O ? so[i] : (i < O + P ? sc[i - O] : 0.0f); }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * PF_W + lane; if (i < O) to[i] = t[u]; else if (i < O + P) tc[i - O] = t[u]; }
      } }
    auto comb = [&](float q, int nch) {
      float s0 = __shfl(q, hb + b, 64);
      const float t1 = __shfl(q, hb + b + 8, 64), t2 = __shfl(q, hb + b + 16, 64), t3 = __shfl(q, hb + b + 24, 64);
      if (b + 8 < nch) s0 += t1;
      if (b + 16 < nch) s0 += t2;
      if (b + 24 < nch) s0 += t3;
      s0 += __shfl_xor(s0, 1, 64); s0 += __shfl_xor(s0, 2, 64); s0 += __shfl_xor(s0, 4, 64);
      return s0;
    };
    float mu[A];
#pragma unroll
    for (int o = 0; o < A; ++o) mu[o] = comb(p[o], nca) + h.d_ba[o];
    const float val = comb(pv, ncc) + h.d_bc[0];
    float lp = 0.0f;
    if (lane < 8) {
#pragma unroll 1
      for (int kk = 0; kk < (A + 7) / 8; ++kk) {
        const int k = lane + 8 * kk;
        if (k < A) {
          float z;
          if (h.d_inject_normal) z = h.d_inject_normal[(size_t)env * A + k];
          else {
            uint32_t r[4];
            pf_philox_raw(h.env_base + (uint32_t)env, h.counter, 1u, (uint32_t)k, h.seed_lo, h.seed_hi, r);
            const float u1 = ((float)(r[0] >> 8) + 1.0f) * (1.0f / 16777216.0f);   // (0,1]
            const float u2 = (float)(r[1] >> 8) * (1.0f / 16777216.0f);
            z = sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
          }
          float m = 0.0f;
#pragma unroll
          for (int o = 0; o < A; ++o) if (o == k) m = mu[o];
          const float sg = h.d_std[k];
          const float a = m + sg * z;
          const float d = (a - m) / sg;
          lp += -0.5f * d * d - logf(sg) - 0.91893853320467274178f;
          h.d_actions_out[(size_t)env * A + k] = a;
          h.s_actions[(size_t)env * A + k] = a; h.s_mu[(size_t)env * A + k] = m; h.s_sigma[(size_t)env * A + k] = sg;
          S[LC::actions + k] = pf_clip(a, -c.clip_actions, c.clip_actions);   // :176-177
        }
      }
    }
    lp += __shfl_xor(lp, 1, 64); lp += __shfl_xor(lp, 2, 64); lp += __shfl_xor(lp, 4, 64);
    if (lane == 0) {
      h.s_values[env] = val; h.s_log_prob[env] = lp;
      if (proc) {      // PPO.process_env_step + the runner's episode bookkeeping (pf_ppo.hip process_store)
        float r = pr;
        if (h.p_ep_ret) {
          float cr = per + r, cl = pel + 1.0f;
          if (pdone) {
            if (h.p_fin_ret) { h.p_fin_ret[env] = cr; h.p_fin_len[env] = cl; }
            else { atomicAdd(h.p_ep_stats, cr); atomicAdd(h.p_ep_stats + 1, cl); atomicAdd(h.p_ep_stats + 2, 1.0f); }
            cr = 0.0f; cl = 0.0f;
          }
          h.p_ep_ret[env] = cr; h.p_ep_len[env] = cl;
        }
        if (h.p_time_outs) r += h.p_gamma * (pval * (pto ? 1.0f : 0.0f));
        h.p_s_rew[env] = r; h.p_s_done[env] = pdone ? 1.0f : 0.0f;
      }
    }
  }
#endif

  // role (DUAL instances): 0 = the wavefront that runs the step (alone: every other instance), 1 = its partner, which runs the
  // contact side of every substep and nothing else
  // shv / slot (DUAL): the workgroup's array of per-robot blocks and this robot's index in it -- see the substep loop
  // hp (pf_step_policy): the policy's tail -- a kernel argument of its own, never written, so that it stays in the kernel-argument segment and is
  // scalar-loaded where the prologue uses it (inside PfStepArgs, which the entry point modifies, its 272 bytes were loaded up front: 113 spilled SGPRs)
  PF_DEV void run(Shared& sh, SharedRO& ro, const PfStepArgs& g, int env, int role = 0, Shared* shv = nullptr, int slot = 0, const PfPolicyHeads* hp = nullptr) {
    const PfConfig& c = *g.cfg; const PfLayout& L = g.L; const PfBuffers& B = g.b; const PfInject& inj = g.inj;
    ModelS& ms = ro.mdl;
    if constexpr (DUAL) { if (role == 1 && (g.mode != 0 || inj.d_root)) return; }      // no substeps: nothing for the partner to do
    if (g.mode == 1 && g.reset_mask && !g.reset_mask[env]) return;      // reset_idx(env_ids): this env is not in the subset
#if defined(PF_STAMPS) && !defined(PF_HOST_EMU)
    if (threadIdx.x == 0) { sh.stamp_last = __builtin_amdgcn_s_memtime(); for (int t = 0; t < PF_NSTAMP; ++t) sh.stamp_acc[t] = 0; }
#endif
    {     // the model in its LDS layout, packed once on the host (pack_model): one flat copy, every load in flight at once
      const uint32_t* src = reinterpret_cast<const uint32_t*>(g.packed);
      uint32_t* dst = reinterpret_cast<uint32_t*>(&ms);
      PF_LANES(i, (int)(sizeof(ModelS) / 4)) dst[i] = src[i];
    }
    if constexpr (LR) { if (c.use_actuator_net && B.d_actuator_params) stage_actuator(ro.act, B.d_actuator_params); }
    const ModelS& m = ms;
    float* S = sh.rec;
    float* grow = B.d_state + (size_t)env * L.rec_floats;
    if constexpr (DUAL) {
      if (role == 1) {      // the contact wavefront: per substep, between the two meetings with its partner
        PF_SYNC();          // (its own copy of the model image is in place)
        for (int sub = 0; sub < c.decimation; ++sub) {
          PF_XSYNC();       // the record holds this substep's state (and, the first time, the step's anchor cell) [SHARED_FK: and the body frames]
          const int ix0 = sh.anchor_x[0], iy0 = sh.anchor_x[1];
          if constexpr (PF_DUAL_SHARED_FK == 0) kinematics<false, true>(sh, m, c, L);
          // the bodies' spatial inertias and their composites depend on positions only, and this wavefront has time to spare:
          // it builds them on its own frames and flags them (the dynamics side picks them up after its velocity pass)
          if constexpr (PF_DUAL_INERTIA != 0) {
            PF_LANES(i, NBD) { float I10[10]; body_inertia<true>(sh, m, i, I10); }
            PF_SYNC();
            composite_inertias(sh, m);
            PF_HANDOFF_SET(sh.flag_x, sub + 1);
          }
          const int nc = substep_contacts<PF_DUAL_SHARED_FK == 0>(sh, m, c, L, B, env, sub, ix0, iy0);
          PF_ONE { sh.ncon_x = nc; }
          PF_XSYNC();       // rows, targets and the contact count are handed over
        }
        // while the partner solves and integrates the last substep: the step's observation noise (state-independent), into the
        // contact scratch, which is dead from here on
        if (c.add_noise) noise_prepare(sh, c, inj, env, g.step_counter, L.r_obs, noise_count(c, B));
        PF_XSYNC();         // the record holds the final state (world coordinates)
        float hs_ = 0.0f;
        if (c.measure_heights) hs_ = height_scan(sh, c, B, env);
        PF_ONE { sh.hsum_x = hs_; }
        PF_XSYNC();         // heights, their sum and the noise draws are handed over (post_physics)
        return;
      }
    }
    // a step (mode 0) reads the read-write part only: everything behind LC::rw_floats is written before it is read
    { const int nload = g.mode == 0 ? LC::rw_floats : L.rec_floats; PF_LANES(i, nload) S[rec_slot(i)] = grow[i]; }
    PF_SYNC();
    if (g.mode == 1) {
      reset_env(sh, m, c, L, B, inj, env, g.step_counter);
      PF_ONE { B.d_episode_length[env] = 0; B.d_reset[env] = 1; }
      PF_LANES(i, LC::rec_used) grow[i] = S[rec_slot(i)];
      return;
    }
    if (g.mode == 2 || g.mode == 3) {      // compute_observations() (:123-129, :320-326): measured_heights are taken afresh (:213 / :324)
      if (c.measure_heights) (void)height_scan(sh, c, B, env);
      PF_SYNC();
      if (g.mode == 2) observations(sh, c, L, B, inj, env, g.step_counter);      // mode 3: a read of the measured_heights attribute, nothing else is written
      return;
    }
#if !defined(PF_HOST_EMU) && !defined(PF_NO_POLICY_HEADS)      // (-DPF_NO_POLICY_HEADS: A/B builds of the launch without the branch, tools/build_variant.sh)
    if (hp && hp->d_ha) policy_heads(sh, c, *hp, env);      // pf_step_policy: the action comes from the policy's heads, computed here
    else
#endif
    PF_LANES(k, c.num_actions) S[LC::actions + k] = pf_clip(g.actions[(size_t)env * c.num_actions + k], -c.clip_actions, c.clip_actions);   // :176-177
    PF_SYNC();
    if (inj.d_root) {
      // Python-layer parity mode: injected sim snapshots stand in for simulate()/refresh (:183-185, :200-202)
      torques(sh, c, L, inj.d_dof_pre + (size_t)env * ND * 2, 2, 1);
      PF_LANES(k, ND) { S[LC::dof_pos + k] = inj.d_dof_post[((size_t)env * ND + k) * 2]; S[LC::dof_vel + k] = inj.d_dof_post[((size_t)env * ND + k) * 2 + 1]; }
      PF_LANES(i, 13) S[LC::root + i] = inj.d_root[(size_t)env * 13 + i];
      PF_LANES(i, NB * 3) S[LC::contact_forces + i] = inj.d_contact[(size_t)env * NB * 3 + i];
      PF_LANES(i, NF * 3) S[LC::feet_pos + i] = inj.d_feet_pos[(size_t)env * NF * 3 + i];
      PF_SYNC();
    } else {
      // The substeps run in coordinates local to the grid cell the base starts the step in: S[root], S[root + 1] hold metres
      // from that cell's corner, so contact geometry is resolved to ~1e-7 m wherever the env sits on the field (float32
      // world coordinates are only good to 4e-6 m at 50 m from the origin).  The world position gets the step's
      // displacement added once at the end.  (Same arithmetic as the oracle's pfo_step.)
      const float gx = S[LC::root], gy = S[LC::root + 1];
      int ix0 = 0, iy0 = 0;
      float lx0 = gx, ly0 = gy;
      if (c.terrain_type != PF_TERRAIN_PLANE && B.d_height_samples) {
#if !defined(PF_HOST_EMU)
#pragma clang fp contract(off)
#endif
        const float cx = (gx + c.border) / c.hscale, cy = (gy + c.border) / c.hscale;
        const float bx = floorf(cx), by = floorf(cy);
        ix0 = (int)bx; iy0 = (int)by;
        lx0 = (cx - bx) * c.hscale; ly0 = (cy - by) * c.hscale;
      }
      PF_SYNC();
      PF_ONE { S[LC::root] = lx0; S[LC::root + 1] = ly0; if constexpr (DUAL) { sh.anchor_x[0] = ix0; sh.anchor_x[1] = iy0; sh.flag_x = 0; sh.park_x[0] = gx; sh.park_x[1] = gy; sh.park_x[2] = lx0; sh.park_x[3] = ly0; } }      // (flag: no substep's token)
      PF_SYNC();
      for (int sub = 0; sub < c.decimation; ++sub) {     // :180-185
        PF_STAMP(sh, 0);
        if constexpr (DUAL && PF_DUAL_SHARED_FK == 0) PF_XSYNC();      // the partner starts this substep's contact side (SHARED_FK: behind the kinematics, substep_dyn)
        bool pd = true;
        if constexpr (LR) { if (c.use_actuator_net && B.d_actuator_params) { actuator_torques(sh, ro.act, c); pd = false; } }
        if (pd) torques(sh, c, L, S + LC::dof_pos, 1, LC::dof_vel - LC::dof_pos);
        PF_STAMP(sh, 12);
        bool ext = sub == 0 && S[LC::ext_force + 3] != 0.0f;
        // register-starved instances (PF_OPAQUE_LANE): the block's LDS address is formed again in every substep behind an opaque copy
        // of the slot index, so that nothing derived from it (base + array offset, + lane stride ...) is hoisted above the loop and
        // carried -- spilled -- across it
        int slot2 = slot;
#if !defined(PF_HOST_EMU)
        if constexpr (PF_OPAQUE_LANE) asm volatile("" : "+v"(slot2));
#endif
        Shared& shl = (PF_OPAQUE_LANE && shv) ? shv[slot2] : sh;
        if constexpr (DUAL) {
          float Lr[NLT], dv[NDV];
          substep_dyn(shl, m, c, L, B, ext, env, Lr, dv, sub + 1);
          PF_XSYNC();                        // ... and has finished it
          PF_STAMP(sh, 5);
          substep_solve(shl, m, c, L, B, env, sub, shl.ncon_x, Lr, dv);
        } else substep(shl, m, c, L, B, ext, env, sub, ix0, iy0);
        if (ext) { PF_ONE { S[LC::ext_force + 3] = 0.0f; } PF_SYNC(); }
      }
      {
        float gx_ = gx, gy_ = gy, lx_ = lx0, ly_ = ly0;
        if constexpr (DUAL) { gx_ = sh.park_x[0]; gy_ = sh.park_x[1]; lx_ = sh.park_x[2]; ly_ = sh.park_x[3]; }
        const float nx = gx_ + (S[LC::root] - lx_), ny = gy_ + (S[LC::root + 1] - ly_);
        PF_SYNC();
        PF_ONE { S[LC::root] = nx; S[LC::root + 1] = ny; }
        PF_SYNC();
      }
      if constexpr (DUAL) PF_XSYNC();      // the partner starts the height scan on the final state
      kinematics<false>(sh, m, c, L);
      PF_LANES(e, NF * 3) {      // origin of the foot's link frame: on its carrier body at feet_off
        // (same operations, same order as the kinematics chain would use for a body of its own: r_foot = r_carrier + R_carrier off)
        const int f = e / 3, t = e - 3 * f, b = m.feet_dyn[f];
        float ro[3]; pf_mat_vec(sh.Rw[b], m.feet_off[f], ro);
        S[LC::feet_pos + e] = S[LC::root + t] + (sh.r[b][t] + ro[t]);
      }
      PF_SYNC();
    }
    PF_STAMP(sh, 13);
    // (everything behind the substeps addresses global memory through an opaque copy of the env index: row pointers are formed
    // here, where they are used, instead of ahead of the loop and carried -- spilled -- across it)
    int env2 = env;
#if !defined(PF_HOST_EMU)
    if constexpr (PF_OPAQUE_LANE) asm volatile("" : "+v"(env2));
#endif
    post_physics(sh, m, c, L, B, inj, env2, g.step_counter, DUAL && !inj.d_root);
    PF_STAMP(sh, 14);
    {
      float* grow2 = B.d_state + (size_t)env2 * L.rec_floats;
      PF_LANES(i, LC::rec_used) grow2[i] = S[rec_slot(i)];      // (the padding behind the last field is never written)
    }
    PF_SYNC();
    PF_STAMP(sh, 15);
#if defined(PF_STAMPS) && !defined(PF_HOST_EMU)
    if (g.stamps && threadIdx.x == 0) for (int t = 0; t < PF_NSTAMP; ++t) g.stamps[(size_t)env * PF_NSTAMP + t] = sh.stamp_acc[t];
#endif
  }

  // ------------------------------------------------------------------ _get_heights (:1048-1075): one scan point per lane
  // fills heights_of(sh); returns the sum over the points of (root z - height) (every lane: the base-height reward's mean)
  PF_DEV float height_scan(Shared& sh, const PfConfig& c, const PfBuffers& B, int env) {
#if !defined(PF_HOST_EMU)
#pragma clang fp contract(off)
#endif
    const float* S = sh.rec;
    float* hts = heights_of(sh);
    const int H = c.num_height_points;
    const float root_z = S[LC::root + 2];
    float qy[4] = {0.0f, 0.0f, S[LC::root + 5], S[LC::root + 6]};
    float nrm = sqrtf(qy[2] * qy[2] + qy[3] * qy[3]);
    nrm = nrm < 1e-9f ? 1e-9f : nrm;
    qy[2] /= nrm; qy[3] /= nrm;
    float hsum = 0.0f;
#if !defined(PF_HOST_EMU)
    if (c.terrain_type != PF_TERRAIN_PLANE) {
      // Same arithmetic per point as the loop below, in two sweeps: every sample of this lane's points (up to PER of them) is
      // requested before the first one is used -- one memory round trip for the scan instead of one per point.
      constexpr int PER = (PF_MAX_HPTS + PF_W - 1) / PF_W;
      const int16_t* hs = B.d_height_samples;
      int at[PER];
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int j = PF_LANE_ID + u * PF_W;
        at[u] = -1;
        if (j < H) {
          float pl[3] = {c.height_px[j], c.height_py[j], 0.0f}, pw[3];
          pf_quat_apply(qy, pl, pw);
          at[u] = height_cell(c, pw[0] + S[LC::root], pw[1] + S[LC::root + 1]);
        }
      }
      int h1[PER], h2[PER], h3[PER];
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int a = at[u] < 0 ? 0 : at[u];
        h1[u] = hs[a]; h2[u] = hs[a + c.hf_cols]; h3[u] = hs[a + 1];
      }
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int j = PF_LANE_ID + u * PF_W;
        if (at[u] >= 0) {
          int h = h1[u] < h2[u] ? h1[u] : h2[u];
          h = h < h3[u] ? h : h3[u];
          const float hv = (float)h * c.vscale;
          hts[j] = hv;
          if (B.d_measured_heights) B.d_measured_heights[(size_t)env * H + j] = hv;      // only when a caller asked for the tensor
          hsum += root_z - hv;
        }
      }
      return PF_WAVE_SUM(hsum);
    }
#endif
    PF_LANES(j, H) {
      float hv = 0.0f;
      if (c.terrain_type != PF_TERRAIN_PLANE) {
        float pl[3] = {c.height_px[j], c.height_py[j], 0.0f}, pw[3];
        pf_quat_apply(qy, pl, pw);
        hv = height_raw(c, B.d_height_samples, pw[0] + S[LC::root], pw[1] + S[LC::root + 1]) * c.vscale;
      }
      hts[j] = hv;
      if (B.d_measured_heights) B.d_measured_heights[(size_t)env * H + j] = hv;      // only when a caller asked for the tensor
      hsum += root_z - hv;
    }
    return PF_WAVE_SUM(hsum);
  }

  // ------------------------------------------------------------------ post_physics_step (:195-230)
  // partner (DUAL): the partner wavefront draws the observation noise and runs the height scan meanwhile; it is met before the rewards
  PF_DEV void post_physics(Shared& sh, const ModelS& m, const PfConfig& c, const PfLayout& L, const PfBuffers& B, const PfInject& inj, int env, long long step, bool partner = false) {
#if !defined(PF_HOST_EMU)
#pragma clang fp contract(off)
#endif
    float* S = sh.rec;
    float* hts = heights_of(sh);
    const int H = c.num_height_points;
    long long ep = B.d_episode_length[env] + 1;      // :204
    if constexpr (LR) {   // legged_robot.py:323-324: periodic command resampling comes first in the callback
      if (c.resample_interval > 0 && (ep % c.resample_interval) == 0) resample_commands(sh, c, L, inj, env, step, L.r_cmd);
    }
    float q[4] = {S[LC::root + 3], S[LC::root + 4], S[LC::root + 5], S[LC::root + 6]};
    // :208-211 (uniform; lane 0 stores)
    float blv[3], bav[3], pg[3];
    {
      float lv[3] = {S[LC::root + 7], S[LC::root + 8], S[LC::root + 9]}, av[3] = {S[LC::root + 10], S[LC::root + 11], S[LC::root + 12]}, gv[3] = {0.0f, 0.0f, -1.0f};
      pf_quat_rotate_dir(q, lv, -1.0f, blv); pf_quat_rotate_dir(q, av, -1.0f, bav); pf_quat_rotate_dir(q, gv, -1.0f, pg);
    }
    // :212-213, :1048-1075 height scan -- one point per lane
    float root_z = S[LC::root + 2];
    float hsum = 0.0f;
    if (c.measure_heights && !partner) hsum = height_scan(sh, c, B, env);
    // :1114-1129 feet state machine (PointFoot only; LeggedRobot keeps it inside _reward_feet_air_time)
    if constexpr (!LR) PF_LANES(f, NF) {
      bool first = S[LC::first_contact + f] != 0.0f, filt = S[LC::contact_filt + f] != 0.0f;
      float air = S[LC::feet_air_time + f];
      S[LC::last_feet_air_time + f] = first ? air : S[LC::last_feet_air_time + f];
      if (filt) air = 0.0f * air;
      if (c.include_feet_height) {
        float cm = S[LC::cur_max_feet_height + f];
        S[LC::last_max_feet_height + f] = first ? cm : S[LC::last_max_feet_height + f];
        if (filt) cm = 0.0f * cm;
        float hb = 0.0f;
        if (c.terrain_type != PF_TERRAIN_PLANE) hb = height_raw(c, B.d_height_samples, S[LC::feet_pos + 3 * f], S[LC::feet_pos + 3 * f + 1]) * c.vscale;
        float fh = S[LC::feet_pos + 3 * f + 2] - hb;
        S[LC::feet_height + f] = fh;
        S[LC::cur_max_feet_height + f] = fmaxf(cm, fh);
      }
      bool contact = S[LC::contact_forces + 3 * m.feet_body[f] + 2] > 1.0f;
      bool nfilt = contact || (S[LC::last_contacts + f] != 0.0f);
      S[LC::contact_filt + f] = nfilt ? 1.0f : 0.0f;
      S[LC::last_contacts + f] = contact ? 1.0f : 0.0f;
      S[LC::first_contact + f] = (air > 0.0f && nfilt) ? 1.0f : 0.0f;
      S[LC::feet_air_time + f] = air + c.dt;
    }
    // :479-493 callback
    float cmd2 = S[LC::commands + 2];
    if (c.heading_command) {
      float fwd[3] = {1.0f, 0.0f, 0.0f}, fw[3];
      pf_quat_apply(q, fwd, fw);
      float heading = atan2f(fw[1], fw[0]);
      cmd2 = pf_clip(0.5f * pf_wrap_to_pi(S[LC::commands + 3] - heading), -1.0f, 1.0f);
    }
    bool push = c.push_robots && (step % c.push_interval == 0);
    float ef[3] = {0.0f, 0.0f, 0.0f};
    float pv[2] = {0.0f, 0.0f};
    if (LR && push) {   // legged_robot.py:414-419: overwrite base lin-vel xy
      for (int i = 0; i < 2; ++i) pv[i] = (c.max_push_vel - (-c.max_push_vel)) * rnd(c, L, inj, env, step, L.r_push + i) + (-c.max_push_vel);
    }
    if (!LR && push) {     // :590-611
      float fr[3];
      for (int i = 0; i < 3; ++i) fr[i] = (c.max_push_force - (-c.max_push_force)) * rnd(c, L, inj, env, step, L.r_push + i) + (-c.max_push_force);
      pf_quat_rotate_dir(q, fr, 1.0f, ef);
      ef[2] *= 0.5f;
    }
    PF_SYNC();
    PF_ONE {
      for (int t = 0; t < 3; ++t) { S[LC::base_lin_vel + t] = blv[t]; S[LC::base_ang_vel + t] = bav[t]; S[LC::projected_gravity + t] = pg[t]; }
      S[LC::commands + 2] = cmd2;
      if (!LR && push) { S[LC::ext_force] = ef[0]; S[LC::ext_force + 1] = ef[1]; S[LC::ext_force + 2] = ef[2]; S[LC::ext_force + 3] = 1.0f; }
      if (LR && push) { S[LC::root + 7] = pv[0]; S[LC::root + 8] = pv[1]; }
    }
    PF_SYNC();
    // :239-247 termination (uniform)
    bool term = false;
    for (int t = 0; t < m.n_term; ++t) {
      const float* Fc = S + LC::contact_forces + 3 * m.term_body[t];
      term = term || (sqrtf(Fc[0] * Fc[0] + Fc[1] * Fc[1] + Fc[2] * Fc[2]) > 1.0f);
    }
    bool time_out = (float)ep > c.max_episode_length;
    bool reset = term || time_out;
    if constexpr (DUAL) { if (partner) { PF_XSYNC(); hsum = sh.hsum_x; } }      // height scan and noise draws are in place
    // :301-318 rewards: one lane per reward term (the terms are independent; a loop over them pays a scalar load of the term's id, a
    // branch and the term's LDS reads one after the other, 19 times); the scaled terms are then added up in term order by every lane --
    // the same sum as the loop's
    float rew = 0.0f;
    float* rterm = &sh.W[0][0];      // (constraint-row storage: physics only)
    static_assert(NROWS * NVP >= PF_MAX_REWARDS, "reward terms do not fit the aliased row storage");
    PF_LANES(ri, c.num_rewards) {
      float v = 0.0f;
      switch (c.reward_id[ri]) {
        case PF_REW_LIN_VEL_Z: v = blv[2] * blv[2]; break;
        case PF_REW_ANG_VEL_XY: v = bav[0] * bav[0] + bav[1] * bav[1]; break;
        case PF_REW_ORIENTATION: v = pg[0] * pg[0] + pg[1] * pg[1]; break;
        case PF_REW_BASE_HEIGHT: { float bh = c.measure_heights ? hsum / (float)H : root_z; v = (bh - c.base_height_target) * (bh - c.base_height_target); } break;
        case PF_REW_TORQUES: for (int k = 0; k < ND; ++k) v += S[LC::torques + k] * S[LC::torques + k]; break;
        case PF_REW_DOF_VEL: for (int k = 0; k < ND; ++k) v += S[LC::dof_vel + k] * S[LC::dof_vel + k]; break;
        case PF_REW_DOF_ACC: for (int k = 0; k < ND; ++k) { float a = (S[LC::last_dof_vel + k] - S[LC::dof_vel + k]) / c.dt; v += a * a; } break;
        case PF_REW_ACTION_RATE: for (int k = 0; k < c.num_actions; ++k) { float a = S[LC::last_actions + k] - S[LC::actions + k]; v += a * a; } break;
        case PF_REW_COLLISION:
          for (int t = 0; t < m.n_pen; ++t) { const float* Fc = S + LC::contact_forces + 3 * m.pen_body[t]; v += (sqrtf(Fc[0] * Fc[0] + Fc[1] * Fc[1] + Fc[2] * Fc[2]) > 0.1f) ? 1.0f : 0.0f; }
          break;
        case PF_REW_TERMINATION: v = (reset && !time_out) ? 1.0f : 0.0f; break;
        case PF_REW_DOF_POS_LIMITS:
          for (int k = 0; k < ND; ++k) { float o = -fminf(S[LC::dof_pos + k] - c.dof_pos_soft_lo[k], 0.0f); o += fmaxf(S[LC::dof_pos + k] - c.dof_pos_soft_hi[k], 0.0f); v += o; }
          break;
        case PF_REW_DOF_VEL_LIMITS: for (int k = 0; k < ND; ++k) v += pf_clip(fabsf(S[LC::dof_vel + k]) - c.dof_vel_limits[k] * c.soft_dof_vel_limit, 0.0f, 1.0f); break;
        case PF_REW_TORQUE_LIMITS: for (int k = 0; k < ND; ++k) v += fmaxf(fabsf(S[LC::torques + k]) - c.torque_limits[k] * c.soft_torque_limit, 0.0f); break;
        case PF_REW_TRACKING_LIN_VEL: { float ex = S[LC::commands] - blv[0], ey = S[LC::commands + 1] - blv[1]; v = expf(-(ex * ex + ey * ey) / c.tracking_sigma); } break;
        case PF_REW_TRACKING_ANG_VEL: { float e = cmd2 - bav[2]; v = expf(-(e * e) / c.tracking_sigma); } break;
        case PF_REW_FEET_AIR_TIME: if constexpr (LR) {   // legged_robot.py:882-893 (state machine lives in the reward)
          float acc = 0.0f;
          for (int f = 0; f < NF; ++f) {
            bool contact = S[LC::contact_forces + 3 * m.feet_body[f] + 2] > 1.0f;
            bool filt = contact || (S[LC::last_contacts + f] != 0.0f);
            float air = S[LC::feet_air_time + f];
            float first = (air > 0.0f && filt) ? 1.0f : 0.0f;
            air += c.dt;
            acc += (air - 0.5f) * first;
            S[LC::last_contacts + f] = contact ? 1.0f : 0.0f; S[LC::feet_air_time + f] = filt ? air * 0.0f : air;      // (this term's lane only)
          }
          float cn = sqrtf(S[LC::commands] * S[LC::commands] + S[LC::commands + 1] * S[LC::commands + 1]);
          v = acc * (cn > 0.1f ? 1.0f : 0.0f);
        } else {
          float lo = 0.0f, hi = 0.0f;
          for (int f = 0; f < NF; ++f) { float fc = S[LC::first_contact + f], at = S[LC::feet_air_time + f]; lo += fminf(at - c.min_feet_air_time, 0.0f) * fc; hi += fminf(c.max_feet_air_time - at, 0.0f) * fc; }
          v = lo + hi;
        } break;
        case PF_REW_NO_FLY: { float n = 0.0f; for (int f = 0; f < NF; ++f) n += (S[LC::contact_forces + 3 * m.feet_body[f] + 2] > 0.1f) ? 1.0f : 0.0f; v = (n == 1.0f) ? 1.0f : 0.0f; } break;
        case PF_REW_UNBALANCE_FEET_AIR_TIME:
        case PF_REW_UNBALANCE_FEET_HEIGHT: {
          const float* a = S + (c.reward_id[ri] == PF_REW_UNBALANCE_FEET_AIR_TIME ? LC::last_feet_air_time : LC::last_max_feet_height);
          float mean = 0.0f; for (int f = 0; f < NF; ++f) mean += a[f]; mean /= (float)NF;
          float s = 0.0f; for (int f = 0; f < NF; ++f) s += (a[f] - mean) * (a[f] - mean);
          v = s / (float)(NF - 1);
        } break;
        case PF_REW_STUMBLE:
          for (int f = 0; f < NF; ++f) { const float* Fc = S + LC::contact_forces + 3 * m.feet_body[f]; if (sqrtf(Fc[0] * Fc[0] + Fc[1] * Fc[1]) > 5.0f * fabsf(Fc[2])) v = 1.0f; }
          break;
        case PF_REW_STAND_STILL: if constexpr (LR) {      // legged_robot.py:900-902
          float sm = 0.0f;
          for (int k = 0; k < ND; ++k) sm += fabsf(S[LC::dof_pos + k] - c.default_dof_pos[k]);
          float cn = sqrtf(S[LC::commands] * S[LC::commands] + S[LC::commands + 1] * S[LC::commands + 1]);
          v = sm * (cn < 0.1f ? 1.0f : 0.0f);
        } else
          v = (fabsf(blv[0]) * (S[LC::commands] < 0.1f ? 1.0f : 0.0f) + fabsf(blv[1]) * (S[LC::commands + 1] < 0.1f ? 1.0f : 0.0f)) + fabsf(bav[2]) * (cmd2 < 0.1f ? 1.0f : 0.0f);
          break;
        case PF_REW_FEET_CONTACT_FORCES:
          for (int f = 0; f < NF; ++f) { const float* Fc = S + LC::contact_forces + 3 * m.feet_body[f]; v += fmaxf(sqrtf(Fc[0] * Fc[0] + Fc[1] * Fc[1] + Fc[2] * Fc[2]) - c.max_contact_force, 0.0f); }
          break;
        case PF_REW_FEET_DISTANCE:
          for (int i = 0; i < NF - 1; ++i) {
            int j = NF - 1;
            float dx = S[LC::feet_pos + 3 * i] - S[LC::feet_pos + 3 * j], dy = S[LC::feet_pos + 3 * i + 1] - S[LC::feet_pos + 3 * j + 1];
            v += pf_clip(c.min_feet_distance - sqrtf(dx * dx + dy * dy), 0.0f, 1.0f);
          }
          break;
        case PF_REW_SURVIVAL: v = (reset ? 0.0f : 1.0f) * c.dt; break;
        default: break;
      }
      float r = v * c.reward_scale[ri];
      rterm[ri] = r;
      S[LC::episode_sums + ri] += r;
    }
    PF_SYNC();
    for (int ri = 0; ri < c.num_rewards; ++ri) rew += rterm[ri];
    if (c.only_positive_rewards) rew = fmaxf(rew, 0.0f);
    if (c.termination_scale != 0.0f) {
      float r = ((reset && !time_out) ? 1.0f : 0.0f) * c.termination_scale;
      rew += r;
      if (PF_LANE_ID == 0) S[LC::episode_sums + c.num_rewards] += r;
    }
    PF_ONE {
      B.d_rew[env] = rew; B.d_terminate[env] = term ? 1 : 0; B.d_time_out[env] = time_out ? 1 : 0; B.d_reset[env] = reset ? 1 : 0;
      B.d_episode_length[env] = reset ? 0 : ep;
    }
    PF_SYNC();
    if (reset) reset_env(sh, m, c, L, B, inj, env, step);      // :221-222 masked reset
    observations(sh, c, L, B, inj, env, step, partner);         // :223
    PF_SYNC();
    PF_LANES(k, c.num_actions) S[LC::last_actions + k] = S[LC::actions + k];    // :225-227
    PF_LANES(k, ND) S[LC::last_dof_vel + k] = S[LC::dof_vel + k];
    PF_LANES(i, 6) S[LC::last_root_vel + i] = S[LC::root + 7 + i];
    PF_SYNC();
  }
};

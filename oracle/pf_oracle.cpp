/*
 * pf_oracle.cpp — TEST INFRASTRUCTURE.  CPU restatement of the north-star env step.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (pointfoot_amd + libpointfoot_hip.so) never does.
 *
 * Two parts:
 *  (1) the reference's PYTHON LAYER, restated line by line from
 *      /root/reference/legged_gym/envs/pointfoot/point_foot.py (cited per function below) in float32
 *      with the same operation order (compile with -ffp-contract=off).  PINNED by the .npz files in tests/golden,
 *      which were produced by running the reference's own PointFoot methods (tests/golden/make_golden.py).
 *  (2) the PHYSICS substep the reference delegates to Isaac Gym / PhysX (point_foot.py:183, closed
 *      binary, absent): an own floating-base articulated-body step (world-frame CRBA + RNEA, Cholesky,
 *      sphere-vs-heightfield contacts, projected Gauss-Seidel with Coulomb cone, joint limits,
 *      semi-implicit Euler).  PARITY UNPINNED — no reference artefact exists; this file *defines* the
 *      semantics the HIP kernel has to reproduce and is validated by physical invariants (tests/).
 *
 * Structs come from include/pointfoot_hip.h (the C ABI) so the tests hand both sides identical inputs.
 */
#include "../include/pointfoot_hip.h"
#include "philox.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ------------------------------------------------------------------------------------------ helpers
struct Ctx {
  const PfModel* m;
  const PfConfig* c;
  const PfLayout* L;
  const PfBuffers* b;   // HOST pointers here
  const PfInject* inj;  // HOST pointers, may be null
  int64_t step_counter;
};

inline float clipf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

// isaacgym.torch_utils.quat_rotate / quat_rotate_inverse, float32, torch op order:
//   a = v*(2w^2-1); b = cross(qv, v)*w*2; c = qv*dot(qv, v)*2; a (+/-) b + c
inline void quat_rotate_dir(const float* q, const float* v, float sign, float* out) {
  float w = q[3];
  float s = 2.0f * (w * w) - 1.0f;
  float cx = q[1] * v[2] - q[2] * v[1], cy = q[2] * v[0] - q[0] * v[2], cz = q[0] * v[1] - q[1] * v[0];
  float d = q[0] * v[0] + q[1] * v[1] + q[2] * v[2];
  float a[3] = {v[0] * s, v[1] * s, v[2] * s};
  float b[3] = {cx * w * 2.0f, cy * w * 2.0f, cz * w * 2.0f};
  float c[3] = {q[0] * d * 2.0f, q[1] * d * 2.0f, q[2] * d * 2.0f};
  for (int i = 0; i < 3; ++i) out[i] = (sign > 0 ? a[i] + b[i] : a[i] - b[i]) + c[i];
}
// isaacgym.torch_utils.quat_apply(a, b): t = 2*cross(a.xyz, b); b + a.w*t + cross(a.xyz, t)
inline void quat_apply(const float* a, const float* b, float* out) {
  float tx = (a[1] * b[2] - a[2] * b[1]) * 2.0f, ty = (a[2] * b[0] - a[0] * b[2]) * 2.0f,
        tz = (a[0] * b[1] - a[1] * b[0]) * 2.0f;
  float ux = a[1] * tz - a[2] * ty, uy = a[2] * tx - a[0] * tz, uz = a[0] * ty - a[1] * tx;
  out[0] = (b[0] + a[3] * tx) + ux;
  out[1] = (b[1] + a[3] * ty) + uy;
  out[2] = (b[2] + a[3] * tz) + uz;
}
// legged_gym/utils/math.py:38-42 quat_apply_yaw: zero x,y; normalize (eps 1e-9); quat_apply
inline void quat_apply_yaw(const float* q, const float* v, float* out) {
  float qy[4] = {0.0f, 0.0f, q[2], q[3]};
  float n = std::sqrt(qy[2] * qy[2] + qy[3] * qy[3]);
  if (n < 1e-9f) n = 1e-9f;
  qy[2] /= n; qy[3] /= n;
  quat_apply(qy, v, out);
}
// legged_gym/utils/math.py:45-48 wrap_to_pi with torch.remainder semantics
inline float wrap_to_pi(float a) {
  const float two_pi = (float)(2.0 * M_PI);
  float mod = std::fmod(a, two_pi);
  if (mod != 0.0f && (mod < 0.0f)) mod += two_pi;
  a = mod;
  if (a > (float)M_PI) a -= two_pi;   // 2*pi*(angles > pi) in float32
  return a;
}

// point_foot.py:1100-1112 _get_terrain_heights_from_points for one point (returns raw int16 min)
inline int terrain_height_raw(const Ctx& x, float px_w, float py_w) {
  const PfConfig& c = *x.c;
  float fx = (px_w + c.border) / c.hscale;
  float fy = (py_w + c.border) / c.hscale;
  long ix = (long)fx, iy = (long)fy;            // .long() truncates toward zero
  ix = std::min<long>(std::max<long>(ix, 0), c.hf_rows - 2);
  iy = std::min<long>(std::max<long>(iy, 0), c.hf_cols - 2);
  const int16_t* hs = x.b->d_height_samples;
  int h1 = hs[ix * c.hf_cols + iy], h2 = hs[(ix + 1) * c.hf_cols + iy], h3 = hs[ix * c.hf_cols + iy + 1];
  return std::min(std::min(h1, h2), h3);
}

inline float rnd(const Ctx& x, int env, int slot) {
  if (x.inj && x.inj->d_rand) return x.inj->d_rand[(size_t)env * x.L->rand_floats + slot];
  return pfo_rand_slot(x.c->seed_lo, x.c->seed_hi, (uint32_t)(x.c->env_index_base + env),
                       (uint32_t)x.step_counter, 0u, (uint32_t)slot);
}

// ------------------------------------------------------------------------------------- LeggedRobot helpers
// legged_robot.py:337-351 _resample_commands for one env; `slot` = first of 3 random slots
void resample_commands(const Ctx& x, int env, float* S, int slot) {
  const PfConfig& c = *x.c; const PfLayout& L = *x.L;
  float cx = (c.cmd_hi[0] - c.cmd_lo[0]) * rnd(x, env, slot) + c.cmd_lo[0];
  float cy = (c.cmd_hi[1] - c.cmd_lo[1]) * rnd(x, env, slot + 1) + c.cmd_lo[1];
  int k = c.heading_command ? 3 : 2;
  float cz = (c.cmd_hi[k] - c.cmd_lo[k]) * rnd(x, env, slot + 2) + c.cmd_lo[k];
  float keep = std::sqrt(cx * cx + cy * cy) > 0.2f ? 1.0f : 0.0f;
  S[L.commands] = cx * keep; S[L.commands + 1] = cy * keep; S[L.commands + k] = cz;
}
// anymal.py:71-78 actuator network (LSTMsea: in_scale -> LSTM(2->8)x2 -> Linear(8->1) -> out_scale), one time step per joint
inline float sigm(float v) { return 1.0f / (1.0f + std::exp(-v)); }
void actuator_torques(const Ctx& x, float* S) {
  const PfConfig& c = *x.c; const PfLayout& L = *x.L; const int nd = x.m->nd;
  const float* P = x.b->d_actuator_params;
  for (int j = 0; j < nd; ++j) {
    float* st = S + L.lstm_state + j * 32;
    float x0 = ((S[L.actions + j] * c.action_scale + c.default_dof_pos[j]) - S[L.dof_pos + j]) * P[PF_ACT_IN_SCALE];
    float x1 = S[L.dof_vel + j] * P[PF_ACT_IN_SCALE + 1];
    for (int layer = 0; layer < 2; ++layer) {
      float G[32];
      for (int g = 0; g < 32; ++g) {
        float acc;
        if (layer == 0) {
          acc = P[PF_ACT_B_IH0 + g] + P[PF_ACT_B_HH0 + g] + P[PF_ACT_W_IH0 + 2 * g] * x0 + P[PF_ACT_W_IH0 + 2 * g + 1] * x1;
          for (int t = 0; t < 8; ++t) acc += P[PF_ACT_W_HH0 + 8 * g + t] * st[t];
        } else {
          acc = P[PF_ACT_B_IH1 + g] + P[PF_ACT_B_HH1 + g];
          for (int t = 0; t < 8; ++t) acc += P[PF_ACT_W_IH1 + 8 * g + t] * st[t] + P[PF_ACT_W_HH1 + 8 * g + t] * st[16 + t];
        }
        G[g] = acc;
      }
      float* s2 = st + layer * 16;
      for (int u = 0; u < 8; ++u) {
        float ig = sigm(G[u]), fg = sigm(G[8 + u]), gg = std::tanh(G[16 + u]), og = sigm(G[24 + u]);
        float cn = fg * s2[8 + u] + ig * gg;
        s2[8 + u] = cn; s2[u] = og * std::tanh(cn);
      }
    }
    float y = P[PF_ACT_B_LIN];
    for (int t = 0; t < 8; ++t) y += P[PF_ACT_W_LIN + t] * st[16 + t];
    S[L.torques + j] = y * P[PF_ACT_OUT_SCALE];
  }
}

// ------------------------------------------------------------------------------------- PD torques
// point_foot.py:523-547 _compute_torques
void compute_torques(const Ctx& x, float* S, const float* qpos, const float* qvel) {
  const PfConfig& c = *x.c; const PfLayout& L = *x.L; const int nd = x.m->nd;
  for (int k = 0; k < nd; ++k) {
    float as = S[L.actions + k] * c.action_scale, t;
    if (c.control_type == PF_CTRL_P)
      t = c.p_gains[k] * ((as + c.default_dof_pos[k]) - qpos[k]) - c.d_gains[k] * qvel[k];
    else if (c.control_type == PF_CTRL_V)
      t = c.p_gains[k] * (as - qvel[k]) - c.d_gains[k] * (qvel[k] - S[L.last_dof_vel + k]) / c.sim_dt;
    else
      t = as;
    S[L.torques + k] = clipf(t, -c.torque_limits[k], c.torque_limits[k]);
  }
}

// ============================================================================================ physics
template <typename R> struct V3 { R x, y, z; };
template <typename R> inline V3<R> operator+(V3<R> a, V3<R> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename R> inline V3<R> operator-(V3<R> a, V3<R> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename R> inline V3<R> operator*(R s, V3<R> a) { return {s * a.x, s * a.y, s * a.z}; }
template <typename R> inline R dot(V3<R> a, V3<R> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename R> inline V3<R> cross(V3<R> a, V3<R> b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <typename R> struct M3 { R m[9]; };
template <typename R> inline V3<R> mul(const M3<R>& A, V3<R> v) {
  return {A.m[0] * v.x + A.m[1] * v.y + A.m[2] * v.z, A.m[3] * v.x + A.m[4] * v.y + A.m[5] * v.z,
          A.m[6] * v.x + A.m[7] * v.y + A.m[8] * v.z};
}
template <typename R> inline M3<R> mul(const M3<R>& A, const M3<R>& B) {
  M3<R> C;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C.m[i * 3 + j] = A.m[i * 3] * B.m[j] + A.m[i * 3 + 1] * B.m[3 + j] + A.m[i * 3 + 2] * B.m[6 + j];
  return C;
}
template <typename R> inline M3<R> quat_to_mat(const R* q) {  // xyzw
  R x = q[0], y = q[1], z = q[2], w = q[3];
  M3<R> A;
  A.m[0] = 1 - 2 * (y * y + z * z); A.m[1] = 2 * (x * y - z * w); A.m[2] = 2 * (x * z + y * w);
  A.m[3] = 2 * (x * y + z * w); A.m[4] = 1 - 2 * (x * x + z * z); A.m[5] = 2 * (y * z - x * w);
  A.m[6] = 2 * (x * z - y * w); A.m[7] = 2 * (y * z + x * w); A.m[8] = 1 - 2 * (x * x + y * y);
  return A;
}
template <typename R> inline M3<R> axis_angle(V3<R> a, R th) {  // Rodrigues
  R c = std::cos(th), s = std::sin(th), t = 1 - c;
  M3<R> A;
  A.m[0] = t * a.x * a.x + c; A.m[1] = t * a.x * a.y - s * a.z; A.m[2] = t * a.x * a.z + s * a.y;
  A.m[3] = t * a.x * a.y + s * a.z; A.m[4] = t * a.y * a.y + c; A.m[5] = t * a.y * a.z - s * a.x;
  A.m[6] = t * a.x * a.z - s * a.y; A.m[7] = t * a.y * a.z + s * a.x; A.m[8] = t * a.z * a.z + c;
  return A;
}
// spatial inertia about the reference point O (world axes): mass, h = m*c, I_O (sym xx,yy,zz,xy,xz,yz)
template <typename R> struct SpI { R m; V3<R> h; R I[6]; };
template <typename R> inline V3<R> symmul(const R* I, V3<R> v) {
  return {I[0] * v.x + I[3] * v.y + I[4] * v.z, I[3] * v.x + I[1] * v.y + I[5] * v.z, I[4] * v.x + I[5] * v.y + I[2] * v.z};
}
template <typename R> struct Sp6 { V3<R> a, l; };   // motion (ang, lin@O) or force (moment@O, force)
template <typename R> inline Sp6<R> spI_mul(const SpI<R>& I, const Sp6<R>& v) {
  return {symmul(I.I, v.a) + cross(I.h, v.l), I.m * v.l - cross(I.h, v.a)};
}

const int MAXC_POINTFOOT = 6, MAXC_QUAD = 10;    // contacts kept per robot (first active in sphere order); the quadruped keeps 10:
                                                 // 4 feet x (floor + riser face), see pf_step_kernel.h PF_MAXC / PF_MAXC_QUAD
const int MAXC = MAXC_QUAD;                      // array bound
const int MAXROWS = 3 * MAXC + PF_MAX_DOF;
const int NVMAX = 6 + PF_MAX_DOF;

// Collision surface (own model; the reference hands PhysX a triangle mesh built by
// isaacgym.terrain_utils.convert_heightfield_to_trimesh(height_field_raw, horizontal_scale, vertical_scale, slope_treshold),
// call site legged_gym/utils/terrain.py:69-73 -- closed package, parity unpinned).  That conversion slides the low vertex of
// every grid edge steeper than slope_treshold under its high neighbour, so the step becomes a vertical face at the high
// vertex's grid line and the ground next to it stays at the low height.  Restated per grid cell:
//   * floor: bilinear over the cell's corner heights, where a corner more than wall_threshold above the cell's lowest
//     corner counts as the top of a face and takes the low height;
//   * faces: on each of the cell's four grid lines a vertical face stands where the raw profile along that line is more
//     than wall_threshold above the floor profile; a sphere touches the nearest point of the face segment (so the lip of
//     a step is a rounded edge).  One face candidate per sphere: the closest.
// wall_threshold == 0 (mesh_type 'heightfield'): plain bilinear field, no faces.
template <typename R>
struct Terrain {
  struct Hit { R h; V3<R> n; bool wall; V3<R> wn; R wdist; long cell; int face; };
  static Hit query(const Ctx& x, int ix0, int iy0, R lx, R ly, R wz, R rad) {
    const PfConfig& c = *x.c;
    Hit o; o.h = 0; o.n = {0, 0, 1}; o.wall = false; o.wn = {0, 0, 1}; o.wdist = (R)1e9; o.cell = 0; o.face = 0;
    if (c.terrain_type == PF_TERRAIN_PLANE || !x.b->d_height_samples) return o;
    // (lx, ly): metres from the corner of grid cell (ix0, iy0) -- see the anchor in pfo_step
    R fx = lx / (R)c.hscale, fy = ly / (R)c.hscale;
    R ox = std::floor(fx), oy = std::floor(fy);
    int ix = ix0 + (int)ox, iy = iy0 + (int)oy;
    R tx = fx - ox, ty = fy - oy;
    if (ix < 0) { ix = 0; tx = 0; } else if (ix > c.hf_rows - 2) { ix = c.hf_rows - 2; tx = (R)1 - (R)1e-3; }
    if (iy < 0) { iy = 0; ty = 0; } else if (iy > c.hf_cols - 2) { iy = c.hf_cols - 2; ty = (R)1 - (R)1e-3; }
    o.cell = (long)ix * c.hf_cols + iy;
    const int16_t* hs = x.b->d_height_samples;
    R vs = (R)c.vscale, hsc = (R)c.hscale;
    R H00 = vs * hs[ix * c.hf_cols + iy], H10 = vs * hs[(ix + 1) * c.hf_cols + iy], H01 = vs * hs[ix * c.hf_cols + iy + 1],
      H11 = vs * hs[(ix + 1) * c.hf_cols + iy + 1];
    R E00 = H00, E10 = H10, E01 = H01, E11 = H11;
    const R thr = (R)c.wall_threshold;
    if (thr > 0) {
      R lo = std::min(std::min(H00, H10), std::min(H01, H11));
      if (H00 - lo > thr) E00 = lo;
      if (H10 - lo > thr) E10 = lo;
      if (H01 - lo > thr) E01 = lo;
      if (H11 - lo > thr) E11 = lo;
    }
    o.h = (1 - tx) * ((1 - ty) * E00 + ty * E01) + tx * ((1 - ty) * E10 + ty * E11);
    R dhdx = ((1 - ty) * (E10 - E00) + ty * (E11 - E01)) / hsc;
    R dhdy = ((1 - tx) * (E01 - E00) + tx * (E11 - E10)) / hsc;
    R inv = 1 / std::sqrt(dhdx * dhdx + dhdy * dhdy + 1);
    o.n = {-dhdx * inv, -dhdy * inv, inv};
    if (thr > 0) {
      // faces on the cell's grid lines: 0: x-far (normal -x), 1: x-near (+x), 2: y-far (-y), 3: y-near (+y)
      const R top[4] = {(1 - ty) * H10 + ty * H11, (1 - ty) * H00 + ty * H01, (1 - tx) * H01 + tx * H11, (1 - tx) * H00 + tx * H10};
      const R bot[4] = {(1 - ty) * E10 + ty * E11, (1 - ty) * E00 + ty * E01, (1 - tx) * E01 + tx * E11, (1 - tx) * E00 + tx * E10};
      const R gap[4] = {(1 - tx) * hsc, tx * hsc, (1 - ty) * hsc, ty * hsc};
      for (int f = 0; f < 4; ++f) {
        if (!(top[f] - bot[f] > thr)) continue;
        R dz = std::max(wz - top[f], (R)0);
        R len = std::sqrt(gap[f] * gap[f] + dz * dz);
        R dist = len - rad;
        if (dist < o.wdist) {
          R il = 1 / std::max(len, (R)1e-9);
          R hn = len > (R)1e-9 ? gap[f] * il : (R)1, zn = len > (R)1e-9 ? dz * il : (R)0;
          o.wall = true; o.wdist = dist; o.face = f;
          o.wn = f == 0 ? V3<R>{-hn, 0, zn} : f == 1 ? V3<R>{hn, 0, zn} : f == 2 ? V3<R>{0, -hn, zn} : V3<R>{0, hn, zn};
        }
      }
    }
    return o;
  }
};

// closest points of the segments p1-q1 and p2-q2 (Ericson, Real-Time Collision Detection 5.1.9); returns c1 - c2 and the
// point c2
template <typename R>
inline V3<R> segment_segment(V3<R> p1, V3<R> q1, V3<R> p2, V3<R> q2, V3<R>* c2_out) {
  const R eps = (R)1e-12;
  V3<R> d1 = q1 - p1, d2 = q2 - p2, r = p1 - p2;
  R a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r);
  R s, t;
  auto clamp01 = [](R v) { return std::min(std::max(v, (R)0), (R)1); };
  if (a <= eps && e <= eps) { s = 0; t = 0; }
  else if (a <= eps) { s = 0; t = clamp01(f / e); }
  else {
    R cc = dot(d1, r);
    if (e <= eps) { t = 0; s = clamp01(-cc / a); }
    else {
      R b = dot(d1, d2), den = a * e - b * b;
      s = den > eps ? clamp01((b * f - cc * e) / den) : (R)0;
      t = (b * s + f) / e;
      if (t < 0) { t = 0; s = clamp01(-cc / a); }
      else if (t > 1) { t = 1; s = clamp01((b - cc) / a); }
    }
  }
  V3<R> c1 = p1 + s * d1, c2 = p2 + t * d2;
  *c2_out = c2;
  return c1 - c2;
}

// One physics substep of sim_dt.  Reads/writes root, dof_pos, dof_vel, contact_forces, feet_pos in S.
template <typename R>
void physics_substep(const Ctx& x, float* S, bool apply_ext, int env, int sub, int ix0, int iy0) {
  const PfModel& m = *x.m; const PfConfig& c = *x.c; const PfLayout& L = *x.L;
  const int nb = m.nb, nd = m.nd, nv = 6 + nd;
  const R dt = (R)c.sim_dt;
  R q[PF_MAX_DOF], qd[PF_MAX_DOF];
  for (int k = 0; k < nd; ++k) { q[k] = S[L.dof_pos + k]; qd[k] = S[L.dof_vel + k]; }
  R quat[4] = {(R)S[L.root + 3], (R)S[L.root + 4], (R)S[L.root + 5], (R)S[L.root + 6]};
  V3<R> pos = {(R)S[L.root], (R)S[L.root + 1], (R)S[L.root + 2]};
  V3<R> vb = {(R)S[L.root + 7], (R)S[L.root + 8], (R)S[L.root + 9]};
  V3<R> wb = {(R)S[L.root + 10], (R)S[L.root + 11], (R)S[L.root + 12]};
  V3<R> grav = {(R)c.gravity[0], (R)c.gravity[1], (R)c.gravity[2]};

  // ---- forward kinematics, velocities, bias accelerations (world axes, origin O = base origin)
  M3<R> Rw[PF_MAX_BODIES]; V3<R> r[PF_MAX_BODIES], ax[PF_MAX_BODIES];
  Sp6<R> Sj[PF_MAX_BODIES], V[PF_MAX_BODIES], A[PF_MAX_BODIES];
  Rw[0] = quat_to_mat(quat); r[0] = {0, 0, 0};
  V[0] = {wb, vb};
  A[0] = {{0, 0, 0}, (R)(-1) * cross(wb, vb) - grav};
  for (int i = 1; i < nb; ++i) {
    int p = m.parent[i];
    M3<R> Rf; for (int t = 0; t < 9; ++t) Rf.m[t] = m.joint_rot[i][t];
    M3<R> Rj = mul(Rw[p], Rf);
    V3<R> jp = {(R)m.joint_pos[i][0], (R)m.joint_pos[i][1], (R)m.joint_pos[i][2]};
    r[i] = r[p] + mul(Rw[p], jp);
    int k = m.dof_of_body[i];
    if (k >= 0) {
      V3<R> al = {(R)m.joint_axis[i][0], (R)m.joint_axis[i][1], (R)m.joint_axis[i][2]};
      ax[i] = mul(Rj, al);
      Rw[i] = mul(Rj, axis_angle(al, q[k]));
      Sj[i] = {ax[i], cross(r[i], ax[i])};
      V[i] = {V[p].a + qd[k] * Sj[i].a, V[p].l + qd[k] * Sj[i].l};
      // A_i = A_p + (V_i x_m S_i) qd
      V3<R> ca = cross(V[i].a, Sj[i].a);
      V3<R> cl = cross(V[i].a, Sj[i].l) + cross(V[i].l, Sj[i].a);
      A[i] = {A[p].a + qd[k] * ca, A[p].l + qd[k] * cl};
    } else {
      ax[i] = {0, 0, 0}; Rw[i] = Rj; Sj[i] = {{0, 0, 0}, {0, 0, 0}}; V[i] = V[p]; A[i] = A[p];
    }
  }
  // ---- per-body spatial inertia about O, bias wrench
  SpI<R> I[PF_MAX_BODIES]; Sp6<R> f[PF_MAX_BODIES];
  for (int i = 0; i < nb; ++i) {
    R mass = m.mass[i];
    V3<R> cl = {(R)m.com[i][0], (R)m.com[i][1], (R)m.com[i][2]};
    R Il[6]; for (int t = 0; t < 6; ++t) Il[t] = m.inertia[i][t];
    if (i == 0) {   // domain randomisation of the base (point_foot.py:466-477); inertia scaled with mass
      R added = S[L.added_mass];
      R sc = (mass + added) / mass;
      mass += added;
      cl = cl + V3<R>{(R)S[L.added_com], (R)S[L.added_com + 1], (R)S[L.added_com + 2]};
      for (int t = 0; t < 6; ++t) Il[t] *= sc;
    }
    V3<R> cw = r[i] + mul(Rw[i], cl);
    // Iw = R Il R^T
    M3<R> Ilm = {{Il[0], Il[3], Il[4], Il[3], Il[1], Il[5], Il[4], Il[5], Il[2]}};
    M3<R> T = mul(Rw[i], Ilm);
    M3<R> Rt = {{Rw[i].m[0], Rw[i].m[3], Rw[i].m[6], Rw[i].m[1], Rw[i].m[4], Rw[i].m[7], Rw[i].m[2], Rw[i].m[5], Rw[i].m[8]}};
    M3<R> Iw = mul(T, Rt);
    R c2 = dot(cw, cw);
    I[i].m = mass; I[i].h = mass * cw;
    I[i].I[0] = Iw.m[0] + mass * (c2 - cw.x * cw.x);
    I[i].I[1] = Iw.m[4] + mass * (c2 - cw.y * cw.y);
    I[i].I[2] = Iw.m[8] + mass * (c2 - cw.z * cw.z);
    I[i].I[3] = Iw.m[1] - mass * cw.x * cw.y;
    I[i].I[4] = Iw.m[2] - mass * cw.x * cw.z;
    I[i].I[5] = Iw.m[5] - mass * cw.y * cw.z;
    Sp6<R> IA = spI_mul(I[i], A[i]);
    Sp6<R> IV = spI_mul(I[i], V[i]);
    // V x* (IV) = (w x n + v x f ; w x f)
    f[i].a = IA.a + cross(V[i].a, IV.a) + cross(V[i].l, IV.l);
    f[i].l = IA.l + cross(V[i].a, IV.l);
  }
  if (apply_ext) {  // push force on body 0 at its origin (ENV_SPACE force, point_foot.py:606-611)
    f[0].l = f[0].l - V3<R>{(R)S[L.ext_force], (R)S[L.ext_force + 1], (R)S[L.ext_force + 2]};
  }
  // ---- composite inertias / wrenches over DFS-contiguous subtrees
  SpI<R> Ic[PF_MAX_BODIES]; Sp6<R> fc[PF_MAX_BODIES];
  for (int i = 0; i < nb; ++i) {
    Ic[i] = I[i]; fc[i] = f[i];
    for (int j = i + 1; j < i + m.subtree[i]; ++j) {
      Ic[i].m += I[j].m; Ic[i].h = Ic[i].h + I[j].h;
      for (int t = 0; t < 6; ++t) Ic[i].I[t] += I[j].I[t];
      fc[i].a = fc[i].a + f[j].a; fc[i].l = fc[i].l + f[j].l;
    }
  }
  // ---- mass matrix (generalised velocity = [v_O(3), w(3), qd]) and bias
  R M[NVMAX][NVMAX]; R bias[NVMAX];
  for (int a = 0; a < nv; ++a) for (int b2 = 0; b2 < nv; ++b2) M[a][b2] = 0;
  {
    const SpI<R>& C = Ic[0];
    M[0][0] = M[1][1] = M[2][2] = C.m;
    // M_ang,ang = I_O
    M[3][3] = C.I[0]; M[4][4] = C.I[1]; M[5][5] = C.I[2];
    M[3][4] = M[4][3] = C.I[3]; M[3][5] = M[5][3] = C.I[4]; M[4][5] = M[5][4] = C.I[5];
    // M_ang,lin = [h]x ; M_lin,ang = -[h]x
    R hx = C.h.x, hy = C.h.y, hz = C.h.z;
    R Hx[3][3] = {{0, -hz, hy}, {hz, 0, -hx}, {-hy, hx, 0}};
    for (int a = 0; a < 3; ++a) for (int b2 = 0; b2 < 3; ++b2) { M[3 + a][b2] = Hx[a][b2]; M[b2][3 + a] = Hx[a][b2]; }
    bias[0] = fc[0].l.x; bias[1] = fc[0].l.y; bias[2] = fc[0].l.z;
    bias[3] = fc[0].a.x; bias[4] = fc[0].a.y; bias[5] = fc[0].a.z;
  }
  for (int k = 0; k < nd; ++k) {
    int i = m.body_of_dof[k];
    Sp6<R> F = spI_mul(Ic[i], Sj[i]);
    M[0][6 + k] = M[6 + k][0] = F.l.x; M[1][6 + k] = M[6 + k][1] = F.l.y; M[2][6 + k] = M[6 + k][2] = F.l.z;
    M[3][6 + k] = M[6 + k][3] = F.a.x; M[4][6 + k] = M[6 + k][4] = F.a.y; M[5][6 + k] = M[6 + k][5] = F.a.z;
    for (int j = i; j > 0; j = m.parent[j]) {
      int kj = m.dof_of_body[j];
      if (kj < 0) continue;
      R v = dot(Sj[j].a, F.a) + dot(Sj[j].l, F.l);
      M[6 + kj][6 + k] = v; M[6 + k][6 + kj] = v;
    }
    bias[6 + k] = dot(Sj[i].a, fc[i].a) + dot(Sj[i].l, fc[i].l);
  }
  // ---- Cholesky M = Lc Lc^T (lower), stored in place
  R Lc[NVMAX][NVMAX];
  for (int j = 0; j < nv; ++j) {
    R d = M[j][j];
    for (int t = 0; t < j; ++t) d -= Lc[j][t] * Lc[j][t];
    d = std::sqrt(std::max(d, (R)1e-12));
    Lc[j][j] = d;
    for (int i = j + 1; i < nv; ++i) {
      R s = M[i][j];
      for (int t = 0; t < j; ++t) s -= Lc[i][t] * Lc[j][t];
      Lc[i][j] = s / d;
    }
  }
  auto solve = [&](const R* rhs, R* out) {
    R y[NVMAX];
    for (int i = 0; i < nv; ++i) { R s = rhs[i]; for (int t = 0; t < i; ++t) s -= Lc[i][t] * y[t]; y[i] = s / Lc[i][i]; }
    for (int i = nv - 1; i >= 0; --i) { R s = y[i]; for (int t = i + 1; t < nv; ++t) s -= Lc[t][i] * out[t]; out[i] = s / Lc[i][i]; }
  };
  // ---- free velocity
  R vel[NVMAX], rhs[NVMAX], dv[NVMAX];
  vel[0] = vb.x; vel[1] = vb.y; vel[2] = vb.z; vel[3] = wb.x; vel[4] = wb.y; vel[5] = wb.z;
  for (int k = 0; k < nd; ++k) vel[6 + k] = qd[k];
  for (int a = 0; a < nv; ++a) rhs[a] = -bias[a];
  for (int k = 0; k < nd; ++k) rhs[6 + k] += (R)S[L.torques + k];
  if (x.b->d_joint_friction) {   // regularised Coulomb joint friction (update_frictions, point_foot.py:103-108)
    for (int k = 0; k < nd; ++k) {
      R u = qd[k] / (R)c.joint_friction_vel;
      rhs[6 + k] -= (R)x.b->d_joint_friction[(size_t)env * nd + k] * std::min(std::max(u, (R)-1), (R)1);
    }
  }
  solve(rhs, dv);
  for (int a = 0; a < nv; ++a) vel[a] += dt * dv[a];

  // ---- constraint rows: contacts (n, t1, t2) then joint limits
  struct Row { R J[NVMAX]; R MinvJ[NVMAX]; R Ainv; R b; R lam; };
  static thread_local Row rows[MAXROWS];
  struct Contact { int body, body_b; V3<R> n, t1, t2; R mu; R A[3][3]; };
  Contact cons[MAXC];
  int ncon = 0, nrows = 0;
  uint64_t cset[2] = {0, 0}, cells = 0;      // candidates 0..63 / 64..127
  const R mu_env = (R)0.5 * ((R)S[L.friction] + (R)c.terrain_friction);
  const R rest = (R)0.5 * (R)c.terrain_restitution;
  // contact candidates in a fixed order -- floor contact of sphere s (candidate s), vertical-face contact of sphere s
  // (nsph + s), self-collision capsule pair p (2 nsph + p) -- the first MAXC active ones are kept
  V3<R> sphc[PF_MAX_SPHERES];                     // sphere centres relative to the base origin
  typename Terrain<R>::Hit hit[PF_MAX_SPHERES];
  for (int s = 0; s < m.nsph; ++s) {
    int i = m.sph_body[s];
    V3<R> sp = {(R)m.sph_pos[s][0], (R)m.sph_pos[s][1], (R)m.sph_pos[s][2]};
    sphc[s] = r[i] + mul(Rw[i], sp);
    hit[s] = Terrain<R>::query(x, ix0, iy0, pos.x + sphc[s].x, pos.y + sphc[s].y, pos.z + sphc[s].z, (R)m.sph_radius[s]);
  }
  const int ncand = 2 * m.nsph + (c.self_collisions ? m.npair : 0);
  const int maxc = c.env_kind == PF_KIND_LEGGED_ROBOT ? MAXC_QUAD : MAXC_POINTFOOT;
  for (int cand = 0; cand < ncand && ncon < maxc; ++cand) {
    int i, ib = -1; V3<R> n, xc; R dist;
    if (cand < 2 * m.nsph) {
      const int s = cand < m.nsph ? cand : cand - m.nsph;
      const bool wall = cand >= m.nsph;
      i = m.sph_body[s];
      if (wall) { if (!hit[s].wall) continue; n = hit[s].wn; dist = hit[s].wdist; }
      else { n = hit[s].n; dist = (pos.z + sphc[s].z - hit[s].h) * n.z - (R)m.sph_radius[s]; }
      xc = sphc[s] - (R)m.sph_radius[s] * n;
    } else {
      const int p = cand - 2 * m.nsph, ca = m.pair_a[p], cb = m.pair_b[p];
      i = m.sph_body[m.cap_s0[ca]]; ib = m.sph_body[m.cap_s0[cb]];
      const R ra = (R)m.sph_radius[m.cap_s0[ca]], rb = (R)m.sph_radius[m.cap_s0[cb]];
      V3<R> c2;
      V3<R> v = segment_segment(sphc[m.cap_s0[ca]], sphc[m.cap_s1[ca]], sphc[m.cap_s0[cb]], sphc[m.cap_s1[cb]], &c2);
      R len = std::sqrt(dot(v, v));
      dist = len - ra - rb;
      n = len > (R)1e-9 ? ((R)1 / len) * v : V3<R>{0, 1, 0};      // from capsule b towards capsule a
      xc = c2 + (rb + (R)0.5 * dist) * n;                          // middle of the gap / overlap
    }
    if (!(dist < (R)c.contact_offset)) continue;
    cset[cand >> 6] |= (uint64_t)1 << (cand & 63);
    if (cand < 2 * m.nsph) { const int s = cand < m.nsph ? cand : cand - m.nsph; cells += (uint64_t)(cand + 1) * (uint64_t)(4 * hit[s].cell + (cand < m.nsph ? 0 : hit[s].face)); }
    // tangent basis
    V3<R> ref = std::fabs(n.x) < (R)0.9 ? V3<R>{1, 0, 0} : V3<R>{0, 1, 0};
    V3<R> t1 = cross(n, ref); R il = 1 / std::sqrt(dot(t1, t1)); t1 = il * t1;
    V3<R> t2 = cross(n, t1);
    Contact& C = cons[ncon];
    C.body = i; C.body_b = ib; C.n = n; C.t1 = t1; C.t2 = t2; C.mu = mu_env;
    V3<R> dirs[3] = {n, t1, t2};
    for (int d = 0; d < 3; ++d) {
      Row& rw = rows[nrows + d];
      for (int a = 0; a < nv; ++a) rw.J[a] = 0;
      V3<R> dd = dirs[d];
      if (ib < 0) {      // against the terrain: the base columns see the contact point
        rw.J[0] = dd.x; rw.J[1] = dd.y; rw.J[2] = dd.z;
        V3<R> xa = cross(xc, dd);
        rw.J[3] = xa.x; rw.J[4] = xa.y; rw.J[5] = xa.z;
      }
      for (int j = i; j > 0; j = m.parent[j]) {
        int kj = m.dof_of_body[j];
        if (kj < 0) continue;
        rw.J[6 + kj] += dot(ax[j], cross(xc - r[j], dd));
      }
      for (int j = ib; j > 0; j = m.parent[j]) {      // relative velocity of body a against body b at the same point
        int kj = m.dof_of_body[j];
        if (kj < 0) continue;
        rw.J[6 + kj] -= dot(ax[j], cross(xc - r[j], dd));
      }
      rw.lam = 0; rw.b = 0;
    }
    R bn;
    if (dist >= 0) bn = -dist / dt;
    else bn = std::min((R)0.2 * (-dist) / dt, (R)c.max_depenetration_vel);
    // restitution on the pre-step approach speed
    R un0 = rows[nrows].J[0] * vb.x + rows[nrows].J[1] * vb.y + rows[nrows].J[2] * vb.z + rows[nrows].J[3] * wb.x +
            rows[nrows].J[4] * wb.y + rows[nrows].J[5] * wb.z;
    for (int k = 0; k < nd; ++k) un0 += rows[nrows].J[6 + k] * qd[k];
    if (un0 < -(R)c.bounce_threshold) bn = std::max(bn, -rest * un0);
    rows[nrows].b = bn;
    nrows += 3; ++ncon;
  }
  const int ncrow = nrows;
  int lim_dof[PF_MAX_DOF]; int nlim = 0;
  uint64_t lset = 0;
  for (int k = 0; k < nd; ++k) {
    R qn = q[k] + dt * vel[6 + k];
    R lo = m.q_lo[k], hi = m.q_hi[k];
    int side = 0;
    if (q[k] < lo || qn < lo) side = 1; else if (q[k] > hi || qn > hi) side = -1;
    if (!side) continue;
    Row& rw = rows[nrows];
    for (int a = 0; a < nv; ++a) rw.J[a] = 0;
    rw.J[6 + k] = (R)side;
    R viol = side > 0 ? lo - q[k] : q[k] - hi;      // > 0 when violated
    rw.b = viol > 0 ? std::min((R)0.2 * viol / dt, (R)c.max_depenetration_vel * 10) : viol / dt;
    rw.lam = 0;
    lim_dof[nlim++] = k; ++nrows;
    lset |= (uint64_t)(side > 0 ? 1 : 2) << (2 * k);
  }
  if (x.b->d_contact_set) {
    uint64_t* o = x.b->d_contact_set + ((size_t)env * c.decimation + sub) * PF_CSET_WORDS;
    o[0] = cset[0]; o[1] = lset; o[2] = cells; o[3] = cset[1];
  }
  for (int rI = 0; rI < nrows; ++rI) {
    solve(rows[rI].J, rows[rI].MinvJ);
    R a = 0; for (int t = 0; t < nv; ++t) a += rows[rI].J[t] * rows[rI].MinvJ[t];
    rows[rI].Ainv = 1 / std::max(a, (R)1e-9);
  }
  for (int ci = 0; ci < ncon; ++ci)
    for (int a = 0; a < 3; ++a)
      for (int b2 = 0; b2 < 3; ++b2) {
        R s = 0; for (int t = 0; t < nv; ++t) s += rows[3 * ci + a].J[t] * rows[3 * ci + b2].MinvJ[t];
        cons[ci].A[a][b2] = s;
      }
  // ---- projected Gauss-Seidel
  for (int it = 0; it < c.solver_iterations; ++it) {
    for (int ci = 0; ci < ncon; ++ci) {
      Row* rw = &rows[3 * ci]; Contact& C = cons[ci];
      R u[3];
      for (int d = 0; d < 3; ++d) { R s = 0; for (int t = 0; t < nv; ++t) s += rw[d].J[t] * vel[t]; u[d] = s; }
      R dl[3];
      R ln = std::max((R)0, rw[0].lam + (rw[0].b - u[0]) * rw[0].Ainv);
      dl[0] = ln - rw[0].lam;
      R u1 = u[1] + C.A[1][0] * dl[0];
      R l1 = rw[1].lam - u1 * rw[1].Ainv;
      R d1 = l1 - rw[1].lam;
      R u2 = u[2] + C.A[2][0] * dl[0] + C.A[2][1] * d1;
      R l2 = rw[2].lam - u2 * rw[2].Ainv;
      R lim = C.mu * ln, nt = std::sqrt(l1 * l1 + l2 * l2);
      if (nt > lim) { R sc = nt > 0 ? lim / nt : 0; l1 *= sc; l2 *= sc; }
      dl[1] = l1 - rw[1].lam; dl[2] = l2 - rw[2].lam;
      rw[0].lam = ln; rw[1].lam = l1; rw[2].lam = l2;
      for (int t = 0; t < nv; ++t) vel[t] += rw[0].MinvJ[t] * dl[0] + rw[1].MinvJ[t] * dl[1] + rw[2].MinvJ[t] * dl[2];
    }
    for (int rI = ncrow; rI < nrows; ++rI) {
      Row& rw = rows[rI];
      R u = 0; for (int t = 0; t < nv; ++t) u += rw.J[t] * vel[t];
      R ln = std::max((R)0, rw.lam + (rw.b - u) * rw.Ainv);
      R d = ln - rw.lam; rw.lam = ln;
      for (int t = 0; t < nv; ++t) vel[t] += rw.MinvJ[t] * d;
    }
  }
  // ---- joint velocity limits, integrate (semi-implicit Euler)
  for (int k = 0; k < nd; ++k) {
    R vl = m.vel_limit[k];
    vel[6 + k] = std::min(std::max(vel[6 + k], -vl), vl);
    q[k] += dt * vel[6 + k];
  }
  pos = pos + dt * V3<R>{vel[0], vel[1], vel[2]};
  V3<R> w = {vel[3], vel[4], vel[5]};
  // q <- normalize(q + 0.5 dt (w,0) (x) q)
  R qx = quat[0], qy = quat[1], qz = quat[2], qw = quat[3];
  R hx = (R)0.5 * dt * w.x, hy = (R)0.5 * dt * w.y, hz = (R)0.5 * dt * w.z;
  R nx = qx + (hx * qw + hy * qz - hz * qy);
  R ny = qy + (-hx * qz + hy * qw + hz * qx);
  R nz = qz + (hx * qy - hy * qx + hz * qw);
  R nw = qw + (-hx * qx - hy * qy - hz * qz);
  R inv = 1 / std::sqrt(nx * nx + ny * ny + nz * nz + nw * nw);
  S[L.root + 0] = (float)pos.x; S[L.root + 1] = (float)pos.y; S[L.root + 2] = (float)pos.z;
  S[L.root + 3] = (float)(nx * inv); S[L.root + 4] = (float)(ny * inv); S[L.root + 5] = (float)(nz * inv); S[L.root + 6] = (float)(nw * inv);
  for (int t = 0; t < 6; ++t) S[L.root + 7 + t] = (float)vel[t];
  for (int k = 0; k < nd; ++k) { S[L.dof_pos + k] = (float)q[k]; S[L.dof_vel + k] = (float)vel[6 + k]; }
  // ---- net contact force per body (world), feet positions (link origins, pre-integration pose advanced by dt*v)
  for (int i = 0; i < nb * 3; ++i) S[L.contact_forces + i] = 0.0f;
  for (int ci = 0; ci < ncon; ++ci) {
    const Contact& C = cons[ci];
    V3<R> F = (rows[3 * ci].lam / dt) * C.n + (rows[3 * ci + 1].lam / dt) * C.t1 + (rows[3 * ci + 2].lam / dt) * C.t2;
    S[L.contact_forces + 3 * C.body + 0] += (float)F.x;
    S[L.contact_forces + 3 * C.body + 1] += (float)F.y;
    S[L.contact_forces + 3 * C.body + 2] += (float)F.z;
    if (C.body_b >= 0) {      // self-contact: the opposite force on the other link
      S[L.contact_forces + 3 * C.body_b + 0] -= (float)F.x;
      S[L.contact_forces + 3 * C.body_b + 1] -= (float)F.y;
      S[L.contact_forces + 3 * C.body_b + 2] -= (float)F.z;
    }
  }
}

// world positions of the feet link origins for the CURRENT state (after the last substep)
template <typename R>
void feet_positions(const Ctx& x, float* S) {
  const PfModel& m = *x.m; const PfLayout& L = *x.L;
  const int nb = m.nb;
  R quat[4] = {(R)S[L.root + 3], (R)S[L.root + 4], (R)S[L.root + 5], (R)S[L.root + 6]};
  M3<R> Rw[PF_MAX_BODIES]; V3<R> r[PF_MAX_BODIES];
  Rw[0] = quat_to_mat(quat); r[0] = {0, 0, 0};
  for (int i = 1; i < nb; ++i) {
    int p = m.parent[i];
    M3<R> Rf; for (int t = 0; t < 9; ++t) Rf.m[t] = m.joint_rot[i][t];
    M3<R> Rj = mul(Rw[p], Rf);
    V3<R> jp = {(R)m.joint_pos[i][0], (R)m.joint_pos[i][1], (R)m.joint_pos[i][2]};
    r[i] = r[p] + mul(Rw[p], jp);
    int k = m.dof_of_body[i];
    if (k >= 0) {
      V3<R> al = {(R)m.joint_axis[i][0], (R)m.joint_axis[i][1], (R)m.joint_axis[i][2]};
      Rw[i] = mul(Rj, axis_angle(al, (R)S[L.dof_pos + k]));
    } else Rw[i] = Rj;
  }
  for (int fI = 0; fI < m.nfeet; ++fI) {
    int i = m.feet_body[fI];
    S[L.feet_pos + 3 * fI + 0] = (float)((R)S[L.root + 0] + r[i].x);
    S[L.feet_pos + 3 * fI + 1] = (float)((R)S[L.root + 1] + r[i].y);
    S[L.feet_pos + 3 * fI + 2] = (float)((R)S[L.root + 2] + r[i].z);
  }
}

// =================================================================================== python layer
// point_foot.py:320-382 compute_observations (+ :328-338 noise, + :189-192 clip in step())
void compute_observations(const Ctx& x, int env, float* S, const float* heights) {
  const PfModel& m = *x.m; const PfConfig& c = *x.c; const PfLayout& L = *x.L;
  const int nd = m.nd, na = c.num_actions, no = c.num_obs, np_ = c.num_priv_obs;
  if (c.env_kind == PF_KIND_LEGGED_ROBOT) {     // legged_robot.py:209-226
    float* obs = x.b->d_obs + (size_t)env * no;
    int w = 0;
    float v[PF_MAX_OBS];
    for (int i = 0; i < 3; ++i) v[w++] = S[L.base_lin_vel + i] * c.obs_scale_lin_vel;
    for (int i = 0; i < 3; ++i) v[w++] = S[L.base_ang_vel + i] * c.obs_scale_ang_vel;
    for (int i = 0; i < 3; ++i) v[w++] = S[L.projected_gravity + i];
    for (int i = 0; i < 3; ++i) v[w++] = S[L.commands + i] * c.commands_scale[i];
    for (int k = 0; k < nd; ++k) v[w++] = (S[L.dof_pos + k] - c.default_dof_pos[k]) * c.obs_scale_dof_pos;
    for (int k = 0; k < nd; ++k) v[w++] = S[L.dof_vel + k] * c.obs_scale_dof_vel;
    for (int k = 0; k < na; ++k) v[w++] = S[L.actions + k];
    if (c.measure_heights)
      for (int j = 0; j < c.num_height_points; ++j) v[w++] = clipf((S[L.root + 2] - 0.5f) - heights[j], -1.0f, 1.0f) * c.obs_scale_height;
    for (int i = 0; i < no; ++i) {
      float t = v[i];
      if (c.add_noise) t += (2.0f * rnd(x, env, L.r_obs + i) - 1.0f) * c.obs_noise_vec[i];
      obs[i] = clipf(t, -c.clip_obs, c.clip_obs);
    }
    return;
  }
  float o[PF_MAX_OBS];
  int w = 0;
  for (int i = 0; i < 3; ++i) o[w++] = S[L.base_ang_vel + i] * c.obs_scale_ang_vel;
  for (int i = 0; i < 3; ++i) o[w++] = S[L.projected_gravity + i];
  for (int k = 0; k < nd; ++k) o[w++] = (S[L.dof_pos + k] - c.default_dof_pos[k]) * c.obs_scale_dof_pos;
  for (int k = 0; k < nd; ++k) o[w++] = S[L.dof_vel + k] * c.obs_scale_dof_vel;
  for (int k = 0; k < na; ++k) o[w++] = S[L.actions + k];
  for (int i = 0; i < 3; ++i) o[w++] = S[L.commands + i] * c.commands_scale[i];
  // cfg.terrain.measure_heights_actor (point_foot.py:361-373): the measurements follow, num_obs = 27 + 121
  for (int j = 0; w < no && j < c.num_height_points; ++j) o[w++] = clipf((S[L.root + 2] - 0.5f) - heights[j], -1.0f, 1.0f) * c.obs_scale_height;
  float* obs = x.b->d_obs + (size_t)env * no;
  float* priv = x.b->d_priv ? x.b->d_priv + (size_t)env * np_ : nullptr;
  for (int i = 0; i < no; ++i) {
    float v = o[i], pv = o[i];
    if (c.add_noise) {
      float nz = (2.0f * rnd(x, env, L.r_obs + i) - 1.0f) * c.obs_noise_vec[i];
      v += nz; pv += nz;
    }
    obs[i] = clipf(v, -c.clip_obs, c.clip_obs);
    if (priv) priv[i] = clipf(pv, -c.clip_obs, c.clip_obs);
  }
  if (priv && c.measure_heights && np_ > no) {      // critic-only heights; with actor heights the copy above already holds them (:340-345)
    for (int j = 0; j < c.num_height_points; ++j) {   // :367-373
      float hgt = clipf((S[L.root + 2] - 0.5f) - heights[j], -1.0f, 1.0f) * c.obs_scale_height;
      if (c.add_noise) hgt += (2.0f * rnd(x, env, L.r_priv + j) - 1.0f) * c.priv_height_noise;
      priv[no + j] = clipf(hgt, -c.clip_obs, c.clip_obs);
    }
  }
}

// point_foot.py:195-230 post_physics_step for one env
void post_physics(const Ctx& x, int env, float* S) {
  const PfModel& m = *x.m; const PfConfig& c = *x.c; const PfLayout& L = *x.L; const PfBuffers& B = *x.b;
  const int nd = m.nd, nf = m.nfeet, H = c.num_height_points;
  int64_t ep = B.d_episode_length[env] + 1;                  // :204
  const bool LR = c.env_kind == PF_KIND_LEGGED_ROBOT;
  if (LR && c.resample_interval > 0 && (ep % c.resample_interval) == 0) resample_commands(x, env, S, L.r_cmd);   // legged_robot.py:323-324
  // :208-211
  const float* q = S + L.root + 3;
  float grav[3] = {0.0f, 0.0f, -1.0f};
  quat_rotate_dir(q, S + L.root + 7, -1.0f, S + L.base_lin_vel);
  quat_rotate_dir(q, S + L.root + 10, -1.0f, S + L.base_ang_vel);
  quat_rotate_dir(q, grav, -1.0f, S + L.projected_gravity);
  // :212-213, 1048-1075
  float heights[PF_MAX_HPTS];
  if (c.measure_heights) {
    for (int j = 0; j < H; ++j) {
      if (c.terrain_type == PF_TERRAIN_PLANE) { heights[j] = 0.0f; continue; }
      float pl[3] = {c.height_px[j], c.height_py[j], 0.0f}, pw[3];
      quat_apply_yaw(q, pl, pw);
      float wx = pw[0] + S[L.root + 0], wy = pw[1] + S[L.root + 1];
      heights[j] = (float)terrain_height_raw(x, wx, wy) * c.vscale;
    }
    if (B.d_measured_heights) std::memcpy(B.d_measured_heights + (size_t)env * H, heights, sizeof(float) * H);
  }
  // :1114-1129 _compute_feet_states (PointFoot only)
  for (int f = 0; f < (LR ? 0 : nf); ++f) {
    bool first = S[L.first_contact + f] != 0.0f, filt = S[L.contact_filt + f] != 0.0f;
    S[L.last_feet_air_time + f] = first ? S[L.feet_air_time + f] : S[L.last_feet_air_time + f];
    if (filt) S[L.feet_air_time + f] = 0.0f * S[L.feet_air_time + f];
    if (c.include_feet_height) {
      S[L.last_max_feet_height + f] = first ? S[L.cur_max_feet_height + f] : S[L.last_max_feet_height + f];
      if (filt) S[L.cur_max_feet_height + f] = 0.0f * S[L.cur_max_feet_height + f];
      float hb = 0.0f;                                         // :1077-1098
      if (c.terrain_type != PF_TERRAIN_PLANE)
        hb = (float)terrain_height_raw(x, S[L.feet_pos + 3 * f], S[L.feet_pos + 3 * f + 1]) * c.vscale;
      S[L.feet_height + f] = S[L.feet_pos + 3 * f + 2] - hb;
      S[L.cur_max_feet_height + f] = std::max(S[L.cur_max_feet_height + f], S[L.feet_height + f]);
    }
    int fb = m.feet_body[f];
    bool contact = S[L.contact_forces + 3 * fb + 2] > 1.0f;
    bool nfilt = contact || (S[L.last_contacts + f] != 0.0f);
    S[L.contact_filt + f] = nfilt ? 1.0f : 0.0f;
    S[L.last_contacts + f] = contact ? 1.0f : 0.0f;
    S[L.first_contact + f] = (S[L.feet_air_time + f] > 0.0f && nfilt) ? 1.0f : 0.0f;
    S[L.feet_air_time + f] += c.dt;
  }
  // :479-493 callback (resampling is `pass`, :521)
  if (c.heading_command) {
    float fwd[3] = {1.0f, 0.0f, 0.0f}, fw[3];
    quat_apply(q, fwd, fw);
    float heading = std::atan2(fw[1], fw[0]);
    S[L.commands + 2] = clipf(0.5f * wrap_to_pi(S[L.commands + 3] - heading), -1.0f, 1.0f);
  }
  // the base-frame velocities used by rewards/obs were taken before the push (legged_robot.py:112-117 vs :334-335)
  float blv_[3] = {S[L.base_lin_vel], S[L.base_lin_vel + 1], S[L.base_lin_vel + 2]};
  (void)blv_;
  if (LR && c.push_robots && (x.step_counter % c.push_interval == 0)) {   // legged_robot.py:414-419
    for (int i = 0; i < 2; ++i) S[L.root + 7 + i] = (c.max_push_vel - (-c.max_push_vel)) * rnd(x, env, L.r_push + i) + (-c.max_push_vel);
  }
  if (!LR && c.push_robots && (x.step_counter % c.push_interval == 0)) {   // :590-611
    float f[3];
    for (int i = 0; i < 3; ++i) f[i] = (c.max_push_force - (-c.max_push_force)) * rnd(x, env, L.r_push + i) + (-c.max_push_force);
    quat_rotate_dir(q, f, +1.0f, S + L.ext_force);
    S[L.ext_force + 2] *= 0.5f;
    S[L.ext_force + 3] = 1.0f;   // pending flag: applied during the next substep only
  }
  // :239-247 check_termination
  bool term = false;
  for (int t = 0; t < m.n_term; ++t) {
    const float* F = S + L.contact_forces + 3 * m.term_body[t];
    term = term || (std::sqrt(F[0] * F[0] + F[1] * F[1] + F[2] * F[2]) > 1.0f);
  }
  bool time_out = (float)ep > c.max_episode_length;
  bool reset = term || time_out;
  B.d_terminate[env] = term; B.d_time_out[env] = time_out;
  // :301-318 compute_reward
  float rew = 0.0f;
  for (int ri = 0; ri < c.num_rewards; ++ri) {
    float v = 0.0f;
    switch (c.reward_id[ri]) {
      case PF_REW_LIN_VEL_Z: v = S[L.base_lin_vel + 2] * S[L.base_lin_vel + 2]; break;
      case PF_REW_ANG_VEL_XY: v = S[L.base_ang_vel] * S[L.base_ang_vel] + S[L.base_ang_vel + 1] * S[L.base_ang_vel + 1]; break;
      case PF_REW_ORIENTATION: v = S[L.projected_gravity] * S[L.projected_gravity] + S[L.projected_gravity + 1] * S[L.projected_gravity + 1]; break;
      case PF_REW_BASE_HEIGHT: {
        float bh;
        if (c.measure_heights) { float s = 0.0f; for (int j = 0; j < H; ++j) s += S[L.root + 2] - heights[j]; bh = s / (float)H; }
        else bh = S[L.root + 2];          // measured_heights == 0 (:775)
        v = (bh - c.base_height_target) * (bh - c.base_height_target);
      } break;
      case PF_REW_TORQUES: for (int k = 0; k < nd; ++k) v += S[L.torques + k] * S[L.torques + k]; break;
      case PF_REW_DOF_VEL: for (int k = 0; k < nd; ++k) v += S[L.dof_vel + k] * S[L.dof_vel + k]; break;
      case PF_REW_DOF_ACC: for (int k = 0; k < nd; ++k) { float a = (S[L.last_dof_vel + k] - S[L.dof_vel + k]) / c.dt; v += a * a; } break;
      case PF_REW_ACTION_RATE: for (int k = 0; k < c.num_actions; ++k) { float a = S[L.last_actions + k] - S[L.actions + k]; v += a * a; } break;
      case PF_REW_COLLISION:
        for (int t = 0; t < m.n_pen; ++t) { const float* F = S + L.contact_forces + 3 * m.pen_body[t]; v += (std::sqrt(F[0] * F[0] + F[1] * F[1] + F[2] * F[2]) > 0.1f) ? 1.0f : 0.0f; }
        break;
      case PF_REW_TERMINATION: v = (reset && !time_out) ? 1.0f : 0.0f; break;
      case PF_REW_DOF_POS_LIMITS:
        for (int k = 0; k < nd; ++k) { float o = -std::min(S[L.dof_pos + k] - c.dof_pos_soft_lo[k], 0.0f); o += std::max(S[L.dof_pos + k] - c.dof_pos_soft_hi[k], 0.0f); v += o; }
        break;
      case PF_REW_DOF_VEL_LIMITS:
        for (int k = 0; k < nd; ++k) v += clipf(std::fabs(S[L.dof_vel + k]) - c.dof_vel_limits[k] * c.soft_dof_vel_limit, 0.0f, 1.0f);
        break;
      case PF_REW_TORQUE_LIMITS:
        for (int k = 0; k < nd; ++k) v += std::max(std::fabs(S[L.torques + k]) - c.torque_limits[k] * c.soft_torque_limit, 0.0f);
        break;
      case PF_REW_TRACKING_LIN_VEL: {
        float ex = S[L.commands] - S[L.base_lin_vel], ey = S[L.commands + 1] - S[L.base_lin_vel + 1];
        v = std::exp(-(ex * ex + ey * ey) / c.tracking_sigma);
      } break;
      case PF_REW_TRACKING_ANG_VEL: { float e = S[L.commands + 2] - S[L.base_ang_vel + 2]; v = std::exp(-(e * e) / c.tracking_sigma); } break;
      case PF_REW_FEET_AIR_TIME: if (LR) {      // legged_robot.py:882-893
        float acc = 0.0f;
        for (int f = 0; f < nf; ++f) {
          bool contact = S[L.contact_forces + 3 * m.feet_body[f] + 2] > 1.0f;
          bool filt = contact || (S[L.last_contacts + f] != 0.0f);
          S[L.last_contacts + f] = contact ? 1.0f : 0.0f;
          float first = (S[L.feet_air_time + f] > 0.0f && filt) ? 1.0f : 0.0f;
          S[L.feet_air_time + f] += c.dt;
          acc += (S[L.feet_air_time + f] - 0.5f) * first;
          if (filt) S[L.feet_air_time + f] = S[L.feet_air_time + f] * 0.0f;
        }
        float cn = std::sqrt(S[L.commands] * S[L.commands] + S[L.commands + 1] * S[L.commands + 1]);
        v = acc * (cn > 0.1f ? 1.0f : 0.0f);
      } else {
        float lo = 0.0f, hi = 0.0f;
        for (int f = 0; f < nf; ++f) {
          float fc = S[L.first_contact + f];
          lo += std::min(S[L.feet_air_time + f] - c.min_feet_air_time, 0.0f) * fc;
          hi += std::min(c.max_feet_air_time - S[L.feet_air_time + f], 0.0f) * fc;
        }
        v = lo + hi;
      } break;
      case PF_REW_NO_FLY: { float n = 0.0f; for (int f = 0; f < nf; ++f) n += (S[L.contact_forces + 3 * m.feet_body[f] + 2] > 0.1f) ? 1.0f : 0.0f; v = (n == 1.0f) ? 1.0f : 0.0f; } break;
      case PF_REW_UNBALANCE_FEET_AIR_TIME:
      case PF_REW_UNBALANCE_FEET_HEIGHT: {
        const float* a = S + (c.reward_id[ri] == PF_REW_UNBALANCE_FEET_AIR_TIME ? L.last_feet_air_time : L.last_max_feet_height);
        float mean = 0.0f; for (int f = 0; f < nf; ++f) mean += a[f]; mean /= (float)nf;
        float s = 0.0f; for (int f = 0; f < nf; ++f) s += (a[f] - mean) * (a[f] - mean);
        v = s / (float)(nf - 1);     // torch.var unbiased
      } break;
      case PF_REW_STUMBLE:
        for (int f = 0; f < nf; ++f) { const float* F = S + L.contact_forces + 3 * m.feet_body[f]; if (std::sqrt(F[0] * F[0] + F[1] * F[1]) > 5.0f * std::fabs(F[2])) v = 1.0f; }
        break;
      case PF_REW_STAND_STILL: if (LR) {      // legged_robot.py:900-902
        float sm = 0.0f;
        for (int k = 0; k < nd; ++k) sm += std::fabs(S[L.dof_pos + k] - c.default_dof_pos[k]);
        float cn = std::sqrt(S[L.commands] * S[L.commands] + S[L.commands + 1] * S[L.commands + 1]);
        v = sm * (cn < 0.1f ? 1.0f : 0.0f);
      } else {
        v = (std::fabs(S[L.base_lin_vel]) * (S[L.commands] < 0.1f ? 1.0f : 0.0f) + std::fabs(S[L.base_lin_vel + 1]) * (S[L.commands + 1] < 0.1f ? 1.0f : 0.0f)) +
            std::fabs(S[L.base_ang_vel + 2]) * (S[L.commands + 2] < 0.1f ? 1.0f : 0.0f);
      }
        break;
      case PF_REW_FEET_CONTACT_FORCES:
        for (int f = 0; f < nf; ++f) { const float* F = S + L.contact_forces + 3 * m.feet_body[f]; v += std::max(std::sqrt(F[0] * F[0] + F[1] * F[1] + F[2] * F[2]) - c.max_contact_force, 0.0f); }
        break;
      case PF_REW_FEET_DISTANCE:   // :1242-1250 (the += sits outside the inner loop)
        for (int i = 0; i < nf - 1; ++i) {
          int j = nf - 1;
          float dx = S[L.feet_pos + 3 * i] - S[L.feet_pos + 3 * j], dy = S[L.feet_pos + 3 * i + 1] - S[L.feet_pos + 3 * j + 1];
          v += clipf(c.min_feet_distance - std::sqrt(dx * dx + dy * dy), 0.0f, 1.0f);
        }
        break;
      case PF_REW_SURVIVAL: v = (reset ? 0.0f : 1.0f) * c.dt; break;
    }
    float r = v * c.reward_scale[ri];
    rew += r;
    S[L.episode_sums + ri] += r;
  }
  if (c.only_positive_rewards) rew = std::max(rew, 0.0f);
  if (c.termination_scale != 0.0f) {
    float r = ((reset && !time_out) ? 1.0f : 0.0f) * c.termination_scale;
    rew += r; S[L.episode_sums + c.num_rewards] += r;
  }
  B.d_rew[env] = rew;
  // :249-299 reset_idx (masked)
  if (reset) {
    if (c.curriculum) {                                    // :613-636
      float dx = S[L.root] - S[L.env_origin], dy = S[L.root + 1] - S[L.env_origin + 1];
      float dist = std::sqrt(dx * dx + dy * dy);
      bool up = dist > c.terrain_length / 2.0f;
      float cn = std::sqrt(S[L.commands] * S[L.commands] + S[L.commands + 1] * S[L.commands + 1]);
      bool down = (dist < cn * c.max_episode_length_s * 0.5f) && !up;
      int lvl = B.d_terrain_level[env] + (up ? 1 : 0) - (down ? 1 : 0);
      if (lvl >= c.max_terrain_level) {
        if (x.inj && x.inj->d_rand) lvl = (int)x.inj->d_rand[(size_t)env * L.rand_floats + L.r_level];
        else lvl = std::min((int)(rnd(x, env, L.r_level) * (float)c.max_terrain_level), c.max_terrain_level - 1);
      } else lvl = std::max(lvl, 0);
      B.d_terrain_level[env] = lvl;
      const float* o = B.d_terrain_origins + ((size_t)lvl * c.terrain_num_cols + B.d_terrain_type[env]) * 3;
      for (int i = 0; i < 3; ++i) S[L.env_origin + i] = o[i];
    }
    for (int k = 0; k < nd; ++k) {                         // :549-559
      S[L.dof_pos + k] = c.default_dof_pos[k] * ((1.5f - 0.5f) * rnd(x, env, L.r_dof + k) + 0.5f);
      S[L.dof_vel + k] = 0.0f;
    }
    for (int i = 0; i < 13; ++i) S[L.root + i] = c.base_init_state[i];     // :566-584
    for (int i = 0; i < 3; ++i) S[L.root + i] += S[L.env_origin + i];
    if (c.custom_origins)
      for (int i = 0; i < 2; ++i) S[L.root + i] += (1.0f - (-1.0f)) * rnd(x, env, L.r_xy + i) + (-1.0f);
    for (int i = 0; i < 6; ++i) S[L.root + 7 + i] = (0.5f - (-0.5f)) * rnd(x, env, L.r_vel + i) + (-0.5f);
    if (LR) { resample_commands(x, env, S, L.r_cmd_reset); for (int e = 0; e < 32 * nd; ++e) S[L.lstm_state + e] = 0.0f; }   // legged_robot.py:168, anymal.py:56-60
    for (int f = 0; f < nf; ++f) {                         // :290-299
      S[L.feet_air_time + f] = 0.0f; S[L.last_feet_air_time + f] = 0.0f;
      S[L.cur_max_feet_height + f] = 0.0f; S[L.last_max_feet_height + f] = 0.0f;
    }
    ep = 0;
    // :276-280 episode statistics
    int nsum = c.num_rewards + (c.termination_scale != 0.0f ? 1 : 0);
    if (B.d_reset_stats) {
#pragma omp critical
      {
        for (int ri = 0; ri < nsum; ++ri) B.d_reset_stats[ri] += S[L.episode_sums + ri];
        B.d_reset_stats[PF_MAX_REWARDS] += 1.0f;
      }
    }
    for (int ri = 0; ri < nsum; ++ri) S[L.episode_sums + ri] = 0.0f;
  }
  B.d_reset[env] = reset;
  B.d_episode_length[env] = ep;
  compute_observations(x, env, S, heights);                 // :223
  for (int k = 0; k < c.num_actions; ++k) S[L.last_actions + k] = S[L.actions + k];   // :225-227
  for (int k = 0; k < nd; ++k) S[L.last_dof_vel + k] = S[L.dof_vel + k];
  for (int i = 0; i < 6; ++i) S[L.last_root_vel + i] = S[L.root + 7 + i];
}

int g_threads = 1;

}  // namespace

extern "C" {

void pfo_set_threads(int n) { g_threads = n > 0 ? n : 1; }

/* PointFoot.step (point_foot.py:170-193).  physics_bits: 32 or 64 (precision of the own physics). */
int pfo_step(const PfModel* model, const PfConfig* cfg, const PfLayout* layout, const PfBuffers* bufs,
             const float* actions, int64_t step_counter, const PfInject* inject, int physics_bits) {
  Ctx x{model, cfg, layout, bufs, inject, step_counter};
  const PfLayout& L = *layout; const PfConfig& c = *cfg;
  const int N = c.num_envs, nd = model->nd;
  const bool injected_sim = inject && inject->d_root;
  if (bufs->d_reset_stats) std::memset(bufs->d_reset_stats, 0, sizeof(float) * (PF_MAX_REWARDS + 4));
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int env = 0; env < N; ++env) {
    float* S = bufs->d_state + (size_t)env * L.rec_floats;
    for (int k = 0; k < c.num_actions; ++k) S[L.actions + k] = clipf(actions[(size_t)env * c.num_actions + k], -c.clip_actions, c.clip_actions);  // :176-177
    if (injected_sim) {
      // Python-layer parity mode: last torque from the injected pre-simulate DOF state, then adopt the
      // injected post-simulate snapshot (what gym.refresh_* would have delivered, :185, :200-202)
      float qp[PF_MAX_DOF], qv[PF_MAX_DOF];
      for (int k = 0; k < nd; ++k) { qp[k] = inject->d_dof_pre[((size_t)env * nd + k) * 2]; qv[k] = inject->d_dof_pre[((size_t)env * nd + k) * 2 + 1]; }
      compute_torques(x, S, qp, qv);
      for (int k = 0; k < nd; ++k) { S[L.dof_pos + k] = inject->d_dof_post[((size_t)env * nd + k) * 2]; S[L.dof_vel + k] = inject->d_dof_post[((size_t)env * nd + k) * 2 + 1]; }
      for (int i = 0; i < 13; ++i) S[L.root + i] = inject->d_root[(size_t)env * 13 + i];
      for (int i = 0; i < model->nb * 3; ++i) S[L.contact_forces + i] = inject->d_contact[(size_t)env * model->nb * 3 + i];
      for (int i = 0; i < model->nfeet * 3; ++i) S[L.feet_pos + i] = inject->d_feet_pos[(size_t)env * model->nfeet * 3 + i];
    } else {
      // The substeps run in coordinates local to the grid cell the base starts the step in: S[root], S[root + 1] hold
      // metres from that cell's corner, so contact geometry is resolved to ~1e-7 m wherever the env sits on the field
      // (float32 world coordinates are only good to 4e-6 m at 50 m from the origin).  The world position gets the step's
      // displacement added once at the end.
      const float gx = S[L.root], gy = S[L.root + 1];
      int ix0 = 0, iy0 = 0;
      if (c.terrain_type != PF_TERRAIN_PLANE && bufs->d_height_samples) {
        const float cx = (gx + c.border) / c.hscale, cy = (gy + c.border) / c.hscale;
        const float bx = std::floor(cx), by = std::floor(cy);
        ix0 = (int)bx; iy0 = (int)by;
        S[L.root] = (cx - bx) * c.hscale; S[L.root + 1] = (cy - by) * c.hscale;
      }
      const float lx0 = S[L.root], ly0 = S[L.root + 1];
      for (int sub = 0; sub < c.decimation; ++sub) {   // :180-185
        if (c.env_kind == PF_KIND_LEGGED_ROBOT && c.use_actuator_net && bufs->d_actuator_params) actuator_torques(x, S);
        else compute_torques(x, S, S + L.dof_pos, S + L.dof_vel);
        bool ext = sub == 0 && S[L.ext_force + 3] != 0.0f;
        if (physics_bits == 64) physics_substep<double>(x, S, ext, env, sub, ix0, iy0); else physics_substep<float>(x, S, ext, env, sub, ix0, iy0);
        if (ext) S[L.ext_force + 3] = 0.0f;
      }
      S[L.root] = gx + (S[L.root] - lx0); S[L.root + 1] = gy + (S[L.root + 1] - ly0);
      if (physics_bits == 64) feet_positions<double>(x, S); else feet_positions<float>(x, S);
    }
    post_physics(x, env, S);
  }
  return 0;
}

/* reset_idx(arange(N)) as called by PointFoot.reset() (:137-139) before init_done bookkeeping matters:
 * curriculum untouched (:620-622 returns while !init_done is only at construction; reset() after
 * construction does run it, with distance 0 -> no change unless commands are set). */
int pfo_compute_observations(const PfModel* model, const PfConfig* cfg, const PfLayout* layout, const PfBuffers* bufs,
                             int64_t step_counter, const PfInject* inject, const float* heights_or_null) {
  Ctx x{model, cfg, layout, bufs, inject, step_counter};
  std::vector<float> zeros(PF_MAX_HPTS, 0.0f);
  for (int env = 0; env < cfg->num_envs; ++env) {
    const float* h = heights_or_null ? heights_or_null + (size_t)env * cfg->num_height_points : zeros.data();
    compute_observations(x, env, bufs->d_state + (size_t)env * layout->rec_floats, h);
  }
  return 0;
}

/* diagnostics for the invariant tests: total mechanical energy and momentum of env `env` */
int pfo_energy(const PfModel* model, const PfConfig* cfg, const PfLayout* layout, const float* state_row, double* out /*[KE, PE, px,py,pz]*/) {
  const PfModel& m = *model; const PfLayout& L = *layout;
  const float* S = state_row;
  typedef double R;
  R quat[4] = {S[L.root + 3], S[L.root + 4], S[L.root + 5], S[L.root + 6]};
  M3<R> Rw[PF_MAX_BODIES]; V3<R> r[PF_MAX_BODIES]; Sp6<R> V[PF_MAX_BODIES];
  Rw[0] = quat_to_mat(quat); r[0] = {0, 0, 0};
  V[0] = {{S[L.root + 10], S[L.root + 11], S[L.root + 12]}, {S[L.root + 7], S[L.root + 8], S[L.root + 9]}};
  double KE = 0, PE = 0; V3<R> P = {0, 0, 0};
  for (int i = 0; i < m.nb; ++i) {
    if (i > 0) {
      int p = m.parent[i];
      M3<R> Rf; for (int t = 0; t < 9; ++t) Rf.m[t] = m.joint_rot[i][t];
      M3<R> Rj = mul(Rw[p], Rf);
      V3<R> jp = {m.joint_pos[i][0], m.joint_pos[i][1], m.joint_pos[i][2]};
      r[i] = r[p] + mul(Rw[p], jp);
      int k = m.dof_of_body[i];
      if (k >= 0) {
        V3<R> al = {m.joint_axis[i][0], m.joint_axis[i][1], m.joint_axis[i][2]};
        V3<R> a = mul(Rj, al);
        Rw[i] = mul(Rj, axis_angle(al, (R)S[L.dof_pos + k]));
        R qd = S[L.dof_vel + k];
        V[i] = {V[p].a + qd * a, V[p].l + qd * cross(r[i], a)};
      } else { Rw[i] = Rj; V[i] = V[p]; }
    }
    R mass = m.mass[i]; V3<R> cl = {m.com[i][0], m.com[i][1], m.com[i][2]};
    R Il[6]; for (int t = 0; t < 6; ++t) Il[t] = m.inertia[i][t];
    if (i == 0) { R added = S[L.added_mass]; R sc = (mass + added) / mass; mass += added; cl = cl + V3<R>{S[L.added_com], S[L.added_com + 1], S[L.added_com + 2]}; for (int t = 0; t < 6; ++t) Il[t] *= sc; }
    V3<R> cw = r[i] + mul(Rw[i], cl);
    V3<R> vc = V[i].l + cross(V[i].a, cw);      // COM velocity
    M3<R> Rt = {{Rw[i].m[0], Rw[i].m[3], Rw[i].m[6], Rw[i].m[1], Rw[i].m[4], Rw[i].m[7], Rw[i].m[2], Rw[i].m[5], Rw[i].m[8]}};
    V3<R> wl = mul(Rt, V[i].a);
    V3<R> Iw = symmul(Il, wl);
    KE += 0.5 * mass * dot(vc, vc) + 0.5 * dot(wl, Iw);
    PE += mass * (-(R)cfg->gravity[2]) * ((R)S[L.root + 2] + cw.z);
    P = P + mass * vc;
  }
  out[0] = KE; out[1] = PE; out[2] = P.x; out[3] = P.y; out[4] = P.z;
  return 0;
}

}  // extern "C"

// pf_lump.h — host side: the robot model the step kernel integrates.  Bodies behind FIXED joints (the feet: 2 of PointFoot's 9
// bodies, 4 of a quadruped's 17; collapse_fixed_joints leaves them as bodies of their own because the URDF marks them
// dont_collapse, point_foot.py:883-909) are rigidly attached to their parents: their mass, centre of mass and inertia are
// lumped into the parent link, their collision spheres move to the parent's frame, and every root-to-leaf chain of the
// kinematics / composite-inertia / bias recursions is one level shorter.  The dynamics are the same rigid-body system; what
// the env REPORTS per body (net contact forces, feet positions: the record layout of pf_layout.h) keeps the original bodies:
// sph_rep[] names the body a sphere's contact force is booked on, feet_dyn[] / feet_off[] give a foot's frame origin on its
// new carrier.  Used by pf_create (device copy) and by the host build of the step program (tests/emu).
#pragma once
#include <cstring>
#include "../../include/pointfoot_hip.h"

struct PfLump {
  int32_t sph_rep[PF_MAX_SPHERES];      // original body of sphere s (index into the record's contact_forces)
  int32_t feet_dyn[PF_MAX_FEET];        // lumped-model body that carries foot f
  float feet_off[PF_MAX_FEET][3];       // origin of the foot's link frame in that body's frame
};

// out: the lumped model (out->nb <= in->nb; everything that is not per body is copied), map: reporting tables.
// Bodies are in depth-first order with parent[b] < b (PfModel contract), which the lumped model keeps.
static inline void pf_lump_model(const PfModel* in, PfModel* out, PfLump* map) {
  const int nb = in->nb;
  std::memcpy(out, in, sizeof(PfModel));
  std::memset(map, 0, sizeof(PfLump));
  int dyn[PF_MAX_BODIES];                    // lumped-model body of original body i
  double R[PF_MAX_BODIES][9], t[PF_MAX_BODIES][3];      // frame of original body i in its lumped body's frame
  // accumulated mass properties of every lumped body: mass, sum of m c, and the parts (for the parallel-axis pass)
  int nd = 0;
  auto matmul = [](const double* a, const double* b, double* c) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += a[3 * i + k] * b[3 * k + j]; c[3 * i + j] = s; }
  };
  auto matvec = [](const double* a, const double* v, double* o) { for (int i = 0; i < 3; ++i) o[i] = a[3 * i] * v[0] + a[3 * i + 1] * v[1] + a[3 * i + 2] * v[2]; };
  for (int i = 0; i < nb; ++i) {
    const int p = in->parent[i];
    const bool fixed = i > 0 && in->dof_of_body[i] < 0;
    double Rj[9], pj[3];
    for (int k = 0; k < 9; ++k) Rj[k] = in->joint_rot[i][k];
    for (int k = 0; k < 3; ++k) pj[k] = in->joint_pos[i][k];
    if (!fixed) {
      dyn[i] = nd;
      for (int k = 0; k < 9; ++k) R[i][k] = (k % 4 == 0) ? 1.0 : 0.0;
      t[i][0] = t[i][1] = t[i][2] = 0.0;
      // the joint hangs off the parent's frame, which may itself sit inside a lumped body
      double Rn[9], tn[3] = {pj[0], pj[1], pj[2]};
      std::memcpy(Rn, Rj, sizeof(Rn));
      if (i > 0) { matmul(R[p], Rj, Rn); matvec(R[p], pj, tn); for (int k = 0; k < 3; ++k) tn[k] += t[p][k]; }
      out->parent[nd] = i > 0 ? dyn[p] : -1;
      out->dof_of_body[nd] = in->dof_of_body[i];
      for (int k = 0; k < 9; ++k) out->joint_rot[nd][k] = (float)Rn[k];
      for (int k = 0; k < 3; ++k) { out->joint_pos[nd][k] = (float)tn[k]; out->joint_axis[nd][k] = in->joint_axis[i][k]; }
      ++nd;
    } else {
      dyn[i] = dyn[p];
      double tt[3];
      matmul(R[p], Rj, R[i]); matvec(R[p], pj, tt);
      for (int k = 0; k < 3; ++k) t[i][k] = tt[k] + t[p][k];
    }
  }
  out->nb = nd;
  // mass properties: mass and centre of mass first, then the inertia about the combined centre of mass
  double M[PF_MAX_BODIES] = {0}, C[PF_MAX_BODIES][3] = {{0}}, I[PF_MAX_BODIES][9] = {{0}};
  double cw[PF_MAX_BODIES][3];
  for (int i = 0; i < nb; ++i) {
    double c[3] = {in->com[i][0], in->com[i][1], in->com[i][2]};
    matvec(R[i], c, cw[i]);
    for (int k = 0; k < 3; ++k) { cw[i][k] += t[i][k]; C[dyn[i]][k] += in->mass[i] * cw[i][k]; }
    M[dyn[i]] += in->mass[i];
  }
  for (int d = 0; d < nd; ++d) for (int k = 0; k < 3; ++k) C[d][k] = M[d] > 0 ? C[d][k] / M[d] : 0.0;
  for (int i = 0; i < nb; ++i) {
    const float* a = in->inertia[i];
    const double Il[9] = {a[0], a[3], a[4], a[3], a[1], a[5], a[4], a[5], a[2]};
    double T[9], Rt[9], Iw[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rt[3 * r + c] = R[i][3 * c + r];
    matmul(R[i], Il, T); matmul(T, Rt, Iw);
    const int d = dyn[i];
    const double e[3] = {cw[i][0] - C[d][0], cw[i][1] - C[d][1], cw[i][2] - C[d][2]}, e2 = e[0] * e[0] + e[1] * e[1] + e[2] * e[2], m = in->mass[i];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) I[d][3 * r + c] += Iw[3 * r + c] + m * ((r == c ? e2 : 0.0) - e[r] * e[c]);
  }
  for (int d = 0; d < nd; ++d) {
    out->mass[d] = (float)M[d];
    for (int k = 0; k < 3; ++k) out->com[d][k] = (float)C[d][k];
    out->inertia[d][0] = (float)I[d][0]; out->inertia[d][1] = (float)I[d][4]; out->inertia[d][2] = (float)I[d][8];
    out->inertia[d][3] = (float)I[d][1]; out->inertia[d][4] = (float)I[d][2]; out->inertia[d][5] = (float)I[d][5];
  }
  // tree bookkeeping of the lumped model (depth-first order is kept: a removed body only ever follows its carrier's subtree start)
  for (int d = 0; d < nd; ++d) { out->subtree[d] = 1; out->depth[d] = d == 0 ? 0 : out->depth[out->parent[d]] + 1; }
  for (int d = nd - 1; d > 0; --d) out->subtree[out->parent[d]] += out->subtree[d];
  for (int d = nd; d < PF_MAX_BODIES; ++d) { out->parent[d] = 0; out->subtree[d] = 0; out->depth[d] = 0; out->dof_of_body[d] = -1; out->mass[d] = 0.0f; }
  for (int k = 0; k < in->nd; ++k) out->body_of_dof[k] = dyn[in->body_of_dof[k]];
  // collision spheres move with their body; contact forces stay booked on the original body
  for (int s = 0; s < in->nsph; ++s) {
    const int b = in->sph_body[s];
    double p[3] = {in->sph_pos[s][0], in->sph_pos[s][1], in->sph_pos[s][2]}, q[3];
    matvec(R[b], p, q);
    out->sph_body[s] = dyn[b];
    for (int k = 0; k < 3; ++k) out->sph_pos[s][k] = (float)(q[k] + t[b][k]);
    map->sph_rep[s] = b;
  }
  for (int f = 0; f < in->nfeet; ++f) {
    const int b = in->feet_body[f];
    map->feet_dyn[f] = dyn[b];
    for (int k = 0; k < 3; ++k) map->feet_off[f][k] = (float)t[b][k];
  }
}

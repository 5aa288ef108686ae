"""URDF -> collapsed articulated model (what `gym.load_asset` derives; reference call site
legged_gym/envs/pointfoot/point_foot.py:883-909 with the asset options of pointfoot_rough_config.py:97-117).

* fixed joints are merged into their parent (`collapse_fixed_joints=True`) unless the joint carries
  `dont_collapse="true"`; inertias are combined with the parallel-axis theorem;
* bodies and DOFs are numbered depth-first in URDF child order (Isaac Gym's convention);
* collision primitives become spheres for the sphere-vs-heightfield contact model: sphere -> itself,
  cylinder (-> capsule, `replace_cylinder_with_capsule=True`) -> its two end spheres, box on the root link -> 8 corners
  (r=0), box on a limb -> two spheres on its long axis.

A parsed model can be saved as / loaded from a small JSON table so the package works without the URDF
file (the GPU box has no /root/reference); `resources/robots/PF_P441A/model.json` was produced by
`python -m pointfoot_amd.utils.urdf <urdf> <json>`.
"""
import json
import math
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np


def _floats(s, n=3):
    if s is None:
        return np.zeros(n)
    return np.array([float(v) for v in s.split()], dtype=np.float64)


def rpy_to_mat(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def _origin(elem):
    o = elem.find("origin") if elem is not None else None
    if o is None:
        return np.zeros(3), np.eye(3)
    return _floats(o.get("xyz")), rpy_to_mat(_floats(o.get("rpy")))


class _Link:
    def __init__(self, name):
        self.name = name
        self.mass = 0.0
        self.com = np.zeros(3)
        self.inertia = np.zeros((3, 3))   # about COM, link axes
        self.spheres = []                 # (pos, radius)
        self.children = []                # joints


def _add_inertial(link, mass, com, I):
    """Merge a rigid piece (mass, com, inertia about its com; all in `link` frame) into the link."""
    if mass <= 0.0:
        return
    m0, c0, I0 = link.mass, link.com, link.inertia
    m = m0 + mass
    c = (m0 * c0 + mass * com) / m

    def shift(Ic, mm, d):
        return Ic + mm * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
    link.inertia = shift(I0, m0, c0 - c) + shift(I, mass, com - c)
    link.mass, link.com = m, c


def _collision_spheres(link_elem, is_root=True):
    out = []
    for col in link_elem.findall("collision"):
        p, R = _origin(col)
        g = col.find("geometry")
        if g is None:
            continue
        if g.find("sphere") is not None:
            out.append((p, float(g.find("sphere").get("radius"))))
        elif g.find("cylinder") is not None:
            c = g.find("cylinder")
            half, rad = 0.5 * float(c.get("length")), float(c.get("radius"))
            axis = R @ np.array([0.0, 0.0, 1.0])
            out.append((p + half * axis, rad))
            out.append((p - half * axis, rad))
        elif g.find("box") is not None and not is_root:
            # limb boxes: two spheres on the long axis, radius = half the middle dimension (capsule-like stand-in)
            dims = _floats(g.find("box").get("size"))
            order = np.argsort(dims)
            rad = 0.5 * dims[order[1]]
            axis = np.zeros(3); axis[order[2]] = 1.0
            half = max(0.5 * dims[order[2]] - rad, 0.0)
            out.append((p + R @ (half * axis), rad))
            out.append((p - R @ (half * axis), rad))
        elif g.find("box") is not None:
            sx, sy, sz = _floats(g.find("box").get("size"))
            for dx in (-0.5, 0.5):
                for dy in (-0.5, 0.5):
                    for dz in (-0.5, 0.5):
                        out.append((p + R @ np.array([dx * sx, dy * sy, dz * sz]), 0.0))
    return out


def parse_urdf(path, collapse_fixed_joints=True):
    root = ET.parse(path).getroot()
    links = {}
    for le in root.findall("link"):
        lk = _Link(le.get("name"))
        ine = le.find("inertial")
        if ine is not None:
            p, R = _origin(ine)
            mass = float(ine.find("mass").get("value"))
            it = ine.find("inertia")
            I = np.array([[float(it.get("ixx")), float(it.get("ixy")), float(it.get("ixz"))],
                          [float(it.get("ixy")), float(it.get("iyy")), float(it.get("iyz"))],
                          [float(it.get("ixz")), float(it.get("iyz")), float(it.get("izz"))]])
            _add_inertial(lk, mass, p, R @ I @ R.T)
        links[lk.name] = lk
    child_names = set()
    for je in root.findall("joint"):
        parent = je.find("parent").get("link")
        child = je.find("child").get("link")
        p, R = _origin(je)
        ax = je.find("axis")
        axis = _floats(ax.get("xyz")) if ax is not None else np.array([1.0, 0.0, 0.0])
        lim = je.find("limit")
        j = dict(name=je.get("name"), type=je.get("type"), child=child, pos=p, rot=R, axis=axis,
                 keep=je.get("dont_collapse") == "true",
                 lower=float(lim.get("lower", 0)) if lim is not None else 0.0,
                 upper=float(lim.get("upper", 0)) if lim is not None else 0.0,
                 effort=float(lim.get("effort", 0)) if lim is not None else 0.0,
                 velocity=float(lim.get("velocity", 0)) if lim is not None else 0.0)
        links[parent].children.append(j)
        child_names.add(child)
    base = [n for n in links if n not in child_names]
    assert len(base) == 1, f"expected one root link, got {base}"
    for le in root.findall("link"):
        links[le.get("name")].spheres = _collision_spheres(le, is_root=(le.get("name") == base[0]))

    bodies, dofs = [], []

    def visit(link, parent_idx, jpos, jrot, joint, depth):
        """Emit `link` as a body, folding collapsible fixed children into it first."""
        idx = len(bodies)
        body = dict(name=link.name, parent=parent_idx, depth=depth, joint_pos=jpos, joint_rot=jrot,
                    joint_axis=np.zeros(3), dof=-1, link=link)
        if joint is not None and joint["type"] in ("revolute", "continuous"):
            n = np.linalg.norm(joint["axis"])
            body["joint_axis"] = joint["axis"] / (n if n > 0 else 1.0)
            body["dof"] = len(dofs)
            dofs.append(dict(name=joint["name"], body=idx, lower=joint["lower"], upper=joint["upper"],
                             effort=joint["effort"], velocity=joint["velocity"]))
        bodies.append(body)
        pending = []   # (child link, pos, rot, joint) expressed in this body's frame

        def fold(lk, T_p, T_R):
            for j in lk.children:
                c = links[j["child"]]
                cp, cR = T_p + T_R @ j["pos"], T_R @ j["rot"]
                if j["type"] == "fixed" and collapse_fixed_joints and not j["keep"]:
                    _add_inertial(link, c.mass, cp + cR @ c.com, cR @ c.inertia @ cR.T)
                    link.spheres += [(cp + cR @ sp, rad) for sp, rad in c.spheres]
                    fold(c, cp, cR)
                else:
                    pending.append((c, cp, cR, j))
        fold(link, np.zeros(3), np.eye(3))
        for c, cp, cR, j in pending:
            visit(c, idx, cp, cR, j, depth + 1)
        body["subtree"] = len(bodies) - idx

    visit(links[base[0]], -1, np.zeros(3), np.eye(3), None, 0)
    return dict(
        body_names=[b["name"] for b in bodies], dof_names=[d["name"] for d in dofs],
        parent=[b["parent"] for b in bodies], subtree=[b["subtree"] for b in bodies],
        depth=[b["depth"] for b in bodies], dof_of_body=[b["dof"] for b in bodies],
        body_of_dof=[d["body"] for d in dofs],
        joint_pos=[b["joint_pos"].tolist() for b in bodies],
        joint_rot=[b["joint_rot"].reshape(-1).tolist() for b in bodies],
        joint_axis=[b["joint_axis"].tolist() for b in bodies],
        mass=[b["link"].mass for b in bodies], com=[b["link"].com.tolist() for b in bodies],
        inertia=[[b["link"].inertia[0, 0], b["link"].inertia[1, 1], b["link"].inertia[2, 2],
                  b["link"].inertia[0, 1], b["link"].inertia[0, 2], b["link"].inertia[1, 2]] for b in bodies],
        q_lo=[d["lower"] for d in dofs], q_hi=[d["upper"] for d in dofs],
        vel_limit=[d["velocity"] for d in dofs], effort_limit=[d["effort"] for d in dofs],
        spheres=[[i, list(map(float, sp)), float(rad)] for i, b in enumerate(bodies) for sp, rad in b["link"].spheres],
    )


def load_model(asset_file):
    """Model table for `cfg.asset.file`: parse the URDF when present, else the packaged model.json."""
    if asset_file and os.path.exists(asset_file):
        return parse_urdf(asset_file)
    stem = os.path.splitext(os.path.basename(asset_file))[0]
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    js = os.path.join(here, "resources", "robots", stem, "model.json")
    if not os.path.exists(js):
        raise ValueError(f"robot asset not found: {asset_file} (and no packaged table {js})")
    with open(js) as f:
        return json.load(f)


if __name__ == "__main__":
    m = parse_urdf(sys.argv[1])
    os.makedirs(os.path.dirname(os.path.abspath(sys.argv[2])), exist_ok=True)
    with open(sys.argv[2], "w") as f:
        json.dump(m, f, indent=1)
    print(m["body_names"], m["dof_names"], "mass", sum(m["mass"]), "spheres", len(m["spheres"]))

"""Shared helpers for the parity tests: build identical inputs for the oracle and the GPU path."""
import glob
import os

import numpy as np

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
SER = sorted(glob.glob(os.path.join(HERE, "golden", "serine_peptidases", "*.pdb")))
Q4CHA = os.path.join(HERE, "golden", "query", "4CHA.pdb")
Q1G2F = os.path.join(HERE, "golden", "query", "1G2F.pdb")


def oracle_structs_to_packed(structs):
    """oracle CompactStructures -> folddisco_amd.PackedStructures (+ resname_std flags)."""
    from folddisco_amd import PackedStructures
    items, std = [], []
    for s in structs:
        a = s.arrays()
        items.append(dict(n_xyz=a["n_xyz"], ca_xyz=a["ca_xyz"], cb_xyz=a["cb_xyz"], aa=a["aa"], cb_ok=a["cb_ok"]))
        names = [bytes(r).decode("latin1") for r in a["resname"]]
        std.append(np.array([1 if (aa < 20 and nm == oracle.lib().fdo_map_u8_to_aa(int(aa)).decode()) else 0
                             for aa, nm in zip(a["aa"], names)], dtype=np.uint8))
    return PackedStructures.concat(items), (np.concatenate(std) if std else np.zeros(0, np.uint8))


def packed_to_oracle_structs(ps):
    off = ps.res_off.astype(np.int64)
    out = []
    for s in range(ps.n_struct):
        a, b = off[s], off[s + 1]
        out.append(oracle.structure_from_packed(ps.n_xyz[a:b], ps.ca_xyz[a:b], ps.cb_xyz[a:b], ps.aa[a:b],
                                                cb_ok=None if ps.cb_valid is None else ps.cb_valid[a:b]))
    return out


def synthetic_packed(n_struct, seed, lengths=None):
    from folddisco_amd import synth
    return synth.to_packed(synth.generate(n_struct, seed=seed, lengths=lengths))


def write_sparse_pdb(path, n_res, seed=0):
    """A PDB file of n_res complete residues + one trailing atom of a further residue (the reference's raw residue count — what its skip test reads,
    controller/mod.rs:313 — is therefore n_res + 1, the compact structure holds n_res) (N, CA, C, CB) laid out as PAIRS of residues 6 A apart on a 30 A grid: every residue has exactly
    one partner inside the 20 A cutoff (residue k pairs with k ^ 1), so a structure of 65,535 residues emits n_res ordered pairs.
    Residue numbers wrap at 9999 (the reference splits residues on a CHANGE of the serial, structure/core.rs:104-200)."""
    rng = np.random.default_rng(seed)
    names = ["ALA", "SER", "HIS", "ASP", "LEU", "LYS", "GLU", "VAL"]
    side = int(np.ceil((n_res / 2.0) ** (1.0 / 3.0))) + 1
    lines, atom = [], 1
    for k in range(n_res):
        cell, half = divmod(k, 2)
        cx, r = divmod(cell, side * side)
        cy, cz = divmod(r, side)
        base = np.array([cx * 30.0, cy * 30.0, cz * 30.0 + half * 6.0]) + rng.uniform(-0.4, 0.4, 3)
        rn = names[int(rng.integers(len(names)))]
        for an, d in ((" N  ", (-1.2, 0.6, 0.1)), (" CA ", (0.0, 0.0, 0.0)), (" C  ", (1.3, 0.7, -0.2)), (" CB ", (0.1, -1.1, 1.0 + 0.2 * half))):
            x, y, z = base + np.array(d)
            lines.append("ATOM  %5d %s %s A%4d    %8.3f%8.3f%8.3f  1.00%6.2f           %s" % (atom % 100000, an, rn, k % 9999 + 1, x, y, z, 50.0 + (k % 40), an.strip()[0]))
            atom += 1
    lines.append("ATOM  %5d  N   GLY A%4d    %8.3f%8.3f%8.3f  1.00 10.00           N" % (atom % 100000, n_res % 9999 + 1, -50.0, -50.0, -50.0))   # the flush needs a next atom
    lines.append("END")
    with open(path, "w") as fh:
        fh.write("\n".join(lines) + "\n")

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

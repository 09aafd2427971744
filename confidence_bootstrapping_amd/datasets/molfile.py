"""Ligand files without rdkit: SDF (V2000) / MOL2 readers, hydrogen removal and the perception the 16 ligand atom features need
(ligand side).

Reference: datasets/process_mols.py:923-977 (`read_molecule`, `read_sdf_or_mol2`), :141-175 (`lig_atom_featurizer`), :567-589
(`get_lig_graph`).  The reference hands these jobs to rdkit, an un-vendored dependency that is absent from this image; what rdkit does
inside is restated here from its published algorithms.  What that means column by column is the table in `LIG_FEATURE_SOURCES` below:
the graph itself (atoms, their order, the bond list and its order, coordinates, formal charges of an SDF) is read from the file and is
exact; ring membership comes from a minimum cycle basis (= rdkit's SSSR except for its symmetrisation of highly symmetric cages);
hydrogen counts, degree and implicit valence follow from the explicit hydrogens of the file plus the default-valence rules;
aromaticity, hybridisation and chirality are PERCEIVED by rdkit and restated here (Aromaticity.cpp's electron-donor model on single
rings and fused pairs / triples, ConjugHybrid.cpp's orbital count and conjugation rule, the sign of the neighbour volume plus a
symmetry-class test) -- parity unpinned at that boundary: nothing in this image can run rdkit to check them.

`Mol` offers the slice of the rdkit Mol API the callers touch (GetAtoms / GetBonds / GetConformer().GetPositions() / GetRingInfo()),
so `molecules_utils.get_symmetry_rmsd(mol, ...)` and `evaluation` take it as they take an rdkit molecule.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

SYMBOLS = ("H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr Rb Sr Y Zr Nb Mo Tc Ru Rh "
           "Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu Hf Ta W Re Os Ir Pt Au Hg Tl Pb Bi Po At Rn "
           "Fr Ra Ac Th Pa U Np Pu Am Cm Bk Cf Es Fm Md No Lr Rf Db Sg Bh Hs Mt Ds Rg Cn Nh Fl Mc Lv Ts Og").split()
_Z_OF = {s.upper(): i + 1 for i, s in enumerate(SYMBOLS)}
_Z_OF.update({"D": 1, "T": 1})
# bond types: the molfile codes; 4 = aromatic.  Index into the reference's `bonds` dictionary (process_mols.py:93)
BOND_TYPE_NAMES = {0: "UNSPECIFIED", 1: "SINGLE", 2: "DOUBLE", 3: "TRIPLE", 4: "AROMATIC"}
BOND_FEATURE_INDEX = {"SINGLE": 0, "DOUBLE": 1, "TRIPLE": 2, "AROMATIC": 3}
_ORDER = {0: 1.0, 1: 1.0, 2: 2.0, 3: 3.0, 4: 1.5}
# default valences (rdkit's periodic table; -1 = no implicit hydrogens) and outer-shell electrons of the elements ligands are made of
_VALENCES = {1: (1,), 5: (3,), 6: (4,), 7: (3,), 8: (2,), 9: (1,), 14: (4,), 15: (3, 5, 7), 16: (2, 4, 6), 17: (1,), 33: (3, 5, 7),
             34: (2, 4, 6), 35: (1,), 52: (2, 4, 6), 53: (1, 3, 5)}
_OUTER = {1: 1, 5: 3, 6: 4, 7: 5, 8: 6, 9: 7, 14: 4, 15: 5, 16: 6, 17: 7, 33: 5, 34: 6, 35: 7, 52: 6, 53: 7}
_ELECTRONEG_ORDER = None

LIG_FEATURE_SOURCES = (
    # column, reference call (process_mols.py:152-167), how it is obtained here, status
    (0, "GetAtomicNum", "element column of the file", "exact"),
    (1, "GetChiralTag", "sign of the neighbour volume in the file's bond order, kept only on atoms whose four substituents fall in "
        "different graph-symmetry classes; parity flip when a hydrogen in front of other neighbours is removed", "restated (rdkit "
        "assignChiralTypesFrom3D + assignStereochemistry); unpinned"),
    (2, "GetTotalDegree", "heavy neighbours + hydrogens (explicit in the file, else from the valence rules)", "exact when the file "
        "carries its hydrogens"),
    (3, "GetFormalCharge", "SDF: charge column / M  CHG lines; MOL2: 0 (rdkit derives charges of a few SYBYL types)", "exact for SDF"),
    (4, "GetImplicitValence", "hydrogens that are not atoms of the graph: removed explicit ones + valence-rule ones", "exact when the "
        "file carries its hydrogens"),
    (5, "GetTotalNumHs", "as column 4 (explicit hydrogen ATOMS that stay in the graph are not counted, as in rdkit)", "as column 4"),
    (6, "GetNumRadicalElectrons", "0", "not derived (rdkit: valence deficit of atoms flagged noImplicit)"),
    (7, "GetHybridization", "orbital count degree + lone pairs, SP3 -> SP2 for conjugated atoms of degree <= 3", "restated "
        "(ConjugHybrid.cpp); unpinned"),
    (8, "GetIsAromatic", "4n+2 test over the electron-donor types of single rings and fused ring pairs / triples; MOL2 `ar` bonds as "
        "given", "restated (Aromaticity.cpp, default model); unpinned"),
    (9, "NumAtomRings", "rings of a minimum cycle basis through the atom", "exact up to rdkit's SSSR symmetrisation"),
    (10, "IsAtomInRingOfSize 3..8 (columns 10-15)", "the same ring set", "as column 9"),
)


class Atom:
    __slots__ = ("idx", "z", "symbol", "charge", "aromatic", "num_hs", "hybridization", "chiral_tag", "radicals", "_mol")

    def __init__(self, idx, z, symbol, charge=0):
        self.idx, self.z, self.symbol, self.charge = idx, z, symbol, charge
        self.aromatic, self.num_hs, self.hybridization, self.chiral_tag, self.radicals = False, 0, "UNSPECIFIED", "CHI_UNSPECIFIED", 0
        self._mol = None

    def GetIdx(self): return self.idx
    def GetAtomicNum(self): return self.z
    def GetSymbol(self): return self.symbol
    def GetFormalCharge(self): return self.charge
    def GetIsAromatic(self): return self.aromatic
    def GetTotalNumHs(self): return self.num_hs
    def GetImplicitValence(self): return self.num_hs
    def GetNumRadicalElectrons(self): return self.radicals
    def GetHybridization(self): return self.hybridization
    def GetChiralTag(self): return self.chiral_tag
    def GetDegree(self): return len(self._mol.neighbors(self.idx))
    def GetTotalDegree(self): return len(self._mol.neighbors(self.idx)) + self.num_hs


class Bond:
    __slots__ = ("a", "b", "type")

    def __init__(self, a, b, t):
        self.a, self.b, self.type = a, b, t

    def GetBeginAtomIdx(self): return self.a
    def GetEndAtomIdx(self): return self.b
    def GetBondType(self): return BOND_TYPE_NAMES[self.type]


class _Conformer:
    def __init__(self, pos): self._pos = pos
    def GetPositions(self): return self._pos.copy()


class RingInfo:
    def __init__(self, rings: List[Tuple[int, ...]], n):
        self.rings = rings
        self._of = [[] for _ in range(n)]
        for r in rings:
            for a in r:
                self._of[a].append(len(r))

    def NumAtomRings(self, idx): return len(self._of[idx])
    def IsAtomInRingOfSize(self, idx, size): return size in self._of[idx]
    def AtomRings(self): return tuple(self.rings)
    def NumRings(self): return len(self.rings)


class Mol:
    """Atoms in file order, bonds in file order, one conformer."""

    def __init__(self, atoms: List[Atom], bonds: List[Bond], pos: np.ndarray, name=""):
        self.atoms, self.bonds, self.pos, self.name = atoms, bonds, np.asarray(pos, dtype=np.float64).reshape(-1, 3), name
        for a in atoms:
            a._mol = self
        self._nbr = None
        self._rings = None
        self.perceived = False

    # ---- rdkit-shaped accessors
    def GetAtoms(self): return self.atoms
    def GetBonds(self): return self.bonds
    def GetNumAtoms(self): return len(self.atoms)
    def GetNumHeavyAtoms(self): return sum(1 for a in self.atoms if a.z > 1)
    def GetNumConformers(self): return 1 if len(self.pos) == len(self.atoms) and len(self.atoms) else 0
    def GetConformer(self, i=0): return _Conformer(self.pos)
    def GetAtomWithIdx(self, i): return self.atoms[i]

    def GetRingInfo(self) -> RingInfo:
        if self._rings is None:
            self._rings = RingInfo(_minimum_cycle_basis(len(self.atoms), [(b.a, b.b) for b in self.bonds]), len(self.atoms))
        return self._rings

    # ---- graph helpers
    def neighbors(self, i):
        if self._nbr is None:
            self._nbr = [[] for _ in self.atoms]
            for k, b in enumerate(self.bonds):
                self._nbr[b.a].append((b.b, k))
                self._nbr[b.b].append((b.a, k))
        return self._nbr[i]

    @property
    def atomicnums(self): return np.asarray([a.z for a in self.atoms])

    @property
    def adjacency_matrix(self):
        am = np.zeros((len(self.atoms), len(self.atoms)), dtype=int)
        for b in self.bonds:
            am[b.a, b.b] = am[b.b, b.a] = 1
        return am


# ---- readers --------------------------------------------------------------------------------------------------------------------------
_SDF_CHARGE = {0: 0, 1: 3, 2: 2, 3: 1, 4: 0, 5: -1, 6: -2, 7: -3}


def _element(sym: str) -> Tuple[int, str]:
    s = sym.strip()
    z = _Z_OF.get(s.upper(), 0)
    return z, (SYMBOLS[z - 1] if z else s)


def parse_mol_block(text: str) -> Mol:
    """One V2000 connection table (the first record of an SDF file)."""
    lines = text.splitlines()
    if len(lines) < 4:
        raise ValueError("not a mol block")
    counts = lines[3]
    if "V3000" in counts:
        raise ValueError("V3000 connection tables are not supported")
    na, nb = int(counts[0:3]), int(counts[3:6])
    atoms, pos = [], []
    for i, l in enumerate(lines[4:4 + na]):
        pos.append((float(l[0:10]), float(l[10:20]), float(l[20:30])))
        z, sym = _element(l[31:34])
        code = int(l[36:39]) if len(l) >= 39 and l[36:39].strip() else 0
        atoms.append(Atom(i, z, sym, _SDF_CHARGE.get(code, 0)))
        if code == 4:
            atoms[-1].radicals = 1
    bonds = []
    for l in lines[4 + na:4 + na + nb]:
        a, b, t = int(l[0:3]) - 1, int(l[3:6]) - 1, int(l[6:9])
        bonds.append(Bond(a, b, t if t in (1, 2, 3, 4) else 0))
    first_chg = True
    for l in lines[4 + na + nb:]:
        if l.startswith("M  END") or l.startswith("$$$$"):
            break
        if l.startswith("M  CHG") or l.startswith("M  RAD"):
            if l.startswith("M  CHG") and first_chg:       # the property block replaces the atom block's charges altogether
                first_chg = False
                for a in atoms:
                    a.charge = 0
            n = int(l[6:9])
            for k in range(n):
                idx, val = int(l[9 + 8 * k:13 + 8 * k]) - 1, int(l[13 + 8 * k:17 + 8 * k])
                if l.startswith("M  CHG"):
                    atoms[idx].charge = val
                else:
                    atoms[idx].radicals = {0: 0, 1: 2, 2: 1, 3: 2}.get(val, 0)
    return Mol(atoms, bonds, np.asarray(pos).reshape(-1, 3), name=lines[0].strip())


def read_sdf(path) -> Mol:
    with open(path) as f:
        text = f.read()
    return parse_mol_block(text.split("$$$$")[0])


def read_mol2(path) -> Mol:
    """TRIPOS MOL2: the first molecule's ATOM and BOND records.  Element = the SYBYL type in front of the dot; bond types 1 / 2 / 3 /
    ar (aromatic) / am (amide: single); `du`, `un` -> unspecified, `nc` records are skipped."""
    section, atoms, pos, bonds, ids, name, seen_mol, sybyl = None, [], [], [], {}, "", 0, []
    with open(path) as f:
        for raw in f:
            line = raw.strip()
            if not line or line.startswith("#"):
                continue
            if line.startswith("@<TRIPOS>"):
                section = line[9:].upper()
                if section == "MOLECULE":
                    seen_mol += 1
                    if seen_mol > 1:
                        break
                    name = None
                continue
            if section == "MOLECULE" and name is None:
                name = line
            elif section == "ATOM":
                t = line.split()
                z, sym = _element(t[5].split(".")[0])
                if z == 0:                                   # e.g. a type column that is not SYBYL: fall back on the atom name's letters
                    z, sym = _element("".join(c for c in t[1] if c.isalpha())[:2])
                    if z == 0:
                        z, sym = _element(t[1][:1])
                ids[int(t[0])] = len(atoms)
                sybyl.append(t[5])
                atoms.append(Atom(len(atoms), z, sym, 0))
                pos.append((float(t[2]), float(t[3]), float(t[4])))
            elif section == "BOND":
                t = line.split()
                kind = t[3].lower()
                if kind == "nc":
                    continue
                code = {"1": 1, "2": 2, "3": 3, "ar": 4, "am": 1}.get(kind, 0)
                bonds.append(Bond(ids[int(t[1])], ids[int(t[2])], code))
    mol = Mol(atoms, bonds, np.asarray(pos).reshape(-1, 3), name=name or "")
    # SYBYL's delocalised acid groups (O.co2 with `ar` / mixed bonds on a carboxylate, phosphate, sulfonate centre): one double bond,
    # the other oxygens single and -1 unless they carry a hydrogen -- rdkit's Mol2 clean-up of the same groups
    for c in range(len(atoms)):
        oxy = [(j, k) for j, k in mol.neighbors(c) if sybyl[j].lower() == "o.co2" and sum(1 for q, _ in mol.neighbors(j) if atoms[q].z > 1) == 1]
        if len(oxy) < 2:
            continue
        for n_, (j, k) in enumerate(oxy):
            bonds[k].type = 2 if n_ == 0 else 1
            if n_ > 0 and not any(atoms[q].z == 1 for q, _ in mol.neighbors(j)):
                atoms[j].charge = -1
    return mol


# ---- graph algorithms -----------------------------------------------------------------------------------------------------------------
def _minimum_cycle_basis(n: int, edges: Sequence[Tuple[int, int]]) -> List[Tuple[int, ...]]:
    """Rings of a minimum cycle basis as ordered atom tuples (networkx's de Pina implementation, each cycle re-ordered along its
    bonds; components and trees contribute nothing)."""
    import networkx as nx
    g = nx.Graph()
    g.add_nodes_from(range(n))
    g.add_edges_from((a, b) for a, b in edges if a != b)
    # only the 2-core can carry cycles: peel the trees off first (cheap, and keeps the basis search small)
    core = nx.k_core(g, 2) if g.number_of_edges() else g
    rings = []
    for cyc in nx.minimum_cycle_basis(core):
        sub = core.subgraph(cyc)
        start = min(cyc)
        order, prev, cur = [start], None, start
        while True:
            nxt = sorted(x for x in sub[cur] if x != prev and (x not in order or (x == start and len(order) > 2)))
            nxt = [x for x in nxt if x != start] or [x for x in nxt if x == start]
            if not nxt or nxt[0] == start:
                break
            prev, cur = cur, nxt[0]
            order.append(cur)
        rings.append(tuple(order))
    rings.sort(key=lambda r: (len(r), sorted(r)))
    return rings


def _symmetry_classes(mol: Mol) -> np.ndarray:
    """Graph-symmetry classes by iterated refinement of (element, charge, degree, hydrogens, aromaticity) over neighbour multisets
    with bond types -- what decides whether two substituents of a potential stereocentre are different."""
    inv = [hash((a.z, a.charge, len(mol.neighbors(a.idx)), a.num_hs, a.aromatic)) for a in mol.atoms]
    n_cls = len(set(inv))
    for _ in range(len(mol.atoms)):
        new = [hash((inv[i], tuple(sorted((inv[j], mol.bonds[k].type) for j, k in mol.neighbors(i))))) for i in range(len(inv))]
        k = len(set(new))
        inv = new
        if k == n_cls:
            break
        n_cls = k
    _, cls = np.unique(np.asarray(inv, dtype=object).astype(str), return_inverse=True)
    return cls


# ---- perception (restated from rdkit; see the module docstring) -----------------------------------------------------------------------
def _default_valences(z, charge):
    """rdkit's charge handling: a charged main-group atom takes the valence list of its isoelectronic neighbour."""
    if z in (5, 6, 7, 8, 14, 15, 16, 33, 34) and charge:
        return _VALENCES.get(z - charge, ())
    return _VALENCES.get(z, ())


def _explicit_valence(mol: Mol, i) -> float:
    return sum(_ORDER[mol.bonds[k].type] for _, k in mol.neighbors(i))


def _rule_hydrogens(mol: Mol, i) -> int:
    """implicit hydrogens by the default-valence rule (the smallest allowed valence that covers the explicit one)"""
    a = mol.atoms[i]
    if a.z <= 1 or any(mol.bonds[k].type == 4 for _, k in mol.neighbors(i)):
        return 0
    ev = int(round(_explicit_valence(mol, i)))
    for v in _default_valences(a.z, a.charge):
        if v >= ev:
            return v - ev
    return 0


def _count_atom_elec(mol: Mol, i) -> int:
    """MolOps::countAtomElec: electrons an atom can give to a pi system (-1: more than three-coordinate)."""
    a = mol.atoms[i]
    vals = _VALENCES.get(a.z, ())
    dv = vals[0] if vals else -1
    if dv <= 1:
        return 0
    degree = len(mol.neighbors(i)) + a.num_hs
    if degree > 3:
        return -1
    nlp = max(_OUTER.get(a.z, 0) - dv - a.charge, 0)
    res = (dv - degree) + nlp - a.radicals
    if res > 1:
        unsat = int(round(_explicit_valence(mol, i))) - len(mol.neighbors(i))
        if unsat > 1:
            res = 1
    return res


_EN = {1: 2.20, 5: 2.04, 6: 2.55, 7: 3.04, 8: 3.44, 9: 3.98, 14: 1.90, 15: 2.19, 16: 2.58, 17: 3.16, 33: 2.18, 34: 2.55, 35: 2.96, 53: 2.66}


def _more_electronegative(za, zb):
    """rdkit compares outer-shell electron counts, then (same column) the lighter element wins"""
    oa, ob = _OUTER.get(za, 0), _OUTER.get(zb, 0)
    return oa > ob or (oa == ob and za < zb)


def perceive(mol: Mol) -> Mol:
    """Fills num_hs / aromatic / hybridization / chiral_tag of every atom and turns the bonds of perceived aromatic rings into
    AROMATIC ones -- what `Chem.SanitizeMol` leaves on an rdkit molecule as far as the 16 features and the bond one-hot see it.
    Call it on the molecule as read (explicit hydrogens still atoms); `remove_hs` keeps the results consistent."""
    n = len(mol.atoms)
    for i, a in enumerate(mol.atoms):
        a.num_hs = _rule_hydrogens(mol, i)
    rings = mol.GetRingInfo().rings
    in_ring_bond = set()
    bond_of = {}
    for k, b in enumerate(mol.bonds):
        bond_of[(b.a, b.b)] = bond_of[(b.b, b.a)] = k
    ring_bonds = []
    for r in rings:
        rb = [bond_of[(r[q], r[(q + 1) % len(r)])] for q in range(len(r))]
        ring_bonds.append(rb)
        in_ring_bond.update(rb)
    ring_atom = set(a for r in rings for a in r)

    def multiple(k):
        return mol.bonds[k].type in (2, 3, 4)

    # ---- electron-donor types (Aromaticity.cpp::getAtomDonorTypeArom, exocyclic bonds steal electrons)
    VAC, ONE, TWO, NO = "vacant", "one", "two", "no"
    donor, cand = {}, {}
    for i in ring_atom:
        a = mol.atoms[i]
        nelec = _count_atom_elec(mol, i)
        exo = [j for j, k in mol.neighbors(i) if multiple(k) and k not in in_ring_bond]
        cyc = any(multiple(k) and k in in_ring_bond for _, k in mol.neighbors(i))
        anym = any(multiple(k) for _, k in mol.neighbors(i))
        if nelec < 0:
            d = NO
        elif nelec == 0:
            d = VAC if exo else (ONE if cyc else NO)
        elif nelec == 1:
            if exo:
                d = VAC if _more_electronegative(mol.atoms[exo[0]].z, a.z) else ONE
            else:
                d = ONE if anym else (VAC if a.charge == 1 else NO)
        else:
            if exo and _more_electronegative(mol.atoms[exo[0]].z, a.z):
                nelec -= 1
            d = ONE if nelec % 2 == 1 else TWO
        donor[i] = d
        # isAtomCandForArom: element set, default valence not exceeded, no radicals on hetero / charged atoms, at most one multiple bond
        ok = d != NO and a.z in (5, 6, 7, 8, 14, 15, 16, 33, 34, 52)
        if ok:
            vals = _default_valences(a.z, a.charge)
            tv = int(round(_explicit_valence(mol, i))) + a.num_hs
            if vals and tv > vals[0] and not any(mol.bonds[k].type == 4 for _, k in mol.neighbors(i)):
                ok = False
            if a.radicals and (a.z != 6 or a.charge):
                ok = False
            if sum(1 for _, k in mol.neighbors(i) if mol.bonds[k].type in (2, 3)) > 1:
                ok = False
        cand[i] = ok

    def huckel(atoms):
        lo = hi = 0
        for i in atoms:
            d = donor[i]
            if d == ONE:
                lo, hi = lo + 1, hi + 1
            elif d == TWO:
                lo, hi = lo + 2, hi + 2
        if hi >= 6:
            return any((e - 2) % 4 == 0 for e in range(lo, hi + 1))
        return hi == 2

    arom_bonds, arom_atoms = set(), set()
    given = [k for k, b in enumerate(mol.bonds) if b.type == 4]            # MOL2 `ar`: taken as given
    for k in given:
        arom_bonds.add(k)
        arom_atoms.update((mol.bonds[k].a, mol.bonds[k].b))
    ok_ring = [all(cand.get(a, False) for a in r) for r in rings]
    done = [False] * len(rings)
    for q, r in enumerate(rings):
        if ok_ring[q] and len(r) <= 24 and huckel(r):
            done[q] = True
            arom_atoms.update(r)
            arom_bonds.update(ring_bonds[q])
    # fused systems: pairs and triples of candidate rings that share bonds; the 4n+2 test runs over the outer envelope
    if not all(done[q] or not ok_ring[q] for q in range(len(rings))):
        share = {q: [p for p in range(len(rings)) if p != q and ok_ring[p] and set(ring_bonds[p]) & set(ring_bonds[q])]
                 for q in range(len(rings)) if ok_ring[q]}
        combos = set()
        for q in share:
            for p in share[q]:
                combos.add(tuple(sorted((q, p))))
                for s in share[p]:
                    if s != q:
                        combos.add(tuple(sorted((q, p, s))))
        for combo in sorted(combos, key=lambda c: (len(c), c)):
            if all(done[q] for q in combo) or len(set(combo)) != len(combo):
                continue
            count = {}
            for q in combo:
                for k in ring_bonds[q]:
                    count[k] = count.get(k, 0) + 1
            env = [k for k, c in count.items() if c == 1]
            env_atoms = set()
            for k in env:
                env_atoms.update((mol.bonds[k].a, mol.bonds[k].b))
            if len(env_atoms) <= 24 and huckel(env_atoms):
                for q in combo:
                    done[q] = True
                arom_atoms.update(env_atoms)
                arom_bonds.update(env)
    for i in range(n):
        mol.atoms[i].aromatic = i in arom_atoms
    # ---- conjugation (ConjugHybrid.cpp::markConjAtomBonds), on the kekule form as read
    conj = set()
    for i, a in enumerate(mol.atoms):
        sbo = len(mol.neighbors(i)) + a.num_hs
        if sbo < 2 or sbo > 3:
            continue
        for j1, k1 in mol.neighbors(i):
            if not multiple(k1):
                continue
            for j2, k2 in mol.neighbors(i):
                if k2 == k1:
                    continue
                a2 = mol.atoms[j2]
                if len(mol.neighbors(j2)) + a2.num_hs > 3:
                    continue
                if _count_atom_elec(mol, j2) > 0:
                    conj.update((k1, k2))
    conj |= arom_bonds
    # ---- hybridisation (ConjugHybrid.cpp::setHybridization)
    for i, a in enumerate(mol.atoms):
        deg = len(mol.neighbors(i)) + a.num_hs
        if a.z <= 1:
            norbs = deg
        else:
            nouter = _OUTER.get(a.z)
            if nouter is None:
                a.hybridization = "UNSPECIFIED" if a.z == 0 else ("SP3D2" if deg > 5 else "SP3D" if deg == 5 else "SP3")
                continue
            tv = int(round(_explicit_valence(mol, i))) + a.num_hs
            free = nouter - (tv + a.charge)
            if tv + nouter - a.charge < 8:
                norbs = deg + (free - a.radicals) // 2 + a.radicals
            else:
                norbs = deg + free // 2
        if norbs <= 1:
            a.hybridization = "S"
        elif norbs == 2:
            a.hybridization = "SP"
        elif norbs == 3:
            a.hybridization = "SP2"
        elif norbs == 4:
            has_conj = any(k in conj for _, k in mol.neighbors(i))
            a.hybridization = "SP3" if deg > 3 or not has_conj else "SP2"
        elif norbs == 5:
            a.hybridization = "SP3D"
        else:
            a.hybridization = "SP3D2"
    # ---- chirality from the 3D coordinates (assignChiralTypesFrom3D) on real stereocentres only
    if mol.GetNumConformers():
        cls = _symmetry_classes(mol)
        for i, a in enumerate(mol.atoms):
            a.chiral_tag = "CHI_UNSPECIFIED"
            nb = [j for j, _ in mol.neighbors(i)]
            if a.z <= 1 or len(nb) + a.num_hs != 4 or len(nb) < 3 or a.num_hs > 1 or a.hybridization != "SP3":
                continue
            classes = [cls[j] for j in nb]
            if len(set(classes)) != len(classes):
                continue
            v = [mol.pos[j] - mol.pos[i] for j in nb[:3]]
            vol = float(np.dot(v[0], np.cross(v[1], v[2])))
            if vol < -0.1:
                a.chiral_tag = "CHI_TETRAHEDRAL_CW"
            elif vol > 0.1:
                a.chiral_tag = "CHI_TETRAHEDRAL_CCW"
    for k in arom_bonds:
        mol.bonds[k].type = 4
    mol.perceived = True
    return mol


def remove_hs(mol: Mol) -> Mol:
    """`Chem.RemoveHs` with its default parameters: a hydrogen goes if it is a plain (charge 0) hydrogen with exactly one neighbour and
    that neighbour is a heavy atom; what stays keeps its relative order (atoms and bonds).  The heavy atom counts the removed hydrogen
    as an implicit one, and a chiral tag flips when the hydrogen's bond was an odd number of places in front of the last one."""
    drop = []
    for i, a in enumerate(mol.atoms):
        nb = mol.neighbors(i)
        if a.z == 1 and a.charge == 0 and len(nb) == 1 and mol.atoms[nb[0][0]].z > 1:
            drop.append(i)
    drop_set = set(drop)
    for i in drop:
        j, k = mol.neighbors(i)[0]
        heavy = mol.atoms[j]
        heavy.num_hs += 1
        if heavy.chiral_tag in ("CHI_TETRAHEDRAL_CW", "CHI_TETRAHEDRAL_CCW"):
            order = [kk for _, kk in mol.neighbors(j) if mol.bonds[kk].a not in drop_set - {i} and mol.bonds[kk].b not in drop_set - {i}]
            if k in order and (len(order) - 1 - order.index(k)) % 2 == 1:
                heavy.chiral_tag = "CHI_TETRAHEDRAL_CCW" if heavy.chiral_tag == "CHI_TETRAHEDRAL_CW" else "CHI_TETRAHEDRAL_CW"
    keep = [i for i in range(len(mol.atoms)) if i not in drop_set]
    remap = {old: new for new, old in enumerate(keep)}
    atoms = []
    for old in keep:
        a = mol.atoms[old]
        a.idx = remap[old]
        atoms.append(a)
    bonds = [Bond(remap[b.a], remap[b.b], b.type) for b in mol.bonds if b.a in remap and b.b in remap]
    out = Mol(atoms, bonds, mol.pos[keep] if len(mol.pos) == len(mol.atoms) else mol.pos, name=mol.name)
    out.perceived = mol.perceived
    return out

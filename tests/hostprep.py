"""Test-side host logic: BED text -> (tid,start,end) regions for the C ABI. (The shipped host layer is C++: ngs-bits_amd/host.)"""
import oracle_lib as O


def chr_norm(name):
    t = name.strip().upper()
    if t.startswith("CHR"):
        t = t[3:]
    if t == "M":
        t = "MT"
    return t


def chr_num(name, _other={}):
    """Chromosome.cpp:133-190 numbering (others get 1004+ in first-seen order)."""
    t = chr_norm(name)
    if t == "":
        return 0
    if t == "X":
        return 1001
    if t == "Y":
        return 1002
    if t == "MT":
        return 1003
    if t.isdigit() and not t.startswith("0") and 0 < int(t) <= 1000:
        return int(t)
    if t not in _other:
        _other[t] = 1004 + len(_other)
    return _other[t]


def tid_map(refs):
    m = {}
    for i, (name, _) in enumerate(refs):
        m.setdefault(chr_num(name), i)
    return m


def nonspecial(refs):
    import numpy as np
    return np.array([1 if 0 < chr_num(n) < 1004 else 0 for n, _ in refs], dtype=np.uint8)


def bed_regions(bed_path, refs, merge_mode):
    """merge_mode: 0 none, 1 merge(), 2 merge(true,true), 3 sort+merge, 4 merge+chunk(100). Returns [(tid,start,end)], annotations."""
    text = O.bed_roundtrip(bed_path, merge_mode)
    tm = tid_map(refs)
    regs, annos = [], []
    for ln in text.splitlines():
        f = ln.split("\t")
        regs.append((tm.get(chr_num(f[0]), -1), int(f[1]) + 1, int(f[2])))
        annos.append(f[3:])
    return regs, annos


def xy_tids(refs):
    tm = tid_map(refs)
    return tm.get(1001, -1), tm.get(1002, -1)

"""Canonical form of an output directory written in the reference's `-t N` (N > 1) format -- see tests/golden/make_t2_golden.py."""
import os

ARITY = ["bi", "tri", "tetra", "penta"]


def canonical(outdir: str, prefix: str) -> dict:
    def read(name):
        with open(os.path.join(outdir, "%s_%s.txt" % (prefix, name))) as f:
            return f.read().splitlines()

    # the threaded reference also NUMBERS the unitigs in an order that depends on timing (Bifrost's read with several threads):
    # every unitig id is replaced by a name derived from its sequence, via the run's own Unitig_Id.txt
    import hashlib
    name = {}
    for r in read("Unitig_Id"):
        uid, seq = r.split("\t")
        name[uid] = hashlib.sha1(seq.encode()).hexdigest()[:12]
    out = {"Unitig_Id": "\n".join(sorted(name.values())) + "\n"}
    sb = read("super_bubble")
    assert sb[0] == "BubbleId\tEntrance\tStrand\tExit\tisSimple\tisComplex"
    ids = sorted(int(r.split("\t", 1)[0]) for r in sb[1:])
    assert ids == list(range(len(ids))), "BubbleIds must be 0 .. n-1"
    rows = []
    for r in sb[1:]:
        _, ent, strand, ex, simple, cx = r.split("\t")
        rows.append("\t".join([name[ent], strand, name[ex], simple, cx]))
    out["super_bubble"] = "\n".join(sorted(rows)) + "\n"
    # alignseq: var_count -> "entrance:exit"
    key = {}
    groups = {}
    for r in read("alignseq"):
        vc, strict, u, ex, row = r.split("\t")
        k = "%s:%s" % (name[u], name[ex])
        assert key.setdefault(vc, k) == k, "one bubble per var_count"
        groups.setdefault(vc, []).append("\t".join([strict, name[u], name[ex], row]))
    vcs = sorted(int(v) for v in key)
    assert vcs == list(range(len(vcs))), "var_counts must be 0 .. n-1"
    assert len(set(key.values())) == len(key), "one var_count per bubble"
    out["alignseq"] = "\n".join(sorted(key[v] + "\n" + "\n".join(rows) for v, rows in groups.items())) + "\n"
    for a in ARITY:
        g = {}
        for r in read(a + "cov"):
            f = r.split("\t")
            vc = f[-4]   # ... strict, indel length, var_count, sites, distance, ""
            f[-4] = key[vc]
            g.setdefault(vc, []).append("\t".join(f))
        out[a + "cov"] = "\n".join(sorted("\n".join(rows) for rows in g.values())) + "\n"
        out[a + "fre"] = "\n".join(sorted(read(a + "fre"))) + "\n"
    out["allele_frequency"] = "\n".join(sorted(read("allele_frequency"))) + "\n"
    return out

"""GFA dialects of one graph: the variants of a fixture's graph.gfa that the reference's GFA_Parser (bifrost/src/GFA_Parser.cpp:380-520)
meets in the wild.  Used by tests/golden/make_dialect_golden.py (which runs the REFERENCE binary on each and stores what it wrote
under tests/golden/dialects/) -- the tests read the stored files, they never rebuild the variants."""
import numpy as np


def variants(gfa_text: str, k: int) -> dict:
    lines = gfa_text.split("\n")[:-1]
    head, body = lines[0], lines[1:]
    seg = [i for i, ln in enumerate(body) if ln.startswith("S\t")]
    rng = np.random.default_rng(5)
    out = {}
    # every line ends "\r\n"; the sequence is the last field, so the '\r' is read as a base (GFA_Parser.cpp:497-503)
    out["crlf"] = "\r\n".join([head] + body) + "\r\n"
    # CRLF again, with the '\r' landing where it reproduces the original graph: a segment longer than k that ends in A loses
    # that A (the '\r' is stored as A, CompressedSequence.cpp:597-614), a k-length segment that ends in T loses its T (Kmer.cpp:92-107),
    # (or is written as its reverse complement without the final T), every other segment carries a tag behind its sequence, which takes the '\r'
    exact = list(body)
    for i in seg:
        f = exact[i].split("\t")
        s = f[2]
        if len(f) == 3 and len(s) > k and s[-1] == "A":
            f[2] = s[:-1]
        elif len(f) == 3 and len(s) == k and (s[-1] == "T" or s[0] == "A"):
            # (a k-length segment is stored as min(sequence, reverse complement) either way: CompactedDBG.tcc:3945-3954)
            f[2] = s[:-1] if s[-1] == "T" else s[::-1].translate(str.maketrans("ACGT", "TGCA"))[:-1]
        else:
            f.append("KC:i:%d" % (i + 1))
        exact[i] = "\t".join(f)
    out["crlf_exact"] = "\r\n".join([head] + exact) + "\r\n"
    lower = list(body)
    for i in seg[::3]:
        f = lower[i].split("\t")
        f[2] = f[2].lower() if i % 2 else "".join(c.lower() if j % 3 == 0 else c for j, c in enumerate(f[2]))
        lower[i] = "\t".join(f)
    out["lowercase"] = "\n".join([head] + lower) + "\n"
    tags = list(body)
    for n, i in enumerate(seg):
        tags[i] += "\tKC:i:%d" % (n * 7) + ("\tDA:Z:%d" % (n % 5) if n % 2 else "") + "\txx:Z:S\tS"
    out["tags"] = "\n".join([head] + tags) + "\n"
    gfa2 = []
    for ln in body:
        f = ln.split("\t")
        gfa2.append("\t".join(["S", f[1], str(len(f[2])), f[2]] + f[3:]) if f[0] == "S" else ln)
    out["gfa2"] = "\n".join([head.replace("VN:Z:1.0", "VN:Z:2.0")] + gfa2) + "\n"
    shuffled = list(body)   # segments and links interleaved, comment lines, an empty line
    rng.shuffle(shuffled)
    shuffled[len(shuffled) // 2:len(shuffled) // 2] = ["# a comment", "", "P\tpath\t1+,2-\t*"]
    out["interleaved"] = "\n".join([head] + shuffled) + "\n"
    out["no_final_newline"] = "\n".join([head] + body)   # the last line is dropped (GFA_Parser.cpp:486)
    return out

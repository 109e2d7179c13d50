#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REAL REFERENCE.

Needs oracle/_ref/simple (built by `make -C oracle ref` from /root/reference/simple.cpp where it
lies; nothing of the reference is copied).  The reference resolves its inputs as
"sequences/<name>" relative to the working directory (simple.cpp:302-309), so it is run with
cwd = a directory holding a `sequences/` folder.

Outputs (all committed, all data):
  ref_fasta_pairs.json   stdout of the reference for ordered pairs of the shipped FASTA files
  ref_fuzz.json          300 synthetic pairs (sequences inline) + reference stdout, varied p
  ref_format_cases.json  FASTA-format edge cases (CRLF, multi-record, lowercase, missing file...)
  ref_random.json        random protein pairs 512/1024/2048 (std::mt19937_64 stream)
  sequences/synth_*.fasta  the inputs of ref_random.json
Restatement-only goldens (inputs the reference cannot run, e.g. titin) are made by
make_oracle_golden.py (oracle_only.json) and are marked as such.

Usage: python tests/golden/make_golden.py            (takes a few minutes)
"""
import base64
import hashlib
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "simple")
ALPHABET = "ACDEFGHIKLMNPQRSTVWY"


class MT19937_64:
    """std::mt19937_64 (Matsumoto & Nishimura 2004), pure Python, for fixture generation."""
    NN, MM = 312, 156
    MATRIX_A, UM, LM = 0xB5026F5AA96619E9, 0xFFFFFFFF80000000, 0x7FFFFFFF

    def __init__(self, seed):
        mt = [0] * self.NN
        mt[0] = seed & 0xFFFFFFFFFFFFFFFF
        for i in range(1, self.NN):
            mt[i] = (6364136223846793005 * (mt[i - 1] ^ (mt[i - 1] >> 62)) + i) & 0xFFFFFFFFFFFFFFFF
        self.mt, self.mti = mt, self.NN

    def __call__(self):
        mt = self.mt
        if self.mti >= self.NN:
            for i in range(self.NN):
                x = (mt[i] & self.UM) | (mt[(i + 1) % self.NN] & self.LM)
                mt[i] = mt[(i + self.MM) % self.NN] ^ (x >> 1) ^ (self.MATRIX_A if x & 1 else 0)
            self.mti = 0
        x = mt[self.mti]
        self.mti += 1
        x ^= (x >> 29) & 0x5555555555555555
        x ^= (x << 17) & 0x71D67FFFEDA60000
        x ^= (x << 37) & 0xFFF7EEE000000000
        x ^= x >> 43
        return x & 0xFFFFFFFFFFFFFFFF


def synth_protein(seed, n):
    g = MT19937_64(seed)
    return "".join(ALPHABET[g() % 20] for _ in range(n))


def write_fasta(path, header, seq, width=60):
    with open(path, "w") as f:
        f.write(">" + header + "\n")
        for k in range(0, len(seq), width):
            f.write(seq[k:k + width] + "\n")


def run_ref(cwd, a, b, p, timeout=3600):
    r = subprocess.run([REF, a, b, str(p)], cwd=cwd, capture_output=True, timeout=timeout)
    return r.returncode, r.stdout, r.stderr


def b64(x):
    return base64.b64encode(x).decode("ascii")


def entry(a, b, p, rc, out, err):
    return {"a": a, "b": b, "p": p, "rc": rc, "stdout_b64": b64(out),
            "stdout_md5": hashlib.md5(out).hexdigest(), "stderr_b64": b64(err)}


def fasta_pairs():
    small = ["insulin_homo.fasta", "insulin_bovin.fasta", "Q16739_homo.fasta",
             "A0A2I3N2W7_baboon.fasta", "Q9CD83.fasta"]
    pairs = [(x, y) for x in small for y in small]
    pairs += [("homo_insulinlike.fasta", "mouse_insulinlike.fasta"),
              ("mouse_insulinlike.fasta", "homo_insulinlike.fasta"),
              ("insulin_homo.fasta", "homo_insulinlike.fasta"),
              ("mouse_insulinlike.fasta", "Q9CD83.fasta")]
    out = []
    ps = [1, 2, 3, 4, 5, 7, 8, 13, 16]
    for k, (x, y) in enumerate(pairs):
        p = 8 if "insulinlike" in x and "insulinlike" in y else ps[k % len(ps)]
        rc, so, se = run_ref(HERE, x, y, p)
        out.append(entry(x, y, p, rc, so, se))
        print("pair", x, y, p, hashlib.md5(so).hexdigest(), file=sys.stderr)
    # the paper's headline pair at several p: output must not depend on p
    for p in (1, 2, 4, 8, 16):
        rc, so, se = run_ref(HERE, "Q16739_homo.fasta", "A0A2I3N2W7_baboon.fasta", p)
        out.append(entry("Q16739_homo.fasta", "A0A2I3N2W7_baboon.fasta", p, rc, so, se))
    return out


def titin_prefix(tmp):
    """3000x3000 prefixes of the titin pair: the largest high-identity case the reference can run."""
    def read(path):
        return "".join(l.strip("\n") for l in open(path) if l.strip("\n") and not l.startswith(">"))
    a = read(os.path.join(HERE, "sequences", "titin_homo(50).fasta"))[:3000]
    b = read(os.path.join(HERE, "sequences", "titin_mus.fasta"))[:3000]
    write_fasta(os.path.join(HERE, "sequences", "synth_titin_homo_3000.fasta"), "titin_homo(50)[:3000]", a)
    write_fasta(os.path.join(HERE, "sequences", "synth_titin_mus_3000.fasta"), "titin_mus[:3000]", b)
    rc, so, se = run_ref(HERE, "synth_titin_homo_3000.fasta", "synth_titin_mus_3000.fasta", 8)
    return entry("synth_titin_homo_3000.fasta", "synth_titin_mus_3000.fasta", 8, rc, so, se)


def fuzz(tmp):
    rnd = random.Random(20251110)
    seqdir = os.path.join(tmp, "sequences")
    os.makedirs(seqdir, exist_ok=True)
    cases = []
    ps = [1, 2, 3, 4, 5, 8, 13]
    for k in range(300):
        nletters = rnd.choice([2, 2, 3, 3, 4, 5, 8, 20])
        alpha = ALPHABET[:nletters]
        kind = k % 6
        if kind == 0:      # tiny
            n, m = rnd.randint(1, 8), rnd.randint(1, 8)
        elif kind == 1:    # wave / tile edges
            n, m = rnd.choice([63, 64, 65, 127, 128, 129, 191, 193]), rnd.choice([1, 31, 63, 64, 65, 97, 128, 130])
        elif kind == 2:    # very unequal
            n, m = rnd.randint(1, 12), rnd.randint(100, 300)
            if rnd.random() < 0.5:
                n, m = m, n
        else:
            n, m = rnd.randint(1, 300), rnd.randint(1, 300)
        a = "".join(rnd.choice(alpha) for _ in range(n))
        if kind in (4, 5):  # b = mutated copy of a (high identity, long diagonal runs + indels)
            b = []
            for ch in a:
                r = rnd.random()
                if r < 0.05:
                    continue
                if r < 0.10:
                    b.append(rnd.choice(alpha))
                if r < 0.20:
                    b.append(rnd.choice(alpha))
                else:
                    b.append(ch)
            b = "".join(b) or rnd.choice(alpha)
        else:
            b = "".join(rnd.choice(alpha) for _ in range(m))
        p = ps[k % len(ps)]
        write_fasta(os.path.join(seqdir, "a.fasta"), "a", a)
        write_fasta(os.path.join(seqdir, "b.fasta"), "b", b)
        rc, so, se = run_ref(tmp, "a.fasta", "b.fasta", p)
        e = entry("a.fasta", "b.fasta", p, rc, so, se)
        e["a_seq"], e["b_seq"] = a, b
        cases.append(e)
    return cases


def format_cases(tmp):
    """FASTA reader behaviour through the binary (read_fasta.cpp:16-19) and degenerate inputs."""
    seqdir = os.path.join(tmp, "sequences")
    os.makedirs(seqdir, exist_ok=True)
    files = {
        "crlf.fasta": b">x\r\nMALWMRLLPL\r\nLALLALWGPD\r\n",
        "plain.fasta": b">y\nMALWMRLLPLLALLALWGPD\n",
        "multi.fasta": b">r1\nMALWM\n>r2\nRLLPL\n\n>r3\nLALLA\n",
        "lower.fasta": b">z\nmalwmrllpllallalwgpd\n",
        "nohdr.fasta": b"MALWMRLLPLLALLALWGPD\n",
        "spaces.fasta": b">w\nMALWM\n   \nRLLPL\n",
        "noeol.fasta": b">v\nMALWMRLLPLLALLALWGPD",
        "empty.fasta": b"",
        "hdronly.fasta": b">only a header\n",
        "one.fasta": b">o\nH\n",
        "two.fasta": b">t\nLC\n",
    }
    for name, data in files.items():
        with open(os.path.join(seqdir, name), "wb") as f:
            f.write(data)
    runs = [("crlf.fasta", "plain.fasta", 2), ("plain.fasta", "crlf.fasta", 2),
            ("multi.fasta", "plain.fasta", 1), ("lower.fasta", "plain.fasta", 3),
            ("nohdr.fasta", "plain.fasta", 1), ("spaces.fasta", "plain.fasta", 2),
            ("noeol.fasta", "plain.fasta", 1), ("empty.fasta", "plain.fasta", 1),
            ("plain.fasta", "empty.fasta", 1), ("empty.fasta", "empty.fasta", 2),
            ("hdronly.fasta", "plain.fasta", 1), ("plain.fasta", "hdronly.fasta", 4),
            ("missing.fasta", "plain.fasta", 1), ("plain.fasta", "missing.fasta", 1),
            ("one.fasta", "two.fasta", 1), ("two.fasta", "one.fasta", 1),
            ("one.fasta", "one.fasta", 1), ("one.fasta", "empty.fasta", 1),
            ("plain.fasta", "one.fasta", 8), ("one.fasta", "plain.fasta", 8),
            ("plain.fasta", "plain.fasta", 32)]
    out = []
    for a, b, p in runs:
        rc, so, se = run_ref(tmp, a, b, p)
        e = entry(a, b, p, rc, so, se)
        e["a_file_b64"] = b64(files[a]) if a in files else None
        e["b_file_b64"] = b64(files[b]) if b in files else None
        out.append(e)
    # usage error (simple.cpp:298-301)
    r = subprocess.run([REF, "plain.fasta"], cwd=tmp, capture_output=True)
    out.append({"argv": ["plain.fasta"], "rc": r.returncode, "stdout_b64": b64(r.stdout),
                "stderr_tail": r.stderr.decode().split(" ", 2)[-1]})
    return out


def random_pairs():
    out = []
    for k, n in ((9, 512), (10, 1024), (11, 2048)):
        a = synth_protein(0xA0000000 + k, n)
        b = synth_protein(0xB0000000 + k, n)
        fa, fb = "synth_A_2p%d.fasta" % k, "synth_B_2p%d.fasta" % k
        write_fasta(os.path.join(HERE, "sequences", fa), "synthetic A seed=0xA0000000+%d n=%d" % (k, n), a)
        write_fasta(os.path.join(HERE, "sequences", fb), "synthetic B seed=0xB0000000+%d n=%d" % (k, n), b)
        rc, so, se = run_ref(HERE, fa, fb, 8)
        print("random", n, hashlib.md5(so).hexdigest(), file=sys.stderr)
        out.append(entry(fa, fb, 8, rc, so, se))
    return out


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=0, sort_keys=True)
        f.write("\n")


def main():
    if not os.path.exists(REF):
        sys.exit("build the reference first: make -C oracle ref")
    tmp = tempfile.mkdtemp(prefix="psa_golden_")
    try:
        dump("ref_fuzz.json", fuzz(tmp))
        dump("ref_format_cases.json", format_cases(tmp))
        pairs = fasta_pairs()
        pairs.append(titin_prefix(tmp))
        dump("ref_fasta_pairs.json", pairs)
        dump("ref_random.json", random_pairs())
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()

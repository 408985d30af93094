#!/usr/bin/env python3
"""Build a Python-3 importable *scratch copy* of the reference under /tmp.

TEST INFRASTRUCTURE ONLY.  The reference (GreenleafLab/NucleoATAC v0.3.4) is
Python 2.7 + two Cython files and cannot be imported as-is in this image.
This script copies /root/reference into a scratch directory OUTSIDE the repo,
mechanically converts it (lib2to3, py2 integer division, renamed scipy/numpy
APIs, a tiny stand-in for the `pysam` readers backed by .npz files) and
compiles `multinomial_cov.pyx` with its accumulator initialised to 0
(the .pyx leaves it uninitialised, nucleoatac/multinomial_cov.pyx:23; the
intended value is pinned by the reference's tests/test_var.py:34-43).

The scratch copy is used by tests/golden/make_golden.py to
  (i)  validate oracle/natac_oracle.py (our restatement), and
  (ii) emit golden input/output vectors (data only) into tests/golden/.
Nothing produced here is ever written into the repository and nothing here
is imported by the product, the tests, bench.py or smoke().

usage: python oracle/make_scratch_ref.py [/tmp/natac_scratch_ref]
"""
import os
import re
import shutil
import subprocess
import sys
import textwrap

REF = "/root/reference"


def sub_file(path, subs, regex=False):
    with open(path) as f:
        s = f.read()
    for old, new in subs:
        if regex:
            s = re.sub(old, new, s)
        else:
            s = s.replace(old, new)
    with open(path, "w") as f:
        f.write(s)


# py2 `int / int` -> `//` : every "/2" and "/3" (not "/3.0") in these files is an
# integer division in the reference
INTDIV = (r"/\s*(?=[23]\b(?!\.))", "//")

PYSAM_STUB = '''
"""Stand-in for the parts of pysam the reference touches; backed by .npz files.

An "alignment file" is an .npz with arrays  chrom_names, chrom_lengths and, per
chromosome c,  pos_<c> (sorted leftmost coordinate of the forward mate),
tlen_<c> (template length).  Every stored record is a forward-strand proper
pair, i.e. exactly the reads pyatac/fragments.pyx keeps.
A "fasta file" is an .npz with chrom_names, chrom_lengths, seq_<c> (bytes).
"""
import gzip
import numpy as np


def _parse_bam(filename):
    """Minimal BGZF/BAM record reader (SAM spec 4.2): keeps forward proper pairs."""
    import struct
    with gzip.open(filename, "rb") as f:
        b = f.read()
    assert b[:4] == b"BAM\x01"
    l_text, = struct.unpack_from("<i", b, 4)
    o = 8 + l_text
    n_ref, = struct.unpack_from("<i", b, o)
    o += 4
    names, lens = [], []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", b, o)
        names.append(b[o + 4:o + 4 + l_name - 1].decode())
        l_ref, = struct.unpack_from("<i", b, o + 4 + l_name)
        lens.append(l_ref)
        o += 8 + l_name
    recs = {n: ([], []) for n in names}
    while o + 4 <= len(b):
        bs, = struct.unpack_from("<i", b, o)
        ref_id, pos, _ln, _mq, _bin, _nc, flag, _ls, _nr, _np, tlen = struct.unpack_from("<iiBBHHHiiii", b, o + 4)
        if ref_id >= 0 and (flag & 0x2) and not (flag & 0x10):
            recs[names[ref_id]][0].append(pos)
            recs[names[ref_id]][1].append(tlen)
        o += 4 + bs
    d = {"chrom_names": np.array(names), "chrom_lengths": np.array(lens)}
    for n in names:
        d["pos_" + n] = np.array(recs[n][0], dtype=np.int64)
        d["tlen_" + n] = np.array(recs[n][1], dtype=np.int64)
    return d


class _Read(object):
    __slots__ = ("pos", "template_length", "is_proper_pair", "is_reverse")

    def __init__(self, pos, tlen):
        self.pos = int(pos)
        self.template_length = int(tlen)
        self.is_proper_pair = True
        self.is_reverse = False


class AlignmentFile(object):
    def __init__(self, filename, mode="rb"):
        if filename.endswith(".bam"):
            self._d = _parse_bam(filename)
        else:
            self._d = np.load(filename, allow_pickle=False)
        self.references = [str(x) for x in self._d["chrom_names"]]
        self.lengths = [int(x) for x in self._d["chrom_lengths"]]

    def fetch(self, chrom=None, start=None, end=None):
        chroms = self.references if chrom is None else [chrom]
        for c in chroms:
            pos = self._d["pos_" + c]
            tlen = self._d["tlen_" + c]
            if start is None:
                lo, hi = 0, len(pos)
            else:
                # htslib fetch returns reads OVERLAPPING [start,end); reads are
                # <= 1000 bp here so widen the left edge generously -- the
                # reference only uses fetch as a superset filter
                # (pyatac/fragments.pyx:24,37).
                lo = int(np.searchsorted(pos, start - 1000, "left"))
                hi = int(np.searchsorted(pos, end, "left"))
            for i in range(lo, hi):
                yield _Read(pos[i], tlen[i])

    def __iter__(self):
        return self.fetch()

    def close(self):
        pass


Samfile = AlignmentFile


class FastaFile(object):
    def __init__(self, filename):
        self._d = np.load(filename, allow_pickle=False)
        self.references = [str(x) for x in self._d["chrom_names"]]
        self.lengths = [int(x) for x in self._d["chrom_lengths"]]

    def fetch(self, chrom, start, end):
        s = self._d["seq_" + chrom]
        return bytes(s[start:end]).decode("ascii")

    def close(self):
        pass


class asTuple(object):
    pass


class TabixFile(object):
    """Linear scan of a (b)gzip text file; good enough for the example tracks."""

    def __init__(self, filename):
        self._rows = {}
        with gzip.open(filename, "rt") as f:
            for line in f:
                t = line.rstrip("\\n").split("\\t")
                self._rows.setdefault(t[0], []).append(t)
        self.contigs = list(self._rows.keys())

    def fetch(self, chrom, start, end, parser=None):
        for t in self._rows.get(chrom, []):
            if int(t[2]) > start and int(t[1]) < end:
                yield t

    def close(self):
        pass


Tabixfile = TabixFile


def tabix_compress(src, dst, force=False):
    """plain gzip: the stub TabixFile above scans the text, no BGZF / index needed"""
    import shutil
    with open(src, "rb") as fi, gzip.open(dst, "wb") as fo:
        shutil.copyfileobj(fi, fo)


def tabix_index(*a, **k):
    return None
'''


def pyx_to_py(src):
    """Strip Cython typing from pyatac/fragments.pyx -> plain Python (same statements)."""
    out = []
    for line in src.splitlines():
        st = line.strip()
        if st.startswith(("cimport", "ctypedef", "@cython", "from pysam", "cdef ")) and "=" not in st:
            continue
        if st.startswith("cimport") or st.startswith("ctypedef") or st.startswith("@cython") \
                or st.startswith("from pysam."):
            continue
        line = re.sub(r"cdef\s+np\.ndarray\[[^\]]*\]\s+", "", line)
        line = re.sub(r"cdef\s+(AlignmentFile|AlignedSegment|int|DTYPE_t)\s+", "", line)
        line = re.sub(r"\b(str|int)\s+(\w+)(\s*[,)=])", r"\2\3", line) if line.lstrip().startswith("def ") else line
        line = line.replace("np.float)", "np.float64)").replace("= np.float\n", "= np.float64\n")
        line = re.sub(r"dtype\s*=\s*DTYPE", "dtype = np.float64", line)
        line = line.replace("(ilen-1)/2", "(ilen-1)//2")
        out.append(line)
    body = "\n".join(out)
    body = body.replace("DTYPE = np.float\n", "DTYPE = np.float64\n")
    body = body.replace("dtype= np.float)", "dtype= np.float64)")
    return "from pysam import AlignmentFile\n" + body + "\n"


def main():
    dst = sys.argv[1] if len(sys.argv) > 1 else "/tmp/natac_scratch_ref"
    assert not os.path.abspath(dst).startswith("/root/repo"), "scratch copy must live outside the repo"
    if os.path.exists(dst):
        shutil.rmtree(dst)
    src = os.path.join(dst, "src")
    shutil.copytree(REF, src)
    subprocess.check_call(["chmod", "-R", "u+w", src])
    subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n", "pyatac", "nucleoatac", "tests"],
                          cwd=src, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    P = lambda *a: os.path.join(src, *a)
    intdiv_files = ["pyatac/VMat.py", "pyatac/bias.py", "pyatac/chunk.py", "pyatac/chunkmat2d.py",
                    "pyatac/tracks.py", "pyatac/utils.py", "nucleoatac/NucleosomeCalling.py",
                    "nucleoatac/Occupancy.py", "nucleoatac/run_occ.py", "nucleoatac/run_nuc.py",
                    "nucleoatac/NFRCalling.py", "nucleoatac/run_nfr.py"]
    for f in intdiv_files:
        sub_file(P(f), [INTDIV], regex=True)
    # API renames (numpy 2 / scipy 1.15 / python 3)
    sub_file(P("pyatac/VMat.py"), [("ndimage.filters.", "ndimage.")])
    sub_file(P("pyatac/utils.py"), [("signal.gaussian(", "signal.windows.gaussian(")])
    sub_file(P("nucleoatac/Occupancy.py"), [("np.float('inf')", "float('inf')")])
    sub_file(P("pyatac/seq.py"), [("string.maketrans", "str.maketrans")])
    sub_file(P("pyatac/chunk.py"), [("list.sort(self, cmp = _chunkCompare)",
                                      "import functools; list.sort(self, key = functools.cmp_to_key(_chunkCompare))")])
    pyx_line = re.compile(r"^.*pyximport.*$", re.M)
    for root, _, files in os.walk(src):
        for fn in files:
            if fn.endswith(".py"):
                p = os.path.join(root, fn)
                with open(p) as f:
                    s = f.read()
                s = pyx_line.sub("", s)
                s = s.replace("from fragments import", "from pyatac.fragments import")
                s = s.replace('gzip.open(bedfile,"r")', 'gzip.open(bedfile,"rt")')      # python 3: text, not bytes
                s = s.replace("import VMat as V\n", "import pyatac.VMat as V\n")
                with open(p, "w") as f:
                    f.write(s)
    for pkg in ("pyatac", "nucleoatac"):
        with open(P(pkg, "__init__.py"), "w") as f:
            f.write('__version__ = "0.3.4"\n')
    # Cython sources
    with open(P("pyatac/fragments.pyx")) as f:
        frag = f.read()
    with open(P("pyatac/fragments.py"), "w") as f:
        f.write(pyx_to_py(frag))
    os.remove(P("pyatac/fragments.pyx"))
    sub_file(P("nucleoatac/multinomial_cov.pyx"),
             [("DTYPE = np.float\n", "DTYPE = np.float64\n"),
              ("cdef DTYPE_t value\n", "cdef DTYPE_t value = 0\n")])
    setup = textwrap.dedent("""
        from setuptools import setup, Extension
        from Cython.Build import cythonize
        import numpy as np
        setup(ext_modules=cythonize([Extension("nucleoatac.multinomial_cov", ["nucleoatac/multinomial_cov.pyx"],
              include_dirs=[np.get_include()])], language_level=2), script_args=["build_ext", "--inplace", "-q"])
    """)
    with open(P("_build_cov.py"), "w") as f:
        f.write(setup)
    subprocess.check_call([sys.executable, "_build_cov.py"], cwd=src, stdout=subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL)
    os.makedirs(os.path.join(dst, "stubs"), exist_ok=True)
    with open(os.path.join(dst, "stubs", "pysam.py"), "w") as f:
        f.write(PYSAM_STUB)
    print("scratch reference ready:", dst)
    print("use: PYTHONPATH=%s/stubs:%s MPLBACKEND=agg python ..." % (dst, src))


if __name__ == "__main__":
    main()
